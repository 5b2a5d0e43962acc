# what bounds a K step of the plane GEMM when a workgroup has its CU to itself (1024^3: 64 tiles) and when two share one (4096^3):
# the micro benchmark with and without the DMA (X3_DBG=1: results wrong, timing only), and with / without the early first
# column of MFMAs (P3_EARLY)
mkdir -p gpurun_out
for e in ${EARLY:-0 1 2}; do for d in ${DBG:-0}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DX3_DBG=$d -DP3_EARLY=$e -Iinclude -Ipsgd_tf_amd/csrc tools/micro/x3_gemm_bench.hip -o /tmp/x3_$e$d 2>/dev/null
  for n in 1024 2048 4096; do echo -n "P3_EARLY=$e "; /tmp/x3_$e$d $n | grep "planes (k_gemm_p3) -> fp32 C  "; done
done; done
