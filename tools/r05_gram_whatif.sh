# What-if builds of the wide Gram kernel (wrong results, timing only): GW_DBG 2 = no MFMA phase, 4 = no split / plane writes (the
# loads stay), 8 = no t / w columns; sums combine (14 = loads and barriers only).  Builds gpurun_out/gw_whatif/lib_<x>.so HERE (CPU box), runs them on the GPU box:
#   bash tools/r05_gram_whatif.sh build ; gpurun ... 'bash tools/r05_gram_whatif.sh run'
C=psgd_tf_amd/csrc
O=gw_whatif
if [ "$1" = build ]; then
  mkdir -p $O
  for x in 2 4 6 8 14; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C -DGW_DBG=$x -c $C/uvd_wide_gram.hip -o $O/gw_$x.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $O/lib_$x.so $(ls $C/build/*.o | grep -v -e uvd_wide_gram.o -e amdgcn) $O/gw_$x.o
  done
else
  for x in 2 4 6 8 14; do
    echo "GW_DBG=$x"; PSGD_HIP_LIB=$PWD/$O/lib_$x.so python tools/r05_wide_gram_time.py 20000000 40 64 2>&1 | tail -2
  done
fi
