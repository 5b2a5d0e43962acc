mkdir -p gpurun_out/chunk
R=$PWD
for v in "" _ch4 _ch16; do
export PSGD_HIP_LIB=$R/psgd_tf_amd/csrc/libpsgd_hip$v.so
echo "== lib$v" >> gpurun_out/chunk/uvd.txt
python tools/uvd_timing.py --N 100000000 --r 20 --iters 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/chunk/uvd.txt
python -m pytest tests/test_splu_gpu.py tests/test_uvd_gpu.py -m gpu -q -x 2>&1 | tail -1 >> gpurun_out/chunk/uvd.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/chunk/stats$v -- python3 $R/tools/splu_timing.py --N 50000000 --r 10 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/chunk/stats$v > $R/gpurun_out/chunk/splu_kernel_stats$v.csv
rm -rf $R/gpurun_out/chunk/stats$v
cd $R
done
cat gpurun_out/chunk/uvd.txt
