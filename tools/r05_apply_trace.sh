# kernel timeline of one 4096^2 fp32 apply with new factors, tile scales off / on: bash tools/r05_apply_trace.sh
R=$PWD
mkdir -p gpurun_out/r05trace
export TMPDIR=/tmp
for key in 0 1; do
  rm -rf /tmp/r05trace && rocprofv3 --kernel-trace --stats -d /tmp/r05trace -- python3 tools/r05_apply_trace.py 4096 4096 $key 12 > $R/gpurun_out/r05trace/apply_out_$key.txt 2>&1
  DB=$(find /tmp/r05trace -name "*_results.db" | head -1)
  python3 tools/trace_timeline.py $DB k_absmax_tri2 8 > gpurun_out/r05trace/apply_timeline_$key.txt 2>&1
done
tail -25 gpurun_out/r05trace/apply_timeline_0.txt gpurun_out/r05trace/apply_timeline_1.txt
