# kernel timeline of one wide-rank UVd update and one fused step (r = 64 and r = 40, N = 20 M): bash tools/r05_wide_trace.sh
R=$PWD
mkdir -p gpurun_out/r05wide
export TMPDIR=/tmp
for what in update step; do
for r in 64 40; do
  rm -rf /tmp/r05wide && rocprofv3 --kernel-trace --stats -d /tmp/r05wide -- python3 tools/r05_wide_trace.py 20000000 $r $what 6 > $R/gpurun_out/r05wide/out_${what}_$r.txt 2>&1
  DB=$(find /tmp/r05wide -name "*_results.db" | head -1)
  python3 tools/trace_timeline.py $DB k_gram_wideILi 3 > gpurun_out/r05wide/timeline_${what}_$r.txt 2>&1
done
done
cat gpurun_out/r05wide/timeline_step_64.txt gpurun_out/r05wide/timeline_step_40.txt
