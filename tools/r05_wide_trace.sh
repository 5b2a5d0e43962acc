# kernel timeline of one wide-rank UVd update (r = 64 and r = 40, N = 20 M): bash tools/r05_wide_trace.sh
R=$PWD
mkdir -p gpurun_out/r05wide
export TMPDIR=/tmp
for r in 64 40; do
  rm -rf /tmp/r05wide && rocprofv3 --kernel-trace --stats -d /tmp/r05wide -- python3 tools/r05_wide_trace.py 20000000 $r update 6 > $R/gpurun_out/r05wide/out_$r.txt 2>&1
  DB=$(find /tmp/r05wide -name "*_results.db" | head -1)
  python3 tools/trace_timeline.py $DB k_gram_wideILi 3 > gpurun_out/r05wide/timeline_$r.txt 2>&1
done
tail -70 gpurun_out/r05wide/timeline_64.txt
