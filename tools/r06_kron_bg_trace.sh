# kernel timelines of one 4096^2 fp32 Kron update with tuning key 30 = 0 / 1 / 32768 (tools/r06_kron_bg.py)
R=$PWD
mkdir -p $R/gpurun_out/kbg
cd /tmp && export TMPDIR=/tmp
for v in ${BGV:-0 1 32768}; do
  rm -rf /tmp/kbg_$v
  KRON_KEYS=30:$v rocprofv3 --kernel-trace -d /tmp/kbg_$v -- python3 $R/tools/kron_update_trace.py ${1:-4096} ${2:-4096} 2 12 > /tmp/kbg_$v.log 2>&1
  db=$(find /tmp/kbg_$v -name '*_results.db' | head -1)
  echo "=== key 30 = $v" > $R/gpurun_out/kbg/trace_$v.txt
  python3 $R/tools/trace_timeline.py $db k_kron_balance 8 >> $R/gpurun_out/kbg/trace_$v.txt 2>&1
  tail -1 $R/gpurun_out/kbg/trace_$v.txt; tail -3 /tmp/kbg_$v.log
done
