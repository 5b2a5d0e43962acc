# kernel timeline of one fp32 Kron apply with new factors (the 9th call): bash tools/r06_kron_apply_trace.sh [M N [route]]  ->  gpurun_out/atrace.txt
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/atrace
rocprofv3 --kernel-trace -d /tmp/atrace -- python3 $R/tools/kron_apply_trace.py ${1:-1024} ${2:-4096} 12 ${3:-reference} > /tmp/atrace.log 2>&1
db=$(find /tmp/atrace -name '*_results.db' | head -1)
python3 $R/tools/trace_timeline.py $db k_kron_balance_planes 8 2>&1 | sed -e 's/_ZN5psgdk[0-9]*//' -e 's/E[vPN].*//' | cut -c1-72 > $R/gpurun_out/atrace.txt
cat $R/gpurun_out/atrace.txt
