# kernel timeline of one fp32 Kron update (the 9th call): bash tools/r06_kron_trace.sh [M N [KRON_KEYS]]  ->  gpurun_out/ktrace.txt
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace
KRON_KEYS=${3:-} rocprofv3 --kernel-trace -d /tmp/ktrace -- python3 $R/tools/kron_update_trace.py ${1:-4096} ${2:-4096} 2 12 > /tmp/ktrace.log 2>&1
db=$(find /tmp/ktrace -name '*_results.db' | head -1)
python3 $R/tools/trace_timeline.py $db ${4:-k_kron_rho} 8 2>&1 | sed -e 's/_ZN5psgdk[0-9]*//' -e 's/E[vPN].*//' | cut -c1-72 > $R/gpurun_out/ktrace.txt
cat $R/gpurun_out/ktrace.txt
