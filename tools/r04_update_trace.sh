# kernel timeline of one 4096^2 Kron update (fp32 or bf16 operands): tools/r04_update_trace.sh [f32|bf16]
R=$PWD
mkdir -p gpurun_out/r04trace
export TMPDIR=/tmp
rm -rf /tmp/r04trace && rocprofv3 --kernel-trace --stats -d /tmp/r04trace -- python3 tools/kron_update_trace.py 4096 4096 2 12 ${1:-f32} > $R/gpurun_out/r04trace/out_${1:-f32}.txt 2>&1
DB=$(find /tmp/r04trace -name "*_results.db" | head -1)
cd $R && python3 tools/trace_timeline.py $DB k_kron_balance 8 > gpurun_out/r04trace/timeline_${1:-f32}.txt 2>&1
tail -3 gpurun_out/r04trace/timeline_${1:-f32}.txt
