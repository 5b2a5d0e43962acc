set -x
mkdir -p gpurun_out/v5
python bench.py > gpurun_out/v5/bench.json.log 2>gpurun_out/v5/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/v5/stats -- python3 $R/bench.py --no-cpu-baseline --no-kron > $R/gpurun_out/v5/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/v5/pmc_f -- python3 $R/bench.py --no-cpu-baseline --no-kron --steps 3 --warmup 1 > $R/gpurun_out/v5/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/v5/pmc_w -- python3 $R/bench.py --no-cpu-baseline --no-kron --steps 3 --warmup 1 > $R/gpurun_out/v5/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/v5/pmc_f gpurun_out/v5/pmc_w > gpurun_out/v5/pmc_traffic.json
find gpurun_out/v5 -name "*kernel_stats.csv" | head
tail -2 gpurun_out/v5/bench.json.log
# drop big raw traces
find gpurun_out/v5 -name "*kernel_trace.csv" -delete
find gpurun_out/v5 -name "*counter_collection.csv" -size +20M -delete
