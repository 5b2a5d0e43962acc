# Round profile: default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, Kron MFMA counters.
# usage (on the GPU box): bash tools/prof_round.sh <tag>      outputs under gpurun_out/<tag>/ ; copy into profiles/
TAG=${1:-v6}
set -x
mkdir -p gpurun_out/$TAG
python bench.py --detail-json gpurun_out/$TAG/bench_detail.json > gpurun_out/$TAG/bench.json.log 2>gpurun_out/$TAG/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
# (the profiled passes skip the exchange_overhead leg: it runs the same r = 20 kernels at 12.5M rows, which would be averaged into the
#  per-kernel figures of the N = 100M workload; tools/sharded_step_trace.sh profiles that leg on its own)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --no-cpu-baseline --no-exchange-leg > $R/gpurun_out/$TAG/stats.log 2>&1
# (round 6: the counter passes run on the packed layout -- bytes do not depend on where the streams live, and the placement probe's
#  13M-row launches of the same kernels would be averaged into the per-launch figures; the stats pass above keeps the probe and
#  tools/rocpd_stats.py adds a row per sweep kernel for its full-size launches only)
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$TAG/pmc_f -- python3 $R/bench.py --placement packed --no-cpu-baseline --no-kron --no-wide-rank --no-exchange-leg --steps 3 --warmup 1 > $R/gpurun_out/$TAG/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$TAG/pmc_w -- python3 $R/bench.py --placement packed --no-cpu-baseline --no-kron --no-wide-rank --no-exchange-leg --steps 3 --warmup 1 > $R/gpurun_out/$TAG/pmc_w.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/$TAG/stats > gpurun_out/$TAG/kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/$TAG/pmc_f gpurun_out/$TAG/pmc_w > gpurun_out/$TAG/pmc_traffic.json
rm -rf gpurun_out/$TAG/stats gpurun_out/$TAG/pmc_f gpurun_out/$TAG/pmc_w
tail -c 400 gpurun_out/$TAG/bench.json.log
# afterwards, in the repo: cp the three files into profiles/ as r03_bench_*_$TAG and run  python tools/curate_pmc_traffic.py gpurun_out/$TAG/pmc_traffic.json $TAG  (never copy the raw table over profiles/pmc_traffic.json: bench.py reads the curated form)
