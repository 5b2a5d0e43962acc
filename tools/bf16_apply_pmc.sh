# Hardware counters of the bf16 apply's kernels (fused triangular pairs) and, with "upd", of the bf16-operand update's bf16 kernels
# at 4096^2: L2 hit rates, waits, LDS conflicts.   bash tools/bf16_apply_pmc.sh [upd]
R=$PWD
mkdir -p gpurun_out/hpmc
export TMPDIR=/tmp
if [ "$1" = "upd" ]; then CMD="tools/kron_update_trace.py 4096 4096 2 6 bf16"; else CMD="tools/kron_bf16_probe.py 4096"; fi
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d $R/gpurun_out/hpmc/$tag -- python3 $CMD > $R/gpurun_out/hpmc/$tag.log 2>&1
done
python3 - <<'PY'
import sqlite3, glob, collections
for d in sorted(glob.glob('gpurun_out/hpmc/*/')):
    dbs = glob.glob(d + '**/*_results.db', recursive=True)
    if not dbs: print(d, "no db"); continue
    con = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if 'psgdh' in name: acc[name.split('(')[0]][ctr].append((val, dur))
    for k, c in sorted(acc.items()):
        for ctr, vals in c.items():
            n = len(vals)
            print("%-44s %-28s launches %3d  mean %.4g  mean_dur_us %.1f" % (k[-44:], ctr, n, sum(v for v, _ in vals) / n, sum(d for _, d in vals) / n / 1e3))
PY
rm -rf gpurun_out/hpmc/*/
