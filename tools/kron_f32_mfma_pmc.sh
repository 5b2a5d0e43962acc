# MFMA utilisation of the fp32 Kron apply and update (operand planes, k_gemm_p3*) from hardware counters, 4096^2.
R=$PWD
mkdir -p gpurun_out/mfma32
cd /tmp && export TMPDIR=/tmp
for c in "MfmaUtil" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d $R/gpurun_out/mfma32/$tag -- python3 $R/tools/kron_f32_probe.py 4096 > $R/gpurun_out/mfma32/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import sqlite3, glob, collections
for d in sorted(glob.glob('gpurun_out/mfma32/*/')):
    dbs = glob.glob(d + '**/*_results.db', recursive=True)
    if not dbs: print(d, "no db"); continue
    con = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if 'psgdk' in name and ('gemm' in name or 'trsm' in name): acc[name.split('(')[0]][ctr].append((val, dur))
    for k, c in acc.items():
        for ctr, vals in c.items():
            n = len(vals)
            print("%-46s %-34s launches %3d  mean %.4g  mean_dur_us %.1f" % (k[:46], ctr, n, sum(v for v, _ in vals) / n, sum(d for _, d in vals) / n / 1e3))
PY
rm -rf gpurun_out/mfma32/*/
