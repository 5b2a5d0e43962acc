# Hardware counters of the wide Gram kernel (r = 64 and r = 40, N = 20 M): bash tools/r05_gram_pmc.sh  -> gpurun_out/gwpmc/summary.txt
R=$PWD
mkdir -p gpurun_out/gwpmc
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d $R/gpurun_out/gwpmc/$tag -- python3 $R/tools/r05_wide_gram_time.py 20000000 40 64 > $R/gpurun_out/gwpmc/$tag.log 2>&1
done
cd $R
python3 - <<'PY' > gpurun_out/gwpmc/summary.txt
import sqlite3, glob, collections
for d in sorted(glob.glob('gpurun_out/gwpmc/*/')):
    dbs = glob.glob(d + '**/*_results.db', recursive=True)
    if not dbs: print(d, "no db"); continue
    con = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val, dur, grid in con.execute("select kernel_name, counter_name, value, duration, grid_size from counters_collection"):
        if 'k_gram_wide<' in name or 'k_gram_wideILi' in name:
            if grid >= 200 * 1024: acc[name.split('(')[0][:40]][ctr].append((val, dur))
    for k, c in acc.items():
        for ctr, vals in c.items():
            vals = vals[-5:]
            n = len(vals)
            print("%-42s %-28s launches %3d  mean %.5g  mean_dur_us %.1f" % (k, ctr, n, sum(v for v, _ in vals) / n, sum(d for _, d in vals) / n / 1e3))
PY
cat gpurun_out/gwpmc/summary.txt
rm -rf gpurun_out/gwpmc/*/
