# Hardware counters of the plane-product kernels inside one 4096^2 fp32 Kron update: the gradient grid (43 % MfmaUtil) against the plain
# products (53 %): L2 hit rates, LDS / VMEM waits.   bash tools/p3_grad_pmc.sh
R=$PWD
mkdir -p gpurun_out/p3pmc
export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d $R/gpurun_out/p3pmc/$tag -- python3 tools/kron_update_trace.py 4096 4096 2 6 > $R/gpurun_out/p3pmc/$tag.log 2>&1
done
python3 - <<'PY'
import sqlite3, glob, collections
for d in sorted(glob.glob('gpurun_out/p3pmc/*/')):
    dbs = glob.glob(d + '**/*_results.db', recursive=True)
    if not dbs: print(d, "no db"); continue
    con = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if 'gemm_p3' in name: acc[name.split('(')[0]][ctr].append((val, dur))
    for k, c in sorted(acc.items()):
        for ctr, vals in c.items():
            big = [v for v in vals if v[1] > 60e3] or vals          # (the long launches: full products, not the inversion levels)
            n = len(big)
            print("%-40s %-30s launches %3d  mean %.4g  mean_dur_us %.1f" % (k[-40:], ctr, n, sum(v for v, _ in big) / n, sum(d for _, d in big) / n / 1e3))
PY
rm -rf gpurun_out/p3pmc/*/
