# Where the triangular-solve strip kernel waits: hardware counters (rocprofv3 --pmc) over one 4096^2 Kron update.
R=$PWD
mkdir -p gpurun_out/trsm
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+)\b" | sort -u > $R/gpurun_out/trsm/avail.txt
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d $R/gpurun_out/trsm/$tag -- python3 $R/tools/kron_big_update.py 4096 > $R/gpurun_out/trsm/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import sqlite3, glob, collections
for d in sorted(glob.glob('gpurun_out/trsm/*/')):
    dbs = glob.glob(d + '**/*_results.db', recursive=True)
    if not dbs: print(d, "no db"); continue
    con = sqlite3.connect(dbs[0])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if 'trsm' in name or 'gemm_x3' in name: acc[name.split('(')[0]][ctr].append((val, dur))
    for k, c in acc.items():
        for ctr, vals in c.items():
            n = len(vals)
            print("%-40s %-36s launches %3d  mean %.5g  mean_dur_us %.1f" % (k[:40], ctr, n, sum(v for v, _ in vals) / n, sum(d for _, d in vals) / n / 1e3))
PY
rm -rf gpurun_out/trsm/*/
