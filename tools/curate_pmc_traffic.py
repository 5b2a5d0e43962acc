"""Turns the raw per-kernel table of tools/pmc_traffic.py (gpurun_out/<tag>/pmc_traffic.json) into
profiles/pmc_traffic.json, the small file bench.py reads for `roofline.traffic` (keys k_update_s2 / k_update_gram with
the row count and rank they were measured at).   python tools/curate_pmc_traffic.py gpurun_out/v9/pmc_traffic.json v9"""
import json
import os
import sys

raw = json.load(open(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else "?"
rows, r = 100_000_000, 20


def pick(prefix):
    ks = [k for k in raw if prefix in k]
    return raw[max(ks, key=lambda k: raw[k]["hbm_bytes_per_launch"])]


s2, gr = pick("k_update_s2<%d" % r), pick("k_update_gram<%d" % r)
out = {
    "_doc": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --placement packed --steps 3 "
            "--warmup 1 --no-kron --no-exchange-leg`; FETCH_SIZE doubled (gfx950: counts 64 B per 128-B request), WRITE_SIZE exact; see "
            "MI355X_MICROARCH.md HBM section and tools/pmc_traffic.py. Source: profiles/r06_bench_pmc_traffic_%s.json "
            "(the raw per-kernel table of the same passes)" % tag,
    "k_update_s2": {"rows": rows, "r": r, "hbm_bytes_per_launch": s2["hbm_bytes_per_launch"],
                    "fetch_bytes_corrected_x2": s2["fetch_bytes_corrected_x2"], "write_bytes": s2["write_bytes"],
                    "variant": "fused"},
    "k_update_gram": {"rows": rows, "r": r, "hbm_bytes_per_launch": gr["hbm_bytes_per_launch"], "alg_bytes": 4 * (2 * r + 3) * rows},
}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, out["k_update_s2"]["hbm_bytes_per_launch"])
