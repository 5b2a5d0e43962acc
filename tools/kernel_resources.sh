#!/bin/bash
# Register / LDS / occupancy table of the UVd sweep kernels of one rank group (development aid).
#   tools/kernel_resources.sh 17 [filter-regex]     -> ranks 17..24
cd "$(dirname "$0")/../psgd_tf_amd/csrc" || exit 1
LO=${1:-17}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DPSGD_RANK_LO=$LO -DPSGD_GROUP_FN=uvd_ops_groupX \
  -Rpass-analysis=kernel-resource-usage -c ${SRC:-uvd_rank_group.hip} -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
import subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void psgd::", "")
    if flt and not re.search(flt, name):
        continue
    print("%-46s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occ %d" % (name, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
          r.get("ScratchSize", -1), r.get("LDS Size", -1), r.get("Occupancy", -1)))
' "$2"
