#!/bin/bash
# Register / scratch / LDS usage of every kernel of one source file, from hipcc's resource-usage remarks.
# Usage: tools/kernel_resources.sh rsn_field_bwd.hip [extra -D flags]
SRC=$1; shift
# the per-file flags of reflect_sampling_nerf_amd/_build.py (SOURCE_FLAGS)
PF=$(python3 -c "import sys; sys.path.insert(0, '.'); from reflect_sampling_nerf_amd._build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$SRC', ())))")
hipcc $PF -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I reflect_sampling_nerf_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c reflect_sampling_nerf_amd/csrc/$SRC -o /dev/null "$@" 2>&1 |
  python3 -c '
import re, sys
cur = None
rows = []
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]", "SGPRs", "VGPRs Spill"):
        m = re.search(key + r": (\d+)", ln)
        if m and cur is not None: cur[key.split(" ")[0] if "Spill" not in key else "VSpill"] = int(m.group(1))
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print("%-62s VGPR %3d AGPR %3d SGPR %3d spill %3d scratch %4d occ %d LDS %6d" % (name[:62], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1), r.get("VSpill", 0), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS", -1)))
'
