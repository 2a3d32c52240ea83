#!/bin/bash
# Compiler's resource remarks for every kernel of the library, one line per kernel:
#   bash tools/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
# (hipcc -Rpass-analysis=kernel-resource-usage with the flags of the product build; no GPU needed)
cd "$(dirname "$0")/.." || exit 1
echo "# kernel | SGPRs | VGPRs | AGPRs | scratch B/lane | occupancy waves/SIMD | SGPR spills | VGPR spills | LDS B"
make -s -C seabreeze_param_amd/csrc resources 2>&1 | python3 -c '
import re, sys, subprocess
cur, rows = None, []
for ln in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void ", ""))
    print(" | ".join([n] + [r.get(k, "?") for k in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "LDS Size")]))
'
