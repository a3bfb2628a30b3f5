#!/bin/bash
# Register / LDS usage of every kernel of one HIP source: scripts/kernel_regs.sh gemm_bf16_big.hip
cd "$(dirname "$0")/../morgana_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS"):
        print("%-90s vgpr %4s agpr %3s scratch %4s spill %3s occ %s lds %7s" % (cur["name"][:90], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("VGPRs Spill"), cur.get("Occupancy [waves/SIMD]"), v))
'
rm -f /tmp/kr_$$.o
