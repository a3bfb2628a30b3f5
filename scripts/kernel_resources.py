"""Per-kernel register / scratch / LDS use of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), as a table.
usage: python scripts/kernel_resources.py morgana_amd/csrc/gemm_bf16_big.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|VGPR Spill|Occupancy \[waves/SIMD\]): (.*?) \[-Rpass', line)
    if not m:
        continue
    key, val = m.group(1), m.group(2)
    if key == 'Function Name':
        cur = subprocess.run(['c++filt', val], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur)
        rows[cur] = {}
    elif cur:
        rows[cur][key.split(' [')[0]] = val
for name, r in rows.items():
    if flt in name:
        print('%-70s vgpr %4s agpr %4s scratch %5s lds %7s occ %s' % (name[:70], r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'),
                                                                 r.get('LDS Size'), r.get('Occupancy')))
