#!/usr/bin/env python
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (scripts/gpu_profile_mfma.sh) -> profiles/<tag>_mfma_busy.csv: per kernel the
mean busy cycles of the matrix pipe per dispatch (summed over the chip's 1 024 SIMDs: 32 cycles per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md), the dispatch's active cycles (GRBM_GUI_ACTIVE is the sum over the 8 XCDs) and the MFMA utilisation
busy / (active / 8 x 1 024 SIMDs).  Usage: python scripts/summarise_mfma.py <tag>"""
import collections
import csv
import glob
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'scripts'))
from summarise_profile import short  # noqa: E402

N_SIMD = 256 * 4


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r5'
    found = sorted(glob.glob(os.path.join(REPO, 'gpurun_out', 'prof_%s_mfma' % tag, '*', '*counter_collection.csv')), key=os.path.getmtime)
    if not found:
        raise SystemExit('no counter_collection.csv under gpurun_out/prof_%s_mfma' % tag)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(found[-1])):
        per[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    rows = []
    for kern, c in per.items():
        busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES', []), c.get('GRBM_GUI_ACTIVE', [])
        if not busy or not act:
            continue
        b, a = sum(busy) / len(busy), sum(act) / len(act)
        rows.append((kern, len(busy), b, a / 8.0, b / max(a / 8.0 * N_SIMD, 1.0)))
    rows.sort(key=lambda r: -r[2])
    path = os.path.join(REPO, 'profiles', '%s_mfma_busy.csv' % tag)
    with open(path, 'w') as f:
        f.write('kernel,dispatches,mean_SQ_VALU_MFMA_BUSY_CYCLES,mean_GRBM_GUI_ACTIVE_per_XCD,mfma_utilisation\n')
        for r in rows:
            f.write('"%s",%d,%.0f,%.0f,%.4f\n' % r)
    for r in rows[:24]:
        print('%-60s %5d busy %12.0f active/XCD %10.0f util %.3f' % (r[0][:60], r[1], r[2], r[3], r[4]))


if __name__ == '__main__':
    main()
