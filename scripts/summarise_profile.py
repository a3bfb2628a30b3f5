#!/usr/bin/env python
"""Turn the rocprofv3 output of scripts/gpu_profile.sh (gpurun_out/prof_<tag>_{stats,fetch,write}) into the summaries kept
under profiles/: per-kernel time statistics, mean FETCH_SIZE / WRITE_SIZE per dispatch, and the HBM bytes per launch that
bench.py quotes as roofline.traffic (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md section HBM, + WRITE_SIZE).
Usage: python scripts/summarise_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """Kernel name without arguments; template arguments reduced to the first (the tile / variant selector)."""
    base = name.split('(')[0].replace('void ', '').strip()
    m = re.match(r'([^<]+)<([^,>]+)', base)
    return '%s<%s>' % (m.group(1), m.group(2)) if m else base


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
    out = os.path.join(REPO, 'profiles')
    def newest(pattern):                          # gpurun merges every call's files into gpurun_out/: take the latest run
        found = sorted(glob.glob(pattern), key=os.path.getmtime)
        return found[-1:] if found else []

    stats = newest(os.path.join(REPO, 'gpurun_out', 'prof_%s_stats' % tag, '*', '*kernel_stats.csv'))
    if stats:
        shutil.copy(stats[0], os.path.join(out, '%s_bench_c2_bf16_kernel_stats.csv' % tag))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for counter, sub in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
        for path in newest(os.path.join(REPO, 'gpurun_out', 'prof_%s_%s' % (tag, sub), '*', '*counter_collection.csv')):
            for row in csv.DictReader(open(path)):
                if row['Counter_Name'] == counter:
                    per[short(row['Kernel_Name'])][counter].append(float(row['Counter_Value']))
    rows, traffic = [], {}
    for kern, counters in sorted(per.items()):
        total = 0.0
        for counter, values in sorted(counters.items()):
            mean_kb = sum(values) / len(values)
            rows.append((kern, counter, len(values), round(mean_kb, 1)))
            total += mean_kb * 1024.0 * (2.0 if counter == 'FETCH_SIZE' else 1.0)
        traffic[kern] = int(total)
    with open(os.path.join(out, '%s_bench_c2_bf16_pmc_hbm.csv' % tag), 'w') as f:
        f.write('kernel,counter,dispatches,mean_value_KB\n')
        for r in rows:
            f.write('"%s",%s,%d,%s\n' % r)
    with open(os.path.join(out, '%s_hbm_traffic.json' % tag), 'w') as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
