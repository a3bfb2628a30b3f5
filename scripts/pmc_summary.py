#!/usr/bin/env python
"""Mean counter values per kernel from a rocprofv3 --pmc run directory (counter_collection.csv): usage pmc_summary.py <dir> [substr]"""
import collections
import csv
import glob
import os
import sys


def main():
    root, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            if want in name:
                per[name][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, counters in sorted(per.items()):
        print(name[:100])
        for counter, values in sorted(counters.items()):
            print('    %-32s %16.0f  (%d dispatches)' % (counter, sum(values) / len(values), len(values)))


if __name__ == '__main__':
    main()
