#!/usr/bin/env python
"""Timeline of the last N kernel dispatches of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous end.
Usage: python scripts/trace_timeline.py <dir-or-csv> [N]"""
import csv
import glob
import os
import sys

target = sys.argv[1]
if os.path.isdir(target):
    target = glob.glob(os.path.join(target, '**', '*kernel_trace.csv'), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(target))), key=lambda r: r[1])
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, a, b in rows:
    gap = '' if prev_end is None else '%7.2f' % ((a - prev_end) / 1e3)
    print('%9.2f  dur %8.2f  gap %7s  %s' % ((a - t0) / 1e3, (b - a) / 1e3, gap, name[:70]))
    prev_end = b if prev_end is None else max(prev_end, b)
