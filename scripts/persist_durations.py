#!/usr/bin/env python
"""Durations (us) of every dispatch of the persistent recurrent kernels in a rocprofv3 kernel trace, and the longest other kernels:
usage persist_durations.py <trace dir>"""
import collections
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
d = collections.defaultdict(list)
other = collections.defaultdict(list)
for s, e, n in rows:
    if 'lstm_stack' in n or 'gru_fwd_persist' in n or 'gru_bwd_persist' in n or 'gru_stack' in n:
        d[n[:50]].append(round((e - s) / 1e3))
    else:
        other[n[:60]].append((e - s) / 1e3)
for k, v in d.items():
    print(k, v)
print('longest single dispatches of other kernels:')
for k, v in sorted(other.items(), key=lambda kv: -max(kv[1]))[:8]:
    print('  %-60s max %9.1f us  calls %d' % (k, max(v), len(v)))
