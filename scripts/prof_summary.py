#!/usr/bin/env python
"""Per-kernel summary (calls, total, average duration) of a rocprofv3 --kernel-trace run, from its results .db (rocpd) or
*_kernel_trace.csv.  Usage: python scripts/prof_summary.py <dir-or-file> [skip_first_n_dispatches_per_kernel]"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = 'select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)
    return [(n, int(a), int(b)) for n, a, b in cur.execute(q)]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    return sorted(out, key=lambda r: r[1])


def main():
    target = sys.argv[1]
    if os.path.isdir(target):
        found = glob.glob(os.path.join(target, '**', '*.db'), recursive=True) + glob.glob(os.path.join(target, '**', '*kernel_trace.csv'), recursive=True)
        target = found[0]
    rows = rows_from_db(target) if target.endswith('.db') else rows_from_csv(target)
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # keep only the last N dispatches of every kernel (the timed steps)
    per = {}
    for name, a, b in rows:
        per.setdefault(name, []).append((b - a) / 1e3)
    print('%-100s %7s %10s %9s' % ('kernel', 'calls', 'total_us', 'avg_us'))
    for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1][-last:] if last else kv[1])):
        ds = ds[-last:] if last else ds
        print('%-100s %7d %10.1f %9.2f' % (name[:100], len(ds), sum(ds), sum(ds) / len(ds)))


if __name__ == '__main__':
    main()
