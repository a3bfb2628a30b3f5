"""Per-step timeline from a rocprofv3 kernel trace of scripts/train_epoch_trace.py: the kernels between two consecutive adam_plan launches
of the last epoch, with start offsets, durations and the idle gaps between them.  usage: python scripts/train_epoch_gaps.py <trace db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute('select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)))
adam = [i for i, r in enumerate(rows) if 'adam_plan' in r[0]]
a, b = adam[-4], adam[-2]
t0 = rows[a][2]
prev_end = t0
for name, s, e in rows[a + 1:b + 1]:
    print('%8.1f us  +%6.1f gap  %6.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:70]))
    prev_end = max(prev_end, e)
print('two steps: %.1f us' % ((rows[b][2] - t0) / 1e3))

# every step of the last epoch: end of one update launch to the end of the next (n = argv[2] steps)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
last = adam[-n:]
spans = [(rows[last[i]][2] - rows[last[i - 1]][2]) / 1e3 for i in range(1, len(last))]
print('steps of the last epoch, us (update end to update end): ' + ' '.join('%.0f' % v for v in spans))
print('epoch on the device: %.1f us for %d steps = %.1f us per step; median step %.1f us' % (
    (rows[last[-1]][2] - rows[last[0]][2]) / 1e3, len(last) - 1, (rows[last[-1]][2] - rows[last[0]][2]) / 1e3 / (len(last) - 1),
    sorted(spans)[len(spans) // 2]))
