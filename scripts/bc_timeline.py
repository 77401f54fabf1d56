"""Timeline of one BC iteration from a rocprofv3 --kernel-trace rocpd database: per kernel name the summed duration inside the window, the
union of busy time (any kernel running), the overlap (sum - union) and the idle gaps.  Usage: bc_timeline.py <results.db> [iteration]"""
import sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
kv = 'kernels' if 'kernels' in views else [v for v in views if 'kernel' in v.lower()][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % kv)]
print('view', kv, cols, file=sys.stderr)
nm = 'name' if 'name' in cols else 'kernel_name'
rows = db.execute('select %s, start, end, %s from %s order by start' % (nm, 'stream_id' if 'stream_id' in cols else 'queue_id', kv)).fetchall()
# iterations are delimited by rmsprop_kernel launches
marks = [i for i, r in enumerate(rows) if 'rmsprop_kernel' in r[0]]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 3
lo, hi = marks[it] + 1, marks[it + 1] + 1
win = rows[lo:hi]
t0, t1 = win[0][1], max(r[2] for r in win)
print('iteration %d: %d launches, wall %.3f ms' % (it, len(win), (t1 - t0) / 1e6))
per = collections.OrderedDict()
for n, s, e, q in win:
    k = n.split('(')[0][-48:]
    a = per.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
ev = sorted([(s, 1) for _, s, e, _ in win] + [(e, -1) for _, s, e, _ in win])
busy = 0.0; depth = 0; last = None; gaps = []; two = 0.0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: two += t - last
    if depth == 0 and last is not None and t - last > 0: gaps.append((t - last) / 1e3)
    depth += d; last = t
print('sum of kernel durations %.3f ms, busy (union) %.3f ms, >=2 kernels at once %.3f ms, idle %.3f ms in %d gaps (largest %s us)' % (
    sum(v[1] for v in per.values()) / 1e3, busy / 1e6, two / 1e6, sum(gaps) / 1e3, len(gaps), [round(g, 1) for g in sorted(gaps)[-5:]]))
for k, (c, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %-50s %5d launches %9.1f us' % (k, c, us))
streams = collections.Counter(q for _, _, _, q in win)
print('streams / queues:', dict(streams))
