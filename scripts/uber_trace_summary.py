"""rocpd database of a stream_embed run -> memory copies (direction, bytes, duration) and the idle gaps of each compute queue: python uber_trace_summary.py <db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
mc = [n for n in names if n.startswith('rocpd_memory_copy')][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % mc)]
print(mc, cols, file=sys.stderr)
rows = db.execute('select start, end, size, name_id from %s order by start' % mc).fetchall() if 'name_id' in cols else db.execute('select start, end, size from %s order by start' % mc).fetchall()
big = [r for r in rows if r[2] > (1 << 20)]
print('%d copies > 1 MB' % len(big))
for r in big[-24:]:
    print('  copy %8.1f MB  %8.3f ms  %6.1f GB/s  start +%.3f ms' % (r[2] / 1e6, (r[1] - r[0]) / 1e6, r[2] / max(1, r[1] - r[0]), (r[0] - big[0][0]) / 1e6))
kv = 'kernels' if 'kernels' in names else [v for v in names if 'kernel' in v.lower() and 'dispatch' in v.lower()][0]
kc = [r[1] for r in db.execute('pragma table_info(%s)' % kv)]
q = 'stream_id' if 'stream_id' in kc else 'queue_id'
ks = db.execute('select start, end, %s from %s order by start' % (q, kv)).fetchall()
per = collections.defaultdict(list)
for s, e, qq in ks: per[qq].append((s, e))
for qq, v in per.items():
    if len(v) < 200: continue
    busy = sum(e - s for s, e in v)
    gaps = sorted(((v[i + 1][0] - v[i][1]) / 1e6 for i in range(len(v) - 1)), reverse=True)
    print('queue %s: %d kernels, span %.1f ms, busy %.1f ms, largest gaps (ms): %s' % (qq, len(v), (v[-1][1] - v[0][0]) / 1e6, busy / 1e6, [round(g, 2) for g in gaps[:10]]))
