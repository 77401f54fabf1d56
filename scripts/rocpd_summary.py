"""Summaries of rocprofv3 (ROCm 7.2 default rocpd / sqlite output) runs, for profiles/:
  rocpd_summary.py stats <results.db> <out.csv>                      per-kernel calls / total / average duration (= --stats)
  rocpd_summary.py pmc <fetch.db> <write.db> <out.json> [<out.txt>]  HBM bytes per launch from two separate --pmc passes
                                                                     (FETCH_SIZE, WRITE_SIZE), gfx950 corrections of
                                                                     /opt/skills/guides/MI355X_MICROARCH.md: KiB units,
                                                                     FETCH_SIZE x2 for wide (16 B/lane) coalesced reads."""
import collections, csv, json, sqlite3, sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc').fetchall()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, int(tot * 1000) if tot < 1e9 else int(tot), round(avg * 1000, 1), round(pct, 3)])
    return rows


def load(db, counter):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, val, dur in c.execute('select name, counter_value, duration from pmc_events where counter_name = ?', (counter,)):
        a = agg[name.split('(')[0]]
        a[0] += 1; a[1] += float(val); a[2] += float(dur)
    return agg


def pmc(fdb, wdb, out_json, out_txt=None):
    f, w = load(fdb, 'FETCH_SIZE'), load(wdb, 'WRITE_SIZE')
    rows = []
    for k in sorted(f, key=lambda k: -f[k][2]):
        n = f[k][0]
        fetch = 2.0 * f[k][1] * 1024 / n
        write = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
        dur = f[k][2] / n * 1e-9
        rows.append(dict(kernel=k[-70:], launches=n, fetch_MB=round(fetch / 1e6, 2), write_MB=round(write / 1e6, 2),
                         avg_us=round(dur * 1e6, 1), hbm_TBps=round((fetch + write) / dur / 1e12, 2)))
    conv = [r for r in rows if any(t in r['kernel'] for t in ('conv_igemm', 'conv_pp256', 'bottleneck_chain'))]
    tot_b = sum((r['fetch_MB'] + r['write_MB']) * r['launches'] for r in conv); tot_n = sum(r['launches'] for r in conv)
    lines = [json.dumps(r) for r in rows[:16]] + ['conv family: avg HBM traffic per launch = %.1f MB over %d launches' % (tot_b / tot_n, tot_n)]
    print('\n'.join(lines))
    if out_txt:
        open(out_txt, 'w').write('\n'.join(lines) + '\n')
    json.dump({'kernel': 'conv_igemm_kernel + conv_pp256_kernel + bottleneck_chain_kernel (all instantiations)', 'launches': tot_n,
               'avg_hbm_bytes_per_launch': tot_b / tot_n * 1e6,
               'correction': 'FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB units, separate --pmc passes', 'per_kernel': rows[:16]},
              open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        for r in stats(sys.argv[2], sys.argv[3])[:12]:
            print(r)
    else:
        pmc(*sys.argv[2:])
