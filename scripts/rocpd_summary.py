"""Summaries of rocprofv3 (ROCm 7.2 default rocpd / sqlite output) runs, for profiles/:
  rocpd_summary.py stats <results.db> <out.csv>                      per-kernel calls / total / average duration (= --stats)
  rocpd_summary.py sq <out.txt> <pass1.db> [<pass2.db> ...]          SQ counters per kernel from separate --pmc passes: MFMA-busy,
                                                                     wait / active fractions, LDS conflicts, instruction mix
  rocpd_summary.py pmc <fetch.db> <write.db> <out.json> [<out.txt> [<plan.json> [<label>]]]  HBM bytes per launch from two separate --pmc passes
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


def pmc(fdb, wdb, out_json, out_txt=None, plan_json=None, label=None):
    f, w = load(fdb, 'FETCH_SIZE'), load(wdb, 'WRITE_SIZE')
    rows = []
    for k in sorted(f, key=lambda k: -f[k][2]):
        n = f[k][0]
        fetch = 2.0 * f[k][1] * 1024 / n
        write = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
        dur = f[k][2] / n * 1e-9
        rows.append(dict(kernel=k[-70:], launches=n, fetch_MB=round(fetch / 1e6, 2), write_MB=round(write / 1e6, 2),
                         avg_us=round(dur * 1e6, 1), hbm_TBps=round((fetch + write) / dur / 1e12, 2)))
    conv = [r for r in rows if any(t in r['kernel'] for t in ('conv_igemm', 'conv_pp256', 'bottleneck_chain', 'chain_wave', 'conv_expand', 'bneck_frame', 'conv_wfrag'))]
    tot_b = sum((r['fetch_MB'] + r['write_MB']) * r['launches'] for r in conv); tot_n = sum(r['launches'] for r in conv)
    lines = [json.dumps(r) for r in rows[:16]] + ['conv family: avg HBM traffic per launch = %.1f MB over %d launches' % (tot_b / tot_n, tot_n)]
    print('\n'.join(lines))
    if out_txt:
        open(out_txt, 'w').write('\n'.join(lines) + '\n')
    out = {'kernel': 'conv_igemm_kernel + conv_pp256_kernel + bottleneck_chain_kernel + chain_wave_kernel + conv_expand_kernel + bneck_frame_kernel + conv_wfrag_kernel (all instantiations)', 'launches': tot_n,
           'avg_hbm_bytes_per_launch': tot_b / tot_n * 1e6,
           'correction': 'FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB units, separate --pmc passes', 'per_kernel': rows[:16]}
    if plan_json:                                           # the launch plan these counters belong to (bench.py --dump-plan): bench.py withholds them for any other plan
        out['plan_launches'] = json.load(open(plan_json))['plan_launches']
    if label:
        out['captured'] = label
    json.dump(out, open(out_json, 'w'), indent=1)


def sq(out_txt, *dbs):
    """per kernel: dispatch-summed SQ counters (rocprofv3 reports one row per dispatch and XCC/SE instance: summed here)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur = collections.Counter(), collections.Counter()
    for db in dbs:
        c = sqlite3.connect(db)
        seen = set()
        for name, disp, cname, val, d in c.execute('select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
            k = name.split('(')[0].replace('void pvr::', '')
            agg[k][cname] += float(val)
            if db == dbs[0] and disp not in seen:
                seen.add(disp); cnt[k] += 1; dur[k] += d
    lines = ['# SQ counters per kernel, summed over dispatches (rocprofv3 --pmc, %d separate passes); ResNet50 batch 256 bf16, one batch in flight' % len(dbs),
             '# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.1 GHz x 1024 SIMDs): the counter sums the matrix-pipe busy cycles of every SIMD',
             '#   (calibration: conv_pp256<256> reads 0.34 here and delivers 0.32 of the dense bf16 peak by its FLOP count; 2.1 GHz = the clock measured under this load, DESIGN 4.1b)',
             '# mfma/wave = the same counter per SQ_WAVE_CYCLES; wait_any / wait_inst / active = fractions of wave cycles; lds_confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE',
             '%-58s %5s %8s | %9s %9s | %8s %8s %8s | %9s | %10s %10s' % ('kernel', 'n', 'avg us', 'mfma_util', 'mfma/wave', 'wait_any', 'wait_ins', 'active', 'lds_confl', 'valu insts', 'mfma insts')]
    for k in sorted(agg, key=lambda k: -dur[k]):
        a = agg[k]; w = a.get('SQ_WAVE_CYCLES', 0) or 1.0; b = a.get('SQ_BUSY_CYCLES', 0) or 1.0
        if cnt[k] == 0 or dur[k] / cnt[k] < 3000:
            continue
        lines.append('%-58s %5d %8.1f | %9.3f %9.4f | %8.2f %8.2f %8.2f | %9.3f | %10.3g %10.3g' % (
            k[-58:], cnt[k], dur[k] / cnt[k] / 1e3, a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (dur[k] * 2.1 * 1024), a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w,
            a.get('SQ_WAIT_ANY', 0) / w, a.get('SQ_WAIT_INST_ANY', 0) / w, a.get('SQ_ACTIVE_INST_ANY', 0) / w,
            a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 0), 1.0),
            a.get('SQ_INSTS_VALU', 0) / max(cnt[k], 1), a.get('SQ_INSTS_MFMA', 0) / max(cnt[k], 1)))
    print('\n'.join(lines))
    open(out_txt, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    if sys.argv[1] == 'sq':
        sq(sys.argv[2], *sys.argv[3:])
    elif sys.argv[1] == 'stats':
        for r in stats(sys.argv[2], sys.argv[3])[:12]:
            print(r)
    else:
        pmc(*sys.argv[2:])
