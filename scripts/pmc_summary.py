"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel, with the gfx950 corrections
of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE reports 1/2 of the bytes of
wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json [plan.json from bench.py --dump-plan [label]]]"""
import csv, sys, collections, json

def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != name:
            continue
        k = r['Kernel_Name'].split('(')[0]
        a = agg[k]
        a[0] += 1; a[1] += float(r['Counter_Value']); a[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg

f = load(sys.argv[1], 'FETCH_SIZE'); w = load(sys.argv[2], 'WRITE_SIZE')
rows = []
for k in sorted(f, key=lambda k: -f[k][2]):
    n = f[k][0]
    fetch = 2.0 * f[k][1] * 1024 / n                  # corrected bytes per launch
    write = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
    dur = f[k][2] / n * 1e-9
    rows.append(dict(kernel=k[-70:], launches=n, fetch_MB=round(fetch / 1e6, 2), write_MB=round(write / 1e6, 2),
                     avg_us=round(dur * 1e6, 1), hbm_TBps=round((fetch + write) / dur / 1e12, 2)))
for r in rows[:14]:
    print(json.dumps(r))
conv = [r for r in rows if any(t in r['kernel'] for t in ('conv_igemm', 'conv_pp256', 'bottleneck_chain', 'chain_wave', 'conv_expand'))]
tot_b = sum((r['fetch_MB'] + r['write_MB']) * r['launches'] for r in conv); tot_n = sum(r['launches'] for r in conv)
print('conv family: avg HBM traffic per launch = %.1f MB over %d launches' % (tot_b / tot_n, tot_n))
if len(sys.argv) > 3:
    out = {'kernel': 'conv_igemm_kernel + conv_pp256_kernel + bottleneck_chain_kernel + chain_wave_kernel + conv_expand_kernel (all instantiations)', 'launches': tot_n,
           'avg_hbm_bytes_per_launch': tot_b / tot_n * 1e6,
           'correction': 'FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB units, separate --pmc passes', 'per_kernel': rows[:16]}
    if len(sys.argv) > 4:                                  # the launch plan these counters belong to (bench.py refuses them for any other plan)
        out['plan_launches'] = json.load(open(sys.argv[4]))['plan_launches']
    if len(sys.argv) > 5:
        out['captured'] = sys.argv[5]
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
