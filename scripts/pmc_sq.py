import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-60:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        cnt[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(agg, key=lambda k: -dur[k])[:8]:
    a = agg[k]; w = a['SQ_WAVE_CYCLES'] or 1
    print('%-62s n=%3d avg %.0f us | wait_any %.2f wait_inst %.2f active %.2f | mfma_busy/wave_cyc(x4) %.3f | lds_conflict/idx_active %.3f | valu insts %.2e' % (
        k, cnt[k], dur[k] / max(cnt[k], 1) / 1e3, a['SQ_WAIT_ANY'] / w, a['SQ_WAIT_INST_ANY'] / w, a['SQ_ACTIVE_INST_ANY'] / w,
        a['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w), a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1), a['SQ_INSTS_VALU'] / max(cnt[k], 1)))
