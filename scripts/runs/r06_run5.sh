#!/bin/bash
# round 6, GPU call 5: SQ counters of the layer2 tails, wave form against block form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06_5; mkdir -p $OUT
for on in 1 0; do
  export PVR_CHAIN_WAVE_L2=$on
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq1_$on -o q -- python3 scripts/fwd_only.py conv5 4 > $OUT/q1_$on.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/sq2_$on -o q -- python3 scripts/fwd_only.py conv5 4 > $OUT/q2_$on.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq3_$on -o q -- python3 scripts/fwd_only.py conv5 4 > $OUT/q3_$on.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq4_$on -o q -- python3 scripts/fwd_only.py conv5 4 > $OUT/q4_$on.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get('OUT', 'gpurun_out/r06_5')
for on in ('1', '0'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('gpurun_out/r06_5/sq*_%s/**/*counter_collection.csv' % on, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-70:]
            if 'chain' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k in sorted(agg):
        a = agg[k]
        line = ' '.join('%s=%.3g' % (c, a[c] / max(cnt[(k, c)], 1)) for c in sorted(a))
        print('L2WAVE=%s %s\n    %s' % (on, k, line))
PY
