#!/bin/bash
# round 6: bneck_frame RUN start-stagger patterns
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "frame_run or default_plan_at_the_bench" > gpurun_out/r06_21_tests.log 2>&1
tail -3 gpurun_out/r06_21_tests.log
FAST="--steps 160 --warmup 10 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
for cfg in "0 0" "1 0" "1 16" "1 20" "1 24" "1 272" "1 276" "1 528" "1 532" "1 536" "1 784" "1 788" "0 0" "1 16"; do
  set -- $cfg
  PVR_FRAME_RUN=$1 PVR_FRAME_RUN_STAGGER=$2 timeout 300 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('run=$1 stagger=$2 (pat %d units %d) value' % ($2 >> 8, $2 & 255), d['value'], 'one_lane', d['one_lane']['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_chunk'], 'layer3', r['stages']['layer3']['ms'])"
done
