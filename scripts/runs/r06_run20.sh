#!/bin/bash
# round 6: layer3.1-3.5 as one launch (bneck_frame RUN): bit-identity, A/B against one launch per bottleneck, stagger sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "frame_run or frame_bottleneck or default_plan_at_the_bench" > gpurun_out/r06_20_tests.log 2>&1
tail -5 gpurun_out/r06_20_tests.log
FAST="--steps 160 --warmup 10 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
for cfg in "1 0" "0 0" "1 4" "1 8" "1 12" "1 16" "1 0" "0 0"; do
  set -- $cfg
  PVR_FRAME_RUN=$1 PVR_FRAME_RUN_STAGGER=$2 timeout 300 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('run=$1 stagger=$2 value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_chunk'], 'layer3', r['stages']['layer3']['ms'])"
done
