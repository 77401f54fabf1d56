#!/bin/bash
# round 6: bneck_frame64 (one wave per SIMD, 64 couts x 13 pixel tiles per wave): bit-identity, isolated timing + stamps, in-network A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "frame64 or frame_bottleneck or default_plan_at_the_bench" > gpurun_out/r06_25_tests.log 2>&1
tail -5 gpurun_out/r06_25_tests.log
timeout 300 python scripts/frame64_time.py f16 > gpurun_out/r06_25_time.txt 2>&1; cat gpurun_out/r06_25_time.txt | tail -12
FAST="--steps 160 --warmup 10 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
for v in 1 0 1 0; do
  PVR_FRAME64=$v timeout 300 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('frame64=$v value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_chunk'], 'layer3', r['stages']['layer3']['ms'], 'parity', d['parity_rel_l2'])"
done
