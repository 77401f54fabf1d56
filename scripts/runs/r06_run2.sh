#!/bin/bash
# round 6, GPU call 2: layer2 wave form (chain_wave128) - bit-identity, then A/B timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "layer2_wave_form or default_plan or chain_wave_equals_block_form" > gpurun_out/r06_2_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_2_tests.log
tail -15 gpurun_out/r06_2_tests.log
for on in 1 0 1 0; do
  PVR_CHAIN_WAVE_L2=$on timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_2_perop_l2wave_$on.txt 2>&1
  grep -E "layer2|total" gpurun_out/r06_2_perop_l2wave_$on.txt
done
for on in 1 0; do
  PVR_CHAIN_WAVE_L2=$on timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-png --no-uber --no-e2e > gpurun_out/r06_2_bench_l2wave_$on.json 2> gpurun_out/r06_2_bench_l2wave_$on.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r06_2_bench_l2wave_$on.json').read().strip().splitlines()[-1])
print('L2WAVE=$on value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', d['roofline']['frac'], 'layer2', d['roofline']['stages']['layer2'], 'bf16', d['bf16']['value'])
PY
done
