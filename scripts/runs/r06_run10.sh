#!/bin/bash
# round 6, GPU call 10: the stem with the max pool in registers
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "stem or five_crop or default_plan" > gpurun_out/r06_10_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_10_tests.log
tail -12 gpurun_out/r06_10_tests.log
for on in 1 0 1 0; do
  PVR_STEM_REGPOOL=$on timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_10_perop_$on.txt 2>&1
  echo "regpool=$on: $(grep -E '^stem' gpurun_out/r06_10_perop_$on.txt | awk '{print $(NF-3)}') ms | $(grep total gpurun_out/r06_10_perop_$on.txt)"
done
for on in 1 0; do
  PVR_STEM_REGPOOL=$on timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-png --no-uber --no-e2e > gpurun_out/r06_10_bench_$on.json 2> gpurun_out/r06_10_bench_$on.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r06_10_bench_$on.json').read().strip().splitlines()[-1])
print('REGPOOL=$on value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', d['roofline']['frac'], 'other_ms', d['roofline']['other_ms_per_chunk'], 'bf16', d['bf16']['value'])
PY
done
