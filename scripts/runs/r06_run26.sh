#!/bin/bash
# round 6 (final build): full GPU suite, then the profile capture and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_26_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_26_tests.log
tail -4 gpurun_out/r06_26_tests.log
bash scripts/capture_profiles.sh r06_prof "round 6 build" > gpurun_out/r06_26_capture.log 2>&1
tail -15 gpurun_out/r06_26_capture.log
timeout 900 python bench.py > gpurun_out/r06_26_bench.json 2> gpurun_out/r06_26_bench.err
tail -c 300 gpurun_out/r06_26_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_26_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', d['roofline']['frac'], 'bf16', d['bf16']['value'], 'parity', d['parity_rel_l2'], 'traffic', d['roofline']['traffic'])
print('stages', {k: v['ms'] for k, v in d['roofline']['stages'].items()}, 'other', d['roofline']['other_ms_per_chunk'], 'conv_ms', d['roofline']['conv_ms_per_chunk'])
u = d['uber5crop']; print('uber f16', u['value'], u['frac_of_mfma_peak'], 'streamed', u['streamed']['value'], 'parity', u.get('parity', {}).get('rel_l2'), '| bf16', u['bf16_throughput_plan']['value'], u['bf16_throughput_plan']['frac_of_mfma_peak'])
print('vit', [(v['value'], v['frac_of_mfma_peak']) for v in d['vit']])
print('pcie', d['pcie_inclusive']['pinned_source']['value'], d['pcie_inclusive']['pageable_source']['value'], 'e2e', d['save_embedded_obs_e2e']['value'], 'png', d['png_source']['value'])
print('bc', d['bc']['value'], d['bc_finetune']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sustained_batch16']['value'])
PY
