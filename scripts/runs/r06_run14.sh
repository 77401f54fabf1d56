#!/bin/bash
# round 6, GPU call 14: the whole GPU suite + a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_14_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_14_tests.log
tail -8 gpurun_out/r06_14_tests.log
timeout 900 python bench.py > gpurun_out/r06_14_bench.json 2> gpurun_out/r06_14_bench.err
tail -c 600 gpurun_out/r06_14_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_14_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', d['roofline']['frac'], 'bf16', d['bf16']['value'], 'parity', d['parity_rel_l2'])
print('stages', {k: v['ms'] for k, v in d['roofline']['stages'].items()}, 'other', d['roofline']['other_ms_per_chunk'])
u = d['uber5crop']; print('uber f16', u['value'], u['frac_of_mfma_peak'], 'streamed', u['streamed']['value'], 'parity', u.get('parity'), '| bf16', u['bf16_throughput_plan']['value'], u['bf16_throughput_plan']['frac_of_mfma_peak'], u['bf16_throughput_plan'].get('parity'))
print('vit', [(v['value'], v['frac_of_mfma_peak']) for v in d['vit']])
print('pcie', d['pcie_inclusive']['pinned_source'], d['pcie_inclusive']['pageable_source'], 'e2e', d['save_embedded_obs_e2e']['value'], 'png', d['png_source']['value'])
print('bc', d['bc']['value'], d['bc_finetune']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sustained_batch16'], d['cpu_baseline']['thread_sweep_frames_per_s'])
PY
