#!/bin/bash
# round 5, call 18: average pool inside the last convolution - tests, bench A/B
mkdir -p gpurun_out/r05_run18
timeout 1500 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "pooled_epilogue or pool_inside or conv_wfrag or fused_bottleneck_chain or low_latency or golden or oracle" 2>&1 | tail -12 > gpurun_out/r05_run18/test.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for w in 1 0 1 0; do
  PVR_POOL_FUSE=$w timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pool $w: value %.0f one_lane %s frac %.4f conv_ms %.3f other %.3f stages %s parity %s' % (d['value'], d.get('one_lane',{}).get('value') if isinstance(d.get('one_lane'),dict) else d.get('one_lane'), r['frac'], r['conv_ms_per_chunk'], r['other_ms_per_chunk'], {k:v['ms'] for k,v in r['stages'].items()}, d.get('parity_rel_l2')))
" >> gpurun_out/r05_run18/ab.txt 2>&1
done
PVR_POOL_FUSE=1 timeout 300 python bench.py $F --per-op 2>&1 >/dev/null | grep -E "^layer4.2|^pool" > gpurun_out/r05_run18/perop.txt
cat gpurun_out/r05_run18/test.txt gpurun_out/r05_run18/ab.txt gpurun_out/r05_run18/perop.txt
