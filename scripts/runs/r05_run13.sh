#!/bin/bash
# round 5, call 13: conv3 & downsample as one two-operand launch - tests, bench A/B
mkdir -p gpurun_out/r05_run13
timeout 1500 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "dual or stride2_downsample or downsample_inside or low_latency or fused" 2>&1 | tail -12 > gpurun_out/r05_run13/test.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for w in 1 0 1 0; do
  PVR_DUAL_DS=$w timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('dual $w: value %.0f one_lane %s frac %.4f conv_ms %.3f stages %s parity %s' % (d['value'], d.get('one_lane',{}).get('value') if isinstance(d.get('one_lane'),dict) else d.get('one_lane'), r['frac'], r['conv_ms_per_chunk'], {k:v['ms'] for k,v in r['stages'].items()}, d.get('parity_rel_l2')))
" >> gpurun_out/r05_run13/ab.txt 2>&1
done
PVR_DUAL_DS=1 timeout 300 python bench.py $F --per-op 2>&1 >/dev/null | grep -E "^layer4.0|^layer3.0" > gpurun_out/r05_run13/perop.txt
cat gpurun_out/r05_run13/test.txt gpurun_out/r05_run13/ab.txt gpurun_out/r05_run13/perop.txt
