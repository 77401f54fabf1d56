#!/bin/bash
# round 5, call 19: does de-phasing the frame kernel's workgroups shorten its phases?  (odd workgroups start late)
mkdir -p gpurun_out/r05_run19
for sg in 0 2 4 6 8 12; do
  echo "STAGGER $sg" >> gpurun_out/r05_run19/stagger.txt
  PVR_FRAME_STAGGER=$sg timeout 200 python scripts/bneck_frame_time.py f16 256 2>&1 | grep -E "whole bottleneck|own conv1|^group" | tail -4 >> gpurun_out/r05_run19/stagger.txt
done
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for sg in 0 3 6 0 3 6; do
  PVR_FRAME_STAGGER=$sg timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('stagger $sg: value %.0f one_lane %s frac %.4f conv_ms %.3f layer3 %.3f' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer3']['ms']))
" >> gpurun_out/r05_run19/ab.txt 2>&1
done
cat gpurun_out/r05_run19/stagger.txt gpurun_out/r05_run19/ab.txt
