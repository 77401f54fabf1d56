#!/bin/bash
# round 5, call 20: frame kernel, front conv1 with three half-chunk regions (DMA two ahead) - tests, stamps, bench
mkdir -p gpurun_out/r05_run20
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "frame_bottleneck" 2>&1 | tail -5 > gpurun_out/r05_run20/test.txt
timeout 200 python scripts/bneck_frame_time.py f16 256 2>&1 | grep -E "whole bottleneck|own conv1|^group" | tail -4 > gpurun_out/r05_run20/time.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for i in 1 2; do
  timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f one_lane %s frac %.4f conv_ms %.3f layer3 %.3f' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer3']['ms']))
" >> gpurun_out/r05_run20/ab.txt 2>&1
done
cat gpurun_out/r05_run20/test.txt gpurun_out/r05_run20/time.txt gpurun_out/r05_run20/ab.txt
