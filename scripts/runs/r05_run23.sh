#!/bin/bash
mkdir -p gpurun_out/r05_run23; rm -f gpurun_out/r05_run23/ab.txt
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "frame_bottleneck or pool_inside or stride2" 2>&1 | tail -5 > gpurun_out/r05_run23/test.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
run() { PVR_FRAME_MULTI=$1 PVR_FRAME_STAGGER=$2 PVR_FRAME_INV=$3 timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('multi $1 stagger $2 inv $3: value %.0f one_lane %s frac %.4f conv_ms %.3f layer3 %.3f parity %s' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer3']['ms'], d.get('parity_rel_l2')))
" >> gpurun_out/r05_run23/ab.txt 2>&1; }
run 0 0 0; run 1 0 0; run 1 0 1; run 1 10 0; run 1 16 0; run 0 0 0; run 1 10 0
cat gpurun_out/r05_run23/test.txt gpurun_out/r05_run23/ab.txt
