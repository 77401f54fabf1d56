#!/bin/bash
set -u
OUT=gpurun_out/r05_run4; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "frame_bottleneck" > $OUT/t_frame.log 2>&1; echo "frame tests rc $?" > $OUT/rc.txt
timeout 300 python3 scripts/bneck_frame_time.py bf16 256 > $OUT/time.txt 2>&1
FAST="--no-cpu-baseline --no-bc --no-vit --no-pcie --no-png --no-uber --no-e2e"
PVR_FRAME_BNECK=0 python3 bench.py $FAST --steps 160 > $OUT/b_sep.json 2> $OUT/b_sep.err
PVR_FRAME_FRONT1=1 python3 bench.py $FAST --steps 160 --per-op > $OUT/b_whole.json 2> $OUT/b_whole.err
PVR_FRAME_FRONT1=0 python3 bench.py $FAST --steps 160 > $OUT/b_tail.json 2> $OUT/b_tail.err
PVR_FRAME_FRONT1=1 python3 bench.py $FAST --steps 160 > $OUT/b_whole2.json 2> $OUT/b_whole2.err
cat $OUT/rc.txt; tail -5 $OUT/t_frame.log; grep -v "^group\|^stamps\|amdgpu" $OUT/time.txt
python3 - <<'PY'
import json
for k in ('b_sep','b_whole','b_tail','b_whole2'):
    try:
        d=json.load(open('gpurun_out/r05_run4/%s.json'%k))
        print(k, d['value'], d['bf16']['value'], 'one_lane', d['one_lane']['value'], 'frac', d['roofline']['frac'], 'conv_ms', d['roofline']['conv_ms_per_chunk'], {s:v['ms'] for s,v in d['roofline']['stages'].items()})
    except Exception as e: print(k, 'failed', e)
PY
grep "layer3" $OUT/b_whole.err | head -12
