#!/bin/bash
# round 5, call 6: chunk-size sweep (does a smaller chunk keep hand-offs in the Infinity Cache?)
mkdir -p gpurun_out/r05_run6
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for cfg in "256 2" "128 2" "128 4" "64 4" "512 2"; do
  set -- $cfg
  timeout 300 python bench.py $F --chunk $1 --batch $1 --lanes $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('chunk $1 lanes $2: value %.0f ms/step %.3f conv_ms_per_chunk %.3f stages %s' % (d['value'], d['ms_per_step'], r['conv_ms_per_chunk'], {k:v['ms'] for k,v in r['stages'].items()}))
" >> gpurun_out/r05_run6/sweep.txt 2>&1
done
cat gpurun_out/r05_run6/sweep.txt
