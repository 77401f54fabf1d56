#!/bin/bash
# round 5, call 26: lanes / chunk sweep with the final kernels
mkdir -p gpurun_out/r05_run26
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
for cfg in "256 2" "256 3" "512 2" "384 2" "256 2"; do
  set -- $cfg
  timeout 300 python bench.py $F --chunk $1 --batch $1 --lanes $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('chunk $1 lanes $2: value %.0f ms/step %.3f frac %.4f' % (d['value'], d['ms_per_step'], r['frac']))
" >> gpurun_out/r05_run26/sweep.txt 2>&1
done
cat gpurun_out/r05_run26/sweep.txt
