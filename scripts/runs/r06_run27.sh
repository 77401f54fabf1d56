#!/bin/bash
# round 6: spread of the uber5crop leg (median of three passes) over three processes on one box
cd "$GRAFT_REPO_ROOT" || exit 1
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-e2e"
for i in 1 2 3; do
  timeout 600 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['uber5crop']
print('run $i: uber f16', u['value'], u['frac_of_mfma_peak'], 'streamed', u['streamed']['value'], '| bf16', u['bf16_throughput_plan']['value'], 'streamed', u['bf16_throughput_plan']['streamed']['value'], '| value', d['value'])"
done
