#!/bin/bash
# round 6: cache policy of the stem's streams now that layer1.0's tail reads BOTH of its inputs (pooled map, t1) from the stem
cd "$GRAFT_REPO_ROOT" || exit 1
FAST="--steps 160 --warmup 10 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
for v in "" nt1024 nt128 nt1152 "" nt1024; do
  if [ -z "$v" ]; then unset PVR_LIB; else export PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_$v.so; fi
  timeout 300 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=${v:-default} value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_chunk'], 'layer1', r['stages']['layer1']['ms'], 'other', r['other_ms_per_chunk'])"
done
