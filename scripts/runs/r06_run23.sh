#!/bin/bash
# round 6: does the frame kernel's time depend on the DRAM locality of its x / identity / y streams?  (layout knock-outs: timing only, wrong results)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
FAST="--steps 160 --warmup 10 --no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
for v in "" ko1 ko2 ko6 ko7 "" ko7; do
  if [ -z "$v" ]; then unset PVR_LIB; else export PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_$v.so; fi
  timeout 300 python bench.py $FAST 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=${v:-default} value', d['value'], 'one_lane', d['one_lane']['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_chunk'], 'layer3', r['stages']['layer3']['ms'])"
done
