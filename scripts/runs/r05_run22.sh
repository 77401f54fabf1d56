#!/bin/bash
# round 5, call 22: frame kernel A/B on one box - A: requests issued together (HEAD), B1: one request per step, B2: + identity loads inside the K loop
mkdir -p gpurun_out/r05_run22
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
L=$GRAFT_REPO_ROOT/pvr_habitat_amd/lib
for v in A B1 B2 A B1 B2; do
  if [ $v = B2 ]; then lib=$L/libpvr_hip.so; else lib=$L/libpvr_hip_$v.so; fi
  PVR_LIB=$lib timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v: value %.0f one_lane %s frac %.4f conv_ms %.3f layer3 %.3f' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer3']['ms']))
" >> gpurun_out/r05_run22/ab.txt 2>&1
done
cat gpurun_out/r05_run22/ab.txt
