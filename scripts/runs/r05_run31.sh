#!/bin/bash
# round 5, call 31: cache hints of the frame kernel's streams (nt on: 1 identity loads, 2 y stores, 4 the x DMA) - same box, interleaved
mkdir -p gpurun_out/r05_run31; rm -f gpurun_out/r05_run31/ab.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
L=$GRAFT_REPO_ROOT/pvr_habitat_amd/lib
for v in 0 1 2 4 5 0 1 5; do
  if [ $v = 0 ]; then lib=$L/libpvr_hip.so; else lib=$L/libpvr_hip_nt$v.so; fi
  PVR_LIB=$lib timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('nt $v: value %.0f one_lane %s frac %.4f conv_ms %.3f layer3 %.3f layer4 %.3f parity %s' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer3']['ms'], r['stages']['layer4']['ms'], d.get('parity_rel_l2')))
" >> gpurun_out/r05_run31/ab.txt 2>&1
done
cat gpurun_out/r05_run31/ab.txt
