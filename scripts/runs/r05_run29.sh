#!/bin/bash
# round 5, call 29: packed epilogue math - bit-identity tests + same-box A/B (library A = before)
mkdir -p gpurun_out/r05_run29; rm -f gpurun_out/r05_run29/ab.txt
timeout 1500 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "chain_wave or fused_bottleneck or downsample_inside or golden or stage" 2>&1 | tail -4 > gpurun_out/r05_run29/test.txt
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
L=$GRAFT_REPO_ROOT/pvr_habitat_amd/lib
for v in A B A B; do
  if [ $v = B ]; then lib=$L/libpvr_hip.so; else lib=$L/libpvr_hip_A.so; fi
  PVR_LIB=$lib timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v: value %.0f one_lane %s frac %.4f conv_ms %.3f layer1 %.3f layer2 %.3f layer3 %.3f layer4 %.3f parity %s' % (d['value'], d['one_lane']['value'], r['frac'], r['conv_ms_per_chunk'], r['stages']['layer1']['ms'], r['stages']['layer2']['ms'], r['stages']['layer3']['ms'], r['stages']['layer4']['ms'], d.get('parity_rel_l2')))
" >> gpurun_out/r05_run29/ab.txt 2>&1
done
cat gpurun_out/r05_run29/test.txt gpurun_out/r05_run29/ab.txt
