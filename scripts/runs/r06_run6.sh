#!/bin/bash
# round 6, GPU call 6: chain_wave128 with hand-pipelined fragment reads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "layer2_wave_form" > gpurun_out/r06_6_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_6_tests.log
tail -3 gpurun_out/r06_6_tests.log
for v in base k8 x4 r4 base; do
  lib=pvr_habitat_amd/lib/libpvr_hip_$v.so; [ $v = base ] && lib=pvr_habitat_amd/lib/libpvr_hip.so
  for n in 248 256; do
  PVR_LIB=$PWD/$lib timeout 300 python scripts/variant_per_op.py conv5 f16 $n 5 > gpurun_out/r06_6_perop_${v}_$n.txt 2>&1
  echo "$v n=$n: $(grep -E 'chain_wave128' gpurun_out/r06_6_perop_${v}_$n.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_6_perop_${v}_$n.txt)"
  done
done
for n in 248 256; do
PVR_CHAIN_WAVE_L2=0 timeout 300 python scripts/variant_per_op.py conv5 f16 $n 5 > gpurun_out/r06_6_perop_block_$n.txt 2>&1
echo "block n=$n: $(grep -E 'layer2.[123].conv2' gpurun_out/r06_6_perop_block_$n.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_6_perop_block_$n.txt)"
done
