#!/bin/bash
# round 6, GPU call 9: chain_wave128, quartets half a round apart
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "layer2_wave_form" > gpurun_out/r06_9_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_9_tests.log
tail -4 gpurun_out/r06_9_tests.log
for v in base rd1 base; do
  lib=pvr_habitat_amd/lib/libpvr_hip_$v.so; [ $v = base ] && lib=pvr_habitat_amd/lib/libpvr_hip.so
  for n in 248 256; do
  PVR_LIB=$PWD/$lib timeout 300 python scripts/variant_per_op.py conv5 f16 $n 5 > gpurun_out/r06_9_perop_${v}_$n.txt 2>&1
  echo "$v n=$n: $(grep -E 'chain_wave128' gpurun_out/r06_9_perop_${v}_$n.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_9_perop_${v}_$n.txt)"
  done
done
PVR_CHAIN_WAVE_L2=0 timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_9_perop_block.txt 2>&1
echo "block: $(grep -E 'layer2.[123].conv2' gpurun_out/r06_9_perop_block.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_9_perop_block.txt)"
PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_stamp.so timeout 300 python scripts/cw8_stamps.py 256 > gpurun_out/r06_9_stamps.txt 2>&1
cut -c1-230 gpurun_out/r06_9_stamps.txt
