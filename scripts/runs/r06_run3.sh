#!/bin/bash
# round 6, GPU call 3: chain_wave128 with tile-less waves skipping the arithmetic; knock-out / prefetch-depth variants (timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "layer2_wave_form" > gpurun_out/r06_3_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_3_tests.log
tail -3 gpurun_out/r06_3_tests.log
for v in base k8 k2 k4 k1 k7 k15 k16 x8 r4 x2r4 base; do
  lib=pvr_habitat_amd/lib/libpvr_hip_$v.so; [ $v = base ] && lib=pvr_habitat_amd/lib/libpvr_hip.so
  PVR_LIB=$PWD/$lib timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_3_perop_$v.txt 2>&1
  echo "$v: $(grep -E 'chain_wave128' gpurun_out/r06_3_perop_$v.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_3_perop_$v.txt)"
done
PVR_CHAIN_WAVE_L2=0 timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_3_perop_block.txt 2>&1
echo "block: $(grep -E 'layer2.[123].conv2' gpurun_out/r06_3_perop_block.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_3_perop_block.txt)"
