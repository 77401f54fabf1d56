#!/bin/bash
# round 6, GPU call 1: the refactored dispatch + conv_split16, first contact
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "split16 or parity_plan or default_plan or pool_inside or stem_reading or compressed or frame_bottleneck_plan or low_latency or stride2_downsample or splitk" > gpurun_out/r06_1_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_1_tests.log
for v in conv3 conv4; do
  timeout 300 python scripts/variant_per_op.py $v f16 256 > gpurun_out/r06_1_perop_${v}_split16.txt 2>&1
  PVR_SPLIT16=0 timeout 300 python scripts/variant_per_op.py $v f16 256 > gpurun_out/r06_1_perop_${v}_f32.txt 2>&1
done
tail -5 gpurun_out/r06_1_tests.log
tail -2 gpurun_out/r06_1_perop_*.txt
