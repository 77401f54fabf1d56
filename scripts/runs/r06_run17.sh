#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "conv1_inside or stem or chain or default_plan or downsample_inside or five_crop or frame_bottleneck_plan or low_latency" > gpurun_out/r06_17_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_17_tests.log
tail -12 gpurun_out/r06_17_tests.log
for on in 1 0 1 0; do
  PVR_STEM_CONV1=$on timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_17_perop_$on.txt 2>&1
  echo "stem_conv1=$on: $(grep -E '^stem|layer1.0.conv1 ' gpurun_out/r06_17_perop_$on.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_17_perop_$on.txt)"
done
