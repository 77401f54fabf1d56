#!/bin/bash
# round 6, GPU call 15: compressed-PVR parity plan without casts, head as one launch; range check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_glue.py -m gpu -x -q -k "split16 or parity_plan or compressed or uber or f16_activation_range or five_crop or glue or splitk or variant" > gpurun_out/r06_15_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_15_tests.log
tail -12 gpurun_out/r06_15_tests.log
for v in conv3 conv4; do
  timeout 300 python scripts/variant_per_op.py $v f16 256 > gpurun_out/r06_15_perop_${v}.txt 2>&1
  grep -E "total|pair|in32" gpurun_out/r06_15_perop_${v}.txt | head -20
done
