#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_glue.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r06_16_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_16_tests.log
tail -12 gpurun_out/r06_16_tests.log
