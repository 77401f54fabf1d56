#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "stem or five_crop or default_plan" > gpurun_out/r06_11_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_11_tests.log
tail -4 gpurun_out/r06_11_tests.log
PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_stemstamp.so timeout 300 python scripts/stem_stamps.py > gpurun_out/r06_11_stem_stamps.txt 2>&1
cat gpurun_out/r06_11_stem_stamps.txt
