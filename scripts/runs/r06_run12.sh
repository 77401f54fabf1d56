#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "stem or five_crop or default_plan" > gpurun_out/r06_12_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r06_12_tests.log
tail -3 gpurun_out/r06_12_tests.log
PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_stemstamp.so timeout 300 python scripts/stem_stamps.py > gpurun_out/r06_12_stem_stamps.txt 2>&1
grep -A4 "wave" gpurun_out/r06_12_stem_stamps.txt | head -20
for on in 1 0 1 0; do
  PVR_STEM_REGPOOL=$on timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_12_perop_$on.txt 2>&1
  echo "regpool=$on: $(grep -E '^stem' gpurun_out/r06_12_perop_$on.txt | awk '{print $(NF-3)}') ms | $(grep total gpurun_out/r06_12_perop_$on.txt)"
done
