#!/bin/bash
# round 6, GPU call 8: chain_wave128 de-phasing sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for st in "1,0" "2,15000" "2,25000" "2,30000" "2,35000" "2,45000" "3,15000" "3,20000" "3,25000" "4,12000" "4,18000" "1,0"; do
  PVR_CW8_STAGGER=$st timeout 300 python scripts/variant_per_op.py conv5 f16 256 5 > gpurun_out/r06_8_perop_$st.txt 2>&1
  echo "stagger $st: $(grep -E 'chain_wave128' gpurun_out/r06_8_perop_$st.txt | awk '{printf "%s ", $(NF-3)}') | $(grep total gpurun_out/r06_8_perop_$st.txt)"
done
PVR_CW8_STAGGER="2,30000" PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_stamp.so timeout 300 python scripts/cw8_stamps.py 256 > gpurun_out/r06_8_stamps.txt 2>&1
cut -c1-200 gpurun_out/r06_8_stamps.txt | head -12
PVR_CW8_STAGGER="2,30000" timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "layer2_wave_form or chain_wave_equals_block_form_at" 2>&1 | tail -2
