#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PVR_LIB=$PWD/pvr_habitat_amd/lib/libpvr_hip_stamp.so timeout 300 python scripts/cw8_stamps.py 256 > gpurun_out/r06_7_stamps.txt 2>&1
cat gpurun_out/r06_7_stamps.txt | cut -c1-260
