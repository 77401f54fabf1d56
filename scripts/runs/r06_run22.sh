#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python scripts/gemm_lib_headroom.py > gpurun_out/r06_22_gemm.txt 2>&1
TORCH_BLAS_PREFER_HIPBLASLT=1 timeout 300 python scripts/gemm_lib_headroom.py >> gpurun_out/r06_22_gemm.txt 2>&1
cat gpurun_out/r06_22_gemm.txt
timeout 300 python scripts/vit_one_lane.py clip_b16 6 2>&1 | tail -5
