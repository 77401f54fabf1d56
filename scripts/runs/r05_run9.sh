#!/bin/bash
# round 5, call 9: conv_wfrag - op-level parity + isolated timing on layer4's shapes
mkdir -p gpurun_out/r05_run9
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "conv_wfrag" 2>&1 | tail -15 > gpurun_out/r05_run9/test.txt
timeout 300 python scripts/conv_wfrag_time.py f16 256 > gpurun_out/r05_run9/time.txt 2>&1
cat gpurun_out/r05_run9/test.txt gpurun_out/r05_run9/time.txt
