#!/bin/bash
mkdir -p gpurun_out/r05_run25
timeout 300 python scripts/bneck_frame_cold.py > gpurun_out/r05_run25/cold.txt 2>&1
cat gpurun_out/r05_run25/cold.txt
