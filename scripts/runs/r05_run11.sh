#!/bin/bash
mkdir -p gpurun_out/r05_run11
timeout 300 python scripts/conv_cold_time.py f16 > gpurun_out/r05_run11/cold.txt 2>&1
cat gpurun_out/r05_run11/cold.txt
