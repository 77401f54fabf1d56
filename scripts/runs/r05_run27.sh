#!/bin/bash
mkdir -p gpurun_out/r05_run27
timeout 300 python scripts/layer2_0_alt.py > gpurun_out/r05_run27/alt.txt 2>&1
cat gpurun_out/r05_run27/alt.txt
