#!/bin/bash
mkdir -p gpurun_out/r05_run14
timeout 300 python scripts/conv_dual_probe.py > gpurun_out/r05_run14/probe.txt 2>&1
PVR_PP_PERSIST=0 timeout 300 python scripts/conv_dual_probe.py >> gpurun_out/r05_run14/probe.txt 2>&1
cat gpurun_out/r05_run14/probe.txt
