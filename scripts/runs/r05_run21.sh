#!/bin/bash
mkdir -p gpurun_out/r05_run21
timeout 200 python scripts/bneck_frame_time.py f16 256 2>&1 | grep -E "whole bottleneck|own conv1|^group" | tail -4 > gpurun_out/r05_run21/time.txt
cat gpurun_out/r05_run21/time.txt
