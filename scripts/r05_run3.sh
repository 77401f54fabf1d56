#!/bin/bash
set -u
OUT=gpurun_out/r05_run3; mkdir -p $OUT
timeout 300 python3 scripts/bneck_frame_time.py bf16 256 > $OUT/time.txt 2>&1
cat $OUT/time.txt
