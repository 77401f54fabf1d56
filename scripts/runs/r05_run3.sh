#!/bin/bash
set -u
OUT=gpurun_out/r05_run3; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "frame_bottleneck_op" > $OUT/t_op.log 2>&1; echo "op test rc $?" > $OUT/rc.txt
timeout 300 python3 scripts/bneck_frame_time.py bf16 256 > $OUT/time.txt 2>&1
cat $OUT/rc.txt; tail -8 $OUT/t_op.log; cat $OUT/time.txt
