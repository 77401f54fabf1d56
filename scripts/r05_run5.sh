#!/bin/bash
# round 5, call 5: per-tile fused layer2 bottleneck - op-level parity + isolated timing
mkdir -p gpurun_out/r05_run5
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "tile_bottleneck" 2>&1 | tail -15 > gpurun_out/r05_run5/test.txt
timeout 300 python scripts/bneck_tile_time.py f16 256 > gpurun_out/r05_run5/time.txt 2>&1
timeout 300 python scripts/bneck_tile_time.py bf16 256 >> gpurun_out/r05_run5/time.txt 2>&1
cat gpurun_out/r05_run5/test.txt gpurun_out/r05_run5/time.txt
