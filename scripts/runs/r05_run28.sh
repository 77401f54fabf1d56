#!/bin/bash
# round 5, call 28: the driver's GPU test command, twice more on a fresh box (stability record)
mkdir -p gpurun_out/r05_run28
for i in 1 2; do
  timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -1 >> gpurun_out/r05_run28/stability.txt
done
cat gpurun_out/r05_run28/stability.txt
