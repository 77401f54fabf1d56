#!/bin/bash
# round 5, call 10: compile-time knock-outs of conv_wfrag (EXPERIMENTS build)
mkdir -p gpurun_out/r05_run10
rm -f gpurun_out/r05_run10/ko.txt
for ko in 0 1 2 3 4 7 8 9 10 11 15; do
  echo "KO $ko" >> gpurun_out/r05_run10/ko.txt
  PVR_WFRAG_KO=$ko timeout 120 python scripts/conv_wfrag_time.py f16 256 2>&1 | grep -E "conv2 3x3 512|conv1 2048" >> gpurun_out/r05_run10/ko.txt
done
cat gpurun_out/r05_run10/ko.txt
