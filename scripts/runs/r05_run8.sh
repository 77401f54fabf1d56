#!/bin/bash
mkdir -p gpurun_out/r05_run8
F="--no-cpu-baseline --no-bc --no-pcie --no-png --no-e2e --no-vit --no-f16 --no-uber --no-dp"
timeout 300 python bench.py $F --per-op > gpurun_out/r05_run8/perop.json 2> gpurun_out/r05_run8/perop.err
grep -E "^(layer|stem|conv1|avgpool|fc)" gpurun_out/r05_run8/perop.err | head -60
tail -5 gpurun_out/r05_run8/perop.err
