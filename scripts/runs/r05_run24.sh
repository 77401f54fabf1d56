#!/bin/bash
# round 5, call 24: the whole GPU suite, smoke, and the default bench line of the current build
mkdir -p gpurun_out/r05_run24
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05_run24/test.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_run24/smoke.txt 2>&1
timeout 1500 python bench.py > gpurun_out/r05_run24/bench.json 2> gpurun_out/r05_run24/bench.err
cat gpurun_out/r05_run24/test.txt; tail -2 gpurun_out/r05_run24/smoke.txt; tail -1 gpurun_out/r05_run24/bench.json | cut -c1-400
