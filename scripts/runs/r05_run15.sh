#!/bin/bash
# round 5, call 15: the whole GPU suite + smoke with conv_wfrag and the two-operand launch in the plan
mkdir -p gpurun_out/r05_run15
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_run15/test.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_run15/smoke.txt 2>&1
cat gpurun_out/r05_run15/test.txt; tail -3 gpurun_out/r05_run15/smoke.txt
