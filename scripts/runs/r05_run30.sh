#!/bin/bash
# round 5, call 30: the full default bench line with two ranks (both on the one GPU of the box: PVR_ONE_GPU=1, gloo) - duration and shape of the N > 1 line
mkdir -p gpurun_out/r05_run30
( time PVR_BENCH_ONE_GPU=1 timeout 1500 python bench.py --gpus 2 > gpurun_out/r05_run30/bench2.json 2> gpurun_out/r05_run30/bench2.err ) 2> gpurun_out/r05_run30/time.txt
tail -3 gpurun_out/r05_run30/time.txt; tail -1 gpurun_out/r05_run30/bench2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print([k for k in d.keys()])
print('pcie', d.get('pcie_inclusive',{}).get('pinned_source'), 'e2e', (d.get('save_embedded_obs_e2e') or {}).get('value'), (d.get('save_embedded_obs_e2e') or {}).get('n_gpus'))
"; tail -3 gpurun_out/r05_run30/bench2.err
