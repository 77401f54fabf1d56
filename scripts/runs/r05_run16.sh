#!/bin/bash
# round 5, call 16: BC iteration timeline (evidence for retiring the >= 250 steps/s target) + default bench line
mkdir -p gpurun_out/r05_run16
O=$GRAFT_REPO_ROOT/gpurun_out/r05_run16
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $O/bct -o b -- python3 scripts/bc_only.py 12 > $O/bct.log 2>&1
python3 scripts/bc_timeline.py $(ls $O/bct/*/*.db $O/bct/*.db 2>/dev/null | head -1) > $O/bc_timeline.txt 2>$O/bc_timeline.err
find $O -name "*.db" -delete
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bc_timeline.txt; tail -1 $O/bench.json | cut -c1-600
