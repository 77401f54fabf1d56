#!/bin/bash
mkdir -p gpurun_out/r05_run17
O=$GRAFT_REPO_ROOT/gpurun_out/r05_run17
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -o t -- python3 scripts/uber_only.py bf16 1024 stream > $O/tr.log 2>&1
python3 scripts/uber_trace_summary.py $(ls $O/tr/*/*.db $O/tr/*.db 2>/dev/null | head -1) > $O/trace.txt 2>&1
find $O -name "*.db" -delete
tail -3 $O/tr.log | grep stream; cat $O/trace.txt
