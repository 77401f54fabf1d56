#!/bin/bash
# Round-2 profile capture on the GPU box: kernel stats (1 and 2 batches in flight), HBM traffic (two separate PMC passes),
# SQ counters of the final kernel set.  Outputs under gpurun_out/$1; summaries are copied into profiles/ by hand afterwards.
set -u
OUT=gpurun_out/${1:-r02_prof}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FAST="--no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png"
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s1 -- python3 bench.py --steps 8 --warmup 2 --lanes 1 $FAST > $OUT/bench_lanes1.json 2> $OUT/s1.err
rocprofv3 --kernel-trace --stats -d $OUT/stats2 -o s2 -- python3 bench.py --steps 8 --warmup 2 --lanes 2 $FAST > $OUT/bench_lanes2.json 2> $OUT/s2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 scripts/fwd_only.py conv5 8 > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python3 scripts/fwd_only.py conv5 8 > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/sq1 -o q1 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/sq2 -o q2 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/sq3 -o q3 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q3.log 2>&1
ls -R $OUT | head -40
