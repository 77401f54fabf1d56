#!/bin/bash
# Profile capture (rounds 2-6) on the GPU box: kernel stats (1 and 2 batches in flight), HBM traffic (two separate PMC passes),
# SQ counters of the final kernel set, BC / finetune / ViT-B/16 kernel stats.  Outputs under gpurun_out/$1; summaries are copied into profiles/ by hand afterwards.
set -u
OUT=gpurun_out/${1:-r06_prof}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FAST="--no-cpu-baseline --no-bc --no-vit --no-pcie --no-f16 --no-png --no-uber --no-e2e"
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s1 -- python3 bench.py --steps 8 --warmup 2 --lanes 1 --dump-plan $OUT/plan.json $FAST > $OUT/bench_lanes1.json 2> $OUT/s1.err
rocprofv3 --kernel-trace --stats -d $OUT/stats2 -o s2 -- python3 bench.py --steps 8 --warmup 2 --lanes 2 $FAST > $OUT/bench_lanes2.json 2> $OUT/s2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 scripts/fwd_only.py conv5 8 > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python3 scripts/fwd_only.py conv5 8 > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/sq1 -o q1 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/sq2 -o q2 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/sq3 -o q3 -- python3 scripts/fwd_only.py conv5 6 > $OUT/q3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/bc -o bc -- python3 scripts/bc_only.py 30 > $OUT/bc.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/ft -o ft -- python3 scripts/bc_only.py 20 conv > $OUT/ft.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/vit -o vit -- python3 scripts/vit_one_lane.py clip_b16 6 > $OUT/vit.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/l3 -o l3 -- python3 scripts/fwd_only.py conv3 6 > $OUT/l3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/l4 -o l4 -- python3 scripts/fwd_only.py conv4 6 > $OUT/l4.log 2>&1
# summaries (rocprofv3 7.2 writes rocpd sqlite databases)
P=$OUT/summary; mkdir -p $P
db() { ls $OUT/$1/*/*.db $OUT/$1/*.db 2>/dev/null | head -1; }
python3 scripts/rocpd_summary.py stats $(db stats1) $P/kernel_stats_lanes1.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db stats2) $P/kernel_stats_lanes2.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db bc) $P/bc_kernel_stats.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db ft) $P/finetune_kernel_stats.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db vit) $P/vit_b16_kernel_stats_one_lane.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db l3) $P/l3_parity_plan_kernel_stats.csv > /dev/null
python3 scripts/rocpd_summary.py stats $(db l4) $P/l4_parity_plan_kernel_stats.csv > /dev/null
python3 scripts/rocpd_summary.py pmc $(db fetch) $(db write) $P/pmc_conv_traffic.json $P/pmc_hbm_traffic_per_kernel.txt $OUT/plan.json "${2:-round 6 build}" > /dev/null
python3 scripts/rocpd_summary.py sq $P/sq_counters_conv.txt $(db sq1) $(db sq2) $(db sq3) > /dev/null
cp $OUT/bench_lanes2.json $P/bench_profile_box.json; cp $OUT/vit.log $P/vit_one_lane.log
find $OUT -name "*.db" -size +20M -delete            # the databases stay on the box side of the 64 MiB merge limit; the summaries travel
ls -la $P
