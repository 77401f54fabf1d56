#!/bin/bash
set -u
OUT=gpurun_out/r05_run2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 scripts/variant_rates.py bf16 > $OUT/variants.txt 2>&1
python3 scripts/uber_only.py bf16 2048 stream > $OUT/uber_stream.txt 2>&1
python3 scripts/uber_only.py bf16 2048 hbm > $OUT/uber_hbm.txt 2>&1
python3 scripts/uber_only.py bf16 1024 stream >> $OUT/uber_stream.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/ub -o ub -- python3 scripts/uber_only.py bf16 1024 hbm > $OUT/ub.log 2>&1
python3 scripts/rocpd_summary.py stats $(ls $OUT/ub/*/*.db $OUT/ub/*.db 2>/dev/null | head -1) $OUT/uber_kernel_stats.csv > /dev/null
find $OUT -name "*.db" -size +20M -delete
cat $OUT/variants.txt $OUT/uber_stream.txt $OUT/uber_hbm.txt
