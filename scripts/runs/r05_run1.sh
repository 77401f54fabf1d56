#!/bin/bash
# round 5, first GPU call: the changed tests + the full default bench line with the per-op table
set -u
OUT=gpurun_out/r05_run1; mkdir -p $OUT
python3 -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu -s > $OUT/t_bench.log 2>&1; echo "bench-contract rc $?" >> $OUT/rc.txt
python3 -m pytest tests/test_gpu_encoder.py -q -m gpu -s -k "bench_batch or preprocess_matches" > $OUT/t_enc.log 2>&1; echo "encoder rc $?" >> $OUT/rc.txt
python3 -m pytest tests/test_gpu_policy.py -q -m gpu -s -k "conv_full or samples or ragged" > $OUT/t_pol.log 2>&1; echo "policy rc $?" >> $OUT/rc.txt
python3 bench.py --per-op > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -3 $OUT/t_bench.log $OUT/t_enc.log $OUT/t_pol.log
