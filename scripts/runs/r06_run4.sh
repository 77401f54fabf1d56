#!/bin/bash
# round 6, GPU call 4: chain_wave128 round quantisation: per-launch time against the number of frames
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for n in 160 192 224 240 248 256 264 288 320; do
  for on in 1 0; do
    PVR_CHAIN_WAVE_L2=$on timeout 300 python scripts/variant_per_op.py conv5 f16 $n 5 > gpurun_out/r06_4_n${n}_$on.txt 2>&1
    echo "n=$n wave=$on: $(grep -E 'layer2.[123].conv2' gpurun_out/r06_4_n${n}_$on.txt | awk '{printf "%s ", $(NF-3)}')"
  done
done
