#!/bin/bash
# A/B of the bottleneck-chain tuning knob (PVR_CHAIN_CFG = 10*RD + OCC): per-op lines of the fused launches + frames/s
for cfg in "$@"; do
  echo "== PVR_CHAIN_CFG=$cfg"
  PVR_CHAIN_CFG=$cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bc --no-vit --no-pcie --per-op 2>&1 | grep -E "\+conv3|downsample|\"value\"" | sed -E 's/.*"value": ([0-9.]+).*/frames\/s \1/' | cut -c1-120
done
