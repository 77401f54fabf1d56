"""Profiling aid: N forwards of a ViT encoder at batch 256 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

variant = sys.argv[1] if len(sys.argv) > 1 else 'clip_b16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print(bench.vit_bench(variant, 256, steps, 2, 'bf16'))
