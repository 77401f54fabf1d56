"""Profiling aid: N BC iterations (T=100, B=16, obs 4096, BN) - run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(bench.bc_bench(int(sys.argv[1]) if len(sys.argv) > 1 else 10, 3, False))
