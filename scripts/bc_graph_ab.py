"""A/B of eager launches vs hipGraph replay for the BC iteration (PVR_POLICY_GRAPH), 5 warm-up + 50 timed steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for g in ('0', '1', '0', '1'):
    os.environ['PVR_POLICY_GRAPH'] = g
    r = bench.bc_bench(50, 5, False)
    print('PVR_POLICY_GRAPH=%s  %.1f steps/s  %.3f ms/step' % (g, r['value'], r['ms_per_step']), flush=True)
