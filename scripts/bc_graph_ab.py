"""A/B of the BC iteration launch strategies: PVR_POLICY_PERSIST (persistent forward recurrence) x PVR_POLICY_PIPELINE (two-lane
layer pipeline) x PVR_POLICY_GRAPH (hipGraph replay).  5 warm-up + 50 timed steps each, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
arms = [a.split(',') for a in (sys.argv[1:] or ['0,1,0', '1,1,0', '1,0,0', '0,1,0', '1,1,0'])]
for persist, pipe, g in arms:
    os.environ['PVR_POLICY_PERSIST'] = persist
    os.environ['PVR_POLICY_PIPELINE'] = pipe
    os.environ['PVR_POLICY_GRAPH'] = g
    r = bench.bc_bench(50, 5, False)
    print('PERSIST=%s PIPELINE=%s GRAPH=%s  %.1f steps/s  %.3f ms/step  loss %.5f' % (persist, pipe, g, r['value'], r['ms_per_step'], r['final_loss']), flush=True)
