"""A/B of the BC iteration launch strategies.  Each arm is a comma list of KEY=VALUE with KEY in WAVEFRONT, PIPELINE, PERSIST, GRAPH
(PVR_POLICY_<KEY>).  5 warm-up + 50 timed steps each, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
arms = sys.argv[1:] or ['WAVEFRONT=0', 'WAVEFRONT=1', 'WAVEFRONT=0', 'WAVEFRONT=1']
for arm in arms:
    for k in ('WAVEFRONT', 'PIPELINE', 'PERSIST', 'GRAPH'):
        os.environ.pop('PVR_POLICY_' + k, None)
    for kv in arm.split(','):
        k, v = kv.split('=')
        os.environ['PVR_POLICY_' + k] = v
    r = bench.bc_bench(50, 5, False)
    print('%-28s %.1f steps/s  %.3f ms/step  loss %.5f' % (arm, r['value'], r['ms_per_step'], r['final_loss']), flush=True)
