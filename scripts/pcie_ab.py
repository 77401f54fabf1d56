"""A/B of the PCIe-inclusive streaming path with one or two compute lanes (PVR_STREAM_LANES)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pvr_habitat_amd import synth
sd = synth.resnet50_state_dict(1, 'conv5')
for lanes in ('1', '2', '1', '2'):
    os.environ['PVR_STREAM_LANES'] = lanes
    r = bench.pcie_bench(sd, 256, 256, 'bf16')
    print('PVR_STREAM_LANES=%s pageable %.0f frames/s, pinned %.0f frames/s' % (lanes, r['pageable_source']['value'], r['pinned_source']['value']), flush=True)
