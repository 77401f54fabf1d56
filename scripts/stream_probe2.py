"""Probe: what costs stream_embed its 14 % against the HBM-resident rate?  Same 8192 frames (256 x 256): device-resident source (no H2D),
pinned host source, pinned + D2H into a reused pinned result; with HSA_ENABLE_SDMA as the environment sets it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import embeddings as E
net = E.EmbeddingNet('resnet50', pretrained=False)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 256, (8192, 256, 256, 3), dtype=np.uint8))
xp = x.pin_memory(); xd = x.cuda()
out = torch.empty((8192, net.out_size), dtype=torch.float32).pin_memory()
print('HSA_ENABLE_SDMA =', os.environ.get('HSA_ENABLE_SDMA'))
for label, src in (('device source', xd), ('pinned source', xp), ('device source', xd), ('pinned source', xp)):
    E.stream_embed(net, src[:1024], 256, out=out[:1024])
    t0 = time.perf_counter(); E.stream_embed(net, src, 256, out=out); el = time.perf_counter() - t0
    print('%-14s: %.1f k frames/s' % (label, 8192 / el / 1e3))
