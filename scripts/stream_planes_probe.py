"""Probe: stream_embed on pinned (rows, 64, 64, 6) blocks with planes=2 (what save_embedded_obs feeds it): rate per call, cost of the
host-side finite check, cost per call vs block size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import embeddings as E

net = E.EmbeddingNet('resnet50', pretrained=False)
rng = np.random.default_rng(0)
for rows in (8192, 32768):
    x = torch.from_numpy(rng.integers(0, 256, (rows, 64, 64, 6), dtype=np.uint8)).pin_memory()
    out = torch.empty((rows, 2 * net.out_size), dtype=torch.float32).pin_memory()
    E.stream_embed(net, x[:1024], 256, out=out[:1024], planes=2)
    for label, patch in (('with check', None), ('no check', lambda a, m: a)):
        saved = E._checked
        if patch: E._checked = patch
        t0 = time.perf_counter()
        for _ in range(3):
            E.stream_embed(net, x, 256, out=out, planes=2)
        el = (time.perf_counter() - t0) / 3
        E._checked = saved
        print('rows %6d %-10s: %.3f s per call = %.1f k frames/s' % (rows, label, el, 2 * rows / el / 1e3))
    t0 = time.perf_counter(); ok = np.isfinite(out.numpy()).all(); t1 = time.perf_counter()
    s = torch.from_numpy(out.numpy()).sum(dtype=torch.float64); t2 = time.perf_counter()
    print('   np.isfinite(...).all(): %.1f ms; torch sum f64: %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# one plane, 256x256 frames for reference
x = torch.from_numpy(rng.integers(0, 256, (4096, 256, 256, 3), dtype=np.uint8)).pin_memory()
E.stream_embed(net, x[:1024], 256)
t0 = time.perf_counter(); E.stream_embed(net, x, 256); el = time.perf_counter() - t0
print('256x256 one plane: %.1f k frames/s' % (4096 / el / 1e3))
x = torch.from_numpy(rng.integers(0, 256, (16384, 64, 64, 3), dtype=np.uint8)).pin_memory()
E.stream_embed(net, x[:1024], 256)
t0 = time.perf_counter(); E.stream_embed(net, x, 256); el = time.perf_counter() - t0
print('64x64 one plane: %.1f k frames/s' % (16384 / el / 1e3))
