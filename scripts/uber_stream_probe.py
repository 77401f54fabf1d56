"""where the 5-crop uber leg loses against its HBM-resident rate: stream_embed from pinned host frames / from frames already in HBM / bare two-lane loop"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import EmbeddingNet, stream_embed, lane_streams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
net = EmbeddingNet('moco_aug_uber_345', pretrained=False, crops=5, max_batch=256, compute_dtype='bf16')
fr = torch.from_numpy(synth.frames(5, n, 256, 256)).pin_memory()
d = fr.cuda()
out = torch.empty((n, net.out_size), dtype=torch.float32).pin_memory()


def t(fn, reps=2):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best


outs = [torch.empty((256, net.out_size), device='cuda') for _ in range(2)]
streams = lane_streams()


def bare():
    for s_ in streams: s_.wait_stream(torch.cuda.current_stream())
    for i in range(n // 256):
        with torch.cuda.stream(streams[i & 1]):
            net.embedding.forward_into(d[i * 256:(i + 1) * 256], outs[i & 1], lane=i & 1)


print('bare two-lane loop, frames and results in HBM: %.0f frames/s' % t(bare), flush=True)
print('stream_embed, frames in HBM (no upload), results to pinned host memory: %.0f' % t(lambda: stream_embed(net, d, batch=256, out=out)), flush=True)
print('stream_embed, pinned host frames: %.0f' % t(lambda: stream_embed(net, fr, batch=256, out=out)), flush=True)
for dep in (2, 3, 4, 6):
    print('  depth %d: %.0f' % (dep, t(lambda: stream_embed(net, fr, batch=256, out=out, depth=dep))), flush=True)
os.environ['PVR_STREAM_LANES'] = '1'
print('stream_embed, pinned host frames, one lane: %.0f' % t(lambda: stream_embed(net, fr, batch=256, out=out)), flush=True)
