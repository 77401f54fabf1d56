"""BASELINE configs[4] alone (for rocprofv3 --kernel-trace --stats and A/B runs): 5-crop moco_aug_uber_345, frames in pinned host memory,
stream_embed.  python scripts/uber_only.py [dtype] [frames] [mode]; mode 'hbm' = frames resident in HBM, forward_into on two lanes (no H2D / D2H)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import EmbeddingNet, stream_embed, lane_streams
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mode = sys.argv[3] if len(sys.argv) > 3 else 'stream'
t0 = time.perf_counter()
net = EmbeddingNet('moco_aug_uber_345', pretrained=False, crops=5, max_batch=256, compute_dtype=dt)
print('construct %.1f s' % (time.perf_counter() - t0), flush=True)
fr = torch.from_numpy(synth.frames(5, n, 256, 256)).pin_memory()
if mode == 'hbm':
    d = fr.cuda()
    outs = [torch.empty((256, net.out_size), device='cuda') for _ in range(2)]
    streams = lane_streams()
    def run():
        for s_ in streams: s_.wait_stream(torch.cuda.current_stream())
        for i in range(n // 256):
            with torch.cuda.stream(streams[i & 1]):
                net.embedding.forward_into(d[i * 256:(i + 1) * 256], outs[i & 1], lane=i & 1)
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter(); run(); el = time.perf_counter() - t0
else:
    out = torch.empty((n, net.out_size), dtype=torch.float32).pin_memory()
    stream_embed(net, fr[:512], batch=256, out=out[:512])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stream_embed(net, fr, batch=256, out=out)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
fps = n / el
print('%s %s n=%d: %.0f frames/s, %.0f trunk-frames/s, %.1f TFLOP/s = %.4f of 2500' % (mode, dt, n, fps, 15 * fps, fps * 115.69 / 1e3, fps * 115.69 / 1e3 / 2500), flush=True)
