"""BASELINE config 5 on one GPU: 5-crop multi-layer ResNet50 PVR (moco_aug_uber_345 x FiveCrop = 15 ResNet50 trunks per frame,
31310 floats per frame), host uint8 frames -> H2D -> encode -> D2H fp32, overlapped streams (stream_embed)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import EmbeddingNet, stream_embed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for name, crops in (('moco_aug_uber_345', 5), ('moco_aug_uber_345', 1), ('moco_aug', 5)):
    net = EmbeddingNet(name, crops=crops, max_batch=256)
    fr = torch.from_numpy(synth.frames(5, n, 256, 256)).pin_memory()
    out = torch.empty((n, net.out_size), dtype=torch.float32).pin_memory()
    stream_embed(net, fr[:512], batch=256, out=out[:512])                       # warm-up (allocations, first-use attributes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stream_embed(net, fr, batch=256, out=out)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('%-20s crops=%d: %d floats/frame, %7.0f frames/s (%.2f GB/s of embeddings to the host), %d ResNet50 trunk forwards per frame -> %.0f trunk-frames/s'
          % (name, crops, net.out_size, n / el, n * net.out_size * 4 / el / 1e9, crops * (3 if 'uber' in name else 1), n / el * crops * (3 if 'uber' in name else 1)), flush=True)
    del net
