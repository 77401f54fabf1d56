"""Online-eval latency: EmbeddingWrapper-style calls with N=2 frames of 64x64 (src/embeddings.py:441-444)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import EmbeddingNet
for name in ('resnet50', 'moco_aug_uber_345', 'clip_vit'):
    net = EmbeddingNet(name, pretrained=False if name == 'resnet50' else True, max_batch=2)
    fr = torch.from_numpy(synth.frames(1, 2, 64, 64))
    for _ in range(5): net(fr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = net(fr)
    el = (time.perf_counter() - t0) / 200
    d = fr.cuda(); o = torch.empty((2, net.out_size), device='cuda')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): net.embedding.forward_into(d, o)
    torch.cuda.synchronize(); el2 = (time.perf_counter() - t0) / 200
    print('%-20s end-to-end call (H2D + forward + D2H sync) %.3f ms | device-resident forward %.3f ms' % (name, el * 1e3, el2 * 1e3), flush=True)
