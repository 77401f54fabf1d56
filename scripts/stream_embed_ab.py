"""stream_embed from a pageable source: staging variants (frames/s over 32 batches of 256 frames)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50, stream_embed
class Net: pass
net = Net(); net.embedding = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype='bf16', max_batch=256); net.out_size = 2048
fr = torch.from_numpy(synth.frames(1, 2048, 256, 256)).repeat(4, 1, 1, 1)
stream_embed(net, fr[:1024], 256)
os.environ['PVR_STREAM_REGISTER'] = '0'
un = torch.from_numpy(np.frombuffer(bytearray(fr.numel() + 64), dtype=np.uint8)[17:17 + fr.numel()].reshape(fr.shape)); un.copy_(fr)
for label, kw, src in (('pageable, registered in place', dict(reg=1), fr), ('pageable UNALIGNED base, registered in place', dict(reg=1), un), ('pageable, stage_threads=4', dict(stage_threads=4), fr), ('pageable, stage_threads=1', dict(stage_threads=1), fr),
                       ('pageable, direct H2D (no staging)', dict(stage_threads=0), fr), ('pinned', {}, fr.pin_memory())):
    os.environ['PVR_STREAM_REGISTER'] = str(kw.pop('reg', 0))
    stream_embed(net, src[:1024], 256, **kw)
    t0 = time.perf_counter(); stream_embed(net, src, 256, **kw); el = time.perf_counter() - t0
    out = torch.empty((fr.shape[0], 2048), dtype=torch.float32).pin_memory()
    t0 = time.perf_counter(); stream_embed(net, src, 256, out=out, **kw); el2 = time.perf_counter() - t0
    print('%-40s %.0f frames/s (%.1f GB/s); with a caller-owned pinned result buffer %.0f frames/s' % (label, fr.shape[0] / el, fr.numel() / el / 1e9, fr.shape[0] / el2), flush=True)
