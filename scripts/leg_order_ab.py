"""Is the second embedding leg of one process slower than the first (same dtype)?  bench.py's f16 leg ran 15 % slower than the same
configuration as the first leg of a fresh process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
sd = synth.resnet50_state_dict(1, 'conv5')
pool = torch.from_numpy(synth.frames(1, 4096, 256, 256)).cuda()
batches = [pool[i:i + 256] for i in range(0, 4096, 256)]
def leg(dt, steps=160, lanes=2, keep=None):
    m = HipResNet50(sd, 'conv5', compute_dtype=dt, max_batch=256)
    outs = [torch.empty((256, 2048), device='cuda') for _ in range(lanes)]
    st = [torch.cuda.Stream() for _ in range(lanes)]
    def run(k):
        for i in range(k):
            with torch.cuda.stream(st[i % lanes]):
                m.forward_into(batches[i % 16], outs[i % lanes], lane=i % lanes)
    torch.cuda.synchronize(); run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return m, steps * 256 / el
for tag, dt, pre in (('1st bf16', 'bf16', 0), ('2nd bf16 (first freed)', 'bf16', 0), ('3rd f16', 'f16', 0), ('4th f16 after 3 s idle', 'f16', 3), ('5th bf16', 'bf16', 0)):
    if pre: time.sleep(pre)
    m, fps = leg(dt)
    print('%-28s %.1f frames/s' % (tag, fps), flush=True)
    del m
keep = []
for tag in ('6th bf16, previous models kept alive', '7th'):
    m, fps = leg('bf16'); keep.append(m)
    print('%-28s %.1f frames/s' % (tag, fps), flush=True)
