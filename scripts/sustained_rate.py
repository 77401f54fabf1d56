"""Is the 1-second headline leg representative of the steady state?  Ten consecutive ~1 s legs of the same model (no re-allocation),
bf16 then f16 then bf16 again; prints frames/s per leg."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
sd = synth.resnet50_state_dict(1, 'conv5')
pool = torch.from_numpy(synth.frames(1, 4096, 256, 256)).cuda()
batches = [pool[i:i + 256] for i in range(0, 4096, 256)]
models = {dt: HipResNet50(sd, 'conv5', compute_dtype=dt, max_batch=256) for dt in ('bf16', 'f16')}
outs = [torch.empty((256, 2048), device='cuda') for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]
def leg(m, steps=320):
    def run(k):
        for i in range(k):
            with torch.cuda.stream(st[i % 2]):
                m.forward_into(batches[i % 16], outs[i % 2], lane=i % 2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    return steps * 256 / (time.perf_counter() - t0)
for dt in ('bf16', 'f16'):
    leg(models[dt], 8)
for tag in ['bf16'] * 5 + ['f16'] * 4 + ['bf16'] * 3 + ['f16'] * 2:
    print('%s %.0f' % (tag, leg(models[tag])), flush=True)
