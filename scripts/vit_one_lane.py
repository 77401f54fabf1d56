"""Profiling aid: N forwards of a ViT encoder at batch 256 with ONE batch in flight (a kernel's duration is its own); run under
rocprofv3 --kernel-trace --stats."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50

variant = sys.argv[1] if len(sys.argv) > 1 else 'clip_b16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sd = synth.clip_vit_state_dict(1, patch=16 if variant == 'clip_b16' else 32)
m = HipResNet50(sd, variant, compute_dtype='bf16', max_batch=256)
fr = torch.from_numpy(synth.frames(3, 256, 224, 224)).cuda()
out = torch.empty((256, 512), dtype=torch.float32, device='cuda')
for _ in range(2):
    m.forward_into(fr, out, lane=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.forward_into(fr, out, lane=0)
torch.cuda.synchronize()
print('%s one lane: %.3f ms per 256-frame forward' % (variant, (time.perf_counter() - t0) / steps * 1e3))
