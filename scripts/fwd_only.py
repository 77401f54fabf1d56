"""Profiling aid: N batch-256 forwards of one encoder variant on one lane (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
v = sys.argv[1] if len(sys.argv) > 1 else 'clip_rn50'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sd = synth.clip_rn50_state_dict(1) if v == 'clip_rn50' else synth.resnet50_state_dict(1, v)
hw = 224 if v == 'clip_rn50' else 256
dt = sys.argv[3] if len(sys.argv) > 3 else 'f16'
m = HipResNet50(sd, v, compute_dtype=dt, max_batch=256)
fr = torch.from_numpy(synth.frames(1, 256, hw, hw)).cuda()
out = torch.empty((256, m.out_size), device='cuda')
for _ in range(n):
    m.forward_into(fr, out)
torch.cuda.synchronize()
