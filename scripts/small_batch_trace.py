"""Profiling aid: 50 device-resident N=2 forwards of ResNet50 (the EmbeddingWrapper pattern); run under rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
m = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype=sys.argv[1] if len(sys.argv) > 1 else 'bf16', max_batch=2)
if len(sys.argv) > 2 and sys.argv[2] == 'fast':
    m.set_low_latency(True)
fr = torch.from_numpy(synth.frames(1, 2, 64, 64)).cuda()
out = torch.empty((2, 2048), device='cuda')
for _ in range(50):
    m.forward_into(fr, out)
torch.cuda.synchronize()
