"""Probe: device-resident ResNet50 forward time for small batches, wave form (PVR_CHAIN_WAVE=1) vs block form (=0) of the layer1 tails."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
sd = synth.resnet50_state_dict(1, 'conv5')
res = {}
for wave in ('1', '0'):
    os.environ['PVR_CHAIN_WAVE'] = wave
    m = HipResNet50(sd, 'conv5', compute_dtype='bf16', max_batch=64)
    for n in (2, 4, 8, 16, 32, 64):
        fr = torch.from_numpy(synth.frames(1, n, 256, 256)).cuda()
        out = torch.empty((n, 2048), device='cuda')
        for _ in range(10): m.forward_into(fr, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m.forward_into(fr, out)
        torch.cuda.synchronize(); res[(wave, n)] = (time.perf_counter() - t0) / 50 * 1e3
    m.close()
for n in (2, 4, 8, 16, 32, 64):
    print('n %3d: wave %.3f ms, block %.3f ms' % (n, res[('1', n)], res[('0', n)]))
