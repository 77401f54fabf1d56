"""Soak test: N rounds of two batch-256 forwards in flight on two lanes; every output must equal the sequential reference bit for bit.
Exercises the hand-counted LDS-DMA hazards of conv_pp256 and the fused chain kernels under co-scheduling."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for variant, sd, hw in (('conv5', synth.resnet50_state_dict(1, 'conv5'), 256), ('clip_b16', synth.clip_vit_state_dict(1, patch=16), 224)):
    m = HipResNet50(sd, variant, compute_dtype='bf16', max_batch=256)
    fa = torch.from_numpy(synth.frames(21, 256, hw, hw)).cuda()
    fb = torch.from_numpy(synth.frames(22, 256, hw, hw)).cuda()
    ra, rb = m(fa).clone(), m(fb).clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    oa, ob = torch.zeros_like(ra), torch.zeros_like(rb)
    torch.cuda.synchronize()             # the zero fills run on the default stream; sa / sb are non-blocking streams
    bad = 0
    n = rounds if variant == 'conv5' else rounds // 4
    for r in range(n):
        with torch.cuda.stream(sa):
            m.forward_into(fa, oa, lane=0)
        with torch.cuda.stream(sb):
            m.forward_into(fb, ob, lane=1)
        if r % 10 == 9:
            torch.cuda.synchronize()
            bad += int(not torch.equal(oa, ra)) + int(not torch.equal(ob, rb))
    torch.cuda.synchronize()
    bad += int(not torch.equal(oa, ra)) + int(not torch.equal(ob, rb))
    print('%s: %d rounds, mismatching checks: %d' % (variant, n, bad), flush=True)
    assert bad == 0
