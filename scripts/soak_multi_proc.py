"""Soak: several PROCESSES share the GPU, each keeps two batch-256 forwards in flight on two lanes and checks every output against
its own quiet-time reference (bit-exact).  python scripts/soak_multi_proc.py [seconds] [variant,variant,...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'worker':
    variant, secs, seed = sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    sys.path.insert(0, ROOT)
    import torch
    from pvr_habitat_amd import synth
    from pvr_habitat_amd.embeddings import HipResNet50
    if variant == 'bc':
        from pvr_habitat_amd.models import PolicyNet, HipRMSprop
        import numpy as np
        torch.manual_seed(0)
        def run():
            torch.manual_seed(0)
            net = PolicyNet((4096,), 4, batch_norm=True).to(device='cuda')
            opt = HipRMSprop(net, lr=1e-4, alpha=0.99, eps=1e-5, max_grad_norm=40.0, max_epochs=100)
            g = torch.Generator().manual_seed(1)
            o = torch.randn((100, 16, 4096), generator=g).cuda(); d = (torch.rand((100, 16), generator=g) < 0.02).cuda(); a = torch.randint(0, 4, (100, 16), generator=g).cuda()
            out = []
            for _ in range(5):
                opt.scheduler_step(); l, gn = opt.step(o, d, a); out.append((float(l), float(gn)))
            return out
        ref = run(); n = bad = 0; t0 = time.time()
        while time.time() - t0 < secs:
            bad += int(run() != ref); n += 1
        print('worker bc: %d five-step runs, %d mismatching' % (n, bad), flush=True)
        sys.exit(1 if bad else 0)
    if variant == 'conv5':
        m = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype='bf16', max_batch=256); hw = 256
    elif variant in ('r18', 'r34', 'conv3', 'conv4'):        # BasicBlock nets (conv3x3_halo), compressed heads (split-K on conv4)
        m = HipResNet50(synth.resnet50_state_dict(2, variant), variant, compute_dtype='f16' if variant == 'conv3' else 'bf16', max_batch=256); hw = 256
    else:
        m = HipResNet50(synth.clip_vit_state_dict(1, patch=16 if variant == 'clip_b16' else 32), variant, compute_dtype='bf16', max_batch=256); hw = 224
    fa = torch.from_numpy(synth.frames(seed, 256, hw, hw)).cuda(); fb = torch.from_numpy(synth.frames(seed + 100, 256, hw, hw)).cuda()
    ra, rb = m(fa).clone(), m(fb).clone()
    oa, ob = torch.zeros_like(ra), torch.zeros_like(rb)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    n = bad = 0; t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(4):
            with torch.cuda.stream(sa): m.forward_into(fa, oa, lane=0)
            with torch.cuda.stream(sb): m.forward_into(fb, ob, lane=1)
        torch.cuda.synchronize()
        bad += int(not torch.equal(oa, ra)) + int(not torch.equal(ob, rb)); n += 8
    print('worker %-8s: %d forwards, %d mismatching checks' % (variant, n, bad), flush=True)
    sys.exit(1 if bad else 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
variants = (sys.argv[2] if len(sys.argv) > 2 else 'conv5,clip_b16,conv5').split(',')
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'worker', v, str(secs), str(11 + i)]) for i, v in enumerate(variants)]
rc = [p.wait() for p in procs]
print('soak %s for %.0f s: exit codes %s' % (variants, secs, rc))
sys.exit(max(rc))
