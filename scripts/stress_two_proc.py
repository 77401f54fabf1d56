"""GPU debugging aid: a second PROCESS keeps the GPU busy with batch-256 forwards while this one repeats the plan up to conv
launch k and reports where the 16-bit workspace buffers differ from the first repeat (pixel-in-tile / channel histograms).
  python scripts/stress_two_proc.py [k] [reps]        (PVR_CHAIN_HALO etc. select the plan)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'noise':
    sys.path.insert(0, ROOT)
    import torch
    from pvr_habitat_amd import synth
    from pvr_habitat_amd.embeddings import HipResNet50
    m = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype='bf16', max_batch=256)
    fr = torch.from_numpy(synth.frames(2, 256, 256, 256)).cuda()
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    o = [torch.empty((256, 2048), device='cuda') for _ in range(2)]
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for i in range(8):
            with torch.cuda.stream(s[i % 2]):
                m.forward_into(fr, o[i % 2], lane=i % 2)
        torch.cuda.synchronize()
    sys.exit(0)
ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '2').split(',')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), 'noise', '100'])      # before this process touches the GPU
sys.path.insert(0, ROOT)
import torch
from collections import Counter
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
n = 256
m = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype='bf16', max_batch=n)
fr = torch.from_numpy(synth.frames(1, n, 256, 256)).cuda()
names = m.op_names()
time.sleep(3)
elems = n * 56 * 56 * 256
for k in ks:
    m.debug_stop_after('#%d' % k)
    m(fr); torch.cuda.synchronize()
    ref = [m.tap('buf%d:%d' % (b, elems), elems).clone() for b in range(5)]
    tot = [0] * 5
    first = None
    for r in range(reps):
        m(fr); torch.cuda.synchronize()
        for b in range(5):
            t = m.tap('buf%d:%d' % (b, elems), elems)
            nb = int((t != ref[b]).sum())
            tot[b] += nb
            if nb and first is None:
                first = (b, t.clone())
    print('%2d %-44s differing elements over %d repeats, buffers X0,X1,T1,T2,DS: %s' % (k, names[k], reps, tot), flush=True)
    if first is not None:
        b, t = first
        for c in (256, 64, 512, 128):           # try the plausible channel counts of that buffer
            px = elems // c
            d = (t.view(px, c) != ref[b].view(px, c)).nonzero()
            p, ch = d[:, 0], d[:, 1]
            print('   buf %d as [%d px][%d ch]: bad %d, tiles %d, pixel-in-tile hist(16s) %s, channel/64 hist %s, ch%%64//8 hist %s' % (
                b, px, c, d.shape[0], len(torch.unique(p // 128)), sorted(Counter(((p % 128) // 16).tolist()).items()),
                sorted(Counter((ch // 64).tolist()).items()), sorted(Counter(((ch % 64) // 8).tolist()).items())))
m.debug_stop_after('')
child.wait()
