"""GPU debugging aid: run the batch-n ResNet50 forward launch by launch and report how many elements of every
16-bit workspace buffer differ between repeats (0 everywhere = deterministic).  PVR_FUSE / PVR_CHAIN_CFG select the plan."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
last = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dt = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
sd = synth.resnet50_state_dict(1, 'conv5')
m = HipResNet50(sd, 'conv5', compute_dtype=dt, max_batch=n)
fr = torch.from_numpy(synth.frames(1, n, 256, 256)).cuda()
names = m.op_names()
print('FUSE=%s CFG=%s n=%d dtype=%s' % (os.environ.get('PVR_FUSE', '1'), os.environ.get('PVR_CHAIN_CFG', 'default'), n, dt))
elems = n * 56 * 56 * 256
for k in range(min(last, len(names))):
    m.debug_stop_after('#%d' % k)
    ref, bad = None, []
    for r in range(reps):
        m(fr)
        t = [m.tap('buf%d:%d' % (b, elems), elems).clone() for b in range(5)]
        if ref is None:
            ref = t
        else:
            bad.append([int((a != b).sum()) for a, b in zip(t, ref)])
    print('%2d %-44s differing per buffer X0,X1,T1,T2,DS: %s' % (k, names[k], bad), flush=True)
m.debug_stop_after('')
outs = [m(fr).clone() for _ in range(reps)]
print('emb     differing: %s' % [int((o != outs[0]).sum()) for o in outs[1:]])
# where do the differences of the first nondeterministic chain launch sit inside its 128-pixel x C tile?
k = int(os.environ.get('PVR_DBG_LAUNCH', '6'))
c = int(os.environ.get('PVR_DBG_C', '512'))
b = int(os.environ.get('PVR_DBG_BUF', '0'))
px = int(os.environ.get('PVR_DBG_PIX', str(n * 28 * 28)))
m.debug_stop_after('#%d' % k)
m(fr); ref = m.tap('buf%d:%d' % (b, px * c), px * c).clone().view(px, c)
from collections import Counter
for r in range(3):
    m(fr); t = m.tap('buf%d:%d' % (b, px * c), px * c).view(px, c)
    d = (t != ref).nonzero()
    if d.numel() == 0:
        print('rep', r, 'clean'); continue
    p, ch = d[:, 0], d[:, 1]
    print('rep', r, 'bad', d.shape[0], 'tiles', len(torch.unique(p // 128)),
          '| (wm,j) hist', sorted(Counter(((p % 128) // 16).tolist()).items()),
          '| group hist', sorted(Counter((ch // 64).tolist()).items()),
          '| (wn,fq) hist', sorted(Counter(((ch % 64) // 8).tolist()).items()))
    tl = (p // 128)
    t0 = int(tl[0]); sel = tl == t0
    print('   tile', t0, 'pixels', sorted(set((p[sel] % 128).tolist()))[:40], 'channels', sorted(set(ch[sel].tolist()))[:64])
    print('   sample values new/ref', t[p[0], ch[0]].item(), ref[p[0], ch[0]].item(), t[p[1], ch[1]].item(), ref[p[1], ch[1]].item())
