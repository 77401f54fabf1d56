import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth, _lib
for (M, K, N, flags, act) in ((100, 768, 768, 3, 0), (100, 768, 768, 2, 0), (100, 768, 768, 1, 0), (300, 768, 3072, 0, 2), (100, 3072, 768, 3, 0)):
    x = torch.from_numpy(synth.normal(1, 'x', (M, K))).half()
    w = torch.from_numpy(synth.normal(1, 'w', (N, K), std=K ** -0.5)).half()
    b = torch.from_numpy(synth.normal(1, 'b', (N,)))
    r = torch.from_numpy(synth.normal(1, 'r', (M, N)))
    ref = x.float() @ w.float().t() + b + (r if flags & 2 else 0)
    if act == 2: ref = ref * torch.sigmoid(1.702 * ref)
    out = torch.full((M, N), float('nan'), dtype=torch.float32 if flags & 1 else torch.float16, device='cuda')
    xd, wd, bd, rd = x.cuda(), w.cuda(), b.cuda(), r.cuda()
    _lib.check(_lib.lib().pvr_op_conv2d(C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(bd.data_ptr()),
               C.c_void_p(rd.data_ptr()) if flags & 2 else None, C.c_void_p(out.data_ptr()), M, 1, 1, K, N, 1, 1, 1, 0, act, flags, 1, _lib.stream_ptr()))
    torch.cuda.synchronize()
    o = out.float().cpu()
    e = (o - ref).abs()
    print(M, K, N, 'flags', flags, 'act', act, 'rel', float((o - ref).norm() / ref.norm()), 'nan', int(torch.isnan(o).sum()),
          'err by col-block(64)', [round(float(e[:, i:i + 64].mean()), 4) for i in range(0, min(N, 512), 64)])
