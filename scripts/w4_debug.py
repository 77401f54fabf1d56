"""Error pattern of conv_w4 against conv_igemm on small 1x1 shapes (debugging aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import _lib
L = _lib.lib()
vp = lambda t: C.c_void_p(t.data_ptr())
for cin in (64, 128, 192, 256, 320, 384, 512):
    n, h, w, cout, k = 1, 16, 14, 256, 1
    torch.manual_seed(cin)
    x = torch.randn((n, h, w, cin), device='cuda').bfloat16()
    wk = (torch.randn((cout, cin), device='cuda') * (2.0 / cin) ** 0.5).bfloat16()
    b = torch.zeros(cout, device='cuda')
    outs = {}
    for algo in (0, 4):
        _lib.check(L.pvr_debug_set_conv_algo(algo))
        out = torch.empty((n, h, w, cout), device='cuda', dtype=torch.bfloat16)
        _lib.check(L.pvr_op_conv2d(vp(x), vp(wk), vp(b), None, vp(out), n, h, w, cin, cout, k, k, 1, 0, 0, 0, _lib.PVR_BF16, _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs[algo] = out.float().reshape(-1, cout)
    _lib.check(L.pvr_debug_set_conv_algo(-1))
    d = (outs[0] - outs[4]).abs()
    bad = d > 0
    print('cin %4d nk %2d: bad %6d of %d; bad rows %s; bad cols %s' % (cin, cin // 32, int(bad.sum()), bad.numel(),
          torch.nonzero(bad.any(1)).flatten()[:12].tolist(), torch.nonzero(bad.any(0)).flatten()[:12].tolist()))
    if bad.any() and cin <= 64:
        # which K slices are missing: out4 vs partial sums
        xf, wf = x.float().reshape(-1, cin), wk.float()
        for s in range(cin // 32):
            part = xf[:, s*32:(s+1)*32] @ wf[:, s*32:(s+1)*32].T
            print('   slice %d: |out4 - slice| max %.3f   |out0 - out4 - slice| max %.3f' % (s, (outs[4] - part).abs().max(), (outs[0] - outs[4] - part).abs().max()))
