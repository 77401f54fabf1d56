"""what makes conv_pp256's two-operand launch slow: stride of the second operand, share of K that is second-operand, persistent form (PVR_PP_PERSIST)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib
tdt, cdt = torch.float16, _lib.PVR_F16
n = 256
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device='cuda', generator=g) * std)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = _lib.stream_ptr


def timed(fn, reps=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ho, cin, cout, cin2, s2 in ((14, 256, 1024, 512, 2), (14, 256, 1024, 512, 1), (14, 704, 1024, 64, 1), (14, 64, 1024, 704, 1), (14, 64, 1024, 704, 2)):
    h2 = s2 * ho
    x = rnd(n, ho, ho, cin).clamp_(min=0).to(tdt); x2 = rnd(n, h2, h2, cin2).clamp_(min=0).to(tdt)
    wc = rnd(cout, cin + cin2, std=(1.0 / (cin + cin2)) ** 0.5).to(tdt)
    b = rnd(cout)
    y2 = torch.empty((n, ho, ho, cout), dtype=tdt, device='cuda'); y3 = torch.empty_like(y2)
    xcat = torch.cat([x, x2[:, ::s2, ::s2, :]], dim=3).contiguous()
    dual = lambda: _lib.check(L.pvr_op_conv2d_dual(vp(x), vp(x2), vp(wc), vp(b), vp(y2), n, ho, ho, cin, cout, 1, 1, 1, 0, h2, h2, cin2, s2, 1, cdt, st()))
    _lib.check(L.pvr_debug_set_conv_algo(3))
    cat = lambda: _lib.check(L.pvr_op_conv2d(vp(xcat), vp(wc), vp(b), None, vp(y3), n, ho, ho, cin + cin2, cout, 1, 1, 1, 0, 1, 0, cdt, st()))
    td, tc = timed(dual), timed(cat)
    _lib.check(L.pvr_debug_set_conv_algo(-1))
    print('PVR_PP_PERSIST=%s cin %d + cin2 %d stride2 %d: dual %.1f us | concatenated (pp256, 224) %.1f us' % (os.environ.get('PVR_PP_PERSIST', 'default'), cin, cin2, s2, td, tc), flush=True)
