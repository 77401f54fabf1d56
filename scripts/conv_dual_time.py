"""Isolated timing of conv_pp256's two-operand launch (conv3 & downsample) against the two launches it replaces and against pvr_op_conv2d on the
channel-concatenated pixels: python scripts/conv_dual_time.py [dtype] [n]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tdt, cdt = {'bf16': (torch.bfloat16, _lib.PVR_BF16), 'f16': (torch.float16, _lib.PVR_F16)}[dt]
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


for name, ho, cin, cout, cin2 in (('layer3.0', 14, 256, 1024, 512), ('layer4.0', 7, 512, 2048, 1024)):
    h2 = 2 * ho
    x = rnd(n, ho, ho, cin).clamp_(min=0).to(tdt); x2 = rnd(n, h2, h2, cin2).clamp_(min=0).to(tdt)
    w3 = rnd(cout, cin, std=(1.0 / cin) ** 0.5).to(tdt); wd = rnd(cout, cin2, std=(1.0 / cin2) ** 0.5).to(tdt)
    wc = torch.cat([w3, wd], dim=1).contiguous()
    b = rnd(cout)
    ds = torch.empty((n, ho, ho, cout), dtype=tdt, device='cuda'); y = torch.empty_like(ds); y2 = torch.empty_like(ds); y3 = torch.empty_like(ds)
    xcat = torch.cat([x, x2[:, ::2, ::2, :]], dim=3).contiguous()
    two = lambda: (_lib.check(L.pvr_op_conv2d(vp(x2), vp(wd), vp(b), None, vp(ds), n, h2, h2, cin2, cout, 1, 1, 2, 0, 0, 0, cdt, st())),
                   _lib.check(L.pvr_op_conv2d(vp(x), vp(w3), vp(b), vp(ds), vp(y), n, ho, ho, cin, cout, 1, 1, 1, 0, 1, 0, cdt, st())))
    dual = lambda: _lib.check(L.pvr_op_conv2d_dual(vp(x), vp(x2), vp(wc), vp(b), vp(y2), n, ho, ho, cin, cout, 1, 1, 1, 0, h2, h2, cin2, 2, 1, cdt, st()))
    cat = lambda: _lib.check(L.pvr_op_conv2d(vp(xcat), vp(wc), vp(b), None, vp(y3), n, ho, ho, cin + cin2, cout, 1, 1, 1, 0, 1, 0, cdt, st()))
    t2, td, tc = timed(two), timed(dual), timed(cat)
    res = [name, t2, td, tc]
    for algo in (1, 2, 3):
        _lib.check(L.pvr_debug_set_conv_algo(algo)); res.append(timed(cat))
    _lib.check(L.pvr_debug_set_conv_algo(-1))
    gf = 2 * n * ho * ho * cout * (cin + cin2) / 1e9
    print('%s: two launches %.1f us | dual %.1f us (%.0f TF) | one launch on concatenated pixels: auto %.1f us, pp256 BM=256 %.1f / 128 %.1f / 224 %.1f us' % (res[0], res[1], res[2], gf / res[2] * 1e3, res[3], res[4], res[5], res[6]), flush=True)
