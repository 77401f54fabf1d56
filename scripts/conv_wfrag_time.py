"""Isolated timing of conv_wfrag.hip against pvr_op_conv2d on layer4's shapes, batch 256, random data, inputs rotating over several buffer sets:
python scripts/conv_wfrag_time.py [dtype] [n]"""
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


def timed(fn, reps=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = [('layer4.x conv2 3x3 512->512', 7, 512, 512, 3, 1, 0), ('layer4.0 conv2 3x3 s2', 14, 512, 512, 3, 2, 0), ('layer4.x conv1 2048->512', 7, 2048, 512, 1, 1, 0),
         ('layer4.x conv3 512->2048 + id', 7, 512, 2048, 1, 1, 1), ('layer4.0 downsample 1024->2048 s2', 14, 1024, 2048, 1, 2, 0),
         ('layer4.0 conv1 1024->512 (14x14)', 14, 1024, 512, 1, 1, 0), ('layer3.0 conv2 3x3 s2 256->256', 28, 256, 256, 3, 2, 0),
         ('layer3.0 conv3 256->1024', 14, 256, 1024, 1, 1, 1)]
for name, hw, cin, cout, k, stride, res in CASES:
    pad = k // 2
    ho = (hw + 2 * pad - k) // stride + 1
    R = 4
    xs = [rnd(n, hw, hw, cin).clamp_(min=0).to(tdt) for _ in range(R)]
    w = rnd(cout, k * k * cin, std=(2.0 / (k * k * cin)) ** 0.5).to(tdt)
    b = rnd(cout)
    rs = [rnd(n, ho, ho, cout).to(tdt) if res else None for _ in range(R)]
    ys = [torch.empty((n, ho, ho, cout), dtype=tdt, device='cuda') for _ in range(R)]
    y2 = torch.empty_like(ys[0])
    wp = torch.empty_like(w)
    _lib.check(L.pvr_op_pack_frag_weights(vp(w), vp(wp), cout, k * k * cin, st()))
    i = [0]

    def ref():
        j = i[0] % R; i[0] += 1
        _lib.check(L.pvr_op_conv2d(vp(xs[j]), vp(w), vp(b), vp(rs[j]), vp(ys[j]), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))

    def wf():
        j = i[0] % R; i[0] += 1
        _lib.check(L.pvr_op_conv_wfrag(vp(xs[j]), vp(wp), vp(b), vp(rs[j]), vp(ys[j]), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))

    t_ref, t_wf = timed(ref), timed(wf)
    _lib.check(L.pvr_op_conv2d(vp(xs[0]), vp(w), vp(b), vp(rs[0]), vp(ys[0]), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))
    _lib.check(L.pvr_op_conv_wfrag(vp(xs[0]), vp(wp), vp(b), vp(rs[0]), vp(y2), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))
    torch.cuda.synchronize()
    gf = 2 * n * ho * ho * cout * k * k * cin / 1e9
    print('%-36s %s n=%d: pvr_op_conv2d %.1f us (%.0f TF) | conv_wfrag %.1f us (%.0f TF)  tiles %d  bit-identical %s'
          % (name, dt, n, t_ref, gf / t_ref * 1e3, t_wf, gf / t_wf * 1e3, (n * ho * ho + 111) // 112 * (cout // 256), bool(torch.equal(ys[0].view(torch.int16), y2.view(torch.int16)))), flush=True)
