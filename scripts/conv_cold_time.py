"""Warm vs cold weights / inputs: one conv launch timed by its own pair of events, with and without a 1 GB fill in front (evicts L2 and the Infinity
Cache): python scripts/conv_cold_time.py [dtype]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
n = 256
tdt, cdt = {'bf16': (torch.bfloat16, _lib.PVR_BF16), 'f16': (torch.float16, _lib.PVR_F16)}[dt]
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device='cuda', generator=g) * std)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = _lib.stream_ptr
big = torch.empty(1 << 28, dtype=torch.float32, device='cuda')


def per_launch(fn, cold, reps=12):
    ts = []
    for _ in range(reps + 2):
        if cold:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


CASES = [('layer4.x conv2 3x3 512->512', 7, 512, 512, 3, 1, 0), ('layer4.x conv1 2048->512', 7, 2048, 512, 1, 1, 0), ('layer4.x conv3 512->2048 + id', 7, 512, 2048, 1, 1, 1),
         ('layer3.x conv2 3x3 256->256', 14, 256, 256, 3, 1, 0)]
for name, hw, cin, cout, k, stride, res in CASES:
    pad = k // 2
    ho = (hw + 2 * pad - k) // stride + 1
    x = rnd(n, hw, hw, cin).clamp_(min=0).to(tdt)
    w = rnd(cout, k * k * cin, std=(2.0 / (k * k * cin)) ** 0.5).to(tdt)
    b = rnd(cout)
    r = rnd(n, ho, ho, cout).to(tdt) if res else None
    y = torch.empty((n, ho, ho, cout), dtype=tdt, device='cuda')
    wp = torch.empty_like(w)
    _lib.check(L.pvr_op_pack_frag_weights(vp(w), vp(wp), cout, k * k * cin, st()))
    ref = lambda: _lib.check(L.pvr_op_conv2d(vp(x), vp(w), vp(b), vp(r), vp(y), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))
    wf = lambda: _lib.check(L.pvr_op_conv_wfrag(vp(x), vp(wp), vp(b), vp(r), vp(y), n, hw, hw, cin, cout, k, k, stride, pad, 1, 0, cdt, st()))
    print('%-32s pvr_op_conv2d warm %.1f us cold %.1f us | conv_wfrag warm %.1f us cold %.1f us' % (name, per_launch(ref, 0), per_launch(ref, 1), per_launch(wf, 0), per_launch(wf, 1)), flush=True)
