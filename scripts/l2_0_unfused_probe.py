"""layer2.0 at batch 256: what would [conv2 3x3 s2] + [conv3 & downsample as the two-operand launch] + [layer2.1.conv1] cost as separate launches,
against the plan's fused tail (bottleneck_chain stride 2: 211 us) + downsample launch (78 us)?   python scripts/l2_0_unfused_probe.py [dtype]"""
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


x = rnd(n, 56, 56, 256).clamp_(min=0).to(tdt)            # block input (layer1 output)
t1 = rnd(n, 56, 56, 128).clamp_(min=0).to(tdt)           # layer2.0.conv1 output
w2 = rnd(128, 9 * 128, std=(1.0 / 1152) ** 0.5).to(tdt)
w3 = rnd(512, 128, std=(1.0 / 128) ** 0.5).to(tdt); wd = rnd(512, 256, std=(1.0 / 256) ** 0.5).to(tdt)
wc = torch.cat([w3, wd], dim=1).contiguous()
w1n = rnd(128, 512, std=(1.0 / 512) ** 0.5).to(tdt)
b128 = rnd(128); b512 = rnd(512)
t2 = torch.empty((n, 28, 28, 128), dtype=tdt, device='cuda'); y = torch.empty((n, 28, 28, 512), dtype=tdt, device='cuda'); t1n = torch.empty((n, 28, 28, 128), dtype=tdt, device='cuda')
ds = torch.empty_like(y)
c2 = lambda: _lib.check(L.pvr_op_conv2d(vp(t1), vp(w2), vp(b128), None, vp(t2), n, 56, 56, 128, 128, 3, 3, 2, 1, 1, 0, cdt, st()))
dual = lambda: _lib.check(L.pvr_op_conv2d_dual(vp(t2), vp(x), vp(wc), vp(b512), vp(y), n, 28, 28, 128, 512, 1, 1, 1, 0, 56, 56, 256, 2, 1, cdt, st()))
c1n = lambda: _lib.check(L.pvr_op_conv2d(vp(y), vp(w1n), vp(b128), None, vp(t1n), n, 28, 28, 512, 128, 1, 1, 1, 0, 1, 0, cdt, st()))
dsl = lambda: _lib.check(L.pvr_op_conv2d(vp(x), vp(wd), vp(b512), None, vp(ds), n, 56, 56, 256, 512, 1, 1, 2, 0, 0, 0, cdt, st()))
c3 = lambda: _lib.check(L.pvr_op_conv2d(vp(t2), vp(w3), vp(b512), vp(ds), vp(y), n, 28, 28, 128, 512, 1, 1, 1, 0, 1, 0, cdt, st()))
allf = lambda: (c2(), dual(), c1n())
print('conv2 3x3 s2 %.1f us | conv3 & downsample (dual) %.1f us | next conv1 %.1f us | the three back to back %.1f us || downsample alone %.1f us, conv3 + res alone %.1f us'
      % (timed(c2), timed(dual), timed(c1n), timed(allf), timed(dsl), timed(c3)))
