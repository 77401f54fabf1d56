"""layer2.0 as separate launches: conv2 3x3 stride 2 (128 -> 128 on 56 x 56), conv3 & downsample as one two-operand launch (K = 128 + 256 -> 512 at 28 x 28),
the next block's conv1 (512 -> 128) - isolated times at batch 256, buffers rotating"""
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
R = 3


def timed(fn, reps=24):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t1 = [rnd(n, 56, 56, 128).clamp_(min=0).to(tdt) for _ in range(R)]
x = [rnd(n, 56, 56, 256).clamp_(min=0).to(tdt) for _ in range(R)]
t2 = [torch.empty((n, 28, 28, 128), dtype=tdt, device='cuda') for _ in range(R)]
y = [torch.empty((n, 28, 28, 512), dtype=tdt, device='cuda') for _ in range(R)]
t1n = [torch.empty((n, 28, 28, 128), dtype=tdt, device='cuda') for _ in range(R)]
w2 = rnd(128, 1152, std=(2.0 / 1152) ** 0.5).to(tdt); b2 = rnd(128)
wc = rnd(512, 128 + 256, std=(1.0 / 384) ** 0.5).to(tdt); bc = rnd(512)
w1 = rnd(128, 512, std=(2.0 / 512) ** 0.5).to(tdt); b1 = rnd(128)
i = [0]


def nxt():
    i[0] += 1
    return i[0] % R


def f_conv2():
    j = nxt(); _lib.check(L.pvr_op_conv2d(vp(t1[j]), vp(w2), vp(b2), None, vp(t2[j]), n, 56, 56, 128, 128, 3, 3, 2, 1, 1, 0, cdt, st()))


def f_dual():
    j = nxt(); _lib.check(L.pvr_op_conv2d_dual(vp(t2[j]), vp(x[j]), vp(wc), vp(bc), vp(y[j]), n, 28, 28, 128, 512, 1, 1, 1, 0, 56, 56, 256, 2, 1, cdt, st()))


def f_conv1():
    j = nxt(); _lib.check(L.pvr_op_conv2d(vp(y[j]), vp(w1), vp(b1), None, vp(t1n[j]), n, 28, 28, 512, 128, 1, 1, 1, 0, 1, 0, cdt, st()))


a, b, c = timed(f_conv2), timed(f_dual), timed(f_conv1)
print('conv2 3x3 s2 %.1f us | conv3 & downsample (two-operand) %.1f us | next conv1 %.1f us | sum %.1f us  (the plan today: downsample 75 + chain 209 = 284 us)' % (a, b, c, a + b + c), flush=True)
