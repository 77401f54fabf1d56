"""whole-bottleneck frame launch: the same weights every launch vs five weight sets in turn (as layer3.1-3.5 in the network), x / y rotating over four sets"""
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
xs = [rnd(n, 14, 14, 1024).clamp_(min=0).to(tdt) for _ in range(4)]
ys = [torch.empty_like(xs[0]) for _ in range(4)]


def wset():
    w1 = rnd(256, 1024, std=(2.0 / 1024) ** 0.5).to(tdt); w2 = rnd(256, 2304, std=(2.0 / 2304) ** 0.5).to(tdt); w3 = rnd(1024, 256, std=(2.0 / 256) ** 0.5).to(tdt)
    out = []
    for w, r, k in ((w1, 256, 1024), (w2, 256, 2304), (w3, 1024, 256)):
        p = torch.empty_like(w); _lib.check(L.pvr_op_pack_frag_weights(vp(w), vp(p), r, k, st())); out.append(p)
    return out + [rnd(256), rnd(256), rnd(1024)]


W = [wset() for _ in range(5)]
k = [0]


def launch(nw):
    i = k[0]; k[0] += 1
    w1, w2, w3, b1, b2, b3 = W[i % nw]
    _lib.check(L.pvr_op_bneck_frame(None, vp(w2), vp(b2), vp(w3), vp(b3), vp(xs[i & 3]), vp(ys[i & 3]), None, None, None, None, vp(w1), vp(b1), n, 11, cdt, st()))


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('one weight set: %.1f us per launch | five weight sets in turn: %.1f us' % (timed(lambda: launch(1)), timed(lambda: launch(5))), flush=True)
big = torch.empty(1 << 27, dtype=torch.float32, device='cuda')


def flushed(nw):
    big.fill_(1.0)
    launch(nw)


tf = timed(lambda: big.fill_(1.0))
print('with a 512 MB fill in front of every launch (fill alone %.1f us): %.1f us' % (tf, timed(lambda: flushed(5)) - tf), flush=True)
