"""Isolated timing of the whole layer3 bottleneck per frame in both tilings (bneck_frame.hip FRONT1: 8 waves x 32 couts; bneck_frame64.hip: 4 waves x 64 couts),
batch 256, random data, warm (inputs in the Infinity Cache) and cold (768 MB touched in front of every launch), with the 64-channel tiling's s_memtime stamps:
python scripts/frame64_time.py [dtype]"""
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
x = rnd(n, 14, 14, 1024).clamp_(min=0).to(tdt)
w1 = rnd(256, 1024, std=(2.0 / 1024) ** 0.5).to(tdt); w2 = rnd(256, 2304, std=(2.0 / 2304) ** 0.5).to(tdt); w3 = rnd(1024, 256, std=(2.0 / 256) ** 0.5).to(tdt)
b1, b2, b3 = rnd(256), rnd(256), rnd(1024)
pk = lambda w: (lambda o: (_lib.check(L.pvr_op_pack_frag_weights(vp(w), vp(o), w.shape[0], w.shape[1], st())), o)[1])(torch.empty_like(w))
w1p, w2p, w3p = pk(w1), pk(w2), pk(w3)
y = {k: torch.empty_like(x) for k in (0, 1)}
flush = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')


def run(mode):
    _lib.check(L.pvr_debug_set_frame64(mode))
    _lib.check(L.pvr_op_bneck_frame(None, vp(w2p), vp(b2), vp(w3p), vp(b3), vp(x), vp(y[mode]), None, None, None, None, vp(w1p), vp(b1), n, 3 | 8, cdt, st()))


def timed(fn, reps=20, cold=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if cold:
            flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


run(0); run(1); torch.cuda.synchronize()
print('bit-identical:', bool(torch.equal(y[0].view(torch.int16), y[1].view(torch.int16))), 'finite:', bool(torch.isfinite(y[1].float()).all()), 'nonzero frac %.3f' % float((y[1] != 0).float().mean()))
gf = 2 * n * 196 * (256 * 1024 + 256 * 2304 + 1024 * 256) / 1e9
for cold in (False, True):
    t0 = timed(lambda: run(0), cold=cold); t1 = timed(lambda: run(1), cold=cold)
    print('%s: 32-channel tiling %.1f us (%.0f TF) | 64-channel tiling %.1f us (%.0f TF)' % ('cold' if cold else 'warm', t0, gf / t0 * 1e3, t1, gf / t1 * 1e3))
for cold in (False, True):
    stamps = torch.zeros(8, dtype=torch.int64, device='cuda')
    for _ in range(10):
        if cold:
            flush.add_(1)
        _lib.check(L.pvr_debug_bneck_frame64_stamps(vp(w1p), vp(b1), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(x), vp(y[1]), n, cdt, vp(stamps), st()))
    torch.cuda.synchronize()
    t = stamps.cpu().numpy()
    names = ['front conv1', 'conv2 loop', 't2 written', 'chunk0 K loop', 'chunk0 epilogue', 'chunks 1-3', 'stores drained']
    print('%s stamps (cycles): ' % ('cold' if cold else 'warm') + ', '.join('%s +%d' % (names[k - 1], t[k] - t[k - 1]) for k in range(1, 8)) + ' | total %d' % (t[7] - t[0]))
_lib.check(L.pvr_debug_set_frame64(-1))
