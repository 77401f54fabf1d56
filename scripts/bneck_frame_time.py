"""Isolated timing of the per-frame fused layer3 bottleneck tail (bneck_frame.hip) against the launches it replaces, batch 256, random data:
python scripts/bneck_frame_time.py [dtype] [n]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib, synth
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tdt, cdt = {'bf16': (torch.bfloat16, _lib.PVR_BF16), 'f16': (torch.float16, _lib.PVR_F16)}[dt]
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device='cuda', generator=g) * std)
x = rnd(n, 14, 14, 256).clamp_(min=0).to(tdt)
xin = rnd(n, 14, 14, 1024).clamp_(min=0).to(tdt)
w1 = rnd(256, 1024, std=(2.0 / 1024) ** 0.5).to(tdt)
w2 = rnd(256, 2304, std=(2.0 / 2304) ** 0.5).to(tdt)
w3 = rnd(1024, 256, std=(2.0 / 256) ** 0.5).to(tdt)
b1, b2, b3 = rnd(256), rnd(256), rnd(1024)
r = xin
t1 = torch.empty((n, 14, 14, 256), dtype=tdt, device='cuda'); t2 = torch.empty_like(t1)
y = torch.empty((n, 14, 14, 1024), dtype=tdt, device='cuda'); y2 = torch.empty_like(y); y3 = torch.empty_like(y)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = _lib.stream_ptr


def conv(i, w, b, res, o, cin, cout, k):
    _lib.check(L.pvr_op_conv2d(vp(i), vp(w), vp(b), vp(res), vp(o), n, 14, 14, cin, cout, k, k, 1, k // 2, 1, 0, cdt, st()))


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


w2p, w3p, w1p = torch.empty_like(w2), torch.empty_like(w3), torch.empty_like(w1)
_lib.check(L.pvr_op_pack_frag_weights(vp(w1), vp(w1p), 256, 1024, st()))
t1n = torch.empty_like(t1)
_lib.check(L.pvr_op_pack_frag_weights(vp(w2), vp(w2p), 256, 2304, st()))
_lib.check(L.pvr_op_pack_frag_weights(vp(w3), vp(w3p), 1024, 256, st()))
sep2 = timed(lambda: conv(x, w2, b2, None, t2, 256, 256, 3))
sep3 = timed(lambda: conv(t2, w3, b3, r, y, 256, 1024, 1))
sep1 = timed(lambda: conv(xin, w1, b1, None, t1, 1024, 256, 1))
both = timed(lambda: (conv(x, w2, b2, None, t2, 256, 256, 3), conv(t2, w3, b3, r, y, 256, 1024, 1)))
f1 = timed(lambda: _lib.check(L.pvr_op_bneck_frame(vp(x), vp(w2p), vp(b2), None, None, None, None, vp(t2), None, None, None, None, None, n, 1, cdt, st())))
f3 = timed(lambda: _lib.check(L.pvr_op_bneck_frame(vp(x), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(r), vp(y2), None, None, None, None, None, None, n, 3, cdt, st())))
f7 = timed(lambda: _lib.check(L.pvr_op_bneck_frame(vp(x), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(r), vp(y2), None, vp(w1p), vp(b1), vp(t1n), None, None, n, 7, cdt, st())))
f11 = timed(lambda: _lib.check(L.pvr_op_bneck_frame(None, vp(w2p), vp(b2), vp(w3p), vp(b3), vp(r), vp(y3), None, None, None, None, vp(w1p), vp(b1), n, 11, cdt, st())))
whole = timed(lambda: (conv(r, w1, b1, None, t1, 1024, 256, 1), conv(t1, w2, b2, None, t2, 256, 256, 3), conv(t2, w3, b3, r, y, 256, 1024, 1)))
torch.cuda.synchronize(); y_whole_ref = y.clone()
conv(x, w2, b2, None, t2, 256, 256, 3); conv(t2, w3, b3, r, y, 256, 1024, 1); conv(y, w1, b1, None, t1, 1024, 256, 1)
all3 = timed(lambda: (conv(x, w2, b2, None, t2, 256, 256, 3), conv(t2, w3, b3, r, y, 256, 1024, 1), conv(y, w1, b1, None, t1, 1024, 256, 1)))
torch.cuda.synchronize()
same = bool(torch.equal(y.view(torch.int16), y2.view(torch.int16))) and bool(torch.equal(t1.view(torch.int16), t1n.view(torch.int16)))
gf1 = 2 * n * 196 * 256 * 1024 / 1e9
gf2, gf3 = 2 * n * 196 * 256 * 2304 / 1e9, 2 * n * 196 * 1024 * 256 / 1e9
same3 = bool(torch.equal(y_whole_ref.view(torch.int16), y3.view(torch.int16)))
print('whole bottleneck (conv1 + conv2 + conv3): one launch %.1f us vs the three launches %.1f us, bit-identical %s' % (f11, whole, same3))
print('%s n=%d: separate conv1 %.1f us | conv2 %.1f us (%.0f TF) + conv3 %.1f us (%.0f TF) = %.1f us back to back %.1f us | fused conv2 only %.1f us (%.0f TF), conv2+conv3 %.1f us (%.0f TF), conv2+conv3+next conv1 %.1f us (%.0f TF) vs the three launches back to back %.1f us  bit-identical (y, t1n): %s'
      % (dt, n, sep1, sep2, gf2 / sep2 * 1e3, sep3, gf3 / sep3 * 1e3, sep2 + sep3, both, f1, gf2 / f1 * 1e3, f3, (gf2 + gf3) / f3 * 1e3, f7, (gf1 + gf2 + gf3) / f7 * 1e3, all3, same), flush=True)

for mode7 in ((0, 1, 2) if n > 8 else ()):
    stamps = torch.zeros(24, dtype=torch.int64, device='cuda')
    cold = os.environ.get('BF_COLD', '0') == '1'        # evict the Infinity Cache in front of every stamped launch (the in-network condition: x comes from HBM)
    flush = torch.empty(768 << 20, dtype=torch.uint8, device='cuda') if cold else None
    for _ in range(20):
        if cold:
            flush.add_(1)
        _lib.check(L.pvr_debug_bneck_frame_stamps(vp(x), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(r), vp(y2), vp(w1p) if mode7 == 1 else None, vp(b1) if mode7 == 1 else None, vp(t1n) if mode7 == 1 else None, vp(w1p) if mode7 == 2 else None, vp(b1) if mode7 == 2 else None, n, cdt, vp(stamps), st()))
    torch.cuda.synchronize()
    extra = stamps.cpu().numpy()[20:]
    t = stamps.cpu().numpy()[:20].reshape(2, 10)
    names = ['start', 'prologue done', 'conv2 loop done', 't2 written', 'conv3 start', 'chunk/round0 K loop', 'chunk/round0 epilogue', 'chunk2/round5 done', 'all issued', 'stores drained']
    print('stamps, %s:' % ('conv2 + conv3', 'conv2 + conv3 + next conv1', 'own conv1 + conv2 + conv3')[mode7])
    if extra.any():
        print('front phase waits (BF_FRONT_STAMPS build): group 0 vmcnt %d barrier %d | group 1 vmcnt %d barrier %d cycles' % tuple(extra))
    for g_ in range(2):
        print('group %d cycles: ' % g_ + ', '.join('%s +%d' % (names[k], t[g_, k] - t[g_, k - 1]) for k in range(1, 10)) + ' | total %d' % (t[g_, 9] - t[g_, 0]))

# timing knock-outs of the conv3 phase (results are wrong by construction): 16 = no y stores, 32 = no identity loads
for ko in (16, 32, 48):
    t_ = timed(lambda: _lib.check(L.pvr_op_bneck_frame(vp(x), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(r), vp(y2), None, None, None, None, None, None, n, 3 | ko, cdt, st())))
    print('knock-out %s: conv2+conv3 %.1f us' % ({16: 'no y stores', 32: 'no identity loads', 48: 'neither'}[ko], t_))
