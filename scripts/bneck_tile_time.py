"""Isolated timing of the per-tile fused layer2 bottleneck (bneck_tile.hip) against the three launches it replaces, batch 256, random data:
python scripts/bneck_tile_time.py [dtype] [n]"""
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
x = rnd(n, 28, 28, 512).clamp_(min=0).to(tdt)
w1 = rnd(128, 512, std=(2.0 / 512) ** 0.5).to(tdt)
w2 = rnd(128, 1152, std=(2.0 / 1152) ** 0.5).to(tdt)
w3 = rnd(512, 128, std=(2.0 / 128) ** 0.5).to(tdt)
b1, b2, b3 = rnd(128), rnd(128), rnd(512)
t1 = torch.empty((n, 28, 28, 128), dtype=tdt, device='cuda'); t2 = torch.empty_like(t1)
y = torch.empty_like(x); y2 = torch.full_like(x, float('nan'))
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = _lib.stream_ptr


def conv(i, w, b, res, o, cin, cout, k):
    _lib.check(L.pvr_op_conv2d(vp(i), vp(w), vp(b), vp(res), vp(o), n, 28, 28, cin, cout, k, k, 1, k // 2, 1, 0, cdt, st()))


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


w1p, w2p, w3p = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(w3)
for src, dst, rows, k in ((w1, w1p, 128, 512), (w2, w2p, 128, 1152), (w3, w3p, 512, 128)):
    _lib.check(L.pvr_op_pack_frag_weights(vp(src), vp(dst), rows, k, st()))
three = lambda: (conv(x, w1, b1, None, t1, 512, 128, 1), conv(t1, w2, b2, None, t2, 128, 128, 3), conv(t2, w3, b3, x, y, 128, 512, 1))
sep = timed(three)
fused = timed(lambda: _lib.check(L.pvr_op_bneck_tile(vp(x), vp(w1p), vp(b1), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(y2), None, None, n, cdt, st())))
torch.cuda.synchronize()
same = bool(torch.equal(y.view(torch.int16), y2.view(torch.int16)))
gf = 2 * n * 784 * (512 * 128 + 1152 * 128 + 128 * 512) / 1e9
mb = 2 * n * 784 * 512 * 2 / 1e6
print('%s n=%d whole layer2 bottleneck: one launch per tile %.1f us (%.0f TF, %.0f GB/s algorithmic) vs the three launches %.1f us, bit-identical %s'
      % (dt, n, fused, gf / fused * 1e3, mb / fused * 1e3, sep, same), flush=True)

# the same launch over four rotating input / output sets (4 x 411 MB): what the kernel sees in the network, where the Infinity Cache (256 MB) holds
# little of a block's input by the time the block runs
xs = [x] + [x.clone() for _ in range(3)]
ys = [torch.empty_like(x) for _ in range(4)]
k = [0]


def rot():
    i = k[0] & 3; k[0] += 1
    _lib.check(L.pvr_op_bneck_tile(vp(xs[i]), vp(w1p), vp(b1), vp(w2p), vp(b2), vp(w3p), vp(b3), vp(ys[i]), None, None, n, cdt, st()))


print('rotating over 4 buffer sets: %.1f us' % timed(rot, reps=32), flush=True)
