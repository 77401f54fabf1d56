"""Per-launch time of single convolutions through pvr_op_conv2d at batch 256 (slope per K tile and intercept of the deep layers' launches):
python scripts/conv_shape_times.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib
L = _lib.lib()
def run(n, h, cin, cout, k, stride, res, reps=30, algo=-1):
    _lib.check(L.pvr_debug_set_conv_algo(algo))
    pad = k // 2
    ho = (h + 2 * pad - k) // stride + 1
    x = torch.randn((n, h, h, cin), device='cuda').to(torch.bfloat16)
    w = (torch.randn((cout, k * k * cin), device='cuda') * 0.02).to(torch.bfloat16)
    b = torch.zeros(cout, device='cuda')
    r = torch.randn((n, ho, ho, cout), device='cuda').to(torch.bfloat16) if res else None
    out = torch.empty((n, ho, ho, cout), device='cuda', dtype=torch.bfloat16)
    def go():
        _lib.check(L.pvr_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(r.data_ptr()) if r is not None else None,
                                   C.c_void_p(out.data_ptr()), n, h, h, cin, cout, k, k, stride, pad, 1, 0, 0, _lib.stream_ptr()))
    for _ in range(5): go()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    M = n * ho * ho
    print('algo %2d  %3dx%-3d cin %4d cout %4d k%d s%d res%d: M=%6d K tiles %3d  %7.1f us  %6.1f TFLOP/s' % (algo, h, h, cin, cout, k, stride, res, M, k * k * cin // 64, us, 2.0 * M * k * k * cin * cout / us / 1e6), flush=True)
    _lib.check(L.pvr_debug_set_conv_algo(-1))
for cin in (64, 128, 256, 512, 1024, 2048):                 # 7x7, cout 512, 1x1: K tiles 1 .. 32 on the layer4 grid (98 x 2 tiles of 128 pixels)
    run(256, 7, cin, 512, 1, 1, 0, algo=2)
run(256, 7, 512, 512, 3, 1, 0, algo=2)                       # layer4 conv2: 72 K tiles
run(256, 7, 512, 512, 3, 1, 0, algo=1)
for cin in (64, 256, 1024):                                   # 14x14, cout 256, 1x1 on the layer3 grid
    run(256, 14, cin, 256, 1, 1, 0, algo=3)
run(256, 14, 256, 256, 3, 1, 0, algo=3)                      # layer3 conv2: 36 K tiles
