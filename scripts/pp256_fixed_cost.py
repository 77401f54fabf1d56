"""Probe: launch duration of conv_pp256 (224-pixel tiles, one tile per CU: batch 256 at 14 x 14, 256 couts) as a function of K - the
intercept is the fixed cost per launch (ramp, prologue, epilogue), the slope the steady-state K loop."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import _lib
L = _lib.lib()
_lib.check(L.pvr_debug_set_conv_algo(3))
n, hw, cout = 256, 14, 256
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k, cins in ((1, (64, 128, 256, 512, 1024, 2048)), (3, (64, 128, 256))):
    for cin in cins:
        x = torch.randn(n, hw, hw, cin, device='cuda').bfloat16()
        w = (torch.randn(cout, k * k * cin, device='cuda') * 0.02).bfloat16()
        b = torch.zeros(cout, device='cuda')
        y = torch.empty(n, hw, hw, cout, device='cuda', dtype=torch.bfloat16)
        def run():
            _lib.check(L.pvr_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(y.data_ptr()),
                                       n, hw, hw, cin, cout, k, k, 1, k // 2, 1, 0, _lib.PVR_BF16, st))
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        kt = k * k * cin // 64
        print('%dx%d cin %5d: K tiles %3d  %.1f us  (%.2f us per K tile)' % (k, k, cin, kt, us, us / kt))
