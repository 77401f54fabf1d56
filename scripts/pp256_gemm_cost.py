"""Probe: the persistent conv_pp256 on the ViT-B/16 GEMM geometry (M = 256 x 197 rows, N = 2304) with K swept: slope = K-loop cost per
64-deep K tile per round, intercept = what a launch spends outside its K loops (per round: epilogue + restart; per launch: ramp, tail)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import _lib
L = _lib.lib()
n, h, w = 256, 197, 1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for cout, label in ((2304, 'N=2304 (QKV, 7 rounds of 256-row tiles)'), (768, 'N=768 (3 rounds of 224-row tiles)')):
    print(label)
    for cin in ((768, 3072) if os.environ.get("PVR_PP_STAGGER_US") else (64, 256, 768, 1536, 3072)):
        x = torch.randn(n, h, w, cin, device='cuda').bfloat16()
        wt = (torch.randn(cout, cin, device='cuda') * 0.02).bfloat16()
        b = torch.zeros(cout, device='cuda')
        y = torch.empty(n, h, w, cout, device='cuda', dtype=torch.bfloat16)
        def run():
            _lib.check(L.pvr_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(y.data_ptr()),
                                       n, h, w, cin, cout, 1, 1, 1, 0, 0, 0, _lib.PVR_BF16, st))
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print('  K %5d (%2d K tiles): %7.1f us   %.2f PFLOP/s' % (cin, cin // 64, us, 2.0 * n * h * cin * cout / us / 1e9))
