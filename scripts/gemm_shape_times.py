"""Per-launch time of the transformer GEMM shapes through pvr_op_conv2d (conv_pp256): python scripts/gemm_shape_times.py"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import _lib, synth
L = _lib.lib()
def run(n, t, cin, cout, act, res, out_f32, reps=30):
    x = torch.randn((n, t, 1, cin), device='cuda').to(torch.bfloat16)
    w = (torch.randn((cout, cin), device='cuda') * 0.03).to(torch.bfloat16)
    b = torch.zeros(cout, device='cuda')
    r = torch.randn((n, t, 1, cout), device='cuda', dtype=torch.float32 if res == 2 else torch.bfloat16) if res else None
    out = torch.empty((n, t, 1, cout), device='cuda', dtype=torch.float32 if out_f32 else torch.bfloat16)
    def go():
        _lib.check(L.pvr_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(r.data_ptr()) if r is not None else None,
                                   C.c_void_p(out.data_ptr()), n, t, 1, cin, cout, 1, 1, 1, 0, act, (1 if out_f32 else 0) | (2 if res == 2 else 0), 0, _lib.stream_ptr()))
    for _ in range(5): go()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * n * t * cin * cout
    tiles = ((n * t + 255) // 256) * ((cout + 255) // 256)
    print('M=%6d K=%4d N=%4d act=%d res=%d f32out=%d: %7.1f us  %6.1f TFLOP/s  %4d tiles (%.2f per CU), %5.2f us per tile-round, K loop alone %5.2f us'
          % (n * t, cin, cout, act, res, out_f32, us, fl / us / 1e6, tiles, tiles / 256, us / -(-tiles // 256), cin / 64 * 2810 / 2100), flush=True)
for args in [(256, 197, 768, 2304, 0, 0, 0), (256, 197, 768, 3072, 2, 0, 0), (256, 197, 768, 3072, 0, 0, 0), (256, 197, 3072, 768, 0, 2, 1), (256, 197, 768, 768, 0, 2, 1),
             (256, 197, 3072, 768, 0, 0, 0), (256, 197, 768, 3072, 0, 0, 1), (256, 196, 768, 3072, 2, 0, 0), (256, 200, 768, 3072, 2, 0, 0), (256, 197, 768, 3072, 3, 0, 0)]:
    run(*args)
