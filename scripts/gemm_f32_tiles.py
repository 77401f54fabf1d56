"""Time the fp32 GEMM shapes of one BC iteration (T=100, B=16, obs 4096) through pvr_op_gemm_f32 for the tile selected by
PVR_GEMM_TILE (0 64x64, 1 128x64, 2 64x128, 3 128x128; unset = the library's own choice).   python scripts/gemm_f32_tiles.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import _lib
from pvr_habitat_amd.models import _plib
L = _plib()
shapes = [('fc1 fwd', 1600, 1024, 4096, 0, 0), ('ih proj', 1600, 4096, 1024, 0, 0), ('fc2 fwd', 1600, 1024, 1024, 0, 0),
          ('dW lstm', 4096, 1024, 1600, 1, 1), ('dW fc1', 1024, 4096, 1600, 1, 1), ('dW fc2', 1024, 1024, 1600, 1, 1),
          ('dx (K=4096)', 1600, 1024, 4096, 0, 1), ('dx chunk', 400, 1024, 4096, 0, 1), ('dz1', 1600, 1024, 1024, 0, 1)]
vp = lambda t: C.c_void_p(t.data_ptr())
tot = 0.0
for name, M, N, K, akm, bkn in shapes:
    A = torch.randn((K, M) if akm else (M, K), device='cuda'); B = torch.randn((K, N) if bkn else (N, K), device='cuda')
    Cc = torch.empty((M, N), device='cuda')
    run = lambda: _lib.check(L.pvr_op_gemm_f32(vp(A), vp(B), None, vp(Cc), M, N, K, akm, bkn, 0, _lib.stream_ptr()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print('tile %s  %-12s M=%4d N=%4d K=%4d  %7.1f us  %6.1f TFLOP/s' % (os.environ.get('PVR_GEMM_TILE', 'auto'), name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
print('tile %s  sum %.1f us' % (os.environ.get('PVR_GEMM_TILE', 'auto'), tot))
