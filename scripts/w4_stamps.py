"""Per-step cycle stamps of one conv_w4 block (PVR_W4_STAMPS): python scripts/w4_stamps.py out.txt"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_path = sys.argv[1]
import torch
from pvr_habitat_amd import _lib
L = _lib.lib()
vp = lambda t: C.c_void_p(t.data_ptr())
for name, n, h, w, cin, cout, k in [('layer3.x.conv2', 256, 14, 14, 256, 256, 3), ('ViT FC2', 256, 197, 1, 3072, 768, 1)]:
    pad = k // 2
    x = torch.randn((n, h, w, cin), device='cuda').bfloat16()
    wk = (torch.randn((cout, k * k * cin), device='cuda') * 0.02).bfloat16()
    b = torch.zeros(cout, device='cuda')
    out = torch.empty((n, h, w, cout), device='cuda', dtype=torch.bfloat16)
    _lib.check(L.pvr_debug_set_conv_algo(4))
    run = lambda: _lib.check(L.pvr_op_conv2d(vp(x), vp(wk), vp(b), None, vp(out), n, h, w, cin, cout, k, k, 1, pad, 0, 0, _lib.PVR_BF16, _lib.stream_ptr()))
    for _ in range(100): run()            # clocks settled under load
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    open(out_path, 'a').write('## %s [%s]: %.1f us per launch\n' % (name, os.path.basename(_lib.LIB_PATH), e0.elapsed_time(e1) / 20 * 1e3))
    os.environ['PVR_W4_STAMPS'] = out_path
    run()
    del os.environ['PVR_W4_STAMPS']
    torch.cuda.synchronize()
_lib.check(L.pvr_debug_set_conv_algo(-1))
