"""Per-shape timing of the deep-K launches of a batch-256 ResNet50 forward (and the ViT-B/16 GEMMs) through pvr_op_conv2d:
automatic choice (conv_pp256 / conv_expand / conv_igemm) against the four-wave kernel (PVR_CONV_ALGO=4 -> conv_w4).
python scripts/w4_vs_pp256.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvr_habitat_amd import _lib
L = _lib.lib()
shapes = [('layer3.x.conv2 3x3', 256, 14, 14, 256, 256, 3, 1, 1, 0), ('layer3.x.conv1', 256, 14, 14, 1024, 256, 1, 1, 1, 0),
          ('layer3.0.conv1', 256, 28, 28, 512, 256, 1, 1, 1, 0), ('layer3.0.downsample', 256, 28, 28, 512, 1024, 1, 2, 0, 0),
          ('layer4.0.conv1', 256, 14, 14, 1024, 512, 1, 1, 1, 0), ('layer4.x.conv2 3x3', 256, 7, 7, 512, 512, 3, 1, 1, 0),
          ('layer4.x.conv1', 256, 7, 7, 2048, 512, 1, 1, 1, 0), ('layer4.0.downsample', 256, 14, 14, 1024, 2048, 1, 2, 0, 0),
          ('layer3.x.conv3 +res', 256, 14, 14, 256, 1024, 1, 1, 1, 1), ('layer4.x.conv3 +res', 256, 7, 7, 512, 2048, 1, 1, 1, 1),
          ('ViT QKV', 256, 197, 1, 768, 2304, 1, 1, 0, 0), ('ViT FC1+QuickGELU', 256, 197, 1, 768, 3072, 1, 1, 2, 0), ('ViT FC2', 256, 197, 1, 3072, 768, 1, 1, 0, 0)]
vp = lambda t: C.c_void_p(t.data_ptr())
for name, n, h, w, cin, cout, k, stride, act, res in shapes:
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn((n, h, w, cin), device='cuda').bfloat16()
    wk = (torch.randn((cout, k * k * cin), device='cuda') * (2.0 / (cin * k * k)) ** 0.5).bfloat16()
    b = torch.randn(cout, device='cuda')
    r = torch.randn((n, ho, wo, cout), device='cuda').bfloat16() if res else None
    out = torch.empty((n, ho, wo, cout), device='cuda', dtype=torch.bfloat16)
    run = lambda: _lib.check(L.pvr_op_conv2d(vp(x), vp(wk), vp(b), vp(r) if res else None, vp(out), n, h, w, cin, cout, k, k, stride, pad, act, 0, _lib.PVR_BF16, _lib.stream_ptr()))
    res_us, outs = {}, {}
    for algo in (-1, 4):
        _lib.check(L.pvr_debug_set_conv_algo(algo))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res_us[algo] = e0.elapsed_time(e1) / 20 * 1e3
        outs[algo] = out.clone()
    _lib.check(L.pvr_debug_set_conv_algo(-1))
    gf = 2.0 * n * ho * wo * cout * k * k * cin
    print('%-22s auto %7.1f us %6.0f TFLOP/s | w4 %7.1f us %6.0f TFLOP/s | %+5.1f %% | equal %s' % (
        name, res_us[-1], gf / res_us[-1] / 1e6, res_us[4], gf / res_us[4] / 1e6, (res_us[4] / res_us[-1] - 1) * 100, bool(torch.equal(outs[-1], outs[4]))), flush=True)
