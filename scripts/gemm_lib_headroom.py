"""How fast does the vendor library run the ViT-B/16 linear layers' GEMM shapes on this box?  (headroom check for conv_pp256 on the transformer GEMMs;
not part of the product: the product's GEMMs are conv_pp256.hip)  M = 256 frames x 197 tokens."""
import os, sys, time
import torch
M = 256 * 197
shapes = [('qkv', 768, 2304), ('proj', 768, 768), ('fc1', 768, 3072), ('fc2', 3072, 768)]
for dt in (torch.float16, torch.bfloat16):
    for name, K, N in shapes:
        a = torch.randn(M, K, device='cuda', dtype=dt); w = torch.randn(N, K, device='cuda', dtype=dt)
        for _ in range(5): torch.mm(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): torch.mm(a, w.t())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%s %-5s M=%d K=%d N=%d  %.1f us  %.0f TFLOP/s' % (str(dt).split('.')[-1], name, M, K, N, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
