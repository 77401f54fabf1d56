"""BASELINE config 4 on one GPU: the end-to-end BC iteration of main_bc_finetune.py (PolicyNetWithConv: 5 x (conv3x3 s2 + ELU) on
every frame of the (T, B) batch + the PolicyNet step), T = 100, B = 16, 64x64x6 uint8 observations (2 frames per observation)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd.models import PolicyNetWithConv, HipRMSprop
T, B = 100, 16
torch.manual_seed(0)
net = PolicyNetWithConv((64, 64, 6), 4, True, max_unroll=T, max_batch=B).to(device='cuda')
opt = HipRMSprop(net, lr=1e-4, alpha=0.99, eps=1e-5, max_grad_norm=40.0, max_epochs=1000)
g = torch.Generator().manual_seed(1)
o = torch.randint(0, 256, (T, B, 64, 64, 6), dtype=torch.uint8, generator=g).cuda()
d = (torch.rand((T, B), generator=g) < 0.02).cuda(); a = torch.randint(0, 4, (T, B), generator=g).cuda()
for _ in range(3):
    opt.scheduler_step(); l, gn = opt.step(o, d, a)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    opt.scheduler_step(); l, gn = opt.step(o, d, a)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print('finetune step (T=%d, B=%d, 64x64x6 uint8, BN): %.1f steps/s, %.2f ms/step, %.0f frames/s through the conv stack; loss %.4f' % (T, B, n / el, el / n * 1e3, n * T * B * 2 / el, float(l)))
