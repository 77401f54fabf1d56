import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from pvr_habitat_amd import synth
from pvr_habitat_amd.models import PolicyNetWithConv, HipRMSprop
torch.manual_seed(0)
net = PolicyNetWithConv((64, 64, 6), 4, batch_norm=True).to(device='cuda')
g = torch.Generator().manual_seed(1)
o = torch.randint(0, 256, (20, 4, 64, 64, 6), generator=g, dtype=torch.uint8).cuda()
d = (torch.rand((20, 4), generator=g) < 0.05).cuda(); a = torch.randint(0, 4, (20, 4), generator=g).cuda()
opt = HipRMSprop(net, lr=1e-4, alpha=0.99, eps=1e-5, max_grad_norm=40.0, max_epochs=100)
out = []
for i in range(6):
    opt.scheduler_step(); l, gn = opt.step(o, d, a); out.append((float(l), float(gn)))
print(out)
