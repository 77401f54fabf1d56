"""The BC iteration alone (bench.py's bc leg: T=100, B=16, obs 4096, BN, RMSprop), for rocprofv3 --kernel-trace --stats.
python scripts/bc_only.py [steps] [conv]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pvr_habitat_amd import synth
from pvr_habitat_amd.models import PolicyNet, PolicyNetWithConv, HipRMSprop
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
conv = len(sys.argv) > 2 and sys.argv[2] == 'conv'
T, B, O, A = 100, 16, 4096, 3
if conv:
    m = PolicyNetWithConv((64, 64, 6), A, True, max_unroll=T, max_batch=B); O = 256
else:
    m = PolicyNet((O,), A, True, max_unroll=T, max_batch=B)
sd = synth.policy_state_dict(1, O, A, True, conv=conv)
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
m = m.to('cuda').train()
opt = HipRMSprop(m, max_epochs=10 ** 6)
obs, done, act = synth.bc_conv_batches(1, T, B, 2, A) if conv else synth.bc_batches(1, T, B, O, A, 2)
o, d, a = torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), torch.from_numpy(act).cuda()
for i in range(4):
    opt.scheduler_step(); opt.step(o[i % 2], d[i % 2], a[i % 2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    opt.scheduler_step(); opt.step(o[i % 2], d[i % 2], a[i % 2])
torch.cuda.synchronize(); el = time.perf_counter() - t0
m.check_status()
print('%s: %.1f steps/s (%.3f ms/step)' % ('PolicyNetWithConv' if conv else 'PolicyNet', steps / el, el / steps * 1e3))
