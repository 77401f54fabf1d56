"""A/B of the persistent BPTT against the per-step launches: per-tensor max |diff| of the first step's gradients, several shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from pvr_habitat_amd import synth
from pvr_habitat_amd.models import HipRMSprop
from test_gpu_policy import _model

for T, B in ((3, 4), (6, 4), (6, 16), (8, 4), (24, 4), (24, 16), (24, 20), (26, 5), (100, 16)):
    O, A = 256, 3
    obs, done, act = synth.bc_batches(6, T, B, O, A, 2)
    res = {}
    for bwd in ('0', '1'):
        os.environ['PVR_POLICY_PERSIST'] = '2'; os.environ['PVR_POLICY_PERSIST_BWD'] = bwd
        m, _ = _model(6, O, A, True, T, B)
        opt = HipRMSprop(m, max_epochs=50); m.train()
        out = []
        for s in range(2):
            opt.scheduler_step(); l, g = opt.step(torch.from_numpy(obs[s]), torch.from_numpy(done[s]), torch.from_numpy(act[s]))
            out.append({k: v.clone() for k, v in m.last_grads().items()})
        torch.cuda.synchronize(); m.check_status(); m.close()
        res[bwd] = out
    for s in range(2):
        bad = {k: float((res['0'][s][k] - res['1'][s][k]).abs().max() / (res['0'][s][k].abs().max() + 1e-30)) for k in res['0'][s]
               if not torch.equal(res['0'][s][k], res['1'][s][k])}
        print('T=%d B=%d step %d:' % (T, B, s), 'identical' if not bad else {k: '%.1e' % v for k, v in bad.items()}, flush=True)
