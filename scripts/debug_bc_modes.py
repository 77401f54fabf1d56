import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from pvr_habitat_amd import synth
from pvr_habitat_amd.models import HipRMSprop
import test_gpu_policy as tp
def run(graph, persist, conv, S=6):
    os.environ['PVR_POLICY_GRAPH'] = graph; os.environ['PVR_POLICY_PERSIST'] = persist
    T, B, O, A = 12, 8, 256, 3
    obs, done, act = synth.bc_conv_batches(5, T, B, S, A) if conv else synth.bc_batches(5, T, B, O, A, S)
    m, _ = tp._model(5, O, A, True, T, B, conv)
    opt = HipRMSprop(m, max_epochs=50); m.train(); st = []
    for s in range(S):
        opt.scheduler_step()
        l, g = opt.step(torch.from_numpy(obs[s]), torch.from_numpy(done[s]), torch.from_numpy(act[s]))
        st.append((round(float(l), 6), round(float(g), 5)))
    return m._flat.clone(), st
for conv in (True, False):
    ref, st0 = run('0', '0', conv)
    for graph, persist in (('0', '2'), ('0', '2'), ('1', '2'), ('1', '0'), ('1', '2')):
        f, st = run(graph, persist, conv)
        print('conv', conv, 'graph', graph, 'persist', persist, 'equal to eager/per-step:', bool(torch.equal(f, ref)), [a == b for a, b in zip(st, st0)])
