"""Pin oracle/policy_oracle.py against fixtures produced by the reference's own src/models.py +
torch.optim.RMSprop (tests/golden/make_golden.py).  CPU only."""
import os, random
import numpy as np
import pytest
import torch

from pvr_habitat_amd import synth
from oracle import policy_oracle as po


def _run(npz, seed, batch_norm, conv=False):
    g = np.load(npz)
    T, B, A, S = int(g['T']), int(g['B']), int(g['A']), int(g['steps'])
    if conv:
        sd = synth.policy_state_dict(seed, 256, A, batch_norm, conv=True)
        obs, done, act = synth.bc_conv_batches(seed, T, B, S, A)
    else:
        O = int(g['O'])
        sd = synth.policy_state_dict(seed, O, A, batch_norm)
        obs, done, act = synth.bc_batches(seed, T, B, O, A, S)
    p = po.to_params(sd)
    opt = po.RMSpropState(p, max_epochs=int(g['max_epochs']))
    for s in range(S):
        loss, gn, logits = po.bc_step(p, opt, torch.from_numpy(obs[s]), torch.from_numpy(done[s]),
                                      torch.from_numpy(act[s]), batch_norm, conv=conv)
        assert loss == pytest.approx(float(g['loss'][s]), rel=2e-5)
        assert gn == pytest.approx(float(g['grad_norm'][s]), rel=2e-4)
        np.testing.assert_allclose(logits.numpy(), g['logits'][s], rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        H = 1024
        out, st = po.forward(p, torch.from_numpy(obs[0]), torch.from_numpy(done[0]),
                             (torch.zeros(2, B, H), torch.zeros(2, B, H)), batch_norm, training=False, conv=conv)
    np.testing.assert_allclose(out['policy_logits'].numpy(), g['eval_logits'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out['baseline'].numpy(), g['eval_baseline'], rtol=1e-4, atol=2e-5)
    assert np.array_equal(out['action'].numpy(), g['eval_action'])          # exact action indices
    np.testing.assert_allclose(st[0].numpy(), g['eval_h'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(st[1].numpy(), g['eval_c'], rtol=1e-4, atol=2e-4)   # 100-step accumulation
    keys = [str(k) for k in g['param_keys']]
    for k, s1, s2 in zip(keys, g['param_sum'], g['param_sq']):
        v = p[k].detach().double()
        assert float(v.sum()) == pytest.approx(float(s1), rel=1e-5, abs=1e-4), k
        assert float((v ** 2).sum()) == pytest.approx(float(s2), rel=1e-5, abs=1e-6), k
    for k in g.files:
        if k.startswith('final/'):
            np.testing.assert_allclose(p[k[6:]].detach().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_policy_small_bn(golden_dir):
    _run(os.path.join(golden_dir, 'policy_small_bn.npz'), 1, True)


def test_policy_small_nobn(golden_dir):
    _run(os.path.join(golden_dir, 'policy_small_nobn.npz'), 2, False)


def test_policy_conv_small(golden_dir):
    _run(os.path.join(golden_dir, 'policy_conv_small.npz'), 3, True, conv=True)


def test_policy_full_bn(golden_dir):
    torch.set_num_threads(8)
    _run(os.path.join(golden_dir, 'policy_full_bn.npz'), 1, True)


def test_sampler(golden_dir):
    g = np.load(os.path.join(golden_dir, 'sampler.npz'))
    rng = random.Random(1)
    for row in g['seed1_n5000_k16_d100']:
        s = po.sample_with_minimum_distance(rng, n=5000, k=16, d=100)
        assert s == list(row)
        ss = sorted(s)
        assert min(b - a for a, b in zip(ss, ss[1:])) >= 100
    rng = random.Random(7)
    for row in g['seed7_n40_k4_d10']:
        assert po.sample_with_minimum_distance(rng, n=40, k=4, d=10) == list(row)
