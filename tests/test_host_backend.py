"""The library's HOST backend (csrc/host_encoder.hip): BASELINE configs[0] embeds saved frames ON CPU, and the reference picks the CPU
when disable_cuda is set or no GPU is present (src/embeddings.py:367-370).  pvr_encoder_set_host_backend keeps the same C-ABI - create /
load_weights / finalize / forward - with host pointers and plain C++ loops.  These tests run WITHOUT a GPU (the CPU suite): the host plan
against the torch oracle (never the other way round: the oracle is only the checker), the EmbeddingNet / save_embedded_obs surface on top
of it, and the refusals."""
import os
import pickle

import numpy as np
import pytest
import torch

from pvr_habitat_amd import _lib, synth


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize('variant,n,h,w', [('conv5', 2, 64, 64), ('conv4', 1, 256, 256), ('conv3', 2, 128, 128), ('r18', 2, 64, 80)])
def test_host_encoder_matches_the_fp32_oracle(variant, n, h, w):
    """fp32 C++ loops over the encoder's own op list (BN folded, eps 1e-5) vs the torch restatement of the reference's model: transforms
    (bilinear Resize with uint8 rounding, CenterCrop, /255, Normalize), stem, maxpool, 53 (or 20) convolutions, pooled or C-major head."""
    from oracle import encoder_oracle as eo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(8)
    sd = synth.resnet50_state_dict(5, variant)
    fr = synth.smooth_frames(17, n, h, w)
    ref = eo.embed(sd, fr, variant, squeeze=False)
    m = HipResNet50(sd, variant, max_batch=4, host=True)
    out = m(torch.from_numpy(fr)).numpy()
    assert out.shape == ref.shape and np.isfinite(out).all()
    l2, mx = _rel(out, ref)
    print('\n[host %s %dx%d] rel-L2 %.2e max-norm %.2e' % (variant, h, w, l2, mx))
    assert l2 < 1e-4 and mx < 1e-3                       # fp32 both sides; a Resize tie may land one uint8 step apart in a few pixels
    again = m(torch.from_numpy(fr)).numpy()
    assert np.array_equal(out, again)                     # fixed summation order whatever the thread schedule
    m.close()


def test_embedding_net_disable_cuda_runs_on_the_host_backend(tmp_path, monkeypatch):
    """EmbeddingNet(..., disable_cuda=True) (reference signature, embeddings.py:345) -> device cpu, numpy fp32 (N, 2048), squeeze for N = 1;
    and the first half of BASELINE configs[0]: save_embedded_obs.run on a scene pickle of 128 x 128 frames with --disable_cuda."""
    monkeypatch.setenv('PVR_SYNTHETIC_WEIGHTS', '1')
    from pvr_habitat_amd.embeddings import EmbeddingNet
    from pvr_habitat_amd import save_embedded_obs as S
    net = EmbeddingNet('resnet50', pretrained=False, disable_cuda=True, max_batch=4)
    assert net.device == torch.device('cpu') and net.out_size == 2048
    fr = synth.smooth_frames(3, 2, 128, 128)
    a = net(torch.from_numpy(fr))
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.shape == (2, 2048)
    assert net(torch.from_numpy(fr[:1])).shape == (2048,)
    net.close()
    rng = np.random.default_rng(1)
    L = 3
    raw = dict(obs=[rng.integers(0, 256, (L, 128, 128, 6), dtype=np.uint8)], action=[np.arange(L)], reward=[np.ones(L)],
               done=[np.arange(L) == L - 1], true_state=[np.zeros((L, 12), np.float32)])
    pickle.dump(raw, open(tmp_path / 'scene.pickle', 'wb'))
    S.run(S.make_parser().parse_args(['--data_path', str(tmp_path), '--env', 'scene', '--embedding_name', 'resnet50', '--disable_pretrained_embedding',
                                      '--disable_cuda', '--source', 'pickle', '--embed_batch', '4']))
    out = pickle.load(open(tmp_path / 'scene_resnet50.pickle', 'rb'))
    assert out['obs'].shape == (L, 4096) and out['obs'].dtype == np.float32 and np.isfinite(out['obs']).all()
    # current frame | goal frame, each embedded by the same network (save_embedded_obs.py:151-156)
    net = EmbeddingNet('resnet50', pretrained=False, disable_cuda=True, max_batch=4)
    cur = net(torch.from_numpy(np.ascontiguousarray(raw['obs'][0][..., :3])))
    goal = net(torch.from_numpy(np.ascontiguousarray(raw['obs'][0][..., 3:])))
    assert np.allclose(out['obs'][:, :2048], cur, rtol=1e-5, atol=1e-6) and np.allclose(out['obs'][:, 2048:], goal, rtol=1e-5, atol=1e-6)


def test_host_backend_refusals():
    """ViT / CLIP encoders and 16-bit plans have no CPU form: finalize says so; GPU-plan features are refused on a host encoder."""
    import ctypes as C
    from pvr_habitat_amd.embeddings import HipResNet50
    L = _lib.lib()
    with pytest.raises(NotImplementedError, match='host backend'):
        HipResNet50(synth.clip_vit_state_dict(1, patch=32), 'clip_b32', max_batch=2, host=True)(torch.zeros((1, 64, 64, 3), dtype=torch.uint8))
    m = HipResNet50(synth.resnet50_state_dict(2, 'conv5'), 'conv5', max_batch=2, host=True)
    m(torch.zeros((1, 64, 64, 3), dtype=torch.uint8))
    cnt = C.c_int64()
    buf = np.zeros(16, np.float32)
    assert L.pvr_encoder_tap(m._handle, b'layer1', buf.ctypes.data, 16, C.byref(cnt), None) != 0 and 'host-backend' in _lib.last_error()
    m.close()


@pytest.mark.parametrize('name,seed,bn', [('policy_small_bn.npz', 1, True), ('policy_small_nobn.npz', 2, False)])
def test_host_policy_matches_the_reference_fixtures(name, seed, bn):
    """pvr_policy_create_host: the fused BC iteration (main_bc_2.py:206-227) and the eval forward (models.py:57-89) as plain C++ on the CPU,
    against the fixtures the REFERENCE's own PolicyNet + torch.optim.RMSprop produced (tests/golden/make_golden.py): loss, gradient norm,
    logits of every update, every parameter's checksum after the updates, eval logits / state and the exact argmax actions."""
    from pvr_habitat_amd.models import PolicyNet, HipRMSprop
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', name))
    T, B, O, A, S = int(g['T']), int(g['B']), int(g['O']), int(g['A']), int(g['steps'])
    sd = synth.policy_state_dict(seed, O, A, bn)
    m = PolicyNet((O,), A, bn, max_unroll=T, max_batch=B)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    assert m._host and m.device == torch.device('cpu')
    with pytest.raises(RuntimeError, match='use_host_backend'):              # parameters left on the CPU are an error unless the host plan is asked for
        m(dict(obs=torch.zeros(T, B, O), done=torch.zeros(T, B, dtype=torch.bool)), m.initial_state(B))
    m.use_host_backend(True)
    m.train()
    opt = HipRMSprop(m, lr=1e-4, alpha=0.99, eps=1e-5, max_grad_norm=40.0, max_epochs=int(g['max_epochs']))
    obs, done, act = synth.bc_batches(seed, T, B, O, A, S)
    for s in range(S):
        opt.scheduler_step()
        loss, gn, logits = opt.step(torch.from_numpy(obs[s]), torch.from_numpy(done[s]), torch.from_numpy(act[s]), return_logits=True)
        assert float(loss) == pytest.approx(float(g['loss'][s]), rel=2e-5), s
        assert float(gn) == pytest.approx(float(g['grad_norm'][s]), rel=3e-4), s
        np.testing.assert_allclose(logits.numpy(), g['logits'][s], rtol=1e-4, atol=5e-5)
    sdm = m.state_dict()
    for k, s1, s2 in zip([str(k) for k in g['param_keys']], g['param_sum'], g['param_sq']):
        v = sdm[k].double()
        assert float(v.sum()) == pytest.approx(float(s1), rel=1e-5, abs=2e-4), k
        assert float((v ** 2).sum()) == pytest.approx(float(s2), rel=1e-5, abs=1e-6), k
    m.eval()
    with torch.no_grad():
        out, st = m(dict(obs=torch.from_numpy(obs[0]), done=torch.from_numpy(done[0])), m.initial_state(B))
    np.testing.assert_allclose(out['policy_logits'].numpy(), g['eval_logits'], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(out['baseline'].numpy(), g['eval_baseline'], rtol=1e-4, atol=5e-5)
    assert np.array_equal(out['action'].numpy(), g['eval_action'])                        # bit-exact action indices
    np.testing.assert_allclose(st[0].numpy(), g['eval_h'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(st[1].numpy(), g['eval_c'], rtol=1e-4, atol=3e-4)
    m.close()


def test_config0_pipeline_without_a_gpu(tmp_path, monkeypatch):
    """BASELINE configs[0] as written: ResNet50 embeds saved 128 x 128 frames ON CPU via save_embedded_obs, then main_bc_2's policy trains on
    the resulting scene - every step on the library's host backend (no GPU in this suite), through the reference's flags and file formats."""
    from pvr_habitat_amd import save_embedded_obs as S, main_bc_2 as M
    from pvr_habitat_amd.arguments import make_parser
    monkeypatch.setenv('PVR_SYNTHETIC_WEIGHTS', '1')
    lens = (14, 11)
    fr = synth.smooth_frames(31, 2 * sum(lens), 128, 128)
    obs_all = np.concatenate([fr[:sum(lens)], fr[sum(lens):]], axis=3)
    cuts = np.cumsum((0,) + lens)
    rng = np.random.default_rng(0)
    raw = dict(obs=[obs_all[a:b] for a, b in zip(cuts[:-1], cuts[1:])], action=[rng.integers(0, 3, L) for L in lens],
               reward=[np.zeros(L, np.float32) for L in lens], done=[np.eye(1, L, L - 1, dtype=bool)[0] for L in lens],
               true_state=[np.zeros((L, 12), np.float32) for L in lens])
    pickle.dump(raw, open(tmp_path / 'scene.pickle', 'wb'))
    args = ['--data_path', str(tmp_path), '--save_path', str(tmp_path / 'bc'), '--env', 'scene', '--to_env', 'scene', '--embedding_name', 'resnet50',
            '--source', 'pickle', '--embed_batch', '8', '--disable_cuda', '--unroll_length', '4', '--batch_size', '2', '--batch_norm',
            '--max_frames', '40', '--eval_frequency', '2']
    S.run(make_parser().parse_args(args))
    out = pickle.load(open(tmp_path / 'scene_resnet50.pickle', 'rb'))
    assert out['obs'].shape == (25, 4096) and np.isfinite(out['obs']).all()
    stats = M.run(make_parser().parse_args(args))
    st = stats['scene']
    assert len(st['frames']) >= 2 and all(np.isfinite(st['training_loss'][1:])) and all(np.isfinite(st['gradient_norm'][1:]))
    ck = torch.load(tmp_path / 'bc' / 'scene_emresnet50_s1_scene.tar', weights_only=False)
    assert ck['actor_model_state_dict']['fc.1.weight'].shape == (1024, 4096)


def _sampling_check(m, dev, T=40, B=16, O=64, calls=12):
    """shared by the host (CPU) and the GPU test: training-mode actions are samples of softmax(logits) (reference models.py:78-80)"""
    g = torch.Generator().manual_seed(5)
    obs = torch.randn(T, B, O, generator=g).to(dev)
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    A = m.num_actions
    m.train()
    counts = np.zeros(A)
    acts = []
    with torch.no_grad():
        for _ in range(calls):
            out, _ = m(dict(obs=obs, done=done), m.initial_state(B))
            a = out['action'].cpu().numpy()
            assert a.shape == (T, B) and a.min() >= 0 and a.max() < A
            acts.append(a)
            counts += np.bincount(a.ravel(), minlength=A)
        p = torch.softmax(out['policy_logits'].float().cpu().view(T * B, A), dim=1).numpy()
    assert p.max() > 0.5 and p.min() < 0.05, 'the test needs skewed rows to mean anything'
    expect = p.sum(0) * calls
    sigma = np.sqrt((p * (1 - p)).sum(0) * calls)
    assert np.all(np.abs(counts - expect) < 5 * sigma + 1), (counts, expect, sigma)
    # rows with one dominant action: the sample is that action (almost) always, a uniform draw would not be
    dom = p.max(1) > 0.999
    if dom.any():
        hit = np.mean([(a.ravel()[dom] == p.argmax(1)[dom]).mean() for a in acts])
        assert hit > 0.99
    assert any(not np.array_equal(acts[0], a) for a in acts[1:]), 'every call must draw new samples'
    # the stream restarts with the seed: same seed -> same actions, another seed -> others
    # the noise is a function of torch's global generator at the call, as the reference's torch.multinomial is: re-seeding restarts it,
    # another seed gives other actions, a second module (or a rebuilt handle) continues the generator's stream instead of replaying
    def draw(seed, net=None):
        if seed is not None:
            torch.manual_seed(seed)
        with torch.no_grad():
            return (net or m)(dict(obs=obs, done=done), m.initial_state(B))[0]['action'].cpu().numpy()
    a1, a1b, a2, a3 = draw(77), draw(None), draw(77), draw(78)
    assert np.array_equal(a1, a2) and not np.array_equal(a1, a3) and not np.array_equal(a1, a1b)
    m2 = _sampling_net()
    m2 = m2.to(dev) if dev != 'cpu' else m2.use_host_backend(True)
    m2.train()
    torch.manual_seed(77)
    b1, b2 = draw(None), draw(None, m2)                       # two modules, one generator: the second does not repeat the first
    assert np.array_equal(b1, a1) and not np.array_equal(b2, b1)
    m._release()                                              # a rebuilt handle does not replay either
    assert not np.array_equal(draw(None), b1)
    m2.close()
    # the library's stream position travels with a handle replacement for hosts that key it once (pvr_policy_action_sampling_call)
    from pvr_habitat_amd.models import _plib
    import ctypes as C
    L = _plib()
    assert L.pvr_policy_set_action_sampling(m._handle, 1, C.c_uint64(5)) == 0 and int(L.pvr_policy_action_sampling_call(m._handle)) == 0
    assert L.pvr_policy_set_action_sampling_call(m._handle, C.c_uint64(41)) == 0 and int(L.pvr_policy_action_sampling_call(m._handle)) == 41
    m.eval()
    with torch.no_grad():
        out, _ = m(dict(obs=obs, done=done), m.initial_state(B))
    assert np.array_equal(out['action'].cpu().numpy().ravel(), out['policy_logits'].float().cpu().view(T * B, A).argmax(1).numpy())


def _sampling_net(O=64, A=6):
    from pvr_habitat_amd.models import PolicyNet
    m = PolicyNet((O,), A, False, max_unroll=40, max_batch=16)
    sd = synth.policy_state_dict(9, O, A, False)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    sd['policy.weight'] = sd['policy.weight'] * 60.0                 # skewed rows: some near one-hot, some spread
    m.load_state_dict(sd)
    return m


def test_host_policy_samples_training_actions_from_softmax():
    """pvr_policy_set_action_sampling on the host plan: the training-mode action is a sample of softmax(logits) drawn in the library
    (no torch.multinomial on the path), frequencies match the probabilities, eval stays argmax."""
    m = _sampling_net()
    m.use_host_backend(True)
    _sampling_check(m, 'cpu')
    m.close()


def test_host_policy_step_rejects_a_bad_batch_before_touching_state():
    """an out-of-range target action fails the step BEFORE the forward has updated the BatchNorm running statistics (the HIP plan and torch reject the
    batch up front too); and BatchNorm in training mode refuses a single row, as nn.BatchNorm1d does"""
    from pvr_habitat_amd.models import PolicyNet, HipRMSprop
    O, A, T, B = 64, 3, 4, 2
    m = PolicyNet((O,), A, True, max_unroll=T, max_batch=B)
    sd = synth.policy_state_dict(3, O, A, True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.use_host_backend(True).train()
    obs, done, act = synth.bc_batches(3, T, B, O, A, 1)
    opt = HipRMSprop(m, max_epochs=10)
    bad = act[0].copy(); bad[1, 1] = A
    before = {k: v.clone() for k, v in m.state_dict().items()}
    with pytest.raises(RuntimeError, match='outside 0'):
        opt.step(torch.from_numpy(obs[0]), torch.from_numpy(done[0]), torch.from_numpy(bad))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k                   # running_mean / running_var / num_batches_tracked / parameters untouched
    with pytest.raises(RuntimeError, match='more than one row'):
        with torch.no_grad():
            m(dict(obs=torch.from_numpy(obs[0][:1, :1]), done=torch.zeros(1, 1, dtype=torch.bool)), m.initial_state(1))
    m.eval()
    with torch.no_grad():
        out, _ = m(dict(obs=torch.from_numpy(obs[0][:1, :1]), done=torch.zeros(1, 1, dtype=torch.bool)), m.initial_state(1))
    assert out['action'].shape == (1, 1)
    m.close()
