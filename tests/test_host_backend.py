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
