"""GPU parity of the CLIP ViT path (B/32 as the reference loads it, B/16 as BASELINE config 3) vs the fp32 oracle."""
import numpy as np
import pytest
import torch

from pvr_habitat_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize('patch,variant', [(32, 'clip_b32'), (16, 'clip_b16')])
@pytest.mark.parametrize('dt,tol', [('f16', 1e-3), ('bf16', 1e-2)])
def test_clip_vit_matches_oracle(patch, variant, dt, tol):
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.clip_vit_state_dict(1, patch=patch)
    fr = synth.smooth_frames(41, 3, 224, 224)
    ref = vo.embed(sd, fr, squeeze=False)
    m = HipResNet50(sd, variant, compute_dtype=dt, max_batch=4)
    out = m(torch.from_numpy(fr).cuda()).cpu().numpy()
    assert out.shape == (3, 512) and np.isfinite(out).all()
    l2, mx = _rel(out, ref)
    print('\n[%s %s] rel-L2 %.2e max-norm %.2e' % (variant, dt, l2, mx))
    assert l2 < tol and mx < 2 * tol
    one = m(torch.from_numpy(fr[1:2]).cuda()).cpu().numpy()
    np.testing.assert_array_equal(one[0], out[1])                # batch-composition invariance, bit-exact


def test_clip_embeddingnet_surface(monkeypatch):
    from pvr_habitat_amd.embeddings import EmbeddingNet
    monkeypatch.setenv('PVR_SYNTHETIC_WEIGHTS', '1')
    monkeypatch.setenv('PVR_DTYPE', 'f16')
    net = EmbeddingNet('clip_vit', max_batch=8)
    assert net.out_size == 512
    fr = synth.frames(3, 2, 256, 224)                            # centre crop of the long side, no resize needed
    out = net(torch.from_numpy(fr))
    assert out.shape == (2, 512) and out.dtype == np.float32


@pytest.mark.parametrize('h,w', [(64, 64), (256, 256), (100, 75), (128, 160)])
def test_clip_antialiased_bicubic_resize(h, w):
    """embeddings.py:310: Resize(224, BICUBIC, antialias=True) on uint8 frames (Habitat renders 64x64)."""
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.clip_vit_state_dict(1, patch=32)
    fr = synth.smooth_frames(43, 2, h, w)
    m = HipResNet50(sd, 'clip_b32', compute_dtype='f16', max_batch=4)
    d = torch.from_numpy(fr).cuda()
    out = m(d).cpu().numpy()
    u8 = vo.preprocess_u8(fr).permute(0, 2, 3, 1).numpy().astype(np.float32)      # (N,224,224,3)
    got = m.tap('resized', u8.size).cpu().numpy().reshape(u8.shape)
    diff = np.abs(got - u8)
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3          # .5 ties may round the other way (fp32 summation order)
    ref = vo.embed(sd, fr, squeeze=False)
    l2 = float(np.linalg.norm(out - ref) / np.linalg.norm(ref))
    assert l2 < 1e-3, l2


@pytest.mark.parametrize('h,w', [(224, 224), (64, 64), (300, 256)])
def test_mae_vit_b16_matches_oracle(h, w):
    """MAE ViT-B/16 encoder (mae.py:202-222), Resize(256, bicubic) + CenterCrop(224), CLS output 768."""
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.mae_vit_state_dict(1)
    fr = synth.smooth_frames(47, 2, h, w)
    ref = vo.mae_embed(sd, fr, squeeze=False)
    m = HipResNet50(sd, 'mae_b16', compute_dtype='f16', max_batch=4)
    out = m(torch.from_numpy(fr).cuda()).cpu().numpy()
    assert out.shape == (2, 768)
    if min(h, w) != 256:                                         # short side 256: Resize is the identity, nothing to tap
        u8 = vo.mae_preprocess_u8(fr).permute(0, 2, 3, 1).numpy().astype(np.float32)
        got = m.tap('resized', u8.size).cpu().numpy().reshape(u8.shape)
        d = np.abs(got - u8)
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    l2, mx = _rel(out, ref)
    print('\n[mae_b16 f16 %dx%d] rel-L2 %.2e max-norm %.2e' % (h, w, l2, mx))
    assert l2 < 1e-3


def test_mae_vit_l16_matches_oracle():
    """MAE ViT-L/16 encoder ('mae_large': width 1024, 24 blocks, 16 heads; mae.py:283-288)."""
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.mae_vit_state_dict(2, width=1024, layers=24)
    fr = synth.smooth_frames(48, 2, 128, 128)
    ref = vo.mae_embed(sd, fr, squeeze=False, heads=16)
    m = HipResNet50(sd, 'mae_l16', compute_dtype='f16', max_batch=4)
    out = m(torch.from_numpy(fr).cuda()).cpu().numpy()
    assert out.shape == (2, 1024)
    l2, mx = _rel(out, ref)
    print('\n[mae_l16 f16] rel-L2 %.2e max-norm %.2e' % (l2, mx))
    assert l2 < 1e-3


def test_mae_vit_h14_matches_oracle():
    """MAE ViT-H/14 encoder ('mae_huge': width 1280, 32 blocks, 16 heads of dim 80, patch 14 -> 257 tokens;
    mae.py:291-296): second attention instantiation (head dim 80 padded to 96 in QK^T, 18 key tiles) and zero-padded patch rows."""
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.mae_vit_state_dict(3, patch=14, width=1280, layers=32)
    fr = synth.smooth_frames(49, 2, 96, 96)
    ref = vo.mae_embed(sd, fr, squeeze=False, heads=16)
    m = HipResNet50(sd, 'mae_h14', compute_dtype='f16', max_batch=4)
    out = m(torch.from_numpy(fr).cuda()).cpu().numpy()
    assert out.shape == (2, 1280)
    l2, mx = _rel(out, ref)
    print('\n[mae_h14 f16] rel-L2 %.2e max-norm %.2e' % (l2, mx))
    assert l2 < 1e-3


@pytest.mark.parametrize('h,w', [(224, 224), (64, 64), (96, 128)])
def test_clip_rn50_matches_oracle(h, w):
    """CLIP RN50 visual tower (embeddings.py:305-306): 3-conv stem, AvgPool2d strides, attention pool -> 1024."""
    from oracle import vit_oracle as vo
    from pvr_habitat_amd.embeddings import HipResNet50
    torch.set_num_threads(16)
    sd = synth.clip_rn50_state_dict(4)
    fr = synth.smooth_frames(50, 3, h, w)
    taps = {}
    ref = vo.clip_rn50_embed(sd, fr, squeeze=False, taps=taps)
    m = HipResNet50(sd, 'clip_rn50', compute_dtype='f16', max_batch=4)
    d = torch.from_numpy(fr).cuda()
    errs = {}
    for name in ('stem3', 'layer1', 'layer2', 'layer3', 'layer4'):
        m.debug_stop_after(name); m(d)
        r = taps['stem' if name == 'stem3' else name].permute(0, 2, 3, 1).contiguous().numpy()
        g = m.tap(name, r.size).cpu().numpy().reshape(r.shape)
        errs[name] = _rel(g, r)[0]
    m.debug_stop_after('')
    out = m(d).cpu().numpy()
    assert out.shape == (3, 1024)
    l2, mx = _rel(out, ref)
    print('\n[clip_rn50 f16 %dx%d] stages %s embedding rel-L2 %.2e max-norm %.2e' % (h, w, {k: '%.1e' % v for k, v in errs.items()}, l2, mx))
    assert l2 < 1e-3 and max(errs.values()) < 1e-3
