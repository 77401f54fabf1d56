"""Pin oracle/encoder_oracle.py: topology + primitive semantics against an independent
implementation (transformers.ResNetModel, v1.5 bottleneck) with name-remapped synthetic
weights, plus properties of the restated torchvision transforms.  CPU only."""
import numpy as np
import pytest
import torch

from pvr_habitat_amd import synth
from oracle import encoder_oracle as eo


def _hf_resnet50(sd):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(layer_type='bottleneck', hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                       embedding_size=64, downsample_in_bottleneck=False)
    m = ResNetModel(cfg).eval()
    new = {}

    def bn(dst, src):
        for a in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
            new[dst + '.' + a] = torch.from_numpy(np.array(sd[src + '.' + a]))

    new['embedder.embedder.convolution.weight'] = torch.from_numpy(sd['conv1.weight'])
    bn('embedder.embedder.normalization', 'bn1')
    for li, nb in enumerate((3, 4, 6, 3)):
        for bi in range(nb):
            src = 'layer%d.%d' % (li + 1, bi)
            dst = 'encoder.stages.%d.layers.%d' % (li, bi)
            for ci in range(3):
                new['%s.layer.%d.convolution.weight' % (dst, ci)] = torch.from_numpy(sd['%s.conv%d.weight' % (src, ci + 1)])
                bn('%s.layer.%d.normalization' % (dst, ci), '%s.bn%d' % (src, ci + 1))
            if bi == 0:
                new[dst + '.shortcut.convolution.weight'] = torch.from_numpy(sd[src + '.downsample.0.weight'])
                bn(dst + '.shortcut.normalization', src + '.downsample.1')
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def test_resnet50_matches_independent_implementation():
    torch.set_num_threads(8)
    sd = synth.resnet50_state_dict(1, 'conv5')
    fr = synth.smooth_frames(3, 2, 256, 256)
    x = eo.preprocess(fr)
    with torch.no_grad():
        ours = eo.resnet50_features(sd, x).reshape(2, 2048).numpy()
        hf = _hf_resnet50(sd)(pixel_values=x).pooler_output.reshape(2, 2048).numpy()
    np.testing.assert_allclose(ours, hf, rtol=2e-4, atol=2e-5)
    n_conv = sum(v.size for k, v in sd.items() if k.endswith('.weight') and v.ndim == 4)
    assert n_conv == 23454912          # 23.455 M conv weights


def test_resnet34_matches_independent_implementation():
    """BasicBlock trunks ('resnet18' / 'resnet34', embeddings.py:112-117) against transformers.ResNetModel(layer_type='basic')."""
    from transformers import ResNetConfig, ResNetModel
    torch.set_num_threads(8)
    sd = synth.resnet50_state_dict(2, 'r34')
    cfg = ResNetConfig(layer_type='basic', hidden_sizes=[64, 128, 256, 512], depths=[3, 4, 6, 3], embedding_size=64)
    m = ResNetModel(cfg).eval()
    new = {}

    def bn(dst, src):
        for a in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
            new[dst + '.' + a] = torch.from_numpy(np.array(sd[src + '.' + a]))

    new['embedder.embedder.convolution.weight'] = torch.from_numpy(sd['conv1.weight'])
    bn('embedder.embedder.normalization', 'bn1')
    for li, nb in enumerate((3, 4, 6, 3)):
        for bi in range(nb):
            src, dst = 'layer%d.%d' % (li + 1, bi), 'encoder.stages.%d.layers.%d' % (li, bi)
            for ci in range(2):
                new['%s.layer.%d.convolution.weight' % (dst, ci)] = torch.from_numpy(sd['%s.conv%d.weight' % (src, ci + 1)])
                bn('%s.layer.%d.normalization' % (dst, ci), '%s.bn%d' % (src, ci + 1))
            if (src + '.downsample.0.weight') in sd:
                new[dst + '.shortcut.convolution.weight'] = torch.from_numpy(sd[src + '.downsample.0.weight'])
                bn(dst + '.shortcut.normalization', src + '.downsample.1')
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    fr = synth.smooth_frames(4, 2, 128, 128)
    x = eo.preprocess(fr)
    with torch.no_grad():
        ours = eo.resnet50_features(sd, x, 'r34').reshape(2, 512).numpy()
        hf = m(pixel_values=x).pooler_output.reshape(2, 512).numpy()
    np.testing.assert_allclose(ours, hf, rtol=2e-4, atol=2e-5)


def test_out_sizes_and_heads():
    torch.set_num_threads(8)
    fr = synth.frames(5, 1, 64, 64)
    for variant, o in (('conv3', 2156), ('conv4', 2058)):
        sd = synth.resnet50_state_dict(2, variant)
        out = eo.embed(sd, fr, variant)
        assert out.shape == (o,)       # N=1 is squeezed (embeddings.py:402)
        assert np.isfinite(out).all() and (out >= 0).all()
    # C-major flatten: (N,11,14,14) -> index c*196 + h*14 + w
    sd = synth.resnet50_state_dict(2, 'conv3')
    with torch.no_grad():
        f = eo.resnet50_features(sd, eo.preprocess(fr), 'conv3')
    assert f.shape == (1, 11, 14, 14)
    assert eo.embed(sd, fr, 'conv3')[3 * 196 + 5 * 14 + 7] == pytest.approx(float(f[0, 3, 5, 7]))


def test_transforms_semantics():
    # 256x256: Resize(256) is the identity, crop offset 16 (embeddings.py:80-85)
    fr = synth.frames(1, 2, 256, 256)
    u8 = eo.preprocess_u8(fr)
    assert u8.shape == (2, 3, 224, 224)
    assert np.array_equal(u8.numpy(), np.transpose(fr[:, 16:240, 16:240, :], (0, 3, 1, 2)))
    # 64x64: x4 bilinear upsample rounded back to uint8; constant image stays constant
    c = np.full((1, 64, 64, 3), 77, np.uint8)
    assert (eo.preprocess_u8(c).numpy() == 77).all()
    # non-square: short side -> 256, long side int(256*long/short)
    assert eo.resize_size(64, 96) == (256, 384)
    assert eo.resize_size(96, 64) == (384, 256)
    assert eo.resize_size(256, 300) == (256, 300)
    x = eo.preprocess(c)
    exp = (77 / 255.0 - np.array(eo.IMAGENET_MEAN)) / np.array(eo.IMAGENET_STD)
    np.testing.assert_allclose(x[0, :, 0, 0].numpy(), exp.astype(np.float32), rtol=1e-6)
    # x2 upsample of a ramp: exact dyadic weights, ties round half to even like torch.round
    ramp = np.tile(np.arange(128, dtype=np.uint8)[None, None, :, None], (1, 128, 1, 3))
    up = eo.resize_u8(torch.from_numpy(ramp).permute(0, 3, 1, 2))
    assert up.shape[-2:] == (256, 256)
    row = up[0, 0, 0].numpy().astype(int)
    assert row[0] == 0 and row[1] == 0 and row[2] == 1 and row[255] == 127      # 0.25->0, 0.75->1


def test_split_embed_concat_order():
    """save_embedded_obs.py:151-156: all current frames first, then all goal frames; output (N, 2*O)."""
    obs = synth.frames(9, 3, 8, 8 * 2).reshape(3, 8, 8, 6)
    calls = []

    def fake(o):
        calls.append(o.copy())
        return o.reshape(o.shape[0], -1)[:, :4].astype(np.float32)

    out = eo.split_embed_concat(fake, obs, 2)
    assert out.shape == (3, 8)
    assert np.array_equal(calls[0][:3], obs[..., :3]) and np.array_equal(calls[0][3:], obs[..., 3:])
    assert np.array_equal(out[:, :4], obs[..., :3].reshape(3, -1)[:, :4].astype(np.float32))
