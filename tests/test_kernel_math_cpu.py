"""Closed-form arithmetic the HIP kernels rely on, restated in numpy and checked exhaustively on the CPU (the kernels themselves are
checked against the oracles on the GPU; these tests pin the FORMULAS, so an edit that breaks one fails without a GPU):
  * csrc/policy_conv.h div255: (float)x / 255.0f for x = 0..255 through one multiply and one Newton step of two FMAs
  * csrc/common.h gelu_erf: GELU through Abramowitz-Stegun 7.1.26
  * csrc/png_decode.hip: DEFLATE length / distance bases and extra bits in closed form (RFC 1951 3.2.5), code-length order (3.2.7)"""
import numpy as np
import torch


def _fma32(a, b, c):
    """float32 fused multiply-add: exact product and sum in float64 (24 + 24 bit products are exact there), one rounding"""
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def test_div255_is_the_correctly_rounded_quotient_for_every_byte():
    inv = np.float32(1.0) / np.float32(255.0)
    for x in range(256):
        xf = np.float32(x)
        q = np.float32(xf * inv)
        got = _fma32(_fma32(-q, np.float32(255.0), xf), inv, q)
        assert got == xf / np.float32(255.0), x
    assert sum(np.float32(np.float32(x) * inv) != np.float32(x) / np.float32(255.0) for x in range(256)) > 100   # the multiply alone is not enough


def test_gelu_through_abramowitz_stegun_is_far_inside_the_storage_rounding():
    v = torch.linspace(-12, 12, 1200001, dtype=torch.float32)
    x = v.abs() * 0.70710678118654752
    t = 1 / (1 + 0.3275911 * x)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    e = 1 - poly * torch.exp(-x * x)
    g = 0.5 * v * (1 + torch.where(v < 0, -e, e))
    ref = torch.nn.functional.gelu(v.double()).float()
    assert float((g - ref).abs().max()) < 1e-6                              # f16 storage rounds at 5e-4 relative, bf16 at 4e-3
    assert float(((g - ref).abs() / v.abs().clamp(min=1e-3)).max()) < 1e-6


def test_deflate_tables_in_closed_form():
    lbase = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    lext = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    dbase = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    dext = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
    for s in range(29):
        assert ((0 if (s < 8 or s == 28) else (s - 4) >> 2)) == lext[s]
        assert (3 + s if s < 8 else (258 if s == 28 else 3 + ((4 + (s & 3)) << ((s - 4) >> 2)))) == lbase[s]
    for d in range(30):
        assert (0 if d < 4 else (d >> 1) - 1) == dext[d]
        assert (1 + d if d < 4 else 1 + ((2 + (d & 1)) << ((d >> 1) - 1))) == dbase[d]
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    lo = 16 | (17 << 5) | (18 << 10) | (0 << 15) | (8 << 20) | (7 << 25) | (9 << 30) | (6 << 35) | (10 << 40) | (5 << 45) | (11 << 50) | (4 << 55)
    hi = 12 | (3 << 5) | (13 << 10) | (2 << 15) | (14 << 20) | (1 << 25) | (15 << 30)
    assert [((lo >> (5 * i)) & 31) if i < 12 else ((hi >> (5 * (i - 12))) & 31) for i in range(19)] == order


def test_bench_byte_model_covers_every_launch_name_of_the_round5_plan():
    """bench.py prices `roofline.stages` / `two_roof` with conv_algorithmic_bytes(names): every launch name the round-5 ResNet50 plan produces must be
    understood (a name the model does not know silently contributes zero bytes), and the fused forms must be cheaper than what they replace."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan = ['layer1.0.conv1', 'layer1.0.conv2+conv3&downsample+layer1.1.conv1', 'layer1.1.conv2+conv3+layer1.2.conv1', 'layer1.2.conv2+conv3+layer2.0.conv1',
            'layer2.0.downsample.0', 'layer2.0.conv2+conv3+layer2.1.conv1', 'layer2.1.conv2+conv3+layer2.2.conv1', 'layer2.2.conv2+conv3+layer2.3.conv1',
            'layer2.3.conv2+conv3', 'layer3.0.conv1', 'layer3.0.conv2', 'layer3.0.conv3&downsample'] + \
           ['layer3.%d.conv1+conv2+conv3' % k for k in range(1, 6)] + \
           ['layer4.0.conv1', 'layer4.0.conv2', 'layer4.0.conv3&downsample', 'layer4.1.conv1', 'layer4.1.conv2', 'layer4.1.conv3',
            'layer4.2.conv1', 'layer4.2.conv2', 'layer4.2.conv3']
    assert len(plan) == 26
    per = [bench.conv_algorithmic_bytes(1, [nm]) for nm in plan]
    assert all(b > 0 for b in per), [nm for nm, b in zip(plan, per) if b <= 0]
    assert sum(per) == bench.conv_algorithmic_bytes(1, plan)
    one = lambda *names: bench.conv_algorithmic_bytes(1, list(names))
    assert one('layer3.0.conv3&downsample') < one('layer3.0.downsample.0', 'layer3.0.conv3')
    assert one('layer3.1.conv1+conv2+conv3') < one('layer3.1.conv1', 'layer3.1.conv2', 'layer3.1.conv3')
    # one launch per convolution (names=None) is the 53-convolution plan minus the stem
    assert bench.conv_algorithmic_bytes(1) > bench.conv_algorithmic_bytes(1, plan)
