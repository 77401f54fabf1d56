"""CPU checks of the C-ABI: the library loads, exports every symbol include/*.h declares, and its
host-side helpers behave (no compute calls without a GPU)."""
import ctypes as C
import glob, os, re
import numpy as np
import pytest
import torch

from pvr_habitat_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = re.sub(r'/\*.*?\*/', '', open(h).read(), flags=re.S)
        names |= set(re.findall(r'\b(pvr_[a-z0-9_]+)\s*\(', src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 15
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    assert b'gfx950' in _lib.lib().pvr_version()


@pytest.mark.parametrize('dtype,tdt', [(_lib.PVR_BF16, torch.bfloat16), (_lib.PVR_F16, torch.float16)])
def test_host_weight_conversion_is_rne(dtype, tdt):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 300.0, 7e4)]
                       + [np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-30, 5.96e-8, 2.98e-8, 2.981e-8], np.float32)])
    out = np.zeros(x.size, np.uint16)
    _lib.check(_lib.lib().pvr_debug_convert(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size, dtype))
    ref = torch.from_numpy(x).to(tdt).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(out, ref)


def test_action_sampler_uniform_is_never_zero_or_one():
    """the Gumbel-max sampler's uniform (csrc/sample_rng.h) stays inside (0, 1) for every 32-bit input: u = 1 would make -log(-log u) = +inf
    and that action win whatever the logits say (with 24 random bits 0xFFFFFFFF rounded up to exactly 1.0)"""
    L = C.CDLL(_lib.LIB_PATH)
    L.pvr_debug_sample_uniform.restype = C.c_float
    L.pvr_debug_sample_uniform.argtypes = [C.c_uint32]
    rng = np.random.default_rng(3)
    edge = [0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFDFF, 0xFFFFFE00, 0xFFFFFFFE, 0xFFFFFFFF]
    us = np.array([L.pvr_debug_sample_uniform(int(b)) for b in edge + list(rng.integers(0, 2 ** 32, 2000))], np.float32)
    assert us.min() > 0.0 and us.max() < 1.0
    g = -np.log(-np.log(us.astype(np.float64)))
    assert np.isfinite(g).all()
    assert L.pvr_debug_sample_uniform(0xFFFFFFFF) == np.float32(1.0 - 2.0 ** -24) and L.pvr_debug_sample_uniform(0) == np.float32(2.0 ** -24)


def test_errors_cross_the_abi_as_status_and_message():
    L = _lib.lib()
    h = C.c_void_p()
    d = _lib.EncoderDesc(arch=99, dtype=0, max_batch=1, chunk=0, resize=256, crop=224)
    assert L.pvr_encoder_create(C.byref(d), C.byref(h)) != 0
    assert 'arch' in _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(L.pvr_encoder_create(C.byref(d), C.byref(h)))
    # a well-formed encoder with no weights reports the first missing key (reference asserts missing_keys == [])
    d = _lib.EncoderDesc(arch=_lib.ARCH_RESNET50, dtype=0, max_batch=1, chunk=0, resize=256, crop=224)
    _lib.check(L.pvr_encoder_create(C.byref(d), C.byref(h)))
    assert L.pvr_encoder_out_size(h) == 2048
    assert L.pvr_encoder_finalize(h) == 2
    assert 'conv1.weight' in _lib.last_error()
    L.pvr_encoder_destroy(h)


def test_registry_matches_reference_names():
    from pvr_habitat_amd import embeddings as E
    assert E._UBER['moco_aug_places_uber_345'] == ['moco_aug_places_l3', 'moco_aug_places_l4', 'moco_aug_places']
    assert E._UBER['moco_croponly_uber_45'] == ['moco_croponly_l4', 'moco_croponly']
    assert len(E._UBER) == 16 and len(E._SINGLE) == 28                 # incl. resnet18 / resnet34 (embeddings.py:112-117)
    with pytest.raises(NotImplementedError, match='Requested model not available'):
        E._get_embedding('nonexistent')
    with pytest.raises(AssertionError):
        E._get_embedding('resnet50', in_channels=4)


def test_checkpoint_key_remapping_like_reference_loaders():
    """moco.py:14-24 / resnet.py:33-42: prefix stripping, fc dropped, missing keys asserted."""
    from pvr_habitat_amd import embeddings as E, synth
    keys = synth.resnet50_state_dict(0, 'conv5', keys_only=True)
    ck = {'module.encoder_q.' + k: torch.zeros(1) for k in keys}
    ck['module.encoder_q.fc.0.weight'] = torch.zeros(1)
    ck['module.encoder_k.conv1.weight'] = torch.zeros(1)
    out = E.remap_checkpoint(ck, 'moco', 'conv5')
    assert set(out) == set(keys)
    del ck['module.encoder_q.layer2.0.bn1.weight']
    with pytest.raises(AssertionError):
        E.remap_checkpoint(ck, 'moco', 'conv5')
    k3 = synth.resnet50_state_dict(0, 'conv3', keys_only=True)
    ck = {'module.' + k: torch.zeros(1) for k in k3}
    ck['module.fc.weight'] = torch.zeros(1); ck['module.layer4.0.conv1.weight'] = torch.zeros(1)
    assert set(E.remap_checkpoint(ck, 'resnet', 'conv3')) == set(k3)
    ck['module.layer1.9.conv1.weight'] = torch.zeros(1)
    with pytest.raises(AssertionError):
        E.remap_checkpoint(ck, 'resnet', 'conv3')
