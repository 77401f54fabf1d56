"""EmbeddingNet / UberModel / EmbeddingWrapper with the reference call surface, HIP inside.

Mirrors reference src/embeddings.py:
  * `_get_embedding(name, in_channels, pretrained, train)`  (:60-332)  name registry + transforms
  * `UberModel`                                              (:44-57)   concat of separate models
  * `EmbeddingNet(embedding_name, in_channels=3, pretrained=True, train=False, disable_cuda=False)`
    `.forward(uint8 (N,H,W,3)) -> np.float32 (N,O).squeeze()`  (:339-402)
  * `EmbeddingWrapper.observation`                           (:441-444)
and the checkpoint loaders of src/vision_models/moco.py:6-113 / resnet.py:6-104 (key remapping and
their asserts).  All arithmetic (transforms + network) runs in libpvr_hip.so on the MI355X; this
module only owns names, state_dicts and tensors.  `disable_cuda=True` selects the library's host backend (ResNet family, fp32 C++ loops);
without that flag there is no CPU path: a
box without a GPU raises instead of silently computing something else.
"""
import ctypes as C
import os
import zlib

import numpy as np
import torch
import torch.nn as nn

from . import _lib, synth

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
OUT_SIZE = {'r18': 512, 'r34': 512, 'conv5': 2048, 'conv4': 2058, 'conv3': 2156, 'clip_b32': 512, 'clip_b16': 512, 'mae_b16': 768, 'mae_l16': 1024, 'mae_h14': 1280, 'clip_rn50': 1024, 'random5': 1568}
_ARCH = {'conv5': _lib.ARCH_RESNET50, 'conv4': _lib.ARCH_RESNET50_L4, 'conv3': _lib.ARCH_RESNET50_L3,
         'clip_b32': 3, 'clip_b16': 4, 'mae_b16': 5, 'random5': 6, 'mae_l16': 7, 'mae_h14': 8, 'clip_rn50': 9, 'r18': 10, 'r34': 11}

# ---------------------------------------------------------------------------------------------
# name registry (reference src/embeddings.py:113-280): name -> (loader family, variant, checkpoint)
# ---------------------------------------------------------------------------------------------
_SINGLE = {
    'resnet18': ('torchvision', 'r18', None),        # embeddings.py:112-117
    'resnet34': ('torchvision', 'r34', None),
    'resnet50': ('torchvision', 'conv5', None),
    'resnet50_places': ('resnet', 'conv5', 'resnet50_places.pth.tar'),
    'resnet50_l4': ('resnet', 'conv4', 'resnet50_l4.pth.tar'),
    'resnet50_l3': ('resnet', 'conv3', 'resnet50_l3.tar'),
    'resnet50_places_l4': ('resnet', 'conv4', 'resnet50_places_l4.tar'),
    'resnet50_places_l3': ('resnet', 'conv3', 'resnet50_places_l3.tar'),
    'demy': ('moco', 'conv5', 'demy.pth'),
    'moco_aug': ('moco', 'conv5', 'moco_aug.pth.tar'),
    'moco_aug_habitat': ('moco', 'conv5', 'moco_aug_habitat_64.pth'),
    'moco_aug_mujoco': ('moco', 'conv5', 'moco_aug_mujoco.pth'),
    'moco_aug_uber': ('moco', 'conv5', 'moco_aug_uber.pth'),
    'moco_aug_places': ('moco', 'conv5', 'moco_aug_places.pth.tar'),
    'moco_aug_l4': ('moco', 'conv4', 'moco_aug_l4.pth'),
    'moco_aug_places_l4': ('moco', 'conv4', 'moco_aug_places_l4.pth'),
    'moco_aug_l3': ('moco', 'conv3', 'moco_aug_l3.pth'),
    'moco_aug_places_l3': ('moco', 'conv3', 'moco_aug_places_l3.pth'),
    'moco_croponly': ('moco', 'conv5', 'moco_croponly.pth'),
    'moco_croponly_places': ('moco', 'conv5', 'moco_croponly_places.pth'),
    'moco_croponly_habitat': ('moco', 'conv5', 'moco_croponly_habitat_64.pth'),
    'moco_croponly_mujoco': ('moco', 'conv5', 'moco_croponly_mujoco.pth'),
    'moco_croponly_uber': ('moco', 'conv5', 'moco_croponly_uber.pth'),
    'moco_croponly_l4': ('moco', 'conv4', 'moco_croponly_l4.pth'),
    'moco_croponly_l3': ('moco', 'conv3', 'moco_croponly_l3.pth'),
    'moco_croponly_places_l4': ('moco', 'conv4', 'moco_croponly_places_l4.pth'),
    'moco_croponly_places_l3': ('moco', 'conv3', 'moco_croponly_places_l3.pth'),
    'moco_coloronly': ('moco', 'conv5', 'moco_coloronly.pth'),
}
_UBER = {}
for _base in ('moco_aug_places', 'moco_aug', 'moco_croponly_places', 'moco_croponly'):
    _m = {'3': _base + '_l3', '4': _base + '_l4', '5': _base}
    for _combo in ('345', '35', '34', '45'):
        _UBER['%s_uber_%s' % (_base, _combo)] = [_m[c] for c in _combo]      # embeddings.py:195-280
# names the reference registers but whose model families are not built yet
_NOT_BUILT = ('maskrcnn_l3',)
# CLIP visual towers: 'clip_vit' is the reference's name (ViT-B/32, embeddings.py:303-304); 'clip_vit_b16' is the
# same block layout at patch 16 (BASELINE config 3), not a reference registry name
_CLIP = {'clip_vit': ('clip_b32', 'ViT-B-32.pt', 32), 'clip_vit_b16': ('clip_b16', 'ViT-B-16.pt', 16),
         'clip_rn50': ('clip_rn50', 'RN50.pt', 0)}            # embeddings.py:305-306 (ModifiedResNet-50 + attention pool)


def _dtype_from_env(compute_dtype=None):
    # Default f16 (round 5): the reference is fp32 and the north star asks for embeddings within 1e-3 of it; f16 storage (11-bit significand)
    # measures 4e-4 at the same rate as bf16 (8-bit significand, 3e-3).  A checkpoint whose activations leave f16's range raises
    # FloatingPointError naming compute_dtype='bf16' as the way out: every output batch is checked for non-finite values (_checked / the device flag of
    # stream_embed), and - because an overflow INSIDE the network can be turned into a finite value by the next ReLU - the first frames a net embeds run
    # once through pvr_encoder_check_range, which checks every convolution's output (EmbeddingNet.validate_range).  Later batches rely on the output check.
    d = (compute_dtype or os.environ.get('PVR_DTYPE', 'f16')).lower()
    if d in ('bf16', 'bfloat16'):
        return _lib.PVR_BF16
    if d in ('f16', 'fp16', 'float16', 'half'):
        return _lib.PVR_F16
    if d in ('f32', 'fp32', 'float32'):
        return _lib.PVR_F32                     # reference-precision mode (ResNet50 family only): f32 MFMA, ~1/8 the speed
    raise ValueError('compute dtype must be bf16, f16 or f32, got %r' % d)


# ---------------------------------------------------------------------------------------------
# checkpoint -> torchvision-named state_dict (reference moco.py / resnet.py key handling)
# ---------------------------------------------------------------------------------------------
def _expected_keys(variant):
    return [k for k in synth.resnet50_state_dict(0, variant, keys_only=True)]


def remap_checkpoint(state_dict, family, variant):
    """Apply the reference loaders' key remapping and asserts; returns {torchvision key: tensor}.

    moco:   keep 'module.encoder_q.*' except 'module.encoder_q.fc*', strip the prefix (moco.py:14-21)
    resnet: strip 'module.' (resnet.py:33-38, 95-99)
    For the compressed variants the checkpoint stores the nn.Sequential nesting
    ('layer4.0.<i>...', 'layer4.1....') because it was saved from the edited model."""
    out = {}
    for k, v in state_dict.items():
        if family == 'moco':
            if k.startswith('module.encoder_q') and not k.startswith('module.encoder_q.fc'):
                out[k[len('module.encoder_q.'):]] = v
        else:
            if k.startswith('module.'):
                out[k[len('module.'):]] = v
    want = _expected_keys(variant)
    missing = [k for k in want if k not in out]
    unexpected = [k for k in out if k not in set(want)]
    if family == 'moco' or variant == 'conv5':
        assert len(missing) == 0, 'missing keys: %s' % missing[:5]                 # moco.py:24, resnet.py:102
    if variant == 'conv3':
        assert all(('fc.' in n or 'layer4.' in n or 'layer3.2' in n) for n in unexpected)   # moco.py:67
    if variant == 'conv4':
        assert all(('fc.' in n or 'layer4.2' in n) for n in unexpected)           # moco.py:110
    return {k: out[k] for k in want if k in out}


def _find_checkpoint(path):
    for d in ('.', os.environ.get('PVR_CHECKPOINT_DIR', '')):
        if d and os.path.isfile(os.path.join(d, path)):
            return os.path.join(d, path)
    return None


def _load_named_state_dict(name, pretrained):
    family, variant, ckpt = _SINGLE[name]
    synthetic_ok = os.environ.get('PVR_SYNTHETIC_WEIGHTS', '0') == '1'
    seed = zlib.crc32(name.encode()) & 0x7fffffff
    if family == 'torchvision':
        # torchvision hub weights cannot be downloaded here; a local torchvision-format file is used if present
        f = _find_checkpoint(name + '.pth') if pretrained else None
        if f is not None:
            sd = torch.load(f, map_location='cpu')
            return {k: v for k, v in sd.items() if not k.startswith('fc.')}, variant
        if pretrained and not synthetic_ok:
            raise FileNotFoundError("pretrained torchvision weights: put '" + name + ".pth' in . or "
                                    "$PVR_CHECKPOINT_DIR (no network), or set PVR_SYNTHETIC_WEIGHTS=1")
        return synth.resnet50_state_dict(seed, variant), variant
    f = _find_checkpoint(ckpt)
    if f is None:
        if synthetic_ok:
            return synth.resnet50_state_dict(seed, variant), variant
        raise FileNotFoundError(ckpt)          # what torch.load(checkpoint_path) raises in the reference
    checkpoint = torch.load(f, map_location=torch.device('cpu'))
    return remap_checkpoint(checkpoint['state_dict'], family, variant), variant


# ---------------------------------------------------------------------------------------------
# module tree holding torchvision-named tensors, so state_dict() keys equal the reference's
# ---------------------------------------------------------------------------------------------
class _Node(nn.Module):
    pass


def _install(root, key, tensor):
    parts = key.split('.')
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    leaf = parts[-1]
    if leaf in ('running_mean', 'running_var', 'num_batches_tracked'):
        node.register_buffer(leaf, tensor)
    elif leaf in node._modules:                      # e.g. 'visual.proj' next to nothing else: keep as parameter
        raise KeyError(key)
    else:
        node.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


class HipResNet50(_Node):
    """One frozen encoder: fp32 host tensors under the reference's state_dict names + a libpvr_hip encoder handle.
    variant: 'conv5' | 'conv4' | 'conv3' (ResNet50 family, torchvision names) or 'clip_b32' | 'clip_b16'
    (openai/CLIP visual tower, `visual.*` names)."""

    def __init__(self, state_dict, variant='conv5', compute_dtype=None, max_batch=None, chunk=None, host=False):
        super().__init__()
        self.variant = variant
        self._host = bool(host)                   # CPU plan of the library (pvr_encoder_set_host_backend): host tensors in and out, fp32
        self.out_size = OUT_SIZE[variant]
        self._clip = variant.startswith('clip')
        self._mae = variant.startswith('mae')
        self._dtype = _dtype_from_env(compute_dtype)
        self._max_batch = int(max_batch or os.environ.get('PVR_MAX_BATCH', 256))
        self._chunk = int(chunk or os.environ.get('PVR_CHUNK', 0))
        self._handle = None
        for k, v in state_dict.items():
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
            _install(self, k, t.detach().clone().cpu())
        self.train(False)

    # -- encoder lifetime ----------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            _lib.lib().pvr_encoder_destroy(self._handle)
            self._handle = None

    def close(self):
        """Free the library handle (folded weights, workspaces) NOW, at a point the caller chooses, instead of whenever the garbage
        collector finds the object.  The module stays usable: the next call re-folds and re-uploads from the host tensors."""
        self._release()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._release()                       # re-fold / re-upload on next use
        return r

    def _build(self):
        L = _lib.lib()
        if self._host and (self._clip or self._mae or self.variant == 'random5'):
            raise NotImplementedError("host backend (disable_cuda / no GPU): only the torchvision ResNet family has a CPU plan, not '%s'" % self.variant)
        tr = transforms_for('clip' if self._clip else 'mae' if self._mae else '')     # (the interpolation mode follows from the arch)
        desc = _lib.EncoderDesc(arch=_ARCH[self.variant], dtype=_lib.PVR_F32 if self._host else self._dtype, max_batch=self._max_batch,
                                chunk=self._chunk, resize=tr.resize, crop=tr.crop)
        desc.mean[:] = tr.mean                                             # embeddings.py:313 / :84
        desc.std_[:] = tr.std
        h = C.c_void_p()
        _lib.check(L.pvr_encoder_create(C.byref(desc), C.byref(h)))
        try:
            if self._host:
                _lib.check(L.pvr_encoder_set_host_backend(h, 1))
            for k, v in self.state_dict().items():
                if k.endswith('num_batches_tracked'):
                    continue
                a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
                shp = (C.c_int64 * max(a.ndim, 1))(*(a.shape or (1,)))
                _lib.check(L.pvr_encoder_load_weights(h, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
            _lib.check(L.pvr_encoder_finalize(h))
        except Exception:
            L.pvr_encoder_destroy(h)
            raise
        self._handle = h
        if self._host:
            return
        if getattr(self, '_low_latency', False):
            self.set_low_latency(True)

    @property
    def max_batch(self):
        return self._max_batch

    @property
    def lanes(self):
        """activation workspaces that can be in flight at once (the 'random' plan has a single workspace)"""
        return 1 if self.variant == 'random5' else 4

    def forward_into(self, frames_u8, out, lane=0):
        """frames_u8: cuda uint8 (N,H,W,3) contiguous; out: cuda fp32 2-D view with row stride out.stride(0).
        lane selects one of the activation workspaces: forwards on different lanes may be in flight at once on different
        streams (pvr_encoder_forward_lane); same-lane forwards issued on different streams are chained by the library
        (a per-lane event), so a workspace is never shared by two forwards in flight."""
        if not self._host:
            _lib.require_gpu()
        if self._handle is None:
            self._build()
        n, h, w, c = frames_u8.shape
        assert c == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
        assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[1] == self.out_size
        if self._host:
            assert not frames_u8.is_cuda and not out.is_cuda, 'host-backend encoder: frames and output live in host memory'
        L = _lib.lib()
        for i in range(0, n, self._max_batch):
            m = min(self._max_batch, n - i)
            _lib.check(L.pvr_encoder_forward_lane(self._handle, lane, C.c_void_p(frames_u8[i:i + m].data_ptr()), m, h, w,
                                                  C.c_void_p(out[i:i + m].data_ptr()), out.stride(0), None if self._host else _lib.stream_ptr()))

    def op_names(self):
        """conv launches of the current HIP plan, in launch order (fused bottleneck tails read 'a.conv2+conv3+b.conv1')."""
        if self._handle is None:
            self._build()
        names, buf, i = [], C.create_string_buffer(256), 3
        while _lib.lib().pvr_encoder_launch_name(self._handle, i, buf, 256) > 0:
            names.append(buf.value.decode())
            i += 1
        return [n for n in names if n != 'pool/flatten']

    def kernel_names(self, n=None):
        """the kernel family each conv launch of the plan runs as in a forward of n frames (default: max_batch), in launch order:
        'bneck_frame(front1)', 'conv_wfrag(pool)', 'conv_pp256(dual)', 'chain', 'conv_split16', 'conv' ... (pvr_encoder_launch_kernel)"""
        if self._handle is None:
            self._build()
        names, buf, i = [], C.create_string_buffer(64), 3
        while _lib.lib().pvr_encoder_launch_kernel(self._handle, int(n or self._max_batch), i, buf, 64) > 0:
            names.append(buf.value.decode())
            i += 1
        return names

    def set_switch(self, name, value):
        """run-time A/B switches of the built plan: 'pool_fuse', 'stem_u8', 'frame_min_n' (pvr_encoder_debug_set_switch); every other PVR_*
        switch is read from the environment when the handle is created"""
        if self._handle is None:
            self._build()
        _lib.check(_lib.lib().pvr_encoder_debug_set_switch(self._handle, name.encode(), int(value)))

    def check_range(self, frames_u8):
        """Load-time validation of the 16-bit storage range on real frames (pvr_encoder_check_range): one forward of the UNFUSED plan with every launch's
        output checked for inf / NaN - an overflow inside the network can be turned into a finite, wrong embedding by the next ReLU, which the finite check of
        the output cannot see.  Returns the name of the first convolution whose output is non-finite, or None.  ResNet family, 16-bit plans."""
        if self._handle is None:
            self._build()
        n = min(int(frames_u8.shape[0]), self._chunk or self._max_batch)
        fr = frames_u8[:n].contiguous()
        out = torch.empty((n, self.out_size), dtype=torch.float32, device=fr.device)
        bad = C.c_int32(-1)
        _lib.check(_lib.lib().pvr_encoder_check_range(self._handle, C.c_void_p(fr.data_ptr()), n, fr.shape[1], fr.shape[2], C.c_void_p(out.data_ptr()),
                                                      out.stride(0), _lib.stream_ptr(), C.byref(bad)))
        if bad.value < 0:
            return None
        if bad.value < 3:
            return 'conv1+bn1+relu+maxpool'
        fused = None
        try:                                      # names of the unfused plan
            self.set_fusion(False)
            names = self.op_names()
        finally:
            self.set_fusion(True)
        return names[bad.value - 3] if bad.value - 3 < len(names) else 'launch %d' % bad.value

    def set_crop(self, pos):
        """0 = centre crop (reference), 1..4 = tl / tr / bl / br corner windows (pvr_encoder_set_crop_position)"""
        if self._handle is None:
            self._build()
        _lib.check(_lib.lib().pvr_encoder_set_crop_position(self._handle, int(pos)))

    def set_low_latency(self, on=True):
        """split-K plan for forwards of <= 4 frames (pvr_encoder_set_low_latency): the online EmbeddingWrapper pattern"""
        self._low_latency = bool(on)
        if self._handle is not None and not (self._clip or self._mae or self.variant == 'random5'):
            _lib.check(_lib.lib().pvr_encoder_set_low_latency(self._handle, int(self._low_latency)))

    def set_fusion(self, on):
        """A/B switch between the fused layer1/layer2 bottleneck-tail plan (default) and one launch per convolution;
        both give bit-identical outputs."""
        if self._handle is None:
            self._build()
        _lib.check(_lib.lib().pvr_encoder_debug_set_fusion(self._handle, 1 if on else 0))

    def tap(self, name, n_elems):
        """fp32 copy of an intermediate activation of the last forward (parity debugging)."""
        buf = torch.empty(n_elems, dtype=torch.float32, device='cuda')
        cnt = C.c_int64()
        _lib.check(_lib.lib().pvr_encoder_tap(self._handle, name.encode(), C.c_void_p(buf.data_ptr()), n_elems,
                                              C.byref(cnt), _lib.stream_ptr()))
        return buf[:cnt.value]

    def debug_stop_after(self, name):
        if self._handle is None:
            self._build()
        _lib.check(_lib.lib().pvr_encoder_debug_stop_after(self._handle, name.encode() if name else None))

    def forward(self, frames_u8):
        out = torch.empty((frames_u8.shape[0], self.out_size), dtype=torch.float32, device=frames_u8.device)
        self.forward_into(frames_u8, out)
        return out


class UberModel(nn.Module):
    """reference src/embeddings.py:44-57.  `models` stays a plain list as in the reference (so, as there,
    its weights are not part of state_dict()); each member writes its columns of one output buffer."""

    def __init__(self, models):
        super().__init__()
        self.models = models
        assert all(models[0].training == m.training for m in models)
        self.training = models[0].training
        self.out_size = sum(m.out_size for m in models)

    def close(self):
        for m in self.models:
            m.close()

    def to(self, device):
        return self

    @property
    def lanes(self):
        return min(m.lanes for m in self.models)

    def set_crop(self, pos):
        for m in self.models:
            m.set_crop(pos)

    def set_low_latency(self, on=True):
        for m in self.models:
            m.set_low_latency(on)

    SMALL_BATCH = 16                  # at or below this many frames a member network leaves most of the GPU idle

    def forward_into(self, frames_u8, out, lane=0):
        """Members write their column blocks of `out`.  Small batches (online evaluation, EmbeddingWrapper: a handful of frames
        per call) run the members CONCURRENTLY on side streams forked from / joined to the caller's stream - each member is its
        own encoder with its own workspace; large batches run them one after the other (each already fills the GPU)."""
        cols, col = [], 0
        for m in self.models:
            cols.append((m, col))
            col += m.out_size
        if frames_u8.shape[0] > self.SMALL_BATCH or len(self.models) == 1:
            for m, c in cols:
                m.forward_into(frames_u8, out[:, c:c + m.out_size], lane=lane)
            return
        cur = torch.cuda.current_stream()
        if getattr(self, '_side', None) is None:
            self._side = [torch.cuda.Stream() for _ in self.models[1:]]
        for (m, c), st in zip(cols[1:], self._side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                m.forward_into(frames_u8, out[:, c:c + m.out_size], lane=lane)
        m, c = cols[0]
        m.forward_into(frames_u8, out[:, c:c + m.out_size], lane=lane)
        for st in self._side:
            cur.wait_stream(st)

    def forward(self, frames_u8):
        out = torch.empty((frames_u8.shape[0], self.out_size), dtype=torch.float32, device=frames_u8.device)
        self.forward_into(frames_u8, out)
        return out


class FiveCrop(nn.Module):
    """Build-defined 5-crop extension (BASELINE config 5; the reference itself only has CenterCrop,
    embeddings.py:82): the wrapped model embeds the four corner windows and the centre window of the Resize(256) frame
    (torchvision FiveCrop order tl, tr, bl, br, centre) and the five embeddings are concatenated along the feature axis:
    (N, 5*O), the last O columns being exactly the reference's centre-crop embedding."""
    ORDER = (1, 2, 3, 4, 0)

    def __init__(self, model):
        super().__init__()
        self.model = [model]                                   # plain list: not a sub-module (as UberModel.models)
        self.training = model.training
        self.out_size = 5 * model.out_size

    def close(self):
        self.model[0].close()

    def to(self, device):
        return self

    @property
    def lanes(self):
        return self.model[0].lanes

    def forward_into(self, frames_u8, out, lane=0):
        m, o = self.model[0], self.model[0].out_size
        try:
            for k, pos in enumerate(self.ORDER):
                m.set_crop(pos)
                m.forward_into(frames_u8, out[:, k * o:(k + 1) * o], lane=lane)
        finally:
            m.set_crop(0)

    def forward(self, frames_u8):
        out = torch.empty((frames_u8.shape[0], self.out_size), dtype=torch.float32, device=frames_u8.device)
        self.forward_into(frames_u8, out)
        return out


class _Transforms(nn.Module):
    """Stand-in for the reference's nn.Sequential of torchvision transforms (embeddings.py:80-85, 309-314): it only records
    the parameters; the Resize / CenterCrop / ConvertImageDtype / Normalize arithmetic is fused into the HIP preprocess + stem
    (or patchify) kernels, which receive exactly these values through pvr_encoder_desc."""

    def __init__(self, resize=256, crop=224, mean=IMAGENET_MEAN, std=IMAGENET_STD, mode='bilinear', antialias=False):
        super().__init__()
        self.resize, self.crop, self.mean, self.std, self.mode, self.antialias = resize, crop, tuple(mean), tuple(std), mode, antialias

    def spec(self):
        """[op, args...] in application order - the form tests/golden/glue_registry.json records from the reference"""
        return [['Resize', self.resize, self.mode, self.antialias], ['CenterCrop', self.crop], ['ConvertImageDtype'],
                ['Normalize', [float(m) for m in self.mean], [float(x) for x in self.std]]]


def transforms_for(embedding_name):
    """The transforms the reference builds for a name: Resize(256) bilinear - bicubic if 'mae' is in the name (embeddings.py:81)
    - CenterCrop(224), /255, ImageNet Normalize (:80-85); the 'clip' names replace them by Resize(224, bicubic, antialias),
    CenterCrop(224), /255, CLIP Normalize (:309-314)."""
    if 'clip' in embedding_name:
        return _Transforms(224, 224, CLIP_MEAN, CLIP_STD, 'bicubic', True)
    return _Transforms(256, 224, IMAGENET_MEAN, IMAGENET_STD, 'bicubic' if 'mae' in embedding_name else 'bilinear', False)


def _get_embedding(embedding_name='random', in_channels=3, pretrained=True, train=False, **hip_kw):
    assert in_channels == 3, 'Current models accept 3-channel inputs only.'          # embeddings.py:87
    if embedding_name == 'true_state':
        return nn.Sequential(nn.Identity()), nn.Sequential(nn.Identity())
    if embedding_name in _SINGLE:
        sd, variant = _load_named_state_dict(embedding_name, pretrained)
        model = HipResNet50(sd, variant, **hip_kw)
    elif embedding_name in _CLIP:
        variant, ckpt, patch = _CLIP[embedding_name]
        f = _find_checkpoint(ckpt) if pretrained else None
        if f is not None:                             # OpenAI's TorchScript archive: keep the visual tower only
            full = torch.jit.load(f, map_location='cpu').state_dict()
            sd = {k: v.float() for k, v in full.items() if k.startswith('visual.')}
        elif os.environ.get('PVR_SYNTHETIC_WEIGHTS', '0') == '1' or not pretrained:
            seed = zlib.crc32(embedding_name.encode()) & 0x7fffffff
            sd = synth.clip_rn50_state_dict(seed) if variant == 'clip_rn50' else synth.clip_vit_state_dict(seed, patch=patch)
        else:
            raise FileNotFoundError(ckpt)
        model = HipResNet50(sd, variant, **hip_kw)
    elif embedding_name == 'random':
        # embeddings.py:90-106: orthogonal(gain=relu) weights, zero bias, same constructor order -> same weights per torch seed
        init_ = lambda m: (nn.init.orthogonal_(m.weight.data, gain=nn.init.calculate_gain('relu')), nn.init.constant_(m.bias.data, 0), m)[2]
        layers, cin = [], in_channels
        for _ in range(5):
            layers += [init_(nn.Conv2d(cin, 32, kernel_size=(3, 3), stride=2, padding=1)), nn.ELU()]
            cin = 32
        model = HipResNet50(nn.Sequential(*layers).state_dict(), 'random5', **hip_kw)
    elif embedding_name in ('mae_base', 'mae_large', 'mae_huge'):   # embeddings.py:137-148; encoder keys only (strict=False there)
        ckpt, variant, patch, width, layers = {                                       # mae.py:275-296
            'mae_base': ('mae_pretrain_vit_base.pth', 'mae_b16', 16, 768, 12),
            'mae_large': ('mae_pretrain_vit_large.pth', 'mae_l16', 16, 1024, 24),
            'mae_huge': ('mae_pretrain_vit_huge.pth', 'mae_h14', 14, 1280, 32)}[embedding_name]
        f = _find_checkpoint(ckpt) if pretrained else None
        if f is not None:
            ck = torch.load(f, map_location='cpu')['model']
            sd = {k: v.float() for k, v in ck.items() if not k.startswith(('decoder', 'mask_token'))}
            sd.setdefault('pos_embed', torch.from_numpy(synth.sincos_2d_pos_embed(width, 224 // patch)[None]))     # fixed buffer
        elif os.environ.get('PVR_SYNTHETIC_WEIGHTS', '0') == '1' or not pretrained:
            sd = synth.mae_vit_state_dict(zlib.crc32(embedding_name.encode()) & 0x7fffffff, patch=patch, width=width, layers=layers)
        else:
            raise FileNotFoundError(ckpt)
        model = HipResNet50(sd, variant, **hip_kw)
    elif embedding_name in _UBER:
        model = UberModel([_get_embedding(n, in_channels, pretrained, train, **hip_kw)[0] for n in _UBER[embedding_name]])
    elif embedding_name in _NOT_BUILT:
        raise NotImplementedError("Requested model not available. ('%s' is registered by the reference but its "
                                  "family is not built in pvr_habitat_amd yet)" % embedding_name)
    else:
        raise NotImplementedError("Requested model not available.")                 # embeddings.py:321
    if train:
        raise NotImplementedError('pvr_habitat_amd runs the encoder frozen (the reference scripts hard-code '
                                  'train=False, main_bc_2.py:68-72); training the embedding is not built')
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model, transforms_for(embedding_name)


class EmbeddingNet(nn.Module):
    """
    Input shape must be (N, H, W, 3), where N is the number of frames.
    The output shape will be (N, O), where O is the embedding size.   (reference embeddings.py:339-402)
    """

    def __init__(self, embedding_name, in_channels=3, pretrained=True, train=False, disable_cuda=False,
                 compute_dtype=None, max_batch=None, chunk=None, crops=1):
        super(EmbeddingNet, self).__init__()
        self.embedding_name = embedding_name
        if self.embedding_name == 'true_state':
            return
        self.in_channels = in_channels
        # disable_cuda: the reference then runs on the CPU (embeddings.py:367-370); here that is the library's host backend
        # (csrc/host_encoder.hip: plain C++ loops behind the same pvr_encoder_* ABI, fp32) - the ResNet family only.
        # It is selected EXPLICITLY (the reference's own flag); a box whose GPU is missing or invisible still fails loudly (no silent fallback).
        self._host = bool(disable_cuda)
        self.embedding, self.transforms = _get_embedding(embedding_name, in_channels, pretrained, train,
                                                         compute_dtype=compute_dtype, max_batch=max_batch, chunk=chunk, host=self._host)
        assert crops in (1, 5), 'crops: 1 (the reference CenterCrop) or 5 (corner + centre windows, FiveCrop order)'
        if crops == 5:
            self.embedding = FiveCrop(self.embedding)
        # the reference discovers these with a dummy CPU forward (embeddings.py:359-363)
        self.in_shape = torch.Size((in_channels, 224, 224))
        self.out_size = int(self.embedding.out_size)
        self.device = torch.device('cpu') if self._host else torch.device('cuda')
        self.training = self.embedding.training

    def _forward(self, observation_u8):
        return self.embedding(observation_u8)

    def close(self):
        """free every library handle under this embedding now (HipResNet50.close); the module stays usable"""
        for m in self.modules():
            if m is not self and hasattr(m, 'close'):
                m.close()

    def embed_device(self, observation):
        """uint8 (N,H,W,3) -> fp32 (N, out_size) on self.device (cuda; host memory for a host-backend encoder); no host sync (for streaming callers)."""
        if not self._host:
            _lib.require_gpu()
        observation = observation.to(device=self.device, non_blocking=True).contiguous()
        if not self._host and not getattr(self, '_range_checked', False):
            self._range_checked = True
            self.validate_range(observation)
        return self._forward(observation)

    def validate_range(self, observation_dev, max_frames=8):
        """ONE-OFF, on the first frames this net embeds (PVR_RANGE_CHECK=0 skips it): every f16 ResNet member runs pvr_encoder_check_range on up to
        `max_frames` of them.  The reference is fp32 (src/embeddings.py:386-402); f16 storage has 5 exponent bits, and an activation that overflows inside the
        network can come out as a finite, wrong embedding (ReLU maps -inf and NaN to 0) - this names the convolution instead.  It validates THESE frames with
        THIS checkpoint, not every later batch; the per-batch finite check of the outputs stays."""
        if os.environ.get('PVR_RANGE_CHECK', '1') == '0':
            return
        members = []
        for m in [self.embedding] + [x for x in getattr(self.embedding, 'model', [])]:
            members += list(getattr(m, 'models', [m]))
        for m in members:
            if isinstance(m, HipResNet50) and m._dtype == _lib.PVR_F16 and m.variant in ('conv5', 'conv4', 'conv3', 'r18', 'r34'):
                bad = m.check_range(observation_dev[:max_frames])
                if bad is not None:
                    raise FloatingPointError("activations of '%s' leave the f16 range on these frames (first non-finite output: %s): "
                                             "use compute_dtype='bf16' (8 exponent bits) or 'f32' for this checkpoint" % (self.embedding_name, bad))

    def forward(self, observation):
        if self.embedding_name == 'true_state':
            return observation.squeeze().cpu().numpy()
        # observation.shape -> (N, H, W, 3); transposes + transforms + model are one HIP plan
        with torch.no_grad():
            out = self.embed_device(observation)
            return _checked(out.view(-1, self.out_size).squeeze().cpu().numpy(), self.embedding)


def _checked(host_out, model):
    """The embeddings are on the host anyway: a non-finite value means an activation left the 16-bit storage range (f16 has
    5 exponent bits; the reference computes in fp32).  Fail loudly instead of handing garbage to the BC stage."""
    if not np.isfinite(host_out).all():
        raise FloatingPointError('non-finite embedding: activations overflowed the encoder storage type (%s); '
                                 "use compute_dtype='bf16' (8 exponent bits) or 'f32' for this checkpoint"
                                 % {_lib.PVR_F16: 'f16', _lib.PVR_BF16: 'bf16', _lib.PVR_F32: 'f32'}.get(getattr(model, '_dtype', None), 'mixed'))
    return host_out


_STREAM_CACHE = {}
_REGISTER_MIN_BYTES = 64 << 20      # stream_embed page-locks a pageable source in place only from this size up (see there)


def _streams(dev_index=None):
    """The four side streams this package uses (two compute lanes, H2D, D2H), created ONCE per device and process, together and in
    this order.  HIP maps streams onto a few hardware queues at creation; which queue a stream lands on changes what overlaps with
    what (scripts/stream_queue_ab.py, stream_embed from pinned memory in a fresh process: creation order lanes-then-copies 77.8 k
    frames/s, copies-then-lanes 56.1 k; bench.py: a leg on two freshly created streams ran at the one-lane rate, 69.8 k instead of
    79.8 k).  Every user of side streams in this package (stream_embed, bench.py's legs) takes them from here, so the process
    has one fixed, known-good set instead of whatever the creation history produced."""
    if dev_index is None:
        dev_index = torch.cuda.current_device()
    if dev_index not in _STREAM_CACHE:
        # a / b = compute lanes, h = H2D, d = D2H, x = a spare stream that is never used.  'xabhd' was the one order that measured the
        # full rate in every context tried (fresh process, after other streams, GPU_MAX_HW_QUEUES 4 and 8; 'abhd' and 'hdab' each lose
        # 25 % in one of them): profiles/experiments/r02_stream_creation_order.txt
        order = os.environ.get('PVR_STREAM_ORDER', 'xabhd')
        made = {}
        with torch.cuda.device(dev_index):
            for ch in order:
                made[ch if ch != 'x' else 'x%d' % len(made)] = torch.cuda.Stream()
        _STREAM_CACHE[dev_index] = (made['h'], made['d'], [made['a'], made['b']], made)
    return _STREAM_CACHE[dev_index][:3]


def lane_streams(dev_index=None):
    """the two compute-lane streams (batch k+1 on lane 1 while batch k drains on lane 0)"""
    return list(_streams(dev_index)[2])


def _stage_rows(dst, x, lo, m, threads):
    """x[lo:lo+m] (uint8, host, possibly a channel-plane view of interleaved frames) -> the pinned tensor dst[:m], by the library's native
    threads (pvr_stage_copy; the call releases the GIL).  Layouts it does not know fall back to torch's copy."""
    src = x[lo:lo + m]
    row = int(np.prod(src.shape[1:]))
    st = src.stride()
    if src.is_contiguous():
        _lib.check(_lib.lib().pvr_stage_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), m, row, row, row, row, threads))
    elif (src.dim() == 4 and src.shape[3] == 3 and st[3] == 1 and st[2] >= 3 and st[1] == src.shape[2] * st[2] and
          (m == 1 or st[0] >= src.shape[1] * st[1])):
        # one 3-channel plane of (N, H, W, 3F) frames: runs of 3 bytes every 3F bytes
        _lib.check(_lib.lib().pvr_stage_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), m, row, st[0] if m > 1 else row, 3, st[2], threads))
    else:
        dst[:m].copy_(src)


def stream_embed(net, frames_u8, batch=256, out=None, depth=4, stage_threads=None, planes=1):
    """Embed a large uint8 (N,H,W,3) array - host-resident, or a CUDA tensor (then only the compute and D2H stages run) - with H2D copies, HIP compute and D2H copies overlapped: a ring of `depth`
    device input / output buffers, one copy stream each way and TWO compute streams (each on its own encoder workspace lane, so
    batch k+1 starts while batch k drains) chained by events.  The ring is deeper than the number of batches in flight on the
    compute side: with only one buffer per lane the upload of batch k+2 cannot start before batch k has been computed, and the lane
    then idles for the whole copy (measured: 60 k frames/s at 11.8 GB/s on a link that sustains 56 GB/s).  Pageable sources are
    staged through pinned buffers by the library's native threads (pvr_stage_copy, `stage_threads` of them, default PVR_STAGE_THREADS
    or 8: one memcpy thread moves ~3 GB/s) on a PRODUCER thread that runs up to `depth` batches ahead of the thread that enqueues the
    GPU work (round 3 staged synchronously between two enqueues: 36 k frames/s from pageable memory).  Same rows, same order, same
    values as calling `net` batch by batch; this is the "embeddings streamed to host" path of BASELINE config 5 and what
    save_embedded_obs uses for big scenes.  Returns np.float32 (N, planes * out_size) (no squeeze).

    planes = F > 1: frames_u8 is (N,H,W,3F) (the scene pickles hold current + goal frame interleaved, save_embedded_obs.py:148-150);
    a batch of whole rows is uploaded ONCE, each 3-channel plane is sliced on the device and embedded into its column block of the
    output row - what the reference builds with np.split / np.concatenate around its model call (:151-156)."""
    _lib.require_gpu()
    if stage_threads is None:
        stage_threads = int(os.environ.get('PVR_STAGE_THREADS', '8'))
    # a strided view (e.g. the channel slice obs[..., 3:6] of a (N,H,W,6) scene) is taken as it is: the staging threads gather it
    # into pinned memory batch by batch, overlapped with the GPU, instead of one np.ascontiguousarray pass over the scene up front
    x = frames_u8 if isinstance(frames_u8, torch.Tensor) else torch.from_numpy(frames_u8 if all(st >= 0 for st in frames_u8.strides)
                                                                               else np.ascontiguousarray(frames_u8))
    F_ = int(planes)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[3] == 3 * F_
    n, osz = x.shape[0], net.out_size
    res = torch.empty((n, F_ * osz), dtype=torch.float32, pin_memory=True) if out is None else out
    assert tuple(res.shape) == (n, F_ * osz)
    dev = torch.device('cuda')
    if os.environ.get('PVR_STREAM_FRESH', '0') == '1':       # experiment: new streams per call
        h2d, d2h, comps = torch.cuda.Stream(), torch.cuda.Stream(), [torch.cuda.Stream(), torch.cuda.Stream()]
    else:
        h2d, d2h, comps = _streams(torch.cuda.current_device())
        comps = list(comps)
    for s_ in (h2d, d2h, *comps):
        s_.wait_stream(torch.cuda.current_stream())          # whatever the caller queued (e.g. a forward on the default stream) comes first
    model = net.embedding
    if hasattr(net, 'validate_range') and not getattr(net, '_range_checked', False) and not getattr(net, '_host', False) and n > 0:
        net._range_checked = True                           # the one-off f16 range validation on the first frames of the stream (EmbeddingNet.validate_range)
        net.validate_range(x[:8, :, :, :3].to(device=dev).contiguous())
    two_lanes = getattr(model, 'lanes', 1) >= 2 and os.environ.get('PVR_STREAM_LANES', '2') != '1'
    depth = max(2, min(int(depth), (n + batch - 1) // batch + 1))
    device_src = x.is_cuda                                  # frames already in HBM (PNG source decoded on the GPU): no upload at all
    pinned_src = device_src or (x.is_pinned() and (F_ > 1 or x.is_contiguous()))   # caller already holds page-locked frames: no staging copy
    registered = None
    if (not pinned_src and stage_threads != 0 and os.environ.get('PVR_STREAM_REGISTER', '0') == '1' and x.is_contiguous()
            and x.numel() >= _REGISTER_MIN_BYTES):
        # OPT-IN (PVR_STREAM_REGISTER=1): page-lock the caller's array IN PLACE for the duration of the call (hipHostRegister), so the
        # frames go straight from the caller's memory to the GPU by DMA - no staging copy.  Round 2 did this by default for any
        # pageable source, i.e. it locked and unlocked pages of the glibc heap that the package does not own, next to unrelated objects
        # (ADVICE round 2); now it needs the opt-in AND a buffer of >= 64 MiB, which glibc serves by a private mmap (above
        # M_MMAP_THRESHOLD_MAX), so the locked pages belong to this array alone.
        try:
            if int(torch.cuda.cudart().cudaHostRegister(x.data_ptr(), x.numel(), 0)) == 0:
                registered = x.data_ptr()
                pinned_src = True
        except Exception:
            registered = None
    shape = (batch,) + tuple(x.shape[1:])
    direct = pinned_src or stage_threads == 0              # stage_threads = 0: hand pageable memory to the driver's own staged copy
    stage_in = None if direct else [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
    dev_in = None if device_src else [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(depth)]
    dev_pl = [torch.empty(shape[:3] + (3,), dtype=torch.uint8, device=dev) for _ in range(2)] if F_ > 1 else None   # one plane, per lane
    dev_out = [torch.empty((batch, F_ * osz), dtype=torch.float32, device=dev) for _ in range(depth)]
    bad_flag = torch.zeros(1, dtype=torch.int32, device=dev)   # set on the device by any batch that holds an inf / NaN (pvr_op_nonfinite_flag); read once at the end
    for s_ in comps:
        s_.wait_stream(torch.cuda.current_stream())          # (the zero fill above)
    in_free = [torch.cuda.Event() for _ in range(depth)]    # compute finished reading dev_in[b]
    out_free = [torch.cuda.Event() for _ in range(depth)]   # D2H finished reading dev_out[b]
    for e in in_free + out_free:
        e.record()
    starts = list(range(0, n, batch))

    # producer thread: pageable -> pinned staging slot, up to `depth` batches ahead; a slot returns to it with the event that marks the end
    # of the H2D copy that read it
    import queue
    import threading
    free_q, ready_q = queue.Queue(), queue.Queue()
    stop = threading.Event()

    def producer():
        try:
            for i, lo in enumerate(starts):
                b, ev = free_q.get()
                if stop.is_set():
                    return
                if ev is not None:
                    ev.synchronize()                             # the upload that last read this slot has finished
                m = min(batch, n - lo)
                _stage_rows(stage_in[b], x, lo, m, max(1, stage_threads))
                ready_q.put((i, b))
        except BaseException as exc:                             # noqa: BLE001 - handed to the consuming thread
            ready_q.put(exc)

    worker = None
    if not direct:
        for b in range(depth):
            free_q.put((b, None))
        worker = threading.Thread(target=producer, name='pvr-stage', daemon=True)
        worker.start()

    try:
        for i, lo in enumerate(starts):
            m = min(batch, n - lo)
            if direct:
                b = i % depth
                src = x[lo:lo + m]
            else:
                got = ready_q.get()
                if isinstance(got, BaseException):
                    raise got
                assert got[0] == i
                b = got[1]
                src = stage_in[b][:m]
            if not device_src:
                with torch.cuda.stream(h2d):
                    h2d.wait_event(in_free[b])
                    dev_in[b][:m].copy_(src, non_blocking=True)
                    ready = torch.cuda.Event(); ready.record(h2d)
                if not direct:
                    free_q.put((b, ready))                        # the staging slot is reusable once this upload has read it
            lane = (i & 1) if two_lanes else 0
            comp = comps[lane]
            with torch.cuda.stream(comp):
                if not device_src:
                    comp.wait_event(ready)
                comp.wait_event(out_free[b])
                rows = src if device_src else dev_in[b][:m]
                if F_ == 1:
                    model.forward_into(rows, dev_out[b][:m], lane=lane)
                else:
                    for f in range(F_):
                        dev_pl[lane][:m].copy_(rows[..., 3 * f:3 * f + 3])       # plane f, contiguous (same stream: ordered before its forward)
                        model.forward_into(dev_pl[lane][:m], dev_out[b][:m, f * osz:(f + 1) * osz], lane=lane)
                in_free[b].record(comp)
                # finite check of this batch on the device, behind its forwards on the same stream (a host pass over the whole result at the
                # end cost 19 % of the 5-crop uber leg's wall clock: 31 310 floats per frame)
                _lib.check(_lib.lib().pvr_op_nonfinite_flag(C.c_void_p(dev_out[b].data_ptr()), m, F_ * osz, dev_out[b].stride(0),
                                                            C.c_void_p(bad_flag.data_ptr()), C.c_void_p(comp.cuda_stream)))
                done = torch.cuda.Event(); done.record(comp)
            with torch.cuda.stream(d2h):
                d2h.wait_event(done)
                res[lo:lo + m].copy_(dev_out[b][:m], non_blocking=True)
                out_free[b].record(d2h)
        torch.cuda.synchronize()
    finally:
        stop.set()
        if worker is not None:
            free_q.put((0, None))                                 # wake a producer that waits for a slot
            worker.join()
        if registered is not None:
            torch.cuda.synchronize()
            rc = int(torch.cuda.cudart().cudaHostUnregister(registered))
            if rc != 0:
                import warnings
                warnings.warn('stream_embed: hipHostUnregister(%#x) returned %d - the source array stays page-locked' % (registered, rc))
    if int(bad_flag.item()) != 0:
        _checked(res.numpy(), model)                          # raises FloatingPointError naming the storage type (and locates nothing else)
        raise FloatingPointError('non-finite embedding in the streamed result')
    return res.numpy() if out is None else res


try:                                          # the reference subclasses gym.ObservationWrapper (embeddings.py:409); gym is optional here
    import gym as _gym
    from gym.spaces.box import Box as _Box
    _WrapperBase = _gym.ObservationWrapper
except Exception:                             # no gym in this image: same attributes, no simulator dependency
    _gym = None

    class _Box(object):
        """the two attributes of gym.spaces.Box the BC scripts read (`.shape`, main_bc_2.py:76; `.low/.high`)"""

        def __init__(self, low, high, shape):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.float32

    class _WrapperBase(object):
        """gym.Wrapper's surface as far as the reference uses it: `.env`, attribute forwarding, reset/step through
        `observation()` (gym.ObservationWrapper.reset / step)."""

        def __init__(self, env):
            self.env = env
            self.observation_space = getattr(env, 'observation_space', None)
            self.action_space = getattr(env, 'action_space', None)

        def __getattr__(self, name):
            if name.startswith('_') or name == 'env':
                raise AttributeError(name)
            return getattr(self.env, name)

        def reset(self, **kwargs):
            return self.observation(self.env.reset(**kwargs))

        def step(self, action):
            observation, reward, done, info = self.env.step(action)
            return self.observation(observation), reward, done, info


class EmbeddingWrapper(_WrapperBase):
    """reference src/embeddings.py:409-444: places the embedding over the observation.  The original observation shape must be
    (H, W, n * 3); each of the n frames passes through the embedding separately and the outputs are stacked:
    `observation((H,W,3n)) -> (n*O,)`, `observation_space = Box(-inf, inf, shape=(O*n,))`.  Subclasses gym.ObservationWrapper
    when gym is importable, otherwise a base with the same surface.  The n frames go to the GPU as ONE batch."""

    def __init__(self, env, embedding):
        _WrapperBase.__init__(self, env)
        in_channels = env.observation_space.shape[2]
        assert in_channels % 3 == 0, \
            """ Only RGB images are supported.
                    Be sure that observation shape is (H, W, n * 3),
                    where n is the number of frames per observation. """
        self.in_channels = 3
        self.n_frames = in_channels // 3
        self.embedding = embedding
        self.observation_space = _Box(low=-np.inf, high=np.inf, shape=(self.embedding.out_size * self.n_frames,))
        # one environment step = one forward of n_frames (2) frames: ask the HIP encoder for its low-latency plan for such calls
        # (larger batches through the same EmbeddingNet are unaffected)
        model = getattr(self.embedding, 'embedding', None)
        if hasattr(model, 'set_low_latency'):
            model.set_low_latency(True)

    def observation(self, observation):
        # (H, W, n_frames * 3) -> (n_frames, H, W, 3); if n_frames > 1, each passes through the embedding separately
        observation = np.stack(np.split(observation, self.n_frames, axis=-1))
        return self.embedding(torch.from_numpy(observation)).flatten()
