"""Counter-based synthetic data: frames, encoder weights, policy weights.

There is no network on the build or GPU boxes, so no checkpoint (torchvision hub, MoCo-v2
`*.pth`, CLIP) can be fetched.  Weights and frames are regenerated on every box from a tiny
counter-based generator (splitmix64 finaliser over `seed, stream, index`), so the same bytes
exist in the build container (where golden fixtures are made) and on the MI355X box (where
they are checked) without shipping 94 MB of fp32 ResNet50 weights.  Not torch RNG: torch's
CPU/GPU generators differ, and their streams are not a stable contract across versions.

State-dict key layout follows torchvision `resnet50` (reference: src/embeddings.py:118-120,
src/vision_models/moco.py:11-12) and the reference policies (src/models.py:22-44, 107-148).
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_id(name):
    return np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)


def bits(seed, name, n):
    """n uint64 words, a pure function of (seed, name, index)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + _stream_id(name))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * np.uint64(0x2545F4914F6CDD1D))


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    u = (bits(seed, name, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape, std=1.0):
    n = int(np.prod(shape))
    m = (n + 1) // 2
    b = bits(seed, name, 2 * m)
    u1 = ((b[:m] >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))
    u2 = (b[m:] >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (std * z).astype(np.float32).reshape(shape)


def frames(seed, n, h, w, name='frames'):
    """uint8 (n,h,w,3) i.i.d. uniform[0,255] (SURVEY 8d synthetic inputs)."""
    b = bits(seed, '%s_%dx%d' % (name, h, w), (n * h * w * 3 + 7) // 8)
    return b.view(np.uint8)[: n * h * w * 3].reshape(n, h, w, 3).copy()


def smooth_frames(seed, n, h, w, name='smooth'):
    """uint8 frames with low-frequency structure (closer to rendered Habitat frames than
    white noise; used by parity tests so the resize path sees non-trivial gradients)."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    ph = uniform(seed, name + '_ph', (n, 3, 4), 0.0, 6.2831853)
    fr = uniform(seed, name + '_fr', (n, 3, 4), 0.5, 6.0)
    out = np.zeros((n, h, w, 3), np.float32)
    for k in range(4):
        ang = (fr[:, :, k, None, None] * (xx[None, None] * (k % 2 + 1) / w + yy[None, None] * ((k + 1) % 2 + 1) / h)
               + ph[:, :, k, None, None])
        out += np.transpose(np.sin(ang), (0, 2, 3, 1))
    out = (out / 4.0 * 0.5 + 0.5) * 255.0
    noise = frames(seed, n, h, w, name + '_n').astype(np.float32) - 127.5
    return np.clip(np.rint(out + 0.08 * noise), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------
# ResNet50 (torchvision naming).  Reference topology: torchvision resnet50 v1.5 reached from
# src/embeddings.py:118-120 / src/vision_models/moco.py:6-26 / resnet.py:86-104.
# ------------------------------------------------------------------------------------------
RESNET50_LAYERS = (3, 4, 6, 3)


_KEYS_ONLY = False


def _bn(sd, seed, prefix, c, gamma=(0.8, 1.2)):
    if _KEYS_ONLY:
        for a in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
            sd[prefix + '.' + a] = None
        return
    sd[prefix + '.weight'] = uniform(seed, prefix + '.weight', (c,), *gamma)
    sd[prefix + '.bias'] = uniform(seed, prefix + '.bias', (c,), -0.2, 0.2)
    sd[prefix + '.running_mean'] = uniform(seed, prefix + '.running_mean', (c,), -0.2, 0.2)
    sd[prefix + '.running_var'] = uniform(seed, prefix + '.running_var', (c,), 0.6, 1.4)
    sd[prefix + '.num_batches_tracked'] = np.zeros((), np.int64)


def _conv(sd, seed, name, co, ci, k, bias=False, gain=2.0):
    fan_in = ci * k * k
    if _KEYS_ONLY:
        sd[name + '.weight'] = None
        if bias:
            sd[name + '.bias'] = None
        return
    sd[name + '.weight'] = normal(seed, name + '.weight', (co, ci, k, k), std=float(np.sqrt(gain / fan_in)))
    if bias:
        sd[name + '.bias'] = uniform(seed, name + '.bias', (co,), -0.1, 0.1)


def resnet50_state_dict(seed=1, variant='conv5', keys_only=False):
    """Synthetic state_dict with torchvision resnet50 keys (keys_only: names without data).

    variant: 'conv5' (-> 2048, moco.py:6-26), 'conv4' (layer4 + BasicBlock(2048->42),
    moco.py:73-113 -> 2058), 'conv3' (layer3 + BasicBlock(1024->11), moco.py:29-70 -> 2156).
    The last BN of every bottleneck gets a small gamma so 16 residual additions keep
    activations O(1) (random-init stand-in for a trained network's statistics).
    """
    global _KEYS_ONLY
    _KEYS_ONLY = bool(keys_only)
    try:
        return _resnet50_state_dict(seed, variant)
    finally:
        _KEYS_ONLY = False


def _resnet_basic_state_dict(seed, layers):
    """torchvision resnet18 / resnet34 keys (BasicBlock trunks, reference embeddings.py:112-117)"""
    sd = {}
    _conv(sd, seed, 'conv1', 64, 3, 7)
    _bn(sd, seed, 'bn1', 64)
    inplanes = 64
    for li in range(4):
        planes = 64 << li
        for bi in range(layers[li]):
            p = 'layer%d.%d' % (li + 1, bi)
            _conv(sd, seed, p + '.conv1', planes, inplanes, 3)
            _bn(sd, seed, p + '.bn1', planes)
            _conv(sd, seed, p + '.conv2', planes, planes, 3)
            _bn(sd, seed, p + '.bn2', planes, gamma=(0.25, 0.45))
            if bi == 0 and li > 0:
                _conv(sd, seed, p + '.downsample.0', planes, inplanes, 1, gain=1.0)
                _bn(sd, seed, p + '.downsample.1', planes)
            inplanes = planes
    return sd


def _resnet50_state_dict(seed, variant):
    if variant in ('r18', 'r34'):
        return _resnet_basic_state_dict(seed, (2, 2, 2, 2) if variant == 'r18' else (3, 4, 6, 3))
    sd = {}
    _conv(sd, seed, 'conv1', 64, 3, 7)
    _bn(sd, seed, 'bn1', 64)
    inplanes = 64
    stages = 4 if variant in ('conv5', 'conv4') else 3
    for li in range(stages):
        planes = 64 << li
        for bi in range(RESNET50_LAYERS[li]):
            p = 'layer%d.%d' % (li + 1, bi)
            if variant == 'conv4' and li == 3:
                p = 'layer4.0.%d' % bi      # nn.Sequential(model.layer4, BasicBlock) nesting
            if variant == 'conv3' and li == 2:
                p = 'layer3.0.%d' % bi
            _conv(sd, seed, p + '.conv1', planes, inplanes, 1)
            _bn(sd, seed, p + '.bn1', planes)
            _conv(sd, seed, p + '.conv2', planes, planes, 3)
            _bn(sd, seed, p + '.bn2', planes)
            _conv(sd, seed, p + '.conv3', planes * 4, planes, 1)
            _bn(sd, seed, p + '.bn3', planes * 4, gamma=(0.25, 0.45))
            if bi == 0:
                _conv(sd, seed, p + '.downsample.0', planes * 4, inplanes, 1, gain=1.0)
                _bn(sd, seed, p + '.downsample.1', planes * 4)
            inplanes = planes * 4
    if variant in ('conv3', 'conv4'):
        cin, c = (1024, 11) if variant == 'conv3' else (2048, 42)
        p = 'layer3.1' if variant == 'conv3' else 'layer4.1'
        _conv(sd, seed, p + '.conv1', c, cin, 3)
        _bn(sd, seed, p + '.bn1', c)
        _conv(sd, seed, p + '.conv2', c, c, 3)
        _bn(sd, seed, p + '.bn2', c)
        _conv(sd, seed, p + '.downsample.0', c, cin, 3, bias=True, gain=1.0)
        _bn(sd, seed, p + '.downsample.1', c)
    return sd


# ------------------------------------------------------------------------------------------
# BC policies (src/models.py:13-44, 96-148).  Keys match the reference modules' state_dict().
# ------------------------------------------------------------------------------------------
def policy_state_dict(seed, obs_size, num_actions, batch_norm, hidden=1024, conv=False):
    sd = {}
    if conv:
        cin = 3
        for i in range(5):
            _conv(sd, seed, 'feat_extract.%d' % (2 * i), 32, cin, 3, bias=True)
            cin = 32
    o = 0
    if batch_norm:
        sd['fc.0.weight'] = uniform(seed, 'fc.0.weight', (obs_size,), 0.8, 1.2)
        sd['fc.0.bias'] = uniform(seed, 'fc.0.bias', (obs_size,), -0.1, 0.1)
        sd['fc.0.running_mean'] = np.zeros((obs_size,), np.float32)
        sd['fc.0.running_var'] = np.ones((obs_size,), np.float32)
        sd['fc.0.num_batches_tracked'] = np.zeros((), np.int64)
        o = 1
    sd['fc.%d.weight' % o] = normal(seed, 'fc.a.weight', (hidden, obs_size), std=float(np.sqrt(2.0 / obs_size)))
    sd['fc.%d.bias' % o] = uniform(seed, 'fc.a.bias', (hidden,), -0.05, 0.05)
    sd['fc.%d.weight' % (o + 2)] = normal(seed, 'fc.b.weight', (hidden, hidden), std=float(np.sqrt(2.0 / hidden)))
    sd['fc.%d.bias' % (o + 2)] = uniform(seed, 'fc.b.bias', (hidden,), -0.05, 0.05)
    k = 1.0 / np.sqrt(hidden)
    for l in range(2):
        sd['core.weight_ih_l%d' % l] = uniform(seed, 'core.weight_ih_l%d' % l, (4 * hidden, hidden), -k, k)
        sd['core.weight_hh_l%d' % l] = uniform(seed, 'core.weight_hh_l%d' % l, (4 * hidden, hidden), -k, k)
        sd['core.bias_ih_l%d' % l] = uniform(seed, 'core.bias_ih_l%d' % l, (4 * hidden,), -k, k)
        sd['core.bias_hh_l%d' % l] = uniform(seed, 'core.bias_hh_l%d' % l, (4 * hidden,), -k, k)
    sd['policy.weight'] = normal(seed, 'policy.weight', (num_actions, hidden), std=float(1.0 / np.sqrt(hidden)))
    sd['policy.bias'] = np.zeros((num_actions,), np.float32)
    sd['baseline.weight'] = normal(seed, 'baseline.weight', (1, hidden), std=float(1.0 / np.sqrt(hidden)))
    sd['baseline.bias'] = np.zeros((1,), np.float32)
    return sd


def bc_batches(seed, T, B, obs_size, A, steps):
    """Synthetic BC batches: |N(0,1)| embedding-like obs, sparse dones, uniform actions."""
    obs = np.abs(normal(seed, 'bc_obs', (steps, T, B, obs_size)))
    done = uniform(seed, 'bc_done', (steps, T, B)) < 0.02
    act = (uniform(seed, 'bc_act', (steps, T, B)) * A).astype(np.int64).clip(0, A - 1)
    return obs, done, act


def bc_conv_batches(seed, T, B, steps, A):
    """Synthetic finetune batches: raw uint8 (T,B,64,64,6) observations (models.py:159-161)."""
    obs = frames(seed, steps * T * B, 64, 64 * 2, 'bc_conv').reshape(steps, T, B, 64, 64, 6)
    done = uniform(seed, 'bc_done', (steps, T, B)) < 0.02
    act = (uniform(seed, 'bc_act', (steps, T, B)) * A).astype(np.int64).clip(0, A - 1)
    return obs, done, act


# ------------------------------------------------------------------------------------------
# CLIP visual transformer (openai/CLIP model.py VisionTransformer; reference src/embeddings.py:303-304
# loads "ViT-B/32": width 768, 12 layers, 12 heads, patch 32, output 512).  Keys as in clip's state_dict.
# ------------------------------------------------------------------------------------------
def clip_vit_state_dict(seed=1, patch=32, width=768, layers=12, out_dim=512, resolution=224):
    sd = {}
    grid = resolution // patch
    s = width ** -0.5
    sd['visual.class_embedding'] = normal(seed, 'vit.cls', (width,), std=s)
    sd['visual.positional_embedding'] = normal(seed, 'vit.pos', (grid * grid + 1, width), std=s)
    sd['visual.proj'] = normal(seed, 'vit.proj', (width, out_dim), std=s)
    sd['visual.conv1.weight'] = normal(seed, 'vit.conv1', (width, 3, patch, patch), std=float(np.sqrt(1.0 / (3 * patch * patch))))
    for nm in ('ln_pre', 'ln_post'):
        sd['visual.%s.weight' % nm] = uniform(seed, 'vit.%s.w' % nm, (width,), 0.8, 1.2)
        sd['visual.%s.bias' % nm] = uniform(seed, 'vit.%s.b' % nm, (width,), -0.1, 0.1)
    for i in range(layers):
        p = 'visual.transformer.resblocks.%d.' % i
        sd[p + 'attn.in_proj_weight'] = normal(seed, p + 'inw', (3 * width, width), std=s)
        sd[p + 'attn.in_proj_bias'] = uniform(seed, p + 'inb', (3 * width,), -0.05, 0.05)
        sd[p + 'attn.out_proj.weight'] = normal(seed, p + 'outw', (width, width), std=s * 0.5)
        sd[p + 'attn.out_proj.bias'] = uniform(seed, p + 'outb', (width,), -0.05, 0.05)
        for ln in ('ln_1', 'ln_2'):
            sd[p + ln + '.weight'] = uniform(seed, p + ln + 'w', (width,), 0.8, 1.2)
            sd[p + ln + '.bias'] = uniform(seed, p + ln + 'b', (width,), -0.1, 0.1)
        sd[p + 'mlp.c_fc.weight'] = normal(seed, p + 'fcw', (4 * width, width), std=s)
        sd[p + 'mlp.c_fc.bias'] = uniform(seed, p + 'fcb', (4 * width,), -0.05, 0.05)
        sd[p + 'mlp.c_proj.weight'] = normal(seed, p + 'pjw', (width, 4 * width), std=float((4 * width) ** -0.5) * 0.5)
        sd[p + 'mlp.c_proj.bias'] = uniform(seed, p + 'pjb', (width,), -0.05, 0.05)
    return sd


# ------------------------------------------------------------------------------------------
# MAE ViT-B/16 encoder (reference src/vision_models/mae.py:74-130, 202-222; timm 0.5.4 Block / PatchEmbed keys)
# ------------------------------------------------------------------------------------------
def sincos_2d_pos_embed(embed_dim, grid_size, cls_token=True):
    """Fixed 2-D sin-cos position embedding of MAE (mae.py:23-70): first half of the channels encodes the
    w coordinate of np.meshgrid(w, h), second half the h coordinate; sin block then cos block per half."""
    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum('m,d->md', pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)     # "here w goes first"
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros((1, embed_dim)), emb], axis=0)
    return emb.astype(np.float32)


def mae_vit_state_dict(seed=1, patch=16, width=768, layers=12, resolution=224):
    sd = {}
    grid = resolution // patch
    s = width ** -0.5
    sd['cls_token'] = normal(seed, 'mae.cls', (1, 1, width), std=0.02)
    sd['pos_embed'] = sincos_2d_pos_embed(width, grid)[None]
    sd['patch_embed.proj.weight'] = normal(seed, 'mae.pe.w', (width, 3, patch, patch), std=float(np.sqrt(1.0 / (3 * patch * patch))))
    sd['patch_embed.proj.bias'] = uniform(seed, 'mae.pe.b', (width,), -0.05, 0.05)
    for i in range(layers):
        p = 'blocks.%d.' % i
        for ln in ('norm1', 'norm2'):
            sd[p + ln + '.weight'] = uniform(seed, p + ln + 'w', (width,), 0.8, 1.2)
            sd[p + ln + '.bias'] = uniform(seed, p + ln + 'b', (width,), -0.1, 0.1)
        sd[p + 'attn.qkv.weight'] = normal(seed, p + 'qkvw', (3 * width, width), std=s)
        sd[p + 'attn.qkv.bias'] = uniform(seed, p + 'qkvb', (3 * width,), -0.05, 0.05)
        sd[p + 'attn.proj.weight'] = normal(seed, p + 'projw', (width, width), std=s * 0.5)
        sd[p + 'attn.proj.bias'] = uniform(seed, p + 'projb', (width,), -0.05, 0.05)
        sd[p + 'mlp.fc1.weight'] = normal(seed, p + 'fc1w', (4 * width, width), std=s)
        sd[p + 'mlp.fc1.bias'] = uniform(seed, p + 'fc1b', (4 * width,), -0.05, 0.05)
        sd[p + 'mlp.fc2.weight'] = normal(seed, p + 'fc2w', (width, 4 * width), std=float((4 * width) ** -0.5) * 0.5)
        sd[p + 'mlp.fc2.bias'] = uniform(seed, p + 'fc2b', (width,), -0.05, 0.05)
    sd['norm.weight'] = uniform(seed, 'mae.norm.w', (width,), 0.8, 1.2)
    sd['norm.bias'] = uniform(seed, 'mae.norm.b', (width,), -0.1, 0.1)
    return sd


# ------------------------------------------------------------------------------------------
# CLIP RN50 visual tower (openai/CLIP ModifiedResNet + AttentionPool2d; reference src/embeddings.py:305-306)
# ------------------------------------------------------------------------------------------
def clip_rn50_state_dict(seed=1, width=64, out_dim=1024, keys_only=False):
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + '.weight'] = None if keys_only else normal(seed, name, (cout, cin, k, k), std=float(np.sqrt(2.0 / (cin * k * k))))

    def bn(name, c, gamma=(0.8, 1.2)):
        if keys_only:
            for f in ('weight', 'bias', 'running_mean', 'running_var'):
                sd[name + '.' + f] = None
            return
        sd[name + '.weight'] = uniform(seed, name + '.w', (c,), *gamma)
        sd[name + '.bias'] = uniform(seed, name + '.b', (c,), -0.1, 0.1)
        sd[name + '.running_mean'] = uniform(seed, name + '.m', (c,), -0.2, 0.2)
        sd[name + '.running_var'] = uniform(seed, name + '.v', (c,), 0.5, 1.5)

    v = 'visual.'
    conv(v + 'conv1', width // 2, 3, 3); bn(v + 'bn1', width // 2)
    conv(v + 'conv2', width // 2, width // 2, 3); bn(v + 'bn2', width // 2)
    conv(v + 'conv3', width, width // 2, 3); bn(v + 'bn3', width)
    inpl = width
    for li, nb in enumerate((3, 4, 6, 3)):
        planes = width << li
        for bi in range(nb):
            p = v + 'layer%d.%d.' % (li + 1, bi)
            stride = 2 if (bi == 0 and li > 0) else 1
            conv(p + 'conv1', planes, inpl, 1); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
            conv(p + 'conv3', planes * 4, planes, 1); bn(p + 'bn3', planes * 4, gamma=(0.2, 0.4))
            if stride > 1 or inpl != planes * 4:
                conv(p + 'downsample.0', planes * 4, inpl, 1); bn(p + 'downsample.1', planes * 4, gamma=(0.6, 0.9))
            inpl = planes * 4
    a, c = v + 'attnpool.', width * 32
    if keys_only:
        for k in ('positional_embedding', 'q_proj.weight', 'q_proj.bias', 'k_proj.weight', 'k_proj.bias', 'v_proj.weight', 'v_proj.bias',
                  'c_proj.weight', 'c_proj.bias'):
            sd[a + k] = None
        return sd
    sd[a + 'positional_embedding'] = normal(seed, a + 'pos', (50, c), std=float(c ** -0.5))
    for nm in ('q', 'k', 'v'):
        sd[a + nm + '_proj.weight'] = normal(seed, a + nm + 'w', (c, c), std=float(c ** -0.5))
        sd[a + nm + '_proj.bias'] = uniform(seed, a + nm + 'b', (c,), -0.05, 0.05)
    sd[a + 'c_proj.weight'] = normal(seed, a + 'cw', (out_dim, c), std=float(c ** -0.5))
    sd[a + 'c_proj.bias'] = uniform(seed, a + 'cb', (out_dim,), -0.05, 0.05)
    return sd
