"""CPU fp32 restatement of the reference's CLIP branch (test infrastructure).

Follows reference src/embeddings.py:298-314 (clip.load("ViT-B/32"), transforms Resize(res, BICUBIC,
antialias=True) -> CenterCrop(res) -> ConvertImageDtype(float) -> Normalize(CLIP mean/std)) and :375-376
(`encode_image`).  openai/CLIP (requirements.txt:19, unpinned git HEAD, NOT under /root/reference) is restated
from its published model.py `VisionTransformer`:
    conv1 (k = s = patch, no bias) -> [class_embedding ; patches] + positional_embedding -> ln_pre ->
    12 x { x += MHA(ln_1(x)) ; x += c_proj(QuickGELU(c_fc(ln_2(x)))) } -> ln_post(x[:,0]) @ proj
with LayerNorm eps 1e-5 computed in fp32, QuickGELU(x) = x * sigmoid(1.702 x), nn.MultiheadAttention packing
(in_proj rows = [q;k;v], heads split the 768 features into 12 x 64, scores scaled by 1/sqrt(64)).

PARITY PINNING: CLIP is not installable here and has no reference test -> "parity unpinned" at this boundary;
the restatement is cross-checked against transformers.CLIPVisionModelWithProjection (quick_gelu, same
parameter count, name-remapped synthetic weights) in tests/test_oracle_vit.py.
torchvision's antialiased bicubic Resize on uint8 = float32 interpolate(bicubic, antialias=True), clamp(0,255)
(bicubic overshoots), round, cast back to uint8 (restated, not executed).
"""
import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))


def resize_size(h, w, size):
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    ns, nl = size, int(size * long_ / short)
    return (nl, ns) if w <= h else (ns, nl)


def preprocess_u8(frames_nhwc_u8, res=224):
    x = _t(frames_nhwc_u8).transpose(1, 2).transpose(1, 3).contiguous()
    n, c, h, w = x.shape
    nh, nw = resize_size(h, w, res)
    if (nh, nw) != (h, w):
        y = F.interpolate(x.float(), size=(nh, nw), mode='bicubic', align_corners=False, antialias=True)
        x = y.clamp(0, 255).round().to(torch.uint8)
    top, left = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
    return x[..., top:top + res, left:left + res]


def preprocess(frames_nhwc_u8, res=224):
    x = preprocess_u8(frames_nhwc_u8, res).float() / 255.0
    m = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - m) / s


def _ln(x, w, b):
    return F.layer_norm(x.float(), (x.shape[-1],), _t(w), _t(b), 1e-5)


def encode_image(sd, x, heads=12, taps=None):
    """x: fp32 (N,3,R,R) normalised -> (N, out_dim)."""
    pre = 'visual.'
    w1 = _t(sd[pre + 'conv1.weight'])
    patch = w1.shape[-1]
    x = F.conv2d(x, w1, None, patch)                              # (N, width, g, g)
    n, width = x.shape[0], x.shape[1]
    x = x.reshape(n, width, -1).permute(0, 2, 1)
    cls = _t(sd[pre + 'class_embedding']).view(1, 1, -1).expand(n, 1, width)
    x = torch.cat([cls, x], dim=1) + _t(sd[pre + 'positional_embedding'])
    x = _ln(x, sd[pre + 'ln_pre.weight'], sd[pre + 'ln_pre.bias'])
    if taps is not None:
        taps['ln_pre'] = x
    T, hd = x.shape[1], width // heads
    i = 0
    while (pre + 'transformer.resblocks.%d.ln_1.weight' % i) in sd:
        p = pre + 'transformer.resblocks.%d.' % i
        y = _ln(x, sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'])
        qkv = y @ _t(sd[p + 'attn.in_proj_weight']).t() + _t(sd[p + 'attn.in_proj_bias'])
        q, k, v = qkv.split(width, dim=-1)
        sh = lambda t: t.reshape(n, T, heads, hd).permute(0, 2, 1, 3)
        a = torch.softmax((sh(q) @ sh(k).transpose(-1, -2)) / (hd ** 0.5), dim=-1) @ sh(v)
        a = a.permute(0, 2, 1, 3).reshape(n, T, width)
        x = x + a @ _t(sd[p + 'attn.out_proj.weight']).t() + _t(sd[p + 'attn.out_proj.bias'])
        y = _ln(x, sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'])
        y = y @ _t(sd[p + 'mlp.c_fc.weight']).t() + _t(sd[p + 'mlp.c_fc.bias'])
        y = y * torch.sigmoid(1.702 * y)                          # QuickGELU
        x = x + y @ _t(sd[p + 'mlp.c_proj.weight']).t() + _t(sd[p + 'mlp.c_proj.bias'])
        if taps is not None:
            taps['block%d' % i] = x
        i += 1
    x = _ln(x[:, 0, :], sd[pre + 'ln_post.weight'], sd[pre + 'ln_post.bias'])
    return x @ _t(sd[pre + 'proj'])


def embed(sd, frames_nhwc_u8, squeeze=True):
    with torch.no_grad():
        out = encode_image(sd, preprocess(frames_nhwc_u8))
        out = out.reshape(out.shape[0], -1)
        return (out.squeeze() if squeeze else out).numpy()


# ------------------------------------------------------------------------------------------------------------
# MAE ViT encoder (reference src/vision_models/mae.py:202-222 forward_encoder with mask_ratio=0; src/embeddings.py:
# 81 Resize(256, interpolation=3 = bicubic) for 'mae' names, :137-140 mae_base, :377-379 CLS token output).
# timm 0.5.4 (requirements.txt:20, not under /root/reference) Block restated: x += proj(MHA(qkv(norm1(x))));
# x += fc2(GELU_erf(fc1(norm2(x)))), LayerNorm eps 1e-6 (mae.py:279).  random_masking(mask_ratio=0) only permutes the
# patch tokens (mae.py:181-192); the CLS output is permutation invariant, so the shuffle is skipped.
# PARITY PINNING: timm is not installable here -> "parity unpinned"; cross-checked against transformers.ViTModel.
# ------------------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def mae_preprocess_u8(frames_nhwc_u8, resize=256, crop=224):
    """torchvision 0.10 tensor Resize(256, bicubic) (antialias off): float32 bicubic, clamp(0,255), round, uint8."""
    x = _t(frames_nhwc_u8).transpose(1, 2).transpose(1, 3).contiguous()
    n, c, h, w = x.shape
    nh, nw = resize_size(h, w, resize)
    if (nh, nw) != (h, w):
        y = F.interpolate(x.float(), size=(nh, nw), mode='bicubic', align_corners=False)
        x = y.clamp(0, 255).round().to(torch.uint8)
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
    return x[..., top:top + crop, left:left + crop]


def mae_preprocess(frames_nhwc_u8):
    x = mae_preprocess_u8(frames_nhwc_u8).float() / 255.0
    m = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (x - m) / s


def mae_encode(sd, x, heads=12, taps=None):
    w1 = _t(sd['patch_embed.proj.weight'])
    patch, width = w1.shape[-1], w1.shape[0]
    x = F.conv2d(x, w1, _t(sd['patch_embed.proj.bias']), patch)
    n = x.shape[0]
    x = x.reshape(n, width, -1).permute(0, 2, 1)
    pos = _t(sd['pos_embed'])
    x = x + pos[:, 1:, :]
    cls = (_t(sd['cls_token']) + pos[:, :1, :]).expand(n, -1, -1)
    x = torch.cat([cls, x], dim=1)
    T, hd = x.shape[1], width // heads
    ln = lambda t, w, b: F.layer_norm(t, (width,), _t(w), _t(b), 1e-6)
    i = 0
    while ('blocks.%d.norm1.weight' % i) in sd:
        p = 'blocks.%d.' % i
        y = ln(x, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'])
        qkv = y @ _t(sd[p + 'attn.qkv.weight']).t() + _t(sd[p + 'attn.qkv.bias'])
        q, k, v = qkv.split(width, dim=-1)
        sh = lambda t: t.reshape(n, T, heads, hd).permute(0, 2, 1, 3)
        a = torch.softmax((sh(q) @ sh(k).transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ sh(v)
        a = a.permute(0, 2, 1, 3).reshape(n, T, width)
        x = x + a @ _t(sd[p + 'attn.proj.weight']).t() + _t(sd[p + 'attn.proj.bias'])
        y = ln(x, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'])
        y = F.gelu(y @ _t(sd[p + 'mlp.fc1.weight']).t() + _t(sd[p + 'mlp.fc1.bias']))
        x = x + y @ _t(sd[p + 'mlp.fc2.weight']).t() + _t(sd[p + 'mlp.fc2.bias'])
        if taps is not None:
            taps['block%d' % i] = x
        i += 1
    x = ln(x, sd['norm.weight'], sd['norm.bias'])
    return x[:, 0, :]                                             # embeddings.py:379


def mae_embed(sd, frames_nhwc_u8, squeeze=True, heads=12):
    """heads: 12 for mae_base (mae.py:277), 16 for mae_large (mae.py:285)"""
    with torch.no_grad():
        out = mae_encode(sd, mae_preprocess(frames_nhwc_u8), heads=heads)
        return (out.squeeze() if squeeze else out).numpy()


# ------------------------------------------------------------------------------------------------------------------
# CLIP RN50 visual tower (reference src/embeddings.py:305-306 clip.load("RN50"), :309-314 transforms, :375-376
# encode_image).  openai/CLIP `ModifiedResNet` restated from its published model.py (not under /root/reference, unpinned
# git HEAD; transformers ships no CLIP-ResNet, so this restatement has NO independent cross-check: parity unpinned):
#   stem: conv1 3x3/2 (3->32) bn relu, conv2 3x3 (32->32) bn relu, conv3 3x3 (32->64) bn relu, AvgPool2d(2)
#   Bottleneck(inplanes, planes, stride): conv1 1x1 bn relu -> conv2 3x3 (stride 1!) bn relu -> AvgPool2d(stride) ->
#       conv3 1x1 bn ; downsample (stride > 1 or inplanes != 4*planes) = AvgPool2d(stride) -> conv 1x1 -> bn ; relu(out + id)
#   layers (3, 4, 6, 3), widths 64/128/256/512 (x4), strides 1/2/2/2 -> (N, 2048, 7, 7)
#   AttentionPool2d(7, 2048, heads 32, out 1024): tokens = [mean ; x_hw] + positional_embedding(50, 2048);
#       multi_head_attention_forward(query = tokens[:1], key = value = tokens) with separate q/k/v projections (+bias),
#       out = c_proj(attn)  -> (N, 1024).  BatchNorm eps 1e-5 (eval).
# ------------------------------------------------------------------------------------------------------------------
def _bn_eval(sd, p, x):
    return F.batch_norm(x, _t(sd[p + '.running_mean']), _t(sd[p + '.running_var']), _t(sd[p + '.weight']), _t(sd[p + '.bias']), False, 0.0, 1e-5)


def clip_rn50_features(sd, x, taps=None):
    v = 'visual.'
    x = F.relu(_bn_eval(sd, v + 'bn1', F.conv2d(x, _t(sd[v + 'conv1.weight']), None, 2, 1)))
    x = F.relu(_bn_eval(sd, v + 'bn2', F.conv2d(x, _t(sd[v + 'conv2.weight']), None, 1, 1)))
    x = F.relu(_bn_eval(sd, v + 'bn3', F.conv2d(x, _t(sd[v + 'conv3.weight']), None, 1, 1)))
    x = F.avg_pool2d(x, 2)
    if taps is not None:
        taps['stem'] = x
    inpl = 64
    for li, nb in enumerate((3, 4, 6, 3)):
        planes = 64 << li
        for bi in range(nb):
            p = v + 'layer%d.%d.' % (li + 1, bi)
            stride = 2 if (bi == 0 and li > 0) else 1
            o = F.relu(_bn_eval(sd, p + 'bn1', F.conv2d(x, _t(sd[p + 'conv1.weight']))))
            o = F.relu(_bn_eval(sd, p + 'bn2', F.conv2d(o, _t(sd[p + 'conv2.weight']), None, 1, 1)))
            if stride > 1:
                o = F.avg_pool2d(o, stride)
            o = _bn_eval(sd, p + 'bn3', F.conv2d(o, _t(sd[p + 'conv3.weight'])))
            idn = x
            if (p + 'downsample.0.weight') in sd:              # Sequential(OrderedDict('-1' AvgPool2d, '0' conv, '1' bn))
                idn = F.avg_pool2d(x, stride) if stride > 1 else x
                idn = _bn_eval(sd, p + 'downsample.1', F.conv2d(idn, _t(sd[p + 'downsample.0.weight'])))
            x = F.relu(o + idn)
            inpl = planes * 4
        if taps is not None:
            taps['layer%d' % (li + 1)] = x
    return x


def clip_rn50_attnpool(sd, x, heads=32):
    a = 'visual.attnpool.'
    n, c, h, w = x.shape
    t = x.reshape(n, c, h * w).permute(0, 2, 1)                        # (N, HW, C)
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + _t(sd[a + 'positional_embedding'])[None]
    q = t[:, :1] @ _t(sd[a + 'q_proj.weight']).t() + _t(sd[a + 'q_proj.bias'])
    k = t @ _t(sd[a + 'k_proj.weight']).t() + _t(sd[a + 'k_proj.bias'])
    vv = t @ _t(sd[a + 'v_proj.weight']).t() + _t(sd[a + 'v_proj.bias'])
    hd = c // heads
    sh = lambda z: z.reshape(n, -1, heads, hd).permute(0, 2, 1, 3)
    att = torch.softmax((sh(q) @ sh(k).transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ sh(vv)      # (N, heads, 1, hd)
    att = att.permute(0, 2, 1, 3).reshape(n, c)
    return att @ _t(sd[a + 'c_proj.weight']).t() + _t(sd[a + 'c_proj.bias'])


def clip_rn50_embed(sd, frames_nhwc_u8, squeeze=True, taps=None):
    with torch.no_grad():
        out = clip_rn50_attnpool(sd, clip_rn50_features(sd, preprocess(frames_nhwc_u8), taps=taps))
        return (out.squeeze() if squeeze else out).numpy()
