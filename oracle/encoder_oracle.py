"""CPU fp32 restatement of the reference's frozen-encoder embedding path (test infrastructure).

Follows, line by line:
  * reference src/embeddings.py:80-85   default transforms: Resize(256) -> CenterCrop(224) ->
                                        ConvertImageDtype(float) -> Normalize(mean, std)
  * reference src/embeddings.py:386-402 EmbeddingNet.forward: NHWC->NCHW, transforms, model,
                                        view(-1, out_size).squeeze().cpu().numpy()
  * reference src/embeddings.py:44-57   UberModel: concat of separately loaded models
  * reference src/vision_models/moco.py:6-113, resnet.py:6-104  topology edits (fc / avgpool /
                                        layer4 emptied, BasicBlock compression heads)
Third-party arithmetic that is NOT under /root/reference (torchvision==0.10.0, requirements.txt:4)
is restated from its public definition:
  * torchvision.transforms.functional_tensor.resize: short side -> size, long side
    int(size*long/short); unchanged if the short side already matches; uint8 input is cast to
    float32, F.interpolate(bilinear, align_corners=False), then round() and cast back to uint8
  * CenterCrop: top = int(round((H-224)/2.0)), left likewise
  * ConvertImageDtype(float): /255 ; Normalize: (x-mean)/std
  * resnet50 v1.5 (stride on the 3x3), BatchNorm eval (eps 1e-5), maxpool 3x3/2 pad 1,
    AdaptiveAvgPool2d(1), BasicBlock (conv3x3-bn-relu-conv3x3-bn + downsample, relu)

PARITY PINNING: torchvision / MoCo checkpoints are not installable/fetchable offline and the
reference has no tests, so
  (a) the ARITHMETIC of this file (torchvision's ResNet / transforms definitions) is "parity unpinned": it is restated from the
      public definitions and cross-checked against an independent implementation (transformers.ResNetModel with name-remapped
      weights, tests/test_oracle_encoder.py);
  (b) the GLUE (layout, transform order and parameters per name, reshape / squeeze, UberModel concat order, the moco.py /
      resnet.py loaders' surgery and key handling, save_embedded_obs split / concat rows and pickle schema, EmbeddingWrapper,
      test()) is pinned by fixtures the reference's OWN code produced: tests/golden/make_glue_golden.py (build container only)
      runs /root/reference/src/embeddings.py etc. with stand-in gym / cv2 / clip / timm / detectron2 / torchvision modules;
      tests/test_glue_golden.py checks this oracle against them.
torchvision's uint8 rounding of Resize is restated, not executed.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))


# --------------------------------------------------------------------------------------------
# transforms  (reference src/embeddings.py:80-85)
# --------------------------------------------------------------------------------------------
def resize_size(h, w, size=256):
    """torchvision resize with int size: smaller edge -> size."""
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def resize_u8(x_nchw_u8, size=256):
    n, c, h, w = x_nchw_u8.shape
    nh, nw = resize_size(h, w, size)
    if (nh, nw) == (h, w):
        return x_nchw_u8
    y = F.interpolate(x_nchw_u8.float(), size=(nh, nw), mode='bilinear', align_corners=False)
    return y.round().to(torch.uint8)


def center_crop(x, size=224):
    h, w = x.shape[-2:]
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return x[..., top:top + size, left:left + size]


def preprocess_u8(frames_nhwc_u8, resize=256, crop=224, crop_pos=0):
    """uint8 (N,H,W,3) -> uint8 (N,3,crop,crop): the integer part of the transforms.
    crop_pos 0 = CenterCrop (the reference, embeddings.py:82); 1..4 = tl, tr, bl, br corner windows of the resized frame
    (torchvision FiveCrop order; build-defined 5-crop extension)."""
    x = _t(frames_nhwc_u8)
    x = x.transpose(1, 2).transpose(1, 3).contiguous()        # embeddings.py:392
    r = resize_u8(x, resize)
    if crop_pos == 0:
        return center_crop(r, crop)
    h, w = r.shape[-2:]
    top = h - crop if crop_pos in (3, 4) else 0
    left = w - crop if crop_pos in (2, 4) else 0
    return r[..., top:top + crop, left:left + crop].contiguous()


def preprocess(frames_nhwc_u8, resize=256, crop=224, mean=IMAGENET_MEAN, std=IMAGENET_STD, crop_pos=0):
    x = preprocess_u8(frames_nhwc_u8, resize, crop, crop_pos).float() / 255.0
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - m) / s


# --------------------------------------------------------------------------------------------
# ResNet50 (functional, eval mode)
# --------------------------------------------------------------------------------------------
def _q(x, q):
    """Optional emulation of the HIP path's storage rounding (bf16 / f16 activations+weights)."""
    if q is None:
        return x
    return x.to(q).float()


def _bn(sd, p, x):
    return F.batch_norm(x, _t(sd[p + '.running_mean']), _t(sd[p + '.running_var']),
                        _t(sd[p + '.weight']), _t(sd[p + '.bias']), False, 0.0, BN_EPS)


def _conv_bn(sd, conv, bn, x, stride=1, pad=0, q=None):
    w = _t(sd[conv + '.weight'])
    b = _t(sd[conv + '.bias']) if (conv + '.bias') in sd else None
    if q is None:
        return _bn(sd, bn, F.conv2d(x, w, b, stride, pad))
    # emulate folded-BN weights rounded to the storage type, fp32 accumulate
    scale = _t(sd[bn + '.weight']) / torch.sqrt(_t(sd[bn + '.running_var']) + BN_EPS)
    shift = _t(sd[bn + '.bias']) - _t(sd[bn + '.running_mean']) * scale
    if b is not None:
        shift = shift + b * scale
    wf = _q(w * scale.view(-1, 1, 1, 1), q)
    return F.conv2d(x, wf, None, stride, pad) + shift.view(1, -1, 1, 1)


def bottleneck(sd, p, x, stride, q=None):
    out = _q(F.relu(_conv_bn(sd, p + '.conv1', p + '.bn1', x, q=q)), q)
    out = _q(F.relu(_conv_bn(sd, p + '.conv2', p + '.bn2', out, stride, 1, q=q)), q)
    out = _conv_bn(sd, p + '.conv3', p + '.bn3', out, q=q)
    if (p + '.downsample.0.weight') in sd:
        idn = _q(_conv_bn(sd, p + '.downsample.0', p + '.downsample.1', x, stride, q=q), q)
    else:
        idn = x
    return F.relu(out + idn)


def basic_block(sd, p, x, q=None):
    """torchvision BasicBlock with a conv3x3(+bias)+BN downsample (moco.py:35-50, 79-94)."""
    out = _q(F.relu(_conv_bn(sd, p + '.conv1', p + '.bn1', x, 1, 1, q=q)), q)
    out = _conv_bn(sd, p + '.conv2', p + '.bn2', out, 1, 1, q=q)
    idn = _q(_conv_bn(sd, p + '.downsample.0', p + '.downsample.1', x, 1, 1, q=q), q)
    return F.relu(out + idn)


def plain_basic_block(sd, p, x, stride, q=None):
    """torchvision BasicBlock of resnet18 / resnet34: stride on conv1, identity or 1x1(stride)+BN downsample."""
    out = _q(F.relu(_conv_bn(sd, p + '.conv1', p + '.bn1', x, stride, 1, q=q)), q)
    out = _conv_bn(sd, p + '.conv2', p + '.bn2', out, 1, 1, q=q)
    idn = x
    if (p + '.downsample.0.weight') in sd:
        idn = _q(_conv_bn(sd, p + '.downsample.0', p + '.downsample.1', x, stride, 0, q=q), q)
    return F.relu(out + idn)


def resnet50_features(sd, x, variant='conv5', q=None, taps=None):
    """x: fp32 (N,3,224,224) normalised.  Returns the model output before flatten."""
    stem = F.relu(_conv_bn(sd, 'conv1', 'bn1', x, 2, 3, q=None if q is None else q))
    x = F.max_pool2d(_q(stem, q), 3, 2, 1)
    if taps is not None:
        taps['conv1'] = stem
        taps['stem'] = x
    if variant in ('r18', 'r34'):                    # embeddings.py:112-117: torchvision resnet18 / resnet34, fc = Identity
        layers = (2, 2, 2, 2) if variant == 'r18' else (3, 4, 6, 3)
        for li in range(4):
            for bi in range(layers[li]):
                x = plain_basic_block(sd, 'layer%d.%d' % (li + 1, bi), x, 2 if (bi == 0 and li > 0) else 1, q=q)
                if not (li == 3 and bi == layers[li] - 1):
                    x = _q(x, q)
            if taps is not None:
                taps['layer%d' % (li + 1)] = x
        return F.adaptive_avg_pool2d(x, 1)
    stages = 4 if variant in ('conv5', 'conv4') else 3
    for li in range(stages):
        nested = (variant == 'conv4' and li == 3) or (variant == 'conv3' and li == 2)
        for bi in range((3, 4, 6, 3)[li]):
            p = ('layer%d.0.%d' if nested else 'layer%d.%d') % (li + 1, bi)
            x = bottleneck(sd, p, x, 2 if (bi == 0 and li > 0) else 1, q=q)
            last = (li == stages - 1 and bi == (3, 4, 6, 3)[li] - 1)
            if not (last and variant == 'conv5'):
                x = _q(x, q)
        if taps is not None:
            taps['layer%d' % (li + 1)] = x
    if variant == 'conv5':
        return F.adaptive_avg_pool2d(x, 1)          # fc = Identity / empty Sequential
    p = 'layer3.1' if variant == 'conv3' else 'layer4.1'
    return basic_block(sd, p, x, q=q)               # avgpool/fc (and layer4) are empty Sequentials


OUT_SIZE = {'conv5': 2048, 'conv4': 2058, 'conv3': 2156, 'r18': 512, 'r34': 512}


def embed(sd, frames_nhwc_u8, variant='conv5', q=None, squeeze=True, crop_pos=0):
    """EmbeddingNet.forward (embeddings.py:386-402) for one ResNet50-family model."""
    with torch.no_grad():
        x = preprocess(frames_nhwc_u8, crop_pos=crop_pos)
        if q is not None:
            # HIP path feeds the stem exact uint8 values with normalisation folded into the
            # weights; emulate only weight/activation storage rounding here.
            pass
        out = resnet50_features(sd, x, variant, q=q)
        out = out.reshape(-1, OUT_SIZE[variant])
        if squeeze:
            out = out.squeeze()
        return out.numpy()


def embed_uber(sds_variants, frames_nhwc_u8, q=None):
    """UberModel (embeddings.py:44-57): concat along dim 1 of separately loaded models."""
    outs = [embed(sd, frames_nhwc_u8, v, q=q, squeeze=False) for sd, v in sds_variants]
    return np.concatenate(outs, axis=1).squeeze()


def split_embed_concat(embed_fn, obs_nhw6c, n_frames):
    """save_embedded_obs.py:151-156: (N,H,W,3n) -> frames stacked on batch -> embed -> (N, n*O)."""
    o = np.concatenate(np.split(obs_nhw6c, n_frames, axis=3), axis=0)
    e = embed_fn(o)
    return np.concatenate(np.split(e, n_frames, axis=0), axis=-1)
