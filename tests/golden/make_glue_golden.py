"""Pin the GLUE of the reference's embedding path by running the reference's own Python (build container only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_glue_golden.py

What runs is the reference's code, unmodified, imported from /root/reference:
  src/embeddings.py            _get_embedding (name registry, transform construction), UberModel, EmbeddingNet.__init__ /
                               forward (:345-402), EmbeddingWrapper (:409-444)
  src/vision_models/moco.py    moco_conv5 / moco_conv3_compressed / moco_conv4_compressed (topology edits + key remapping)
  src/vision_models/resnet.py  resnet_conv5 / resnet_conv3_compressed / resnet_conv4_compressed
  src/vision_models/mae.py     get_2d_sincos_pos_embed (:23-70)
  behavioral_cloning/save_embedded_obs.py   read_habitat_data_from_pickle / _png, run (:29-172)
  src/test_model.py            test (:4-22)
What does NOT exist in this image and is replaced by stand-in modules in sys.modules (defined below, nothing else):
  gym (ObservationWrapper / Box: attribute holders), cv2 (imread through PIL with cv2's channel order), clip, timm,
  detectron2 (import-only stubs), and torchvision.  The torchvision stand-in is a restatement of torchvision's public
  ResNet / transforms definitions (models.resnet{18,34,50}, models.resnet.BasicBlock, T.Resize / CenterCrop /
  ConvertImageDtype / Normalize): it pins NOTHING about torchvision's arithmetic - that boundary stays "parity unpinned"
   - but with it in place every line of the reference's glue executes for real: NHWC->NCHW transposes, transform
  order, reshape(-1,*in_shape), view(-1,out_size).squeeze(), UberModel concat order, the moco/resnet loaders' surgery on the
  model and their key handling of real-layout checkpoints, save_embedded_obs' split/stack/concat rows and pickle schema,
  EmbeddingWrapper.observation, and test()'s episode / state semantics.
Inputs and weights come from pvr_habitat_amd.synth (regenerable on the GPU box): only OUTPUTS are stored, in
  glue_registry.json   name -> loader calls + transform construction log, for every registry name
  glue_embed.npz       EmbeddingNet outputs / attributes for a handful of names and frame shapes
  glue_save_obs.npz    save_embedded_obs.run outputs (pickle and png sources), .tar keys, EmbeddingWrapper, test()
  mae_sincos.npz       get_2d_sincos_pos_embed tables (full for a small grid, checksums + samples for B/16, L/16, H/14)
/root/reference is never read by tests at run time.
"""
import json
import os
import pickle
import random
import sys
import tempfile
import types
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from pvr_habitat_amd import synth                                   # noqa: E402
sys.path.insert(0, HERE)
import glue_inputs as GI                                            # noqa: E402

LOG = []                # construction / call log the stand-ins append to


# ------------------------------------------------------------------------------------------------------------------
# stand-in modules
# ------------------------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AnyModule(types.ModuleType):
    """import-only stub: every attribute is a dummy class (detectron2 / timm names the reference imports at module level)"""
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return type(k, (), {'__init__': lambda self, *a, **kw: None})


def install_stubs():
    # ---- gym: attribute holders only --------------------------------------------------------------------------
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class ObservationWrapper:
        def __init__(self, env):
            self.env = env
            self.observation_space = env.observation_space

    box_mod = _module('gym.spaces.box', Box=Box)
    spaces = _module('gym.spaces', box=box_mod, Box=Box)
    _module('gym', ObservationWrapper=ObservationWrapper, spaces=spaces)

    # ---- cv2.imread: the file's RGB reversed (what cv2 returns), None for a missing file -----------------------
    def imread(path):
        from PIL import Image
        if not os.path.isfile(path):
            return None
        return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[..., ::-1])
    _module('cv2', imread=imread)

    # ---- clip / timm / detectron2: import-only ------------------------------------------------------------------
    def clip_load(name, device='cpu'):
        LOG.append(('clip.load', name, str(device)))
        visual = types.SimpleNamespace(input_resolution=224)
        return types.SimpleNamespace(visual=visual, eval=lambda: None, train=lambda: None, parameters=lambda: []), None
    _module('clip', load=clip_load)
    for n in ('timm', 'timm.models', 'timm.models.vision_transformer', 'detectron2', 'detectron2.layers', 'detectron2.config',
              'detectron2.modeling', 'detectron2.modeling.meta_arch', 'detectron2.modeling.anchor_generator',
              'detectron2.modeling.backbone', 'detectron2.modeling.backbone.resnet', 'detectron2.modeling.box_regression',
              'detectron2.modeling.matcher', 'detectron2.modeling.poolers', 'detectron2.modeling.proposal_generator',
              'detectron2.modeling.roi_heads'):
        sys.modules[n] = _AnyModule(n)
    np.float = float                                                # mae.py:59 uses the removed alias

    # ---- torchvision.transforms: restated public definitions (torchvision 0.10 functional_tensor) ---------------
    class InterpolationMode:
        BILINEAR, BICUBIC = 'bilinear', 'bicubic'

    class Resize(nn.Module):
        def __init__(self, size, interpolation='bilinear', max_size=None, antialias=None):
            super().__init__()
            mode = {2: 'bilinear', 3: 'bicubic'}.get(interpolation, interpolation)
            LOG.append(('T.Resize', int(size), mode, bool(antialias)))
            self.size, self.mode, self.antialias = int(size), mode, bool(antialias)

        def forward(self, img):
            h, w = img.shape[-2:]
            short, long_ = (w, h) if w <= h else (h, w)
            if short == self.size:
                return img
            ns, nl = self.size, int(self.size * long_ / short)
            nh, nw = (nl, ns) if w <= h else (ns, nl)
            was_u8 = img.dtype == torch.uint8
            x = img.float() if not img.is_floating_point() else img
            x = F.interpolate(x, size=(nh, nw), mode=self.mode, align_corners=False, antialias=self.antialias)
            if was_u8:
                if self.mode == 'bicubic':
                    x = x.clamp(0, 255)
                x = x.round().to(torch.uint8)
            return x

    class CenterCrop(nn.Module):
        def __init__(self, size):
            super().__init__()
            LOG.append(('T.CenterCrop', int(size)))
            self.size = int(size)

        def forward(self, img):
            h, w = img.shape[-2:]
            top, left = int(round((h - self.size) / 2.0)), int(round((w - self.size) / 2.0))
            return img[..., top:top + self.size, left:left + self.size]

    class ConvertImageDtype(nn.Module):
        def __init__(self, dtype):
            super().__init__()
            LOG.append(('T.ConvertImageDtype', str(dtype)))
            assert dtype == torch.float

        def forward(self, img):
            return img.float() / 255.0 if img.dtype == torch.uint8 else img.float()

    class Normalize(nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            LOG.append(('T.Normalize', [float(m) for m in mean], [float(s) for s in std]))
            self.mean, self.std = list(mean), list(std)

        def forward(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            s = torch.tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            return (x - m) / s

    T = _module('torchvision.transforms', Resize=Resize, CenterCrop=CenterCrop, ConvertImageDtype=ConvertImageDtype,
                Normalize=Normalize, InterpolationMode=InterpolationMode)

    # ---- torchvision.models: restated public ResNet definition ---------------------------------------------------
    def conv3x3(i, o, stride=1):
        return nn.Conv2d(i, o, 3, stride, 1, bias=False)

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
            super().__init__()
            norm_layer = norm_layer or nn.BatchNorm2d
            self.conv1, self.bn1 = conv3x3(inplanes, planes, stride), norm_layer(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2, self.bn2 = conv3x3(planes, planes), norm_layer(planes)
            self.downsample, self.stride = downsample, stride

        def forward(self, x):
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            idn = x if self.downsample is None else self.downsample(x)
            return self.relu(out + idn)

    class Bottleneck(nn.Module):
        expansion = 4

        def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
            super().__init__()
            norm_layer = norm_layer or nn.BatchNorm2d
            self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, 1, bias=False), norm_layer(planes)
            self.conv2, self.bn2 = conv3x3(planes, planes, stride), norm_layer(planes)          # v1.5: stride on the 3x3
            self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), norm_layer(planes * 4)
            self.relu = nn.ReLU(inplace=True)
            self.downsample, self.stride = downsample, stride

        def forward(self, x):
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
            idn = x if self.downsample is None else self.downsample(x)
            return self.relu(out + idn)

    class ResNet(nn.Module):
        def __init__(self, block, layers, num_classes=1000):
            super().__init__()
            self._norm_layer = nn.BatchNorm2d
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = self._make_layer(block, 64, layers[0])
            self.layer2 = self._make_layer(block, 128, layers[1], 2)
            self.layer3 = self._make_layer(block, 256, layers[2], 2)
            self.layer4 = self._make_layer(block, 512, layers[3], 2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)

        def _make_layer(self, block, planes, blocks, stride=1):
            downsample = None
            if stride != 1 or self.inplanes != planes * block.expansion:
                downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                           nn.BatchNorm2d(planes * block.expansion))
            seq = [block(self.inplanes, planes, stride, downsample)]
            self.inplanes = planes * block.expansion
            seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
            return nn.Sequential(*seq)

        def forward(self, x):
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            x = torch.flatten(self.avgpool(x), 1)
            return self.fc(x)

    def _hub(name, block, layers, variant):
        def make(pretrained=False, progress=True):
            LOG.append(('models.' + name, bool(pretrained)))
            m = ResNet(block, layers)
            if make.synthetic:       # stands in for the hub download / random init: the build's synthetic weights for this name
                sd = synth.resnet50_state_dict(zlib.crc32(name.encode()) & 0x7fffffff, variant)
                msg = m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=False)
                assert all(k.startswith('fc.') for k in msg.missing_keys) and not msg.unexpected_keys
            return m
        make.synthetic = True
        return make

    r18, r34, r50 = (_hub('resnet18', BasicBlock, (2, 2, 2, 2), 'r18'), _hub('resnet34', BasicBlock, (3, 4, 6, 3), 'r34'),
                     _hub('resnet50', Bottleneck, (3, 4, 6, 3), 'conv5'))

    def plain_resnet50(pretrained=False, progress=True):                 # models.resnet.resnet50 of moco.py / resnet.py
        return ResNet(Bottleneck, (3, 4, 6, 3))
    resnet_mod = _module('torchvision.models.resnet', resnet50=plain_resnet50, BasicBlock=BasicBlock, Bottleneck=Bottleneck, ResNet=ResNet)
    models = _module('torchvision.models', resnet18=r18, resnet34=r34, resnet50=r50, resnet=resnet_mod)
    _module('torchvision', models=models, transforms=T)


# ------------------------------------------------------------------------------------------------------------------
# synthetic checkpoints in the layouts the reference loaders read (moco.py:7-8,14-21; resnet.py:7-8,33-38)
# ------------------------------------------------------------------------------------------------------------------
def write_checkpoint(path, name, family, variant):
    sd = synth.resnet50_state_dict(zlib.crc32(name.encode()) & 0x7fffffff, variant)
    prefix = 'module.encoder_q.' if family == 'moco' else 'module.'
    out = {prefix + k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    if family == 'moco':                                             # keys the loader must drop
        out['module.encoder_q.fc.0.weight'] = torch.zeros(4, 4)
        out['module.encoder_k.conv1.weight'] = torch.zeros(64, 3, 7, 7)
    else:
        out['module.fc.weight'] = torch.zeros(4, 2048)
    torch.save({'state_dict': out}, path)


def main():
    install_stubs()
    sys.path.insert(0, '/root/reference')
    sys.path.insert(0, '/root/reference/behavioral_cloning')
    import src.embeddings as E                                       # the reference
    from src.vision_models import mae as ref_mae
    from src.test_model import test as ref_test
    from pvr_habitat_amd import embeddings as P                       # only for the list of names to walk
    torch.set_num_threads(8)
    work = tempfile.mkdtemp(prefix='glue_')
    os.chdir(work)                                                   # the reference opens checkpoints relative to cwd

    # ---- (A) registry: which loader / checkpoint / transforms every name gets ------------------------------------
    real = {}
    for fn in ('moco_conv5', 'moco_conv3_compressed', 'moco_conv4_compressed', 'resnet_conv5', 'resnet_conv3_compressed',
               'resnet_conv4_compressed', 'mae_vit_base_patch16', 'mae_vit_large_patch16', 'mae_vit_huge_patch14', 'mask_rcnn_model'):
        real[fn] = getattr(E, fn)

        def logged(*a, _fn=fn, **kw):
            LOG.append((_fn, kw.get('checkpoint_path', a[0] if a else None)))
            return nn.Sequential()
        setattr(E, fn, logged)
    orig_load = torch.load
    torch.load = lambda *a, **k: (LOG.append(('torch.load', os.path.basename(str(a[0])))), {'model': {}})[1]
    for m in ('resnet18', 'resnet34', 'resnet50'):
        getattr(E.models, m).synthetic = False
    names = list(P._SINGLE) + list(P._UBER) + ['random', 'clip_vit', 'clip_rn50', 'mae_base', 'mae_large', 'mae_huge', 'maskrcnn_l3', 'true_state']
    registry = {}
    def describe(tr):
        """the transforms a name ends up with (the nn.Sequential _get_embedding returns), from the stand-ins' attributes"""
        out = []
        for t in tr:
            nm = type(t).__name__
            if nm == 'Resize':
                out.append(['Resize', t.size, t.mode, t.antialias])
            elif nm == 'CenterCrop':
                out.append(['CenterCrop', t.size])
            elif nm == 'Normalize':
                out.append(['Normalize', [float(m) for m in t.mean], [float(x) for x in t.std]])
            else:
                out.append([nm])
        return out

    for n in names:
        del LOG[:]
        entry = {}
        try:
            _, tr = E._get_embedding(n, 3, True, False)
            entry['transforms'] = describe(tr)
        except Exception as e:
            entry['raised'] = type(e).__name__
        entry['loaders'] = [list(x) for x in LOG if not x[0].startswith('T.')]
        registry[n] = entry
    del LOG[:]
    try:
        E._get_embedding('not_a_model', 3, True, False)
    except NotImplementedError as e:
        registry['__unknown__'] = {'raised': 'NotImplementedError', 'message': str(e)}
    for fn, f in real.items():
        setattr(E, fn, f)
    torch.load = orig_load
    for m in ('resnet18', 'resnet34', 'resnet50'):
        getattr(E.models, m).synthetic = True
    json.dump(registry, open(os.path.join(HERE, 'glue_registry.json'), 'w'), indent=0, sort_keys=True)
    print('registry:', len(registry), 'names')

    # ---- (B) EmbeddingNet end to end through the reference loaders ------------------------------------------------
    for name in ('moco_aug', 'moco_aug_l3', 'moco_aug_l4', 'resnet50_places_l3', 'resnet50_l4'):
        family, variant, ckpt = P._SINGLE[name]
        write_checkpoint(ckpt, name, family, variant)
    out = {}
    f64 = GI.frames()['f64']
    for name, tags in GI.EMBED_CASES:
        frames_by_tag = {t: GI.case_frames(name, t) for t in tags}
        net = E.EmbeddingNet(name, in_channels=3, pretrained=False, train=False, disable_cuda=True)
        out[name + '/out_size'] = np.int64(net.out_size)
        out[name + '/in_shape'] = np.array(tuple(net.in_shape))
        out[name + '/training'] = np.bool_(net.training)
        out[name + '/state_dict_keys'] = np.array(list(net.state_dict().keys()))
        for tag, fr in frames_by_tag.items():
            o = net(torch.from_numpy(fr))
            assert isinstance(o, np.ndarray) and o.dtype == np.float32
            out['%s/%s' % (name, tag)] = o
            print(name, tag, fr.shape, '->', o.shape)
        one = net(torch.from_numpy(f64[:1]))
        out[name + '/f64_single'] = one                               # N=1: squeezed to (O,)
        print(name, 'single ->', one.shape)
    # 'random' PVR: seed-dependent weights from torch's generator (embeddings.py:90-106), as main_bc_1 / save_embedded_obs seed it
    torch.manual_seed(3)
    net = E.EmbeddingNet('random', in_channels=3, pretrained=True, train=False, disable_cuda=True)
    out['random/out_size'] = np.int64(net.out_size)
    out['random/in_shape'] = np.array(tuple(net.in_shape))
    out['random/f64'] = net(torch.from_numpy(f64[:2]))
    out['random/state_dict_keys'] = np.array(list(net.state_dict().keys()))
    ts = E.EmbeddingNet('true_state')
    out['true_state/passthrough'] = ts(torch.arange(24, dtype=torch.float32).reshape(2, 1, 12))
    np.savez_compressed(os.path.join(HERE, 'glue_embed.npz'), **out)
    print('wrote glue_embed.npz', len(out), 'arrays')

    # ---- (C) save_embedded_obs.run: pickle and png sources; EmbeddingWrapper; test() -------------------------------
    import save_embedded_obs as S                                    # behavioral_cloning/save_embedded_obs.py (the reference)
    so = {}
    data_dir = os.path.join(work, 'data')
    raw, trajs, goals = GI.write_scene(data_dir)
    for source in ('pickle', 'png'):
        flags = S.parser.parse_args(['--data_path', data_dir, '--env', 'scene', '--embedding_name', 'resnet50',
                                     '--disable_pretrained_embedding', '--disable_cuda', '--source', source, '--batch_size', '4'])
        save_name = os.path.join(data_dir, 'scene_resnet50.pickle')
        if os.path.isfile(save_name):
            os.remove(save_name)
        S.run(flags)
        res = pickle.load(open(save_name, 'rb'))
        so[source + '/keys'] = np.array(list(res.keys()))
        for k, v in res.items():
            so['%s/%s' % (source, k)] = np.array([os.path.relpath(q, data_dir) for q in v]) if k == 'png' else np.array(v)
        tar = torch.load(os.path.join(data_dir, 'resnet50.tar'), map_location='cpu')
        so[source + '/tar_top_keys'] = np.array(list(tar.keys()))
        so[source + '/tar_state_keys'] = np.array(list(tar['embedding_model_state_dict'].keys()))
        print(source, {k: np.array(v).shape for k, v in res.items()})
    np.testing.assert_allclose(so['pickle/obs'], so['png/obs'], rtol=0, atol=1e-5)     # same frames either way (batch composition differs)
    S.run(flags)                                                      # idempotent: existing output -> immediate return

    # EmbeddingWrapper (embeddings.py:409-444) on a stub env with a (64,64,6) observation space
    env = types.SimpleNamespace(observation_space=types.SimpleNamespace(shape=(64, 64, 6)))
    net = E.EmbeddingNet('resnet50', in_channels=3, pretrained=False, train=False, disable_cuda=True)
    w = E.EmbeddingWrapper(env, net)
    so['wrapper/space_shape'] = np.array(w.observation_space.shape)
    so['wrapper/n_frames'] = np.int64(w.n_frames)
    so['wrapper/obs'] = w.observation(trajs[0][1])                    # (64,64,6) -> (2*2048,)
    print('wrapper', so['wrapper/space_shape'], so['wrapper/obs'].shape)

    # test() (src/test_model.py:4-22) on a scripted env / model: which calls happen in which order, what the stats hold
    calls = []
    stats = ref_test(GI.ScriptedModel(calls), GI.ScriptedEnv(calls), GI.STAT_KEYS, n_episodes=3)
    so['test/calls'] = np.array(calls)
    for k, v in stats.items():
        so['test/' + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, 'glue_save_obs.npz'), **so)
    print('wrote glue_save_obs.npz', len(so), 'arrays;', 'test() calls:', calls[:6], '...')

    # ---- (D) MAE fixed sin-cos position table (mae.py:23-70) --------------------------------------------------------
    ms = {'small_d64_g3': ref_mae.get_2d_sincos_pos_embed(64, 3, cls_token=True)}
    for tag, (dim, grid) in dict(b16=(768, 14), l16=(1024, 14), h14=(1280, 16)).items():
        tab = ref_mae.get_2d_sincos_pos_embed(dim, grid, cls_token=True)
        idx = (synth.bits(9, 'sincos_' + tag, 256) % np.uint64(tab.size)).astype(np.int64)
        ms[tag + '/shape'] = np.array(tab.shape)
        ms[tag + '/sum'] = np.float64(tab.sum())
        ms[tag + '/sq'] = np.float64((tab ** 2).sum())
        ms[tag + '/row_sums'] = tab.sum(1)
        ms[tag + '/idx'] = idx
        ms[tag + '/samples'] = tab.reshape(-1)[idx]
    np.savez_compressed(os.path.join(HERE, 'mae_sincos.npz'), **ms)
    print('wrote mae_sincos.npz')


if __name__ == '__main__':
    torch.manual_seed(1); random.seed(1); np.random.seed(1)
    main()
