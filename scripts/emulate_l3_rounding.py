"""CPU emulation of rounding placements for the compressed PVRs (`*_l3` = variant conv3, `*_l4` = conv4): which tensors may stay 16-bit
before the element-wise output (no final average pool) leaves the north-star 1e-3.  Each policy says, per ResNet stage, whether conv
WEIGHTS are rounded to f16 ('w'), conv-operand ACTIVATIONS are rounded ('a') and the RESIDUAL stream is rounded ('r'); the head is fp32.
python scripts/emulate_l3_rounding.py [variant] [seeds...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from pvr_habitat_amd import synth
from oracle import encoder_oracle as eo

H = torch.float16


def q(x, on):
    return x.to(H).float() if on else x


def conv_bn(sd, conv, bn, x, stride, pad, wq, two_term=False):
    w = eo._t(sd[conv + '.weight'])
    b = eo._t(sd[conv + '.bias']) if (conv + '.bias') in sd else None
    scale = eo._t(sd[bn + '.weight']) / torch.sqrt(eo._t(sd[bn + '.running_var']) + eo.BN_EPS)
    shift = eo._t(sd[bn + '.bias']) - eo._t(sd[bn + '.running_mean']) * scale
    if b is not None:
        shift = shift + b * scale
    wf = w * scale.view(-1, 1, 1, 1)
    if wq:
        hi = wf.to(H).float()
        wf = hi + (wf - hi).to(H).float() if two_term else hi
    return F.conv2d(x, wf, None, stride, pad) + shift.view(1, -1, 1, 1)


def run(sd, x, variant, pol):
    """pol[stage] = set of flags among 'w','a','r','2' (two-term weights), stage in 0..4 (0 = stem)"""
    p0 = pol[0]
    stem = F.relu(conv_bn(sd, 'conv1', 'bn1', x, 2, 3, 'w' in p0, '2' in p0))
    x = F.max_pool2d(q(stem, 'a' in p0), 3, 2, 1)
    stages = 4 if variant == 'conv4' else 3
    for li in range(stages):
        P = pol[li + 1]
        nested = (variant == 'conv4' and li == 3) or (variant == 'conv3' and li == 2)
        xo = q(x, 'a' in P)                                  # 16-bit operand copy of the (possibly fp32) stream
        for bi in range((3, 4, 6, 3)[li]):
            p = ('layer%d.0.%d' if nested else 'layer%d.%d') % (li + 1, bi)
            Pb = P
            if isinstance(P, dict):
                Pb = P.get(bi, P['*'])
                xo = q(x, 'a' in Pb)
            stride = 2 if (bi == 0 and li > 0) else 1
            o = q(F.relu(conv_bn(sd, p + '.conv1', p + '.bn1', xo, 1, 0, 'w' in Pb, '2' in Pb)), 'a' in Pb)
            o = q(F.relu(conv_bn(sd, p + '.conv2', p + '.bn2', o, stride, 1, 'w' in Pb, '2' in Pb)), 'a' in Pb)
            o = conv_bn(sd, p + '.conv3', p + '.bn3', o, 1, 0, 'w' in Pb, '2' in Pb)
            if (p + '.downsample.0.weight') in sd:
                idn = q(conv_bn(sd, p + '.downsample.0', p + '.downsample.1', xo, stride, 0, 'w' in Pb, '2' in Pb), 'r' in Pb)
            else:
                idn = x
            x = q(F.relu(o + idn), 'r' in Pb)
            xo = q(x, 'a' in Pb)
    p = 'layer3.1' if variant == 'conv3' else 'layer4.1'
    return eo.basic_block(sd, p, x, q=None)                 # fp32 head


if __name__ == '__main__':
    variant = sys.argv[1] if len(sys.argv) > 1 else 'conv3'
    seeds = [int(a) for a in sys.argv[2:]] or [2, 3, 4]
    torch.set_num_threads(8)
    A = {'w', 'a', 'r'}
    pols = {
        'all f16 (+fp32 head)': [A] * 5,
        'built: fp32 residual from layer3 on': [A, A, A, {'w', 'a'}, {'w', 'a'}],
        'two-term weights in layer3': [A, A, A, {'w', 'a', '2'}, {'w', 'a', '2'}],
        'two-term weights in layer2+3': [A, A, {'w', 'a', 'r', '2'}, {'w', 'a', '2'}, {'w', 'a', '2'}],
        'two-term weights everywhere': [{'w', 'a', 'r', '2'}] * 3 + [{'w', 'a', '2'}] * 2,
        'two-term weights everywhere + fp32 residual everywhere': [{'w', 'a', '2'}] * 5,
        'last 2 blocks of layer3 fp32 (weights + operands)': [A, A, A, {'*': {'w', 'a'}, 4: set(), 5: set()}, {'w', 'a'}],
        'last 3 blocks of layer3 fp32': [A, A, A, {'*': {'w', 'a'}, 3: set(), 4: set(), 5: set()}, {'w', 'a'}],
        'layer3 all fp32': [A, A, A, set(), set()],
        'two-term weights in last 3 blocks of layer3': [A, A, A, {'*': {'w', 'a'}, 3: {'w', 'a', '2'}, 4: {'w', 'a', '2'}, 5: {'w', 'a', '2'}}, {'w', 'a'}],
        'layer3 all fp32 + fp32 residual in layer2': [A, A, {'w', 'a'}, set(), set()],
        'layer3 all fp32 + fp32 residual in layer1+2': [A, {'w', 'a'}, {'w', 'a'}, set(), set()],
        'layer2+3 all fp32': [A, A, set(), set(), set()],
        'fp32 residual everywhere (f16 operands + weights)': [{'w', 'a'}] * 5,
        'l4: layer4 all fp32, fp32 residual from layer3': [A, A, A, {'w', 'a'}, set()],
        'l4: layer4 all fp32, fp32 residual from layer2': [A, A, {'w', 'a'}, {'w', 'a'}, set()],
        'l4: layer3+4 all fp32, fp32 residual in layer2': [A, A, {'w', 'a'}, set(), set()],
        'layer2+3 fp32 residual, layer3 two-term': [A, A, {'w', 'a'}, {'w', 'a', '2'}, {'w', 'a', '2'}],
    }
    for seed in seeds:
        sd = synth.resnet50_state_dict(seed, variant)
        fr = synth.smooth_frames(20 + seed, 2, 128, 128)
        with torch.no_grad():
            x = eo.preprocess(fr)
            ref = eo.resnet50_features(sd, x, variant).reshape(2, -1).numpy()
            for name, pol in pols.items():
                if os.environ.get('ONLY') and os.environ['ONLY'] not in name:
                    continue
                if variant == 'conv3':
                    pol = pol[:4]
                out = run(sd, x, variant, pol).reshape(2, -1).numpy()
                l2 = np.linalg.norm(out - ref) / np.linalg.norm(ref)
                mx = np.abs(out - ref).max() / np.abs(ref).max()
                print('seed %d  %-58s rel-L2 %.2e  max-norm %.2e' % (seed, name, l2, mx), flush=True)
