"""GPU debugging aid: stage-wise error of the HIP CLIP-ViT plan vs the oracle."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
from oracle import vit_oracle as vo

torch.set_num_threads(16)
patch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variant = 'clip_b32' if patch == 32 else 'clip_b16'
sd = synth.clip_vit_state_dict(1, patch=patch)
n = 2
fr = synth.smooth_frames(41, n, 224, 224)
x = vo.preprocess(fr)
taps = {}
with torch.no_grad():
    ref = vo.encode_image(sd, x, taps=taps).numpy()
    pe = F.conv2d(x, torch.from_numpy(sd['visual.conv1.weight']), None, patch)
    pe = pe.reshape(n, 768, -1).permute(0, 2, 1).reshape(-1, 768)
    x0 = taps['ln_pre']
    y = F.layer_norm(x0, (768,), torch.from_numpy(sd['visual.transformer.resblocks.0.ln_1.weight']), torch.from_numpy(sd['visual.transformer.resblocks.0.ln_1.bias']), 1e-5)
    qkv = y @ torch.from_numpy(sd['visual.transformer.resblocks.0.attn.in_proj_weight']).t() + torch.from_numpy(sd['visual.transformer.resblocks.0.attn.in_proj_bias'])
    T = x0.shape[1]
    q, k, v = qkv.split(768, -1)
    sh = lambda t: t.reshape(n, T, 12, 64).permute(0, 2, 1, 3)
    att = (torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / 8.0, -1) @ sh(v)).permute(0, 2, 1, 3).reshape(n * T, 768)
    blk = 'visual.transformer.resblocks.0.'
    res0 = x0.reshape(-1, 768) + att @ torch.from_numpy(sd[blk + 'attn.out_proj.weight']).t() + torch.from_numpy(sd[blk + 'attn.out_proj.bias'])
    y2 = F.layer_norm(res0, (768,), torch.from_numpy(sd[blk + 'ln_2.weight']), torch.from_numpy(sd[blk + 'ln_2.bias']), 1e-5)
    fc = y2 @ torch.from_numpy(sd[blk + 'mlp.c_fc.weight']).t() + torch.from_numpy(sd[blk + 'mlp.c_fc.bias'])
    fc = fc * torch.sigmoid(1.702 * fc)
refs = {'res0': res0.numpy(), 'fc0': fc.numpy(), 'pe': pe.numpy(), 'ln_pre': x0.reshape(-1, 768).numpy(), 'qkv0': qkv.reshape(-1, 2304).numpy(), 'att0': att.numpy(),
        'block0': taps['block0'].reshape(-1, 768).numpy(), 'block5': taps['block5'].reshape(-1, 768).numpy(), 'block11': taps['block11'].reshape(-1, 768).numpy()}
m = HipResNet50(sd, variant, compute_dtype='f16', max_batch=4)
d = torch.from_numpy(fr).cuda()
for name, r in refs.items():
    m.debug_stop_after(name); m(d)
    g = m.tap(name, r.size).cpu().numpy().reshape(r.shape)
    print('%-8s rel-L2 %.3e  max|ref| %.3f' % (name, np.linalg.norm(g - r) / np.linalg.norm(r), np.abs(r).max()), flush=True)
    if name == 'att0' and np.linalg.norm(g - r) / np.linalg.norm(r) > 1e-2:
        e = np.abs(g - r).reshape(n, T, 12, 64)
        print('   att err by token(8)', e.mean((0, 2, 3))[:8], 'by head', e.mean((0, 1, 3)), 'by d(8)', e.mean((0, 1, 2))[::8])
m.debug_stop_after(''); out = m(d).cpu().numpy()
print('final rel-L2 %.3e' % (np.linalg.norm(out - ref) / np.linalg.norm(ref)))
