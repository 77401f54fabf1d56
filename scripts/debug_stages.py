"""GPU debugging aid: stage-wise error of the HIP encoder vs the oracle for several (frame size, n, max_batch)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50
from oracle import encoder_oracle as eo

torch.set_num_threads(16)
sd = synth.resnet50_state_dict(1, 'conv5')


def rel(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


for (h, n, mb) in ((64, 4, 4), (64, 3, 8), (256, 4, 4), (256, 2, 4), (128, 2, 4), (64, 1, 4)):
    fr = synth.smooth_frames(5, n, h, h)
    taps = {}
    with torch.no_grad():
        ref_out = eo.resnet50_features(sd, eo.preprocess(fr), 'conv5', taps=taps).reshape(n, 2048).numpy()
    u8 = eo.preprocess_u8(fr).permute(0, 2, 3, 1).numpy().astype(np.float32)
    m = HipResNet50(sd, 'conv5', compute_dtype='f16', max_batch=mb)
    d = torch.from_numpy(fr).cuda()
    res = {}
    m.debug_stop_after('pre'); m(d)
    pre = m.tap('pre', n * 230 * 232 * 4).cpu().numpy().reshape(n, 230, 232, 4)
    res['pre'] = float(np.abs(pre[:, 3:227, 3:227, :3] + 128.0 - u8).max())
    for name, key in (('stem', 'conv1'), ('pool', 'stem'), ('layer1', 'layer1'), ('layer2', 'layer2'), ('layer3', 'layer3'), ('layer4', 'layer4')):
        m.debug_stop_after(name); m(d)
        r = taps[key].permute(0, 2, 3, 1).contiguous().numpy()
        g = m.tap(name, r.size).cpu().numpy().reshape(r.shape)
        res[name] = rel(g, r)
        if res[name] > 0.01 and name == 'stem':
            e = np.abs(g - r).reshape(n, 112, 112, 64)
            print('   stem err by frame', e.mean((1, 2, 3)), 'by row(8)', e.mean((0, 2, 3))[::14], 'by col', e.mean((0, 1, 3))[::14])
    m.debug_stop_after(''); out = m(d).cpu().numpy()
    res['emb'] = rel(out, ref_out)
    print('h=%d n=%d mb=%d' % (h, n, mb), {k: '%.2e' % v for k, v in res.items()}, flush=True)
