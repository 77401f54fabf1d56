"""Where the PNG-source loop spends its time (per decode group of 16 trajectories): python scripts/png_loop_breakdown.py"""
import os, pickle, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from PIL import Image
from pvr_habitat_amd import synth, save_embedded_obs as S
from pvr_habitat_amd.embeddings import EmbeddingNet, stream_embed
T, L = 64, 250
d = tempfile.mkdtemp(prefix='pngb_')
fr = synth.smooth_frames(3, 512, 64, 64)
for t in range(T):
    for s in range(L):
        Image.fromarray(fr[(t * L + s) % 512][..., ::-1]).save(os.path.join(d, '%d_%d.png' % (t, s)))
    Image.fromarray(fr[t][..., ::-1]).save(os.path.join(d, '%d_goal.png' % t))
    pickle.dump(dict(action=np.zeros(L, np.int64), reward=np.zeros(L), done=np.zeros(L, bool), true_state=np.zeros((L, 12))), open(os.path.join(d, '%d.pickle' % t), 'wb'))
net = EmbeddingNet('resnet50', pretrained=False, max_batch=256)
S.read_habitat_data_from_png(d, net, 17, batch=256)
listing = frozenset(os.listdir(d))
for rep in range(2):
    t0 = time.perf_counter(); g, _ = S._load_png_trajectories(d, 0, 16, 16, True, listing); torch.cuda.synchronize(); t1 = time.perf_counter()
    dec = g[0].pop()
    emb = stream_embed(net, dec, 256); t2 = time.perf_counter()
    obs = np.empty((len(dec) - 16, 4096), np.float32); obs[:, :2048] = emb[16:]; obs[:, 2048:] = emb[0]; t3 = time.perf_counter()
    print('group of 16 trajectories (%d files): list + read + decode %.1f ms, embed from HBM %.1f ms (%.0f frames/s), assemble rows %.1f ms'
          % (len(dec), (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(dec) / (t2 - t1), (t3 - t2) * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); data = S.read_habitat_data_from_png(d, net, -1, batch=256); el = time.perf_counter() - t0; pr.disable()
print('whole loop: %.0f frames/s' % (T * L / el))
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
