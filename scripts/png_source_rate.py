"""how fast does the per-frame PNG source feed the encoder?  Synthetic PNG tree in the reference's layout
(save_opt_trajectories_png.py:44-58: <t>_<s>.png, <t>_goal.png, <t>.pickle; 64x64 frames as habitat_config/nav_task.yaml renders them),
read by save_embedded_obs.read_habitat_data_from_png: decode only (host thread pool) and decode + embed (ResNet50 on the GPU)."""
import os, pickle, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
from PIL import Image
from pvr_habitat_amd import synth, save_embedded_obs as S
from pvr_habitat_amd.embeddings import EmbeddingNet
T, L = int(os.environ.get("PNG_T", "24")), 250                   # 6000 frames (PNG_T=96: 24000, steady state)
d = tempfile.mkdtemp(prefix='png_')
fr = synth.smooth_frames(3, 512, 64, 64)
for t in range(T):
    for s in range(L):
        Image.fromarray(fr[(t * L + s) % 512][..., ::-1]).save(os.path.join(d, '%d_%d.png' % (t, s)))
    Image.fromarray(fr[t][..., ::-1]).save(os.path.join(d, '%d_goal.png' % t))
    pickle.dump(dict(action=np.zeros(L, np.int64), reward=np.zeros(L), done=np.zeros(L, bool), true_state=np.zeros((L, 12))), open(os.path.join(d, '%d.pickle' % t), 'wb'))
net = EmbeddingNet('resnet50', pretrained=False, max_batch=256)
net(torch.from_numpy(fr[:256]))
for workers in (1, 8, 32, 64):      # worker processes (png_decode.decode_parallel); first call per count = untimed pool start-up
    S.read_habitat_data_from_png(d, None, 2, decode_workers=workers)
    t0 = time.perf_counter(); data = S.read_habitat_data_from_png(d, None, -1, decode_workers=workers); el = time.perf_counter() - t0
    print('decode only, %2d worker processes: %6.0f frames/s' % (workers, T * L / el), flush=True)
for workers in (32,):
    S.read_habitat_data_from_png(d, None, 2, decode_workers=workers)
    t0 = time.perf_counter(); data = S.read_habitat_data_from_png(d, net, -1, batch=256, decode_workers=workers, gpu_decode=False); el = time.perf_counter() - t0
    print('decode + embed (ResNet50 bf16), host decode, %2d worker processes: %6.0f frames/s, obs %s' % (workers, T * L / el, data['obs'].shape), flush=True)
host_obs = data['obs']
# frames decoded ON the GPU (csrc/png_decode.hip): the host only reads file bytes
from pvr_habitat_amd import png_gpu
import glob
names = sorted(glob.glob(os.path.join(d, '*_*.png')))[:4000]
png_gpu.decode_files(names[:64])
for n in (250, 1000, 4000):
    torch.cuda.synchronize(); t0 = time.perf_counter(); o = png_gpu.decode_files(names[:n], threads=16); torch.cuda.synchronize(); el = time.perf_counter() - t0
    t1 = time.perf_counter(); png_gpu.read_files(names[:n], 16); rd = time.perf_counter() - t1  # native threads
    print('GPU decode only, %4d files per call: %6.0f frames/s (reading the file bytes alone: %6.0f files/s)' % (n, n / el, n / rd), flush=True)
for threads in (4, 16):
    S.read_habitat_data_from_png(d, net, 2, batch=256, decode_workers=threads, gpu_decode=True)
    t0 = time.perf_counter(); data = S.read_habitat_data_from_png(d, net, -1, batch=256, decode_workers=threads, gpu_decode=True); el = time.perf_counter() - t0
    print('decode + embed (ResNet50 bf16), GPU decode, %2d reader threads: %6.0f frames/s, bit-identical to the host-decoded run: %s'
          % (threads, T * L / el, bool(np.array_equal(data['obs'], host_obs))), flush=True)
x = torch.from_numpy(np.stack([fr[i % 512] for i in range(T * L)]))
from pvr_habitat_amd.embeddings import stream_embed
stream_embed(net, x[:1024], 256); t0 = time.perf_counter(); stream_embed(net, x, 256); el = time.perf_counter() - t0
print('same frames already decoded in host memory (stream_embed): %6.0f frames/s' % (T * L / el))
