"""Probe: how fast is the scene reader alone (scene_pickle.scene_rows into a BlockRing whose blocks are released at once)?"""
import os, sys, time, pickle, tempfile, shutil, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pvr_habitat_amd import scene_pickle as SP
from pvr_habitat_amd.save_embedded_obs import BlockRing

n, L, hw = 100000, 500, 64
d = tempfile.mkdtemp(prefix='pvr_reader_probe_')
try:
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (L, hw, hw, 6), dtype=np.uint8)
    raw = dict(obs=[np.roll(base, t, axis=1) for t in range(n // L)], action=[np.zeros(L, np.int64)] * (n // L), reward=[np.zeros(L)] * (n // L),
               done=[np.zeros(L, bool)] * (n // L), true_state=[np.zeros((L, 12), np.float32)] * (n // L))
    p = os.path.join(d, 'scene.pickle')
    with open(p, 'wb') as f:
        pickle.dump(raw, f, protocol=pickle.HIGHEST_PROTOCOL)
    del raw
    gb = os.path.getsize(p) / 1e9
    t0 = time.perf_counter(); SP.scene_index(p); t1 = time.perf_counter()
    print('index pass: %.2f s (%.2f GB/s)' % (t1 - t0, gb / (t1 - t0)))
    t0 = time.perf_counter(); SP.scene_rows(p, 0, n, lambda b: None); t1 = time.perf_counter()
    print('row pass, null sink: %.2f s (%.2f GB/s)' % (t1 - t0, gb / (t1 - t0)))
    for pinned in (False, True):
        ring = BlockRing(8192, (hw, hw, 6), count=2, pinned=pinned and torch.cuda.is_available())

        def reader():
            SP.scene_rows(p, 0, n, ring.put); ring.close()
        th = threading.Thread(target=reader); t0 = time.perf_counter(); th.start()
        while True:
            g = ring.get()
            if g is None:
                break
            ring.release(g[0])
        th.join(); t1 = time.perf_counter()
        print('row pass into a BlockRing (pinned=%s), blocks released at once: %.2f s (%.2f GB/s)' % (pinned, t1 - t0, gb / (t1 - t0)))
    t0 = time.perf_counter()
    with open(p, 'rb') as f:
        while f.read(64 << 20):
            pass
    t1 = time.perf_counter()
    print('plain file read: %.2f s (%.2f GB/s)' % (t1 - t0, gb / (t1 - t0)))
finally:
    shutil.rmtree(d, ignore_errors=True)
