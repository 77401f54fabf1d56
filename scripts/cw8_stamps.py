"""s_memtime stamps of chain_wave128 (diagnostic library variant built with -DCW8_STAMP; PVR_LIB points at it): per round and weight unit, cycles from the
unit's top to the end of its arithmetic and the time spent at its barrier, workgroup 13, waves 0 and 4.  python scripts/cw8_stamps.py [n]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pvr_habitat_amd import synth, _lib
from pvr_habitat_amd.embeddings import HipResNet50
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = synth.resnet50_state_dict(1, 'conv5')
m = HipResNet50(sd, 'conv5', compute_dtype='f16', max_batch=n)
fr = torch.from_numpy(synth.frames(2, n, 256, 256)).cuda()
out = torch.empty((n, 2048), device='cuda')
for _ in range(3):
    m.forward_into(fr, out)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
st = np.zeros((2, 9, 9, 3), np.uint64)
assert L.pvr_debug_cw8_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.astype(np.int64)                      # (the LAST wave128 launch of the forward: layer2.3, Cmn = 0)
base = st[st > 0].min()
for w in range(2):
    print('wave %d (quartet %d)' % (4 * w, w))
    for hr in range(9):
        if st[w, hr, 0, 0] == 0:
            continue
        comp = st[w, hr, :, 1] - st[w, hr, :, 0]
        barr = st[w, hr, :, 2] - st[w, hr, :, 1]
        print('  half-round %d: starts at %7d, %6d cycles | work per step: %s | at the barrier: %s' % (
            hr, st[w, hr, 0, 0] - base, st[w, hr, 8, 2] - st[w, hr, 0, 0], ' '.join('%5d' % c for c in comp), ' '.join('%4d' % c for c in barr)))
