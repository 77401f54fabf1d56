"""s_memtime stamps of the fused stem (diagnostic library variant built with -DSTEM_STAMP; PVR_LIB points at it): per image of workgroup (13, 1), per wave:
[loop top, past barrier 1, end of the MFMA + pooling phase, past the vmcnt / lgkmcnt wait, past barrier 2, end of the convert + copy-out phase]."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pvr_habitat_amd import synth, _lib
from pvr_habitat_amd.embeddings import HipResNet50
n = 256
sd = synth.resnet50_state_dict(1, 'conv5')
m = HipResNet50(sd, 'conv5', compute_dtype='f16', max_batch=n)
fr = torch.from_numpy(synth.frames(2, n, 256, 256)).cuda()
out = torch.empty((n, 2048), device='cuda')
for _ in range(3):
    m.forward_into(fr, out)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
st = np.zeros((8, 64, 6), np.int64)
assert L.pvr_debug_stem_stamps(st.ctypes.data_as(C.c_void_p)) == 0
for w in (0, 3, 7):
    print('wave %d: per image [barrier 1 wait | mfma + pool | memory wait | barrier 2 wait | convert + copy-out] and the image period' % w)
    for i in range(2, 12):
        a = st[w, i]
        if a[0] == 0:
            continue
        print('  image %2d: %5d %5d %5d %5d %5d | %6d' % (i, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], st[w, i + 1, 0] - a[0] if st[w, i + 1, 0] else 0))
