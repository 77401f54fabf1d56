"""Per-launch HIP-event times of one variant's plan at batch N (pvr_encoder_profile), with the kernel family each launch runs as.
python scripts/variant_per_op.py conv3 f16 256 [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pvr_habitat_amd import synth, _lib
from pvr_habitat_amd.embeddings import HipResNet50

variant = sys.argv[1] if len(sys.argv) > 1 else 'conv3'
dt = sys.argv[2] if len(sys.argv) > 2 else 'f16'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sd = synth.resnet50_state_dict(1, variant)
m = HipResNet50(sd, variant, compute_dtype=dt, max_batch=n)
fr = torch.from_numpy(synth.frames(2, n, 256, 256)).cuda()
out = torch.empty((n, m.out_size), device='cuda')
m.forward_into(fr, out)
torch.cuda.synchronize()
names = ['preprocess', 'stem', 'maxpool'] + m.op_names() + ['pool/flatten']
kn = ['', '', ''] + m.kernel_names(n) + ['']
cap = 160
op_ms = (C.c_float * cap)(); op_fl = (C.c_double * cap)(); n_ops = C.c_int32()
acc = [0.0] * cap
for r in range(reps):
    _lib.check(_lib.lib().pvr_encoder_profile(m._handle, C.c_void_p(fr.data_ptr()), n, 256, 256, C.c_void_p(out.data_ptr()), out.stride(0), _lib.stream_ptr(),
                                              op_ms, op_fl, cap, C.byref(n_ops)))
    for i in range(n_ops.value):
        acc[i] += op_ms[i] / reps
tot = 0.0
for i in range(n_ops.value):
    tf = op_fl[i] / (acc[i] * 1e-3) / 1e12 if acc[i] > 0 else 0.0
    tot += acc[i]
    print('%-44s %-22s %8.3f ms %8.1f TFLOP/s' % (names[i] if i < len(names) else '?', kn[i] if i < len(kn) else '', acc[i], tf))
print('total %.3f ms for %d frames = %.1f frames/s (one lane, per-launch events)' % (tot, n, n / tot * 1e3))
