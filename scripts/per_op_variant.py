"""Per-launch HIP-event timings of one batch-256 forward of a ResNet50-family variant (conv5 / conv4 = *_l4 / conv3 = *_l3 ...)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth, _lib
from pvr_habitat_amd.embeddings import HipResNet50
v = sys.argv[1] if len(sys.argv) > 1 else 'conv4'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = HipResNet50(synth.resnet50_state_dict(1, v), v, compute_dtype='bf16', max_batch=n)
fr = torch.from_numpy(synth.frames(1, n, 256, 256)).cuda()
out = torch.empty((n, m.out_size), device='cuda')
m.forward_into(fr, out); torch.cuda.synchronize()
cap = 128
op_ms = (C.c_float * cap)(); op_fl = (C.c_double * cap)(); n_ops = C.c_int32()
for _ in range(2):
    _lib.check(_lib.lib().pvr_encoder_profile(m._handle, C.c_void_p(fr.data_ptr()), n, 256, 256, C.c_void_p(out.data_ptr()), out.stride(0),
                                              _lib.stream_ptr(), op_ms, op_fl, cap, C.byref(n_ops)))
names = ['preprocess', 'stem', 'maxpool'] + m.op_names() + ['pool/flatten']
tot = 0.0
for i in range(n_ops.value):
    tot += op_ms[i]
    if len(sys.argv) > 3 or i >= n_ops.value - 8:
        print('%-40s %8.3f ms %8.1f TFLOP/s' % (names[i] if i < len(names) else '?', op_ms[i], op_fl[i] / (op_ms[i] * 1e-3) / 1e12 if op_ms[i] > 0 else 0))
print('total %.3f ms' % tot)
