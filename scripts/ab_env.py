"""A/B of an environment switch on the ResNet50 batch-256 forward: per-op times of both arms + bit-identity of the embeddings.
usage: ab_env.py VAR [dtype]"""
import os, subprocess, sys
var, dt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'bf16')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
from pvr_habitat_amd import synth, _lib
from pvr_habitat_amd.embeddings import HipResNet50
m = HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype=%r, max_batch=256)
fr = torch.from_numpy(synth.frames(1, 256, 256, 256)).cuda()
out = m(fr); torch.cuda.synchronize()
np.save(sys.argv[1], out.cpu().numpy())
cap = 128; op_ms = (C.c_float * cap)(); op_fl = (C.c_double * cap)(); n_ops = C.c_int32(); acc = np.zeros(cap)
for _ in range(5):
    _lib.check(_lib.lib().pvr_encoder_profile(m._handle, C.c_void_p(fr.data_ptr()), 256, 256, 256, C.c_void_p(out.data_ptr()), out.stride(0), _lib.stream_ptr(), op_ms, op_fl, cap, C.byref(n_ops)))
    acc[:n_ops.value] += np.array(op_ms[:n_ops.value])
names = ['preprocess', 'stem', 'maxpool'] + m.op_names() + ['pool']
for i in range(n_ops.value): print('%%-40s %%.4f' %% (names[i], acc[i] / 5))
print('TOTAL %%.4f' %% (acc[:n_ops.value].sum() / 5))
''' % (root, dt)
res = {}
for v in ('0', '1'):
    r = subprocess.run([sys.executable, '-c', code, '/tmp/ab_%s.npy' % v], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res[v] = [l.rsplit(' ', 1) for l in r.stdout.strip().splitlines() if not l.startswith('/opt')]
import numpy as np
print('%s: embeddings bit-identical between arms: %s' % (var, np.array_equal(np.load('/tmp/ab_0.npy'), np.load('/tmp/ab_1.npy'))))
for (n0, a), (n1, b) in zip(res['0'], res['1']):
    if abs(float(a) - float(b)) > 0.003 or n0.startswith('TOTAL'):
        print('%-40s %s=0: %.4f ms   %s=1: %.4f ms' % (n0.strip(), var, float(a), var, float(b)))
