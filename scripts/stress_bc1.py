"""Stress for the round-2 driver abort (GPUTEST_r02: SIGABRT in tests/test_gpu_pipeline.py::test_main_bc_1_random_pvr_in_process):
runs that test's body N times in ONE process, outside pytest (so nothing captures stderr: an HSA memory-fault line, a glibc heap
message or a C++ terminate() text lands in the log), optionally with garbage cycles in between so that handle destruction happens
at GC time.   python scripts/stress_bc1.py [iterations] [gc|nogc]"""
import faulthandler, gc, os, sys, tempfile, pathlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
faulthandler.enable()
import numpy as np
import torch
import pickle
from pvr_habitat_amd import synth, main_bc_1 as M1
from pvr_habitat_amd.arguments import make_parser


def body(tmp_path):
    """the round-2 form of the test: both runs IN this process"""
    lens = (60, 50)
    fr = synth.smooth_frames(41, sum(lens), 64, 128).reshape(sum(lens), 64, 64, 6)
    cuts = np.cumsum((0,) + lens)
    rng = np.random.default_rng(1)
    raw = dict(obs=[fr[a:b] for a, b in zip(cuts[:-1], cuts[1:])], action=[rng.integers(0, 3, L) for L in lens],
               reward=[np.zeros(L, np.float32) for L in lens], done=[np.eye(1, L, L - 1, dtype=bool)[0] for L in lens],
               true_state=[np.zeros((L, 12), np.float32) for L in lens])
    pickle.dump(raw, open(tmp_path / 'scene.pickle', 'wb'))
    for mode in ('fused', 'autograd'):
        args = ['--data_path', str(tmp_path), '--save_path', str(tmp_path / mode), '--env', 'scene', '--to_env', 'scene',
                '--embedding_name', 'random', '--run_id', '3', '--unroll_length', '8', '--batch_size', '4', '--batch_norm',
                '--max_frames', '320', '--eval_frequency', '5'] + (['--autograd_step'] if mode == 'autograd' else [])
        M1.run(make_parser().parse_args(args))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else 'gc'
if mode == 'nogc':
    gc.disable()
t0 = time.time()
for i in range(n):
    with tempfile.TemporaryDirectory() as d:
        body(pathlib.Path(d))
    if mode == 'cycles':                                      # dead handles that only the cyclic collector can reclaim
        from pvr_habitat_amd.models import PolicyNet
        for _ in range(3):
            m = PolicyNet((256,), 3, True, max_unroll=4, max_batch=2).to(device='cuda')
            m(dict(obs=torch.zeros(4, 2, 256), done=torch.zeros(4, 2, dtype=torch.bool)), m.initial_state(2))
            m._self = m
            del m
    print('iteration %d ok (%.1f s)' % (i, time.time() - t0), flush=True)
print('stress_bc1: %d iterations clean, mode %s, env %s' % (n, mode, {k: v for k, v in os.environ.items() if k.startswith(('PVR_', 'MALLOC'))}))
