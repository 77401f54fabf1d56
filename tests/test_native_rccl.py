"""A host written in C++ against include/pvr_policy.h drives the data-parallel hook of the BC path with RCCL itself: ncclAllReduce goes
into the library as a C function pointer through pvr_policy_set_data_parallel (tests/native/rccl_binding.cpp; BASELINE config 4, SURVEY 8e;
the reference's main_bc_finetune.py:167-208 is single-GPU).  The CPU test proves that the example compiles and links against the shipped
library and librccl; the GPU test runs it (one rank on a one-GPU box: the averaged gradient of a world of two is exactly half the local one)."""
import os
import shutil
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'native', 'rccl_binding.cpp')
OUT = os.path.join(ROOT, 'tests', 'native', 'build', 'rccl_binding')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    lib = os.path.join(ROOT, 'pvr_habitat_amd', 'lib')
    assert os.path.exists(os.path.join(lib, 'libpvr_hip.so')), 'build the library first (__graft_entry__.build())'
    cmd = [HIPCC, '-O2', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-result', '-I', os.path.join(ROOT, 'include'), SRC, '-L', lib, '-lpvr_hip', '-lrccl',
           '-Wl,-rpath,' + lib, '-o', OUT]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_rccl_binding_example_compiles_and_links():
    _build()
    assert os.path.exists(OUT)
    r = subprocess.run(['nm', '-D', '--undefined-only', OUT], capture_output=True, text=True)
    assert 'ncclAllReduce' in r.stdout and 'pvr_policy_set_data_parallel' in r.stdout and 'pvr_policy_backward' in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')
def test_rccl_binding_example_runs():
    if not os.path.exists(OUT):
        _build()
    ranks = '2' if torch.cuda.device_count() >= 2 else '1'
    r = subprocess.run([OUT], capture_output=True, text=True, timeout=600, env=dict(os.environ, RANKS=ranks))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert 'rccl_binding: ok' in r.stdout
