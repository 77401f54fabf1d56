"""bench.py prints ONE JSON line with the driver's contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` object; the N > 1 path (one process
per GPU, barrier + MAX over ranks) is exercised with two ranks sharing the one test GPU (PVR_BENCH_ONE_GPU=1, gloo)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ['--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-bc', '--no-vit', '--no-pcie', '--png-traj', '3']


def _check(line, n):
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    assert d['metric'].startswith('frames/sec embedded (ResNet50') and d['unit'] == 'frames/s' and d['n_gpus'] == n and d['steps'] == 4
    # the headline is the product's default storage type, the one inside the north-star 1e-3 bound (round 5); bf16 runs beside it at the same length
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f16'
    assert 'workload' in d['config'] and 'model' not in d['config'] and d['config']['global_batch'] == 256 * n
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['peak'] == 2500.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['frac'] > 0.05
    assert d['value'] > 8000 * n                                       # north-star floor: 8 k frames/s per GPU
    assert 'frame_pool' in d['config'] and 'traffic_source' in r
    assert d['bf16']['steps'] == d['steps'] and d['bf16']['value'] > 8000 * n and d['bf16']['dtype'] == 'bf16'
    # parity of BOTH timed storage types is part of the line: rel-L2, max-norm and the element-wise relative error distribution
    assert 0 < d['parity_rel_l2'] < 1e-3 and 0 < d['bf16']['parity_rel_l2'] < 1e-2
    for dt, bound in (('f16', 1e-3), ('bf16', 1e-2)):
        pe = d['parity'][dt]
        assert 0 < pe['rel_l2'] < bound and 0 < pe['max_norm'] < bound
        ew = pe['elementwise']
        assert 0 < ew['p50'] <= ew['p99'] <= ew['max'] and ew['n_elements'] > 1000 and ew['p50'] < bound
    if n == 1:
        assert d['png_source']['files'] == 750 and d['png_source']['value'] > 1000      # the PNG tree through the GPU decoder
    # BASELINE configs[4] at every N (round 6): 5-crop uber_345, `value` = the compliant f16 plan (HBM-resident) with the bf16 throughput plan beside it,
    # both also streamed to the host
    u = d['uber5crop']
    assert 'error' not in u, u
    assert u['dtype'] == 'f16' and u['floats_per_frame'] == 31310 and u['value'] > 800 * n and 0.03 < u['frac_of_mfma_peak'] < 1
    assert u['bf16_throughput_plan']['dtype'] == 'bf16' and u['bf16_throughput_plan']['value'] > u['value']
    assert 300 < u['streamed']['value'] <= u['value'] * 1.05 and u['streamed']['d2h_GBps'] > 0
    pr = d['per_rank_ms_per_step']                                     # a straggler shows as max >> min; value uses the slowest rank
    assert 0 < pr['min'] <= pr['max'] and abs(pr['max'] - d['ms_per_step']) < 1e-6
    if n > 1:
        assert u['n_gpus'] == n and u['frames'] % n == 0 and 0 < u['per_rank_s']['streamed']['min'] <= u['per_rank_s']['streamed']['max']
    return d


def test_bench_line_single_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + FAST, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    _check(lines[0], 1)


def test_bench_line_two_ranks_on_one_gpu():
    """N = 2 (two ranks sharing the one test GPU): the headline, the data-parallel finetune leg AND - round 5 - the host-fed legs run by all ranks
    concurrently: the PCIe-inclusive stream of every rank's own pool and ONE scene embedded by save_embedded_obs.run in per-rank shards
    that rank 0 stitches - and, round 6, configs[4] (5-crop uber, resident + streamed) and configs[2] (ViT-B/16) with per-rank min / max times.
    These are the keys the multi-GPU driver line carries beside the HBM-resident `value`."""
    env = dict(os.environ, PVR_BENCH_ONE_GPU='1')
    flags = [f for f in FAST if f not in ('--no-pcie', '--no-vit')] + ['--e2e-samples', '3000', '--pool', '1024']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + flags
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                             # rank 0 only
    d = _check(lines[0], 2)
    dp = d['bc_finetune_dp']                                           # BASELINE config 4 leg: data-parallel finetune over the two ranks
    assert 'error' not in dp, dp
    assert dp['n_gpus'] == 2 and dp['value'] > 0 and dp['allreduce_ms'] > 0
    assert dp['allreduce_bytes'] == 4 * 18148868                       # every trainable fp32 parameter of PolicyNetWithConv((64,64,6), 4 actions, BN): SURVEY 8e's 72.6 MB
    pc, e2e = d['pcie_inclusive'], d['save_embedded_obs_e2e']
    assert 'error' not in pc and pc['n_gpus'] == 2 and pc['frames'] == 2 * 2 * 1024 and pc['pageable_source']['value'] > 2000 and pc['pinned_source']['value'] > 2000
    assert 'error' not in e2e and e2e['n_gpus'] == 2 and e2e['samples'] == 3000 and e2e['value'] > 1000 and len(e2e['runs_frames_per_s']) == 3
    # round 6: configs[2] (ViT-B/16) runs on every rank at once too: aggregate + the slowest / fastest rank
    v = d['vit']
    assert len(v) == 1 and v[0]['n_gpus'] == 2 and v[0]['value'] > 2000 and 0 < v[0]['per_rank_ms_per_step']['min'] <= v[0]['per_rank_ms_per_step']['max']
    assert 'aborted' not in d


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` as a plain command (how a driver without a launcher would type it): the parent starts
    torch.distributed.run as a child process, relays rank 0's single line and the exit code, and never initialises a GPU itself."""
    env = dict(os.environ, PVR_BENCH_ONE_GPU='1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--no-dp'] + FAST, capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    _check(lines[0], 2)
