"""CPU tests of the host logic around the hot path: flags, batch assembly, row sharding (gloo, world size 2),
save_embedded_obs split/concat order with a stand-in embedder."""
import json
import os
import pickle
import subprocess
import sys
import time
import numpy as np
import pytest
import torch

from pvr_habitat_amd import utils_bc
from pvr_habitat_amd.arguments import make_parser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flag_names_and_defaults_match_reference():
    f = make_parser().parse_args([])
    ref = dict(max_frames=200000000, n_episodes_test=50, eval_frequency=200, to_env='HabitatImageNav-apartment_0', debug=False,
               disable_save=False, essential_save_only=False, save_path='bc', data_path='behavioral_cloning',
               embedding_name='resnet50', train_embedding=False, pretrained_embedding=True, batch_norm=False,
               env='HabitatImageNav-apartment_0', num_input_frames=1, xpid=None, run_id=1, seed=1, total_frames=50000000,
               batch_size=32, unroll_length=100, mp_start='spawn', disable_cuda=False, learning_rate=0.0001, alpha=0.99,
               momentum=0, epsilon=1e-5, max_grad_norm=40., n_trajectories=-1, source='png')      # src/arguments.py:3-68
    for k, v in ref.items():
        assert getattr(f, k) == v, k


def test_gather_unrolls_matches_reference_loop():
    rng = np.random.default_rng(0)
    obs, act = rng.standard_normal((50, 7)).astype(np.float32), rng.integers(0, 3, 50)
    starts = [3, 47, 20]
    o, a = utils_bc.gather_unrolls([obs, act], starts, 10, 50)
    ro = np.stack([obs[np.mod(np.arange(i, i + 10), 50)] for i in starts], axis=1)      # main_bc_2.py:194-201
    ra = np.stack([act[np.mod(np.arange(i, i + 10), 50)] for i in starts], axis=1)
    assert np.array_equal(o, ro) and np.array_equal(a, ra) and o.shape == (10, 3, 7)


def test_sampler_raises_like_reference_when_data_too_short():
    import random
    random.seed(1)
    with pytest.raises(ValueError):
        utils_bc.sample_with_minimum_distance(n=500, k=16, d=100)      # negative population


def test_embed_rows_order_and_shapes():
    from pvr_habitat_amd.save_embedded_obs import embed_rows
    obs = np.arange(5 * 4 * 4 * 6, dtype=np.uint8).reshape(5, 4, 4, 6)
    calls = []

    def fake(t):
        calls.append(tuple(t.shape))
        e = t.numpy().reshape(t.shape[0], -1)[:, :3].astype(np.float32)
        return e.squeeze()                                     # EmbeddingNet squeezes N=1

    out = embed_rows(fake, obs, 2, 2)
    assert out.shape == (5, 6) and calls == [(4, 4, 4, 3), (4, 4, 4, 3), (2, 4, 4, 3)]
    np.testing.assert_array_equal(out[:, :3], obs[..., :3].reshape(5, -1)[:, :3])
    np.testing.assert_array_equal(out[:, 3:], obs[..., 3:].reshape(5, -1)[:, :3])


def test_png_source_reader_rows_and_order(tmp_path):
    """PNG layout of the reference (save_opt_trajectories_png.py:44-58: <t>_<s>.png, <t>_goal.png, <t>.pickle, written with
    cv2.imwrite of RGB arrays and read back with cv2.imread): the threaded / prefetching reader returns the rows a serial reader
    would, in order - each observation = frame channels then goal channels, ragged trajectory lengths, stops at the first gap."""
    import pickle
    from PIL import Image
    from pvr_habitat_amd import save_embedded_obs as seo
    rng = np.random.RandomState(0)
    lens, frames, goals = [5, 1, 7], [], []
    for t, L in enumerate(lens):
        g = rng.randint(0, 256, (16, 16, 3), dtype=np.uint8)
        fr = rng.randint(0, 256, (L, 16, 16, 3), dtype=np.uint8)
        # cv2.imwrite(array) stores array[..., ::-1] as the file's RGB; cv2.imread returns the array again
        Image.fromarray(np.ascontiguousarray(g[..., ::-1])).save(tmp_path / ('%d_goal.png' % t))
        for s_ in range(L):
            Image.fromarray(np.ascontiguousarray(fr[s_][..., ::-1])).save(tmp_path / ('%d_%d.png' % (t, s_)))
        with open(tmp_path / ('%d.pickle' % t), 'wb') as f:
            pickle.dump(dict(action=np.arange(L), reward=np.ones(L), done=np.arange(L) == L - 1, true_state=np.zeros((L, 12))), f)
        frames.append(fr); goals.append(g)
    Image.fromarray(goals[0]).save(tmp_path / '4_goal.png')                      # trajectory 3 is missing: 4 must not be read
    for workers in (1, 4):
        d = seo.read_habitat_data_from_png(str(tmp_path), None, -1, decode_workers=workers)
        assert d['obs'].shape == (sum(lens), 16, 16, 6) and len(d['action']) == sum(lens) and len(d['png']) == sum(lens)
        row = 0
        for t, L in enumerate(lens):
            np.testing.assert_array_equal(d['obs'][row:row + L, ..., :3], frames[t])
            np.testing.assert_array_equal(d['obs'][row:row + L, ..., 3:], np.broadcast_to(goals[t], frames[t].shape))
            row += L
    d2 = seo.read_habitat_data_from_png(str(tmp_path), None, 2)
    assert d2['obs'].shape[0] == lens[0] + lens[1]


def test_shard_bounds_cover_rows_in_order():
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 3, 8):
            b = [utils_bc.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


_WORKER = r'''
import os, sys, pickle, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
import pvr_habitat_amd.save_embedded_obs as S
class Fake:                                    # stand-in embedder: deterministic function of the frame bytes
    out_size = 4
    def __init__(self, *a, **k): pass
    def state_dict(self): return {}
    def __call__(self, t):
        x = t.numpy().astype(np.float32).reshape(t.shape[0], -1)
        return np.stack([x.sum(1), x[:, 0], x[:, -1], x.mean(1)], 1).squeeze()
S.EmbeddingNet = Fake
flags = S.make_parser().parse_args(['--data_path', sys.argv[1], '--env', 'scene', '--embedding_name', 'fake', '--source', 'pickle', '--embed_batch', '6'])
S.run(flags)
dist.barrier()
'''


def test_save_embedded_obs_sharded_gloo_world2(tmp_path):
    """N>1 path: two gloo ranks embed contiguous row shards; rank 0 writes the same pickle a single process does."""
    rng = np.random.default_rng(1)
    trajs = [rng.integers(0, 255, (L, 8, 8, 6), dtype=np.uint8) for L in (5, 9, 3)]
    raw = dict(obs=trajs, action=[rng.integers(0, 3, len(t)) for t in trajs], reward=[np.zeros(len(t)) for t in trajs],
               done=[np.zeros(len(t), bool) for t in trajs], true_state=[np.zeros((len(t), 12)) for t in trajs])
    outs = []
    for world in (1, 2):
        d = tmp_path / ('w%d' % world)
        d.mkdir()
        pickle.dump(raw, open(d / 'scene.pickle', 'wb'))
        script = d / 'worker.py'
        script.write_text(_WORKER % dict(root=ROOT))
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29731 + world))
            procs.append(subprocess.Popen([sys.executable, str(script), str(d)], env=env))
        assert all(p.wait(timeout=120) == 0 for p in procs)
        outs.append(pickle.load(open(d / 'scene_fake.pickle', 'rb')))
    assert outs[0]['obs'].shape == (17, 8) and outs[0]['obs'].dtype == np.float32
    np.testing.assert_array_equal(outs[0]['obs'], outs[1]['obs'])
    np.testing.assert_array_equal(outs[0]['action'], outs[1]['action'])
    assert set(outs[0]) == {'obs', 'action', 'reward', 'done', 'true_state'}      # save_embedded_obs.py:165


_DP_WORKER = r'''
import ctypes as C, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from pvr_habitat_amd.models import make_allreduce_fn
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
r = dist.get_rank()
fn, errors = make_allreduce_fn(None, device='cpu')
g = (np.arange(10, dtype=np.float32) * (r + 1))               # rank 0: k, rank 1: 2k -> sum 3k
assert fn(g.ctypes.data, 10, None, None) == 0 and not errors
assert np.allclose(g, np.arange(10, dtype=np.float32) * 3), g
# a failing collective must not vanish inside the ctypes callback: non-zero status + the exception kept for the caller
bad, berr = make_allreduce_fn('not a process group', device='cpu')
assert bad(g.ctypes.data, 10, None, None) == 1 and len(berr) == 1
dist.barrier()
'''


def test_allreduce_callback_gloo_world2(tmp_path):
    """The one collective the library calls in the finetune configuration (pvr_allreduce_fn): in-place SUM over the
    ranks of a raw buffer, on gloo; errors come back as a status, not as a swallowed exception."""
    script = tmp_path / 'dp.py'
    script.write_text(_DP_WORKER % dict(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29741')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    assert all(p.wait(timeout=120) == 0 for p in procs)


_PNG_WORKER = r'''
import os, sys, pickle, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
os.environ['PVR_DIST_BACKEND'] = 'gloo'
import pvr_habitat_amd.save_embedded_obs as S
from test_glue_golden import OracleEmbeddingNet
S.EmbeddingNet = OracleEmbeddingNet
S.main(['--data_path', sys.argv[1], '--env', 'scene', '--embedding_name', 'resnet50', '--disable_pretrained_embedding',
        '--source', sys.argv[2], '--embed_batch', '4'])
'''


@pytest.mark.parametrize('source', ['png', 'pickle'])
def test_save_embedded_obs_cli_entry_shards_and_stitches(tmp_path, source):
    """`python -m pvr_habitat_amd.save_embedded_obs` under torch.distributed.run's environment (RANK / WORLD_SIZE / MASTER_*), gloo,
    world 2: main() creates the process group itself, the png source shards on trajectory boundaries, every rank writes
    <env>_<emb>.rank<r>.pickle, rank 0 stitches them in rank order and removes them - same file as the reference fixture."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import glue_inputs as GI
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'glue_save_obs.npz'))
    GI.write_scene(str(tmp_path))
    script = tmp_path / 'w.py'
    script.write_text(_PNG_WORKER % dict(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29761' if source == 'png' else '29763',
                   OMP_NUM_THREADS='4')
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path), source], env=env))
    assert all(p.wait(timeout=600) == 0 for p in procs)
    res = pickle.load(open(tmp_path / 'scene_resnet50.pickle', 'rb'))
    assert list(res.keys()) == list(g[source + '/keys'])
    assert res['obs'].shape == g[source + '/obs'].shape
    err = np.linalg.norm(res['obs'] - g[source + '/obs'], axis=1) / np.linalg.norm(g[source + '/obs'], axis=1)
    assert err.max() < 5e-5, err
    for k in ('action', 'reward', 'done', 'true_state'):
        np.testing.assert_array_equal(res[k], g['%s/%s' % (source, k)])
    if source == 'png':
        assert [os.path.relpath(q, str(tmp_path)) for q in res['png']] == list(g['png/png'])
    assert not [f for f in os.listdir(tmp_path) if '.rank' in f]


def test_native_file_reader(tmp_path):
    """pvr_file_sizes / pvr_read_files (host side of the per-file PNG source): bytes land at the prefix-sum offsets, whatever the
    thread count; a missing file is an error that names it."""
    from pvr_habitat_amd import png_gpu
    paths = []
    for i in range(70):
        p = tmp_path / ('f%d.bin' % i)
        p.write_bytes(bytes([i]) * (i * 13 % 97) + b'x')
        paths.append(str(p))
    for threads in (1, 3, 64):
        buf, off = png_gpu.read_files(paths, threads)
        assert off[0] == 0 and off[-1] == sum(i * 13 % 97 + 1 for i in range(70))
        for i in (0, 1, 33, 69):
            assert buf[off[i]:off[i + 1]].numpy().tobytes() == bytes([i]) * (i * 13 % 97) + b'x'
    with pytest.raises(Exception, match='missing.bin'):
        png_gpu.read_files(paths + [str(tmp_path / 'missing.bin')], 4)


def _scene(rng, lens, hw=8, c=6):
    trajs = [rng.integers(0, 255, (L, hw, hw, c), dtype=np.uint8) for L in lens]
    return dict(obs=trajs, action=[rng.integers(0, 3, len(t)) for t in trajs], reward=[np.zeros(len(t)) for t in trajs],
                done=[np.zeros(len(t), bool) for t in trajs], true_state=[np.zeros((len(t), 12)) for t in trajs])


@pytest.mark.parametrize('protocol', [3, 4, 5])
def test_scene_pickle_reads_row_ranges_without_holding_the_scene(tmp_path, protocol):
    """scene_pickle: the per-scene pickle of save_opt_trajectories.py:100-106 read as (index pass, row-range pass) - same rows as
    np.concatenate(data['obs'])[lo:hi] of the reference's loader (save_embedded_obs.py:29-47), with a peak of a few trajectories
    instead of the whole file (what lets each of 8 ranks read only its slice of a 1 M-frame scene)."""
    import tracemalloc
    from pvr_habitat_amd import scene_pickle as SP
    rng = np.random.default_rng(3)
    raw = _scene(rng, (5, 9, 3, 7))
    raw['obs'][2] = np.asfortranarray(raw['obs'][2])            # a non-contiguous block takes numpy's other pickle path
    p = str(tmp_path / 's.pickle')
    pickle.dump(raw, open(p, 'wb'), protocol=protocol)
    rows = np.concatenate(raw['obs'])
    lengths, frame_shape, small = SP.scene_index(p)
    assert lengths == [5, 9, 3, 7] and frame_shape == (8, 8, 6)
    for k in ('action', 'reward', 'done', 'true_state'):
        np.testing.assert_array_equal(np.concatenate(small[k]), np.concatenate(raw[k]))
    for lo, hi in ((0, 24), (3, 17), (14, 14), (20, 24), (5, 6), (0, 0)):
        got = []
        assert SP.scene_rows(p, lo, hi, lambda b: got.append(np.array(b))) == hi - lo
        np.testing.assert_array_equal(np.concatenate(got) if got else rows[:0], rows[lo:hi])
    big = _scene(rng, (150,) * 24, hw=64, c=6)                  # 24 x 3.7 MB of frames
    p = str(tmp_path / 'big.pickle')
    pickle.dump(big, open(p, 'wb'), protocol=protocol)
    del big
    tracemalloc.start()
    SP.scene_index(p)
    assert SP.scene_rows(p, 1000, 1400, lambda b: None) == 400
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < 0.25 * os.path.getsize(p), (peak, os.path.getsize(p))


def test_scene_pickle_protocol_2_reads_correctly_and_warns_about_its_memory_bound(tmp_path):
    """a protocol-2 scene (Python-2 era pickles): rows are still right, and the reader says that its one-trajectory memory bound does not
    hold there (the payload travels as a latin-1 str that the unpickler memoizes)."""
    from pvr_habitat_amd import scene_pickle as SP
    raw = _scene(np.random.default_rng(4), (4, 6, 2))
    p = str(tmp_path / 's2.pickle')
    pickle.dump(raw, open(p, 'wb'), protocol=2)
    with pytest.warns(RuntimeWarning, match='protocol-2'):
        lengths, frame_shape, _ = SP.scene_index(p)
    assert lengths == [4, 6, 2] and frame_shape == (8, 8, 6)
    got = []
    with pytest.warns(RuntimeWarning, match='protocol-2'):
        assert SP.scene_rows(p, 3, 11, lambda b: got.append(np.array(b))) == 8
    np.testing.assert_array_equal(np.concatenate(got), np.concatenate(raw['obs'])[3:11])


def test_uint8_stem_refuses_a_crop_window_outside_the_frame():
    """stem.hip's uint8-reading form takes the crop window's rows by a range-checked DMA: a window outside the frame would read zeros, not
    fail, so the geometry predicate (also behind the launcher's PVR_REQUIRE) must say no.  Host-side, no GPU needed."""
    import ctypes as C
    from pvr_habitat_amd import _lib
    ok = _lib.lib().pvr_debug_stem_u8_geometry_ok
    buf = C.create_string_buffer(64)
    base = (C.addressof(buf) + 15) & ~15                         # a 16-byte aligned address stands in for the frames pointer
    assert ok(C.c_void_p(base), 256, 256, 16, 16) == 1           # the bench geometry: centre crop of a 256 x 256 frame
    assert ok(C.c_void_p(base), 224, 224, 0, 0) == 1
    for h, w, top, left in ((256, 256, -16, 16), (256, 256, 16, -16), (256, 256, 48, 16), (256, 256, 16, 48), (200, 256, 0, 16), (256, 200, 16, 0)):
        assert ok(C.c_void_p(base), h, w, top, left) == 0, (h, w, top, left)


def test_direct_pickle_writer_equals_pickle_dump(tmp_path):
    """save_embedded_obs.DirectPickleWriter (single-rank runs): rows appended block by block between the two halves of a pre-computed
    pickle stream give the file pickle.dump would have written for the finished dict - byte for byte once the matrix is larger than the
    pickler's 64 KiB frame target (what every real run is), and an equal, writable array for tiny ones; a short run is refused."""
    from pvr_habitat_amd.save_embedded_obs import DirectPickleWriter
    rng = np.random.default_rng(0)
    keys = ('obs', 'action', 'reward', 'done', 'true_state')
    for n, w in ((5000, 64), (7, 3), (1, 1)):
        obs = rng.standard_normal((n, w)).astype(np.float32)
        small = dict(action=rng.integers(0, 3, n), reward=np.zeros(n), done=np.zeros(n, bool), true_state=np.zeros((n, 12), np.float32))
        p = str(tmp_path / ('o_%d.pickle' % n))
        wr = DirectPickleWriter(p, n, w, small, keys)
        assert not os.path.exists(p)                                  # nothing under the final name until finish()
        for a in range(0, n, 999):
            wr.append(obs[a:a + 999])
        wr.finish()
        with open(p, 'rb') as f:
            got = pickle.load(f)
        assert list(got.keys()) == list(keys) and got['obs'].dtype == np.float32 and got['obs'].flags.writeable
        np.testing.assert_array_equal(got['obs'], obs)
        for k in keys[1:]:
            np.testing.assert_array_equal(got[k], small[k])
        if obs.nbytes > (64 << 10):
            assert open(p, 'rb').read() == pickle.dumps(dict(obs=obs, **small), protocol=pickle.HIGHEST_PROTOCOL)
    wr = DirectPickleWriter(str(tmp_path / 'short.pickle'), 10, 4, dict(action=np.zeros(10), reward=np.zeros(10), done=np.zeros(10, bool), true_state=np.zeros((10, 12))), keys)
    wr.append(np.zeros((6, 4), np.float32))
    with pytest.raises(AssertionError, match='embedded 6 rows of 10'):
        wr.finish()
    assert not os.path.exists(str(tmp_path / 'short.pickle'))


def test_stitch_shards_streams_rows_from_disk(tmp_path):
    """Rank 0's stitch (save_embedded_obs.stitch_shards): shard row files -> the reference's single pickle through a file-backed
    memmap; peak Python-side memory stays far below the size of the embedding matrix (round 2 concatenated every shard in RAM)."""
    import tracemalloc
    from pvr_habitat_amd import save_embedded_obs as S
    save = str(tmp_path / 'scene_fake.pickle')
    rng = np.random.default_rng(0)
    parts = [rng.standard_normal((n, 2048)).astype(np.float32) for n in (3000, 0, 2500, 1700)]      # 59 MB in all; one empty rank
    for r, rows in enumerate(parts):
        w = S.ShardWriter(save, r)
        for a in range(0, len(rows), 700):
            w.append(rows[a:a + 700])
        n = len(rows)
        w.finish(dict(action=np.full(n, r), reward=np.zeros(n), done=np.zeros(n, bool), true_state=np.zeros((n, 12))), dict(rank=r, world=4))
    want = np.concatenate(parts)
    del parts
    tracemalloc.start()
    S.stitch_shards(save, 4, ('obs', 'action', 'reward', 'done', 'true_state'), copy_bytes=4 << 20)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < 0.25 * want.nbytes, (peak, want.nbytes)
    out = pickle.load(open(save, 'rb'))
    assert type(out['obs']) is np.ndarray and out['obs'].dtype == np.float32
    np.testing.assert_array_equal(out['obs'], want)
    np.testing.assert_array_equal(out['action'], np.concatenate([np.full(n, r) for r, n in enumerate((3000, 0, 2500, 1700))]))
    assert sorted(os.listdir(tmp_path)) == ['scene_fake.pickle']


_RESUME_WORKER = r'''
import os, sys, pickle, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank = int(os.environ['RANK'])
dist.init_process_group('gloo', rank=rank, world_size=int(os.environ['WORLD_SIZE']))
import pvr_habitat_amd.save_embedded_obs as S
calls = os.path.join(sys.argv[1], 'calls.rank%%d' %% rank)
class Fake:                                    # stand-in embedder: deterministic function of the frame bytes; counts its calls on disk
    out_size = 4
    def __init__(self, *a, **k): pass
    def state_dict(self): return {}
    def __call__(self, t):
        n = int(open(calls).read()) if os.path.isfile(calls) else 0
        open(calls, 'w').write(str(n + 1))
        if rank == 1 and os.environ.get('DIE_AT') and n + 1 >= int(os.environ['DIE_AT']):
            os._exit(17)                       # the rank dies in the middle of its shard
        x = t.numpy().astype(np.float32).reshape(t.shape[0], -1)
        return np.stack([x.sum(1), x[:, 0], x[:, -1], x.mean(1)], 1).squeeze()
S.EmbeddingNet = Fake
flags = S.make_parser().parse_args(['--data_path', sys.argv[1], '--env', 'scene', '--embedding_name', 'fake', '--source', 'pickle',
                                    '--embed_batch', '4', '--embed_block', '6'])
S.run(flags)
dist.barrier()
'''


def test_save_embedded_obs_resumes_after_a_rank_died(tmp_path):
    """SURVEY section 5 / save_embedded_obs.py:97-101: rank 1 is killed in the middle of its shard (rank 0 is then stopped by the
    launcher, as torch.distributed.run does).  The relaunch finds rank 0's finished shard and does NOT embed it again, rank 1
    starts its shard over, and the stitched file equals a single-process run."""
    rng = np.random.default_rng(5)
    raw = _scene(rng, (11, 14, 9, 12))
    pickle.dump(raw, open(tmp_path / 'scene.pickle', 'wb'))
    script = tmp_path / 'worker.py'
    script.write_text(_RESUME_WORKER % dict(root=ROOT))

    def launch(port, **extra):
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **extra)
            procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env))
        return procs
    p0, p1 = launch(29771, DIE_AT='3')
    assert p1.wait(timeout=120) == 17
    deadline = time.time() + 60
    while not os.path.isfile(tmp_path / 'scene_fake.rank0.pickle') and time.time() < deadline:
        time.sleep(0.1)                                         # rank 0 finishes its shard, then waits at the barrier for a dead peer
    p0.kill()
    p0.wait()
    assert os.path.isfile(tmp_path / 'scene_fake.rank0.pickle') and not os.path.isfile(tmp_path / 'scene_fake.rank1.pickle')
    assert not os.path.isfile(tmp_path / 'scene_fake.pickle')
    calls0 = int(open(tmp_path / 'calls.rank0').read())
    assert all(p.wait(timeout=120) == 0 for p in launch(29773))
    assert int(open(tmp_path / 'calls.rank0').read()) == calls0          # rank 0 skipped its finished shard
    out = pickle.load(open(tmp_path / 'scene_fake.pickle', 'rb'))
    x = np.concatenate(raw['obs']).astype(np.float32)
    want = np.concatenate([np.stack([f.reshape(len(f), -1).sum(1), f.reshape(len(f), -1)[:, 0], f.reshape(len(f), -1)[:, -1],
                                     f.reshape(len(f), -1).mean(1)], 1) for f in (x[..., :3], x[..., 3:])], axis=1)
    np.testing.assert_array_equal(out['obs'], want)
    np.testing.assert_array_equal(out['action'], np.concatenate(raw['action']))
    assert not [f for f in os.listdir(tmp_path) if '.rank' in f and not f.startswith('calls')]


class _FakeNet(object):
    """stand-in embedder for the host-logic tests: a deterministic function of the frame bytes"""
    out_size = 4

    def __init__(self, *a, **k):
        pass

    def state_dict(self):
        return {}

    def __call__(self, t):
        x = t.numpy().astype(np.float32).reshape(t.shape[0], -1)
        return np.stack([x.sum(1), x[:, 0], x[:, -1], x.mean(1)], 1).squeeze()


def test_pickle_source_ignores_n_trajectories_like_the_reference(tmp_path, monkeypatch):
    """save_embedded_obs.py:142-145 calls read_habitat_data_from_pickle WITHOUT the trajectory count: the pickle source always embeds
    the whole scene, whatever --n_trajectories says (the build used to pass the flag on)."""
    import pvr_habitat_amd.save_embedded_obs as S
    monkeypatch.setattr(S, 'EmbeddingNet', _FakeNet)
    raw = _scene(np.random.default_rng(2), (4, 6, 5))
    pickle.dump(raw, open(tmp_path / 'scene.pickle', 'wb'))
    S.run(S.make_parser().parse_args(['--data_path', str(tmp_path), '--env', 'scene', '--embedding_name', 'fake', '--source', 'pickle',
                                      '--n_trajectories', '1', '--embed_batch', '4', '--embed_block', '3']))
    out = pickle.load(open(tmp_path / 'scene_fake.pickle', 'rb'))
    assert out['obs'].shape == (15, 8) and len(out['action']) == 15
    np.testing.assert_array_equal(out['action'], np.concatenate(raw['action']))
    assert sorted(f for f in os.listdir(tmp_path) if 'scene' in f) == ['scene.pickle', 'scene_fake.pickle']        # shard files are gone


def test_png_source_fails_loudly_on_an_undecodable_frame(tmp_path, monkeypatch):
    """Deliberate difference: where the reference's `except: break` (save_embedded_obs.py:69-79) silently ends a
    trajectory at a frame that does not decode - leaving obs shorter than action / reward / done, which main_bc_2.py:118 then rejects
    with 'data length does not match' - this build stops at the frame and names the file."""
    from PIL import Image
    import pvr_habitat_amd.save_embedded_obs as S
    monkeypatch.setattr(S, 'EmbeddingNet', _FakeNet)
    d = tmp_path / 'scene'
    d.mkdir()
    rng = np.random.default_rng(4)
    for t in range(2):
        pickle.dump(dict(action=np.zeros(3, int), reward=np.zeros(3), done=np.zeros(3, bool), true_state=np.zeros((3, 12))), open(d / ('%d.pickle' % t), 'wb'))
        Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)).save(d / ('%d_goal.png' % t))
        for s_ in range(3):
            Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)).save(d / ('%d_%d.png' % (t, s_)))
    (d / '1_1.png').write_bytes(b'\x89PNG\r\n\x1a\n' + b'garbage' * 10)
    with pytest.raises(Exception, match='1_1.png'):
        S.run(S.make_parser().parse_args(['--data_path', str(tmp_path), '--env', 'scene', '--embedding_name', 'fake', '--source', 'png', '--embed_batch', '4']))
    assert not os.path.isfile(tmp_path / 'scene_fake.pickle')


def test_block_ring_reuses_blocks_and_hands_over_a_ragged_last_block():
    """save_embedded_obs.BlockRing (round 4): rows deposited by a producer thread arrive at the consumer block by block, in order, the
    two blocks are reused, puts larger than a block are split, the last block is ragged, and a producer failure reaches the consumer."""
    import threading
    from pvr_habitat_amd.save_embedded_obs import BlockRing
    rng = np.random.default_rng(3)
    rows = rng.integers(0, 256, (53, 2, 3, 6), dtype=np.uint8)
    ring = BlockRing(8, (2, 3, 6), count=2, pinned=False)
    cuts = [0, 5, 6, 19, 20, 41, 53]                              # trajectory-sized puts, one of them (13 rows) larger than a block

    def producer():
        for a, b in zip(cuts[:-1], cuts[1:]):
            ring.put(rows[a:b])
        ring.close()
    th = threading.Thread(target=producer)
    th.start()
    got, used = [], []
    while True:
        g = ring.get()
        if g is None:
            break
        idx, blk = g
        used.append(idx)
        got.append(blk.numpy().copy())
        ring.release(idx)
    th.join()
    assert [len(g) for g in got] == [8] * 6 + [5] and set(used) == {0, 1}
    assert np.array_equal(np.concatenate(got), rows)
    ring = BlockRing(4, (1,), count=2, pinned=False)
    ring.put(np.zeros((2, 1), np.uint8))
    ring.close(ValueError('scene is corrupt'))
    with pytest.raises(ValueError, match='corrupt'):
        ring.get()
    # a consumer that fails must not leave the producer blocked in put() (both blocks full, none released): abort() wakes it with an error
    ring = BlockRing(2, (1,), count=2, pinned=False)
    failed = []

    def stuck_producer():
        try:
            ring.put(np.zeros((6, 1), np.uint8))                  # fills both blocks, then waits for a free one
        except RuntimeError as e:
            failed.append(str(e))
    th = threading.Thread(target=stuck_producer)
    th.start()
    assert ring.get() is not None                                 # the consumer takes one block and dies without releasing it
    ring.abort()
    th.join(timeout=20)
    assert not th.is_alive() and failed and 'consumer stopped' in failed[0]


def test_native_stage_copy_gathers_rows_and_channel_planes():
    """pvr_stage_copy (host threads of libpvr_hip.so): plain rows and the 3-byte-run gather of a channel plane of (N,H,W,3F) frames, every
    thread count, ragged sizes; invalid geometry is refused with a message."""
    from pvr_habitat_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (37, 5, 7, 3), dtype=np.uint8)
    for th in (1, 3, 16):
        b = np.zeros_like(a)
        _lib.check(L.pvr_stage_copy(b.ctypes.data, a.ctypes.data, 37, 105, 105, 105, 105, th))
        assert np.array_equal(a, b)
    for F in (2, 3):
        s6 = rng.integers(0, 256, (29, 4, 9, 3 * F), dtype=np.uint8)
        for f in range(F):
            v, o = s6[..., 3 * f:3 * f + 3], np.zeros((29, 4, 9, 3), np.uint8)
            _lib.check(L.pvr_stage_copy(o.ctypes.data, v.ctypes.data, 29, 108, s6.strides[0], 3, 3 * F, 4))
            assert np.array_equal(o, v), (F, f)
    big = rng.integers(0, 256, (3, 1 << 20), dtype=np.uint8)      # contiguous rows are split into 1 MiB pieces
    o = np.zeros_like(big)
    _lib.check(L.pvr_stage_copy(o.ctypes.data, big.ctypes.data, 3, 1 << 20, 1 << 20, 1 << 20, 1 << 20, 8))
    assert np.array_equal(o, big)
    assert L.pvr_stage_copy(o.ctypes.data, big.ctypes.data, 3, 100, 100, 7, 7, 2) != 0 and 'pvr_stage_copy' in _lib.last_error()


def test_weights_fingerprint_covers_every_tensor():
    """the shard-resume cover's `weights` field: a checkpoint that differs only in its LAST layers (a fine-tuned layer4, another compression
    head) must not be mistaken for the one a half-finished shard was made with"""
    from pvr_habitat_amd.save_embedded_obs import _weights_fingerprint

    class _M:
        def __init__(self, sd): self.sd = sd
        def state_dict(self): return self.sd
    g = torch.Generator().manual_seed(0)
    sd = {'layer%02d.%d.weight' % (i // 4, i % 4): torch.randn(64, 300, generator=g) for i in range(90)}
    sd['bn.num_batches_tracked'] = torch.tensor(7)
    base = _weights_fingerprint(_M(sd))
    assert base == _weights_fingerprint(_M({k: v.clone() for k, v in sd.items()}))
    lk = sorted(k for k in sd if k.startswith('layer'))
    for key, idx in ((lk[-1], -1), (lk[70], 5000), (lk[0], 0)):
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2[key].view(-1)[idx] += 1e-3
        assert _weights_fingerprint(_M(sd2)) != base, key
    sd3 = dict(sd); sd3['bn.num_batches_tracked'] = torch.tensor(8)
    assert _weights_fingerprint(_M(sd3)) != base


def test_weights_fingerprint_sees_the_members_of_an_uber_model():
    """UberModel keeps its member encoders in a plain list (reference src/embeddings.py:44-57): their weights are not in state_dict(), so a
    fingerprint of state_dict() alone is a constant for every `*_uber_*` PVR.  Members are walked explicitly: two uber models
    whose members differ in one weight - also under the 5-crop wrapper and inside an EmbeddingNet-like holder - get different fingerprints."""
    import torch.nn as nn
    from pvr_habitat_amd.embeddings import UberModel, FiveCrop
    from pvr_habitat_amd.save_embedded_obs import _weights_fingerprint

    class _Member(nn.Module):
        def __init__(self, seed):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.w = nn.Parameter(torch.randn(8, 8, generator=g), requires_grad=False)
            self.register_buffer('running_mean', torch.zeros(8))
            self.out_size = 8

    class _Net(nn.Module):                      # EmbeddingNet's shape: the (wrapped) model is a registered sub-module named `embedding`
        def __init__(self, emb):
            super().__init__()
            self.embedding = emb

    def uber(seeds, bump=None, crops=1):
        ms = [_Member(s) for s in seeds]
        if bump is not None:
            ms[bump].w.data.view(-1)[3] += 1e-3
        u = UberModel(ms)
        return _Net(FiveCrop(u) if crops == 5 else u)

    for crops in (1, 5):
        base = _weights_fingerprint(uber((1, 2, 3), crops=crops))
        assert base is not None and base == _weights_fingerprint(uber((1, 2, 3), crops=crops))
        assert len(uber((1, 2, 3), crops=crops).state_dict()) == 0                      # the reference's blind spot, kept
        for i in range(3):
            assert _weights_fingerprint(uber((1, 2, 3), bump=i, crops=crops)) != base, (crops, i)
        assert _weights_fingerprint(uber((1, 3, 2), crops=crops)) != base                # member order matters (column order of the concat)
    # numpy values are hashed by content, not by name only
    class _NP:
        def __init__(self, a): self.a = a
        def state_dict(self): return {'w': self.a}
    assert _weights_fingerprint(_NP(np.arange(6.0))) != _weights_fingerprint(_NP(np.arange(6.0) + 1))


_GUARD_WORKER = r'''
import json, os, sys, time, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank = int(os.environ['RANK'])
dist.init_process_group('gloo', rank=rank, world_size=2)
import bench
line = {'metric': 'headline', 'value': 1.0}
guard = bench.LegGuard(dist, rank, lambda: line)
guard.enter('host_leg', 600.0)
try:
    if rank == 1:
        raise MemoryError('pin_memory failed on this rank only')
    dist.barrier()                      # rank 0: blocked in the leg's collective - rank 1 never arrives
    print('rank 0 passed the barrier?!', flush=True)
except Exception as e:
    guard.failed('host_leg', e)
'''


def test_bench_leg_guard_ends_a_one_sided_failure_promptly(tmp_path):
    """bench.py, N > 1: a rank that raises inside a host-fed leg used to leave the other ranks blocked in the leg's collectives
    until the launcher's 3000 s time-out - and no JSON line.  With LegGuard the failing rank posts its error to the rendezvous store, rank 0's
    watcher prints the line so far with the error in it and every rank exits non-zero within seconds."""
    script = tmp_path / 'guard_worker.py'
    script.write_text(_GUARD_WORKER % dict(root=ROOT))
    procs = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29747')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [3, 3], [p.returncode for p in procs]
    assert time.time() - t0 < 90
    lines = [l for l in outs[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1].splitlines() if l.startswith('{')]
    d = json.loads(lines[0])
    assert d['value'] == 1.0 and 'rank 1, leg host_leg: MemoryError' in d['aborted']
