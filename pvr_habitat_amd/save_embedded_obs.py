"""Offline embedding of saved trajectories: same `run(flags)` contract, flags and output files as reference
behavioral_cloning/save_embedded_obs.py:96-172, with the frames pushed through the HIP encoder in large
batches and, under torch.distributed, sharded across the GPUs of one node with NO collective on the data
path: each rank embeds a contiguous row range (pickle source) or a contiguous range of trajectories (png source: the goal frame
is per trajectory) and writes its own shard file `<env>_<emb>.rank<r>.pickle`; after a barrier rank 0 stitches the shard files
in rank order into the reference's output file and removes them.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m pvr_habitat_amd.save_embedded_obs ...

  pickle source: <data_path>/<env>.pickle  {obs:[(L,H,W,3n) u8], action, reward, done, true_state}  (:29-47)
  png source:    <data_path>/<env>/<t>_<s>.png, <t>_goal.png, <t>.pickle                             (:50-93)
  output:        <data_path>/<env>_<embedding_name>.pickle {obs f32 (N, n*O), action, reward, done, true_state}
                 <data_path>/<embedding_name>[_<run_id>].tar  {'embedding_model_state_dict': ...}      (:126-131)
"""
import os
import pickle
import random

import numpy as np
import torch

from .arguments import make_parser
from .embeddings import EmbeddingNet, stream_embed
from .utils_bc import shard_bounds
from .dist_utils import init_distributed, finalize_distributed


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def read_habitat_data_from_pickle(data_path, n_trajectories=-1):
    print('loading %s ...' % data_path)
    with open(data_path + '.pickle', 'rb') as f:
        data = pickle.load(f)
    if n_trajectories == -1:
        n_trajectories = len(data['reward'])
    for k in ('obs', 'action', 'reward', 'done', 'true_state'):
        data[k] = np.concatenate(data[k][:n_trajectories])
    n_samples = len(data['reward'])
    print('  ', '%d trajectories for a total of %d samples' % (n_trajectories, n_samples))
    print('  ', 'avg. return is', data['reward'].sum() / n_trajectories)
    return data


from .png_decode import imread as _imread, decode_parallel


def embed_rows(embed_fn, obs, n_frames, batch):
    """save_embedded_obs.py:151-157: (N,H,W,3n) -> per batch: split frames, stack on the batch axis (all first
    frames, then all second frames, ...), embed, re-split, concat on the feature axis -> (N, n*O)."""
    out = []
    for i in range(0, obs.shape[0], batch):
        o = obs[i:i + batch]
        m = o.shape[0]
        o = np.concatenate(np.split(o, n_frames, axis=3), axis=0)
        e = embed_fn(torch.from_numpy(np.ascontiguousarray(o)))
        e = np.asarray(e).reshape(n_frames * m, -1)            # undo the N=1 squeeze of EmbeddingNet.forward
        out.append(np.concatenate(np.split(e, n_frames, axis=0), axis=-1))
    return np.concatenate(out) if out else np.zeros((0, 0), np.float32)


def _load_png_trajectories(data_path, t0, t1, workers, gpu=False, listing=None, device=None):
    """Decode trajectories t0 .. t1-1 (stopping at the first missing one): a list of (goal, meta dict, frames (L,H,W,3) uint8 or None,
    file names).  All frames of the group are decoded in ONE call: on the GPU (gpu=True: png_gpu.decode_files, the host only reads
    file bytes and the frames never leave HBM; goal and frames are uint8 CUDA tensors then), or by the worker processes
    (png_decode.decode_parallel: PNG decoding holds the GIL, threads do not scale it; a group gives every worker a task)."""
    group, all_names, goal_names = [], [], []
    # `listing` (the directory's names, read once): the reference probes up to 500 file names per trajectory with the file system
    exists = (lambda name: name in listing) if listing is not None else (lambda name: os.path.isfile(os.path.join(data_path, name)))
    for t in range(t0, t1):
        meta_path = os.path.join(data_path, '%d.pickle' % t)
        goal_path = os.path.join(data_path, '%d_goal.png' % t)
        if gpu:
            goal = goal_path if (exists('%d.pickle' % t) and exists('%d_goal.png' % t)) else None
        else:
            goal = _imread(goal_path) if exists('%d.pickle' % t) else None
        if goal is None:
            break
        with open(meta_path, 'rb') as f:
            tmp = pickle.load(f)
        names = []
        for s in range(500):                                   # max steps per trajectory (habitat_config/nav_task.yaml:4)
            if not exists('%d_%d.png' % (t, s)):
                break
            names.append(os.path.join(data_path, '%d_%d.png' % (t, s)))
        group.append([goal, tmp, None, names])
        all_names += names
        goal_names.append(goal_path)
    if gpu and group:
        from . import png_gpu
        # this runs in the read-ahead thread: the current device is per THREAD (a rank's GPU is not device 0), and the decode stays off
        # the encoder's stream
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            with torch.cuda.stream(_png_stream()):
                dec = png_gpu.decode_files(goal_names + all_names, threads=max(1, min(16, workers)))      # returns after its stream has drained
        for i, g in enumerate(group):
            g[0] = dec[i]
        group[0].append(dec)                                   # (the whole group in one tensor: goals first, then every frame in order)
        frames, lo = dec[len(group):], 0
        for g in group:
            if g[3]:
                g[2] = frames[lo:lo + len(g[3])]
                lo += len(g[3])
    elif all_names:
        frames, lo = decode_parallel(all_names, workers), 0
        for g in group:
            if g[3]:
                g[2] = frames[lo:lo + len(g[3])]
                lo += len(g[3])
    return group, len(group) < t1 - t0                          # (trajectories, "the scene ends inside this group")


_PNG_STREAM = {}


def _png_stream():
    dev = torch.cuda.current_device()
    if dev not in _PNG_STREAM:
        _PNG_STREAM[dev] = torch.cuda.Stream(device=dev)
    return _PNG_STREAM[dev]


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(a)


def count_png_trajectories(data_path, n_trajectories=-1):
    """trajectories the reference's reader would visit: 0, 1, ... up to the first missing <t>.pickle / <t>_goal.png (:57-68)"""
    t = 0
    while (n_trajectories < 0 or t < n_trajectories) and os.path.isfile(os.path.join(data_path, '%d.pickle' % t)) \
            and os.path.isfile(os.path.join(data_path, '%d_goal.png' % t)):
        t += 1
    return t


def read_habitat_data_from_png(data_path, model=None, n_trajectories=-1, batch=256, decode_workers=None, t_range=None, gpu_decode=None,
                               row_sink=None):
    """PNG layout of save_opt_trajectories_png.py:44-58.  The reference decodes and embeds one frame per forward (:69-77);
    here the frames of four trajectories are decoded in one call and embedded together (same rows, same order), and the next group
    is decoded while this one is on the encoder (keeping the GPU fed from the PNG source).  With an encoder on a GPU
    the files are decoded ON the GPU (csrc/png_decode.hip; PVR_PNG_GPU=0 or gpu_decode=False selects the host decoders);
    decode_workers: host decoder processes (default min(32, cores); <= 1 decodes in this process) / file-reader threads.
    row_sink: called with every trajectory's finished (L, 2*O) rows in order instead of keeping them (run(): the rank's shard file),
    so the host never holds more than the group in flight; data['obs'] is then empty."""
    from concurrent.futures import ThreadPoolExecutor
    print('loading %s ...' % data_path)
    data = dict(obs=[], action=[], reward=[], done=[], true_state=[], png=[])
    if n_trajectories == -1:
        n_trajectories = 100000
    workers = decode_workers if decode_workers is not None else min(32, os.cpu_count() or 1)
    if gpu_decode is None:
        gpu_decode = model is not None and torch.cuda.is_available() and os.environ.get('PVR_PNG_GPU', '1') != '0'
    t_lo, t_hi = t_range if t_range is not None else (0, n_trajectories)     # a rank's shard: trajectories [t_lo, t_hi)

    def join(e, g):                                             # rows = [frame | goal of its trajectory] (:73-77)
        rows = np.empty((len(e),) + e.shape[1:-1] + (e.shape[-1] + g.shape[-1],), e.dtype)
        rows[..., :e.shape[-1]] = e
        rows[..., e.shape[-1]:] = g
        return rows

    def keep(eg):
        if row_sink is not None and model is not None:
            row_sink(join(*eg))
        else:
            data['obs'].append(eg)
    G = 16 if gpu_decode else 4                                 # trajectories decoded per call (<= 8000 / 2000 frames of 12 KB in flight)
    t = t_lo
    listing = frozenset(os.listdir(data_path)) if os.path.isdir(data_path) else frozenset()
    device = torch.cuda.current_device() if gpu_decode else None
    with ThreadPoolExecutor(max_workers=1) as ahead:
        nxt = ahead.submit(_load_png_trajectories, data_path, t_lo, min(t_lo + G, t_hi), workers, gpu_decode, listing, device) if t_hi > t_lo else None
        g0 = t_lo
        while nxt is not None:
            group, ended = nxt.result()
            g0 += G
            nxt = ahead.submit(_load_png_trajectories, data_path, g0, min(g0 + G, t_hi), workers, gpu_decode, listing, device) if (not ended and g0 < t_hi) else None
            emb = None
            if group and len(group[0]) == 5 and isinstance(model, EmbeddingNet):
                # frames decoded on the GPU: the whole group goes through the encoder in one pipelined pass (two lanes, D2H overlapped)
                dec = group[0].pop()
                dec.record_stream(torch.cuda.current_stream())      # decoded on the side stream: tell the allocator who reads it
                emb, lo = stream_embed(model, dec, batch), len(group)
            for gi, (goal, tmp, frames, names) in enumerate(g_[:4] for g_ in group):
                t += 1
                for k in data.keys():
                    if k in tmp:
                        data[k].append(tmp[k])
                if frames is None:
                    continue
                if emb is not None:
                    keep((emb[lo:lo + len(names)], emb[gi]))                    # (frame rows, goal row): joined once, at the end
                    lo += len(names)
                elif model is not None:
                    if torch.is_tensor(frames):                 # decoded on the side stream: tell the allocator who reads them
                        frames.record_stream(torch.cuda.current_stream()); goal.record_stream(torch.cuda.current_stream())
                    g = np.asarray(model(_t(goal)[None, :])).reshape(-1,)
                    e = np.concatenate([np.asarray(model(_t(frames[i:i + batch]))).reshape(min(batch, len(frames) - i), -1)
                                        for i in range(0, len(frames), batch)])
                    keep((e, g))
                else:
                    keep((frames, goal))
                data['png'] += names
    n_trajectories = t - t_lo
    if data['obs']:                                             # rows = [frame | goal of its trajectory] (:73-77), written once
        e0, g0_ = data['obs'][0]
        obs = np.empty((sum(len(e) for e, _ in data['obs']),) + e0.shape[1:-1] + (e0.shape[-1] + g0_.shape[-1],), e0.dtype)
        r = 0
        for e, g in data['obs']:
            obs[r:r + len(e), ..., :e.shape[-1]] = e
            obs[r:r + len(e), ..., e.shape[-1]:] = g
            r += len(e)
        data['obs'] = obs
    else:
        data['obs'] = np.zeros((0, 0), np.float32)
    for k in ('action', 'reward', 'done', 'true_state'):
        data[k] = np.concatenate(data[k]) if data[k] else np.zeros((0,))
    n_samples = len(data['reward'])
    print('  ', '%d trajectories for a total of %d samples' % (n_trajectories, n_samples))
    return data


class ShardWriter(object):
    """One rank's output while a run is in progress ("per-shard output files make a crashed 8-GPU run resumable"):
        <base>.rank<r>.obs.f32   raw float32 rows, appended block by block as they come off the encoder (never all in RAM)
        <base>.rank<r>.pickle    written LAST, through a rename: {'_shard': {...what this shard covers...}, action, reward, done,
                                 true_state[, png]} - its existence means the shard is complete
    A relaunch finds the pickle, compares `_shard` with what it would compute itself (rank, world size, source, embedding, row or
    trajectory range) and skips the work: only ranks that did not finish run again."""

    def __init__(self, save_name, rank):
        self.meta_path, self.obs_path = shard_name(save_name, rank), shard_name(save_name, rank)[:-len('.pickle')] + '.obs.f32'
        for q in (self.meta_path, self.obs_path):               # leftovers of an interrupted attempt at this shard
            if os.path.isfile(q):
                os.remove(q)
        self.f, self.rows, self.width = open(self.obs_path, 'wb'), 0, None

    def append(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.shape[0] == 0:
            return
        assert rows.ndim == 2 and (self.width is None or rows.shape[1] == self.width), rows.shape
        self.width = rows.shape[1]
        self.f.write(memoryview(rows).cast('B'))
        self.rows += rows.shape[0]

    def finish(self, small, cover):
        self.f.flush()
        os.fsync(self.f.fileno())
        self.f.close()
        meta = dict(small, _shard=dict(cover, rows=self.rows, width=self.width or 0))
        with open(self.meta_path + '.tmp', 'wb') as handle:
            pickle.dump(meta, handle, protocol=pickle.HIGHEST_PROTOCOL)
            handle.flush()
            os.fsync(handle.fileno())
        os.replace(self.meta_path + '.tmp', self.meta_path)


class _ObsStandIn(object):
    """Pickles the way numpy pickles a C-contiguous (n, width) float32 array under protocol 5 with in-band buffers -
    `_frombuffer(<payload>, dtype, shape, 'C')` - but with a 64 KiB sentinel bytearray where the n * width * 4 payload bytes belong."""

    def __init__(self, shape, sentinel):
        self.shape, self.sentinel = tuple(int(v) for v in shape), sentinel

    def __reduce_ex__(self, protocol):
        frombuffer = np.zeros(1, np.float32).__reduce_ex__(5)[0]        # numpy's own reconstructor, whatever module this numpy keeps it in
        return frombuffer, (self.sentinel, np.dtype(np.float32), self.shape, 'C')


class DirectPickleWriter(object):
    """Single-rank runs: embedding rows go STRAIGHT into the reference's output pickle (save_embedded_obs.py:165-172) - no shard row file,
    no stitch copy of the (N, D) matrix (1.6 GB for 100 k samples of two ResNet50 embeddings).  Possible because a pickle keeps a large
    buffer outside its frames as [opcode, 8-byte length, payload]: the stream around the payload is produced once, up front, by pickling
    the final dict with a stand-in for `obs` (same reconstructor, dtype, shape and order as numpy's own reduce) around a sentinel, and
    split at the sentinel; rows are then appended between the two halves.  The file is written under a temporary name and renamed at
    the end, so an interrupted run leaves no output (and restarts from scratch, as a single rank did before).  Same append / finish
    interface as ShardWriter."""

    def __init__(self, save_name, n_rows, width, small, keys):
        import struct
        self.save_name, self.tmp = save_name, save_name + '.tmp'
        self.n_rows, self.width, self.rows = int(n_rows), int(width), 0
        sentinel = bytearray(os.urandom(32)) * 2049                        # 65 568 B: above the pickler's 64 KiB frame target -> written outside the frames
        data = {'obs': _ObsStandIn((self.n_rows, self.width), sentinel)}
        for k in keys[1:]:
            data[k] = small[k]
        blob = pickle.dumps(data, protocol=pickle.HIGHEST_PROTOCOL)
        pos = blob.find(bytes(sentinel))
        assert pos >= 9 and blob.find(bytes(sentinel), pos + 1) < 0, 'sentinel not found exactly once'
        assert blob[pos - 9] == 0x96 and struct.unpack('<Q', blob[pos - 8:pos])[0] == len(sentinel), 'expected BYTEARRAY8 in front of the payload'
        self.suffix = blob[pos + len(sentinel):]
        self.f = open(self.tmp, 'wb')
        self.f.write(blob[:pos - 8] + struct.pack('<Q', 4 * self.n_rows * self.width))

    def append(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.shape[0] == 0:
            return
        assert rows.ndim == 2 and rows.shape[1] == self.width and self.rows + rows.shape[0] <= self.n_rows, (rows.shape, self.rows, self.n_rows, self.width)
        self.f.write(memoryview(rows).cast('B'))
        self.rows += rows.shape[0]

    def finish(self, small=None, cover=None):
        assert self.rows == self.n_rows, 'embedded %d rows of %d' % (self.rows, self.n_rows)
        self.f.write(self.suffix)
        self.f.flush()
        os.fsync(self.f.fileno())
        self.f.close()
        os.replace(self.tmp, self.save_name)


def load_complete_shard(save_name, rank, cover):
    """the shard's small arrays if <base>.rank<r>.pickle exists, covers exactly `cover` and its row file has the size it states"""
    path = shard_name(save_name, rank)
    if not os.path.isfile(path):
        return None
    try:
        with open(path, 'rb') as handle:
            meta = pickle.load(handle)
        sh = meta['_shard']
        ok = all(sh.get(k) == v for k, v in cover.items()) and \
            os.path.getsize(path[:-len('.pickle')] + '.obs.f32') == 4 * sh['rows'] * sh['width']
    except Exception:
        return None
    return meta if ok else None


def stitch_shards(save_name, world, keys, copy_bytes=256 << 20):
    """Rank 0, after the barrier: the reference's single output pickle (save_embedded_obs.py:165-172) from the shard files, in rank
    order (= the reference's row order), without ever holding the embeddings in RAM: the (N, D) float32 matrix is assembled in a
    file-backed np.memmap by bounded copies from the shards' row files and pickled from there (protocol 5 writes an array's
    buffer to the file as it is, no intermediate copy).  Round 2 np.concatenate'd every shard in memory: 125 GB for BASELINE
    config 5 (1 M frames x 31 310 floats)."""
    metas = []
    for r in range(world):
        with open(shard_name(save_name, r), 'rb') as handle:
            metas.append(pickle.load(handle))
    live = [(r, q) for r, q in enumerate(metas) if q['_shard']['rows'] > 0]
    n = sum(q['_shard']['rows'] for _, q in live)
    assert n > 0, 'no data found'
    widths = set(q['_shard']['width'] for _, q in live)
    assert len(widths) == 1, 'shards disagree on the embedding width: %s' % sorted(widths)
    width = widths.pop()
    obs_file = lambda r: shard_name(save_name, r)[:-len('.pickle')] + '.obs.f32'
    if len(live) == 1:
        tmp, obs = None, np.memmap(obs_file(live[0][0]), np.float32, 'r+', shape=(n, width))        # nothing to copy
    else:
        tmp = save_name + '.obs.tmp'
        obs = np.memmap(tmp, np.float32, 'w+', shape=(n, width))
        step, row = max(1, copy_bytes // (4 * width)), 0
        for r, q in live:
            src = np.memmap(obs_file(r), np.float32, 'r', shape=(q['_shard']['rows'], width))
            for a in range(0, src.shape[0], step):
                obs[row + a:row + min(a + step, src.shape[0])] = src[a:a + step]
            row += src.shape[0]
            del src
        obs.flush()
    small = [q for _, q in live]
    data = {'obs': np.asarray(obs)}
    for k in keys[1:]:
        data[k] = sum((list(q[k]) for q in small), []) if k == 'png' else np.concatenate([q[k] for q in small])
    assert len(data['reward']) == n, 'data length does not match'
    print('  ', 'total number of samples', n)
    with open(save_name + '.tmp', 'wb') as handle:
        pickle.dump(data, handle, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(save_name + '.tmp', save_name)
    del data, obs
    for q in ([tmp] if tmp else []) + [f for r in range(world) for f in (shard_name(save_name, r), obs_file(r))]:
        if os.path.isfile(q):
            os.remove(q)


def _weights_fingerprint(model):
    """identity of the embedding's weights for the shard-resume check: a shard left by a run with other weights must not be reused.  EVERY
    tensor is hashed in full (name, shape, dtype, bytes): a checkpoint that differs only in its last layers - a fine-tuned layer4, another
    compression head - must get another fingerprint.  state_dict() alone does not see every weight: UberModel keeps its members in a plain
    list (as the reference does, src/embeddings.py:44-57), FiveCrop wraps its model the same way - so those containers are walked
    explicitly (`models` / `model`), member by member, under an index prefix.  numpy values are hashed by their bytes too.  ~100 MB of sha1
    for a ResNet50, a fraction of a second next to the embedding pass it guards."""
    import hashlib
    h = hashlib.sha1()

    def feed(prefix, mod, depth=0):
        try:
            sd = mod.state_dict()
        except Exception:
            return False
        for k in sorted(sd):
            v = sd[k]
            h.update((prefix + k).encode())
            if hasattr(v, 'detach'):
                v = v.detach().cpu().contiguous().numpy() if v.numel() else np.zeros(0, np.uint8)
            if isinstance(v, np.ndarray):
                a = np.ascontiguousarray(v)
                h.update(str(tuple(a.shape)).encode() + str(a.dtype).encode())
                h.update(a.tobytes())
            else:
                h.update(repr(v).encode())
        if depth < 4:
            # containers whose members are not registered sub-modules: plain lists named `models` (UberModel) / `model` (FiveCrop)
            subs = [mod] + [m for m in mod.modules() if m is not mod] if hasattr(mod, 'modules') else [mod]
            for sm in subs:
                for attr in ('models', 'model'):
                    members = sm.__dict__.get(attr)
                    if isinstance(members, (list, tuple)):
                        for i, mem in enumerate(members):
                            if hasattr(mem, 'state_dict'):
                                feed('%s%s[%d].' % (prefix, attr, i), mem, depth + 1)
        return True

    return h.hexdigest()[:16] if feed('', model) else None


class BlockRing(object):
    """A ring of package-owned, page-locked row blocks between a producer that deposits observation rows (scene_rows' `take`) and a
    consumer that embeds whole blocks: rows are copied ONCE, from the unpickled trajectory buffer straight into pinned memory the
    H2D DMA reads (round 3 copied every trajectory to a pageable array, concatenated the block and staged strided channel planes:
    three host passes and 2x the block in memory).  put() blocks while every block is full and still being embedded; blocks come
    back through release().  The last block may be ragged.  (Pinned only when a GPU is present: the ring logic is CPU-testable.)"""

    def __init__(self, rows, row_shape, count=2, pinned=None):
        import queue
        pinned = torch.cuda.is_available() if pinned is None else pinned
        self.blocks = [torch.empty((rows,) + tuple(row_shape), dtype=torch.uint8, pin_memory=bool(pinned)) for _ in range(count)]
        self._np = [b.numpy() for b in self.blocks]             # put() copies through these views: one memcpy per deposit (see put)
        self.rows = rows
        self._free, self._full = queue.Queue(), queue.Queue()
        for i in range(count):
            self._free.put(i)
        self._cur, self._fill = None, 0

    # ---- producer side -----------------------------------------------------------------------------------------
    def put(self, rows):
        """rows: uint8 array (k,) + row_shape (a view is fine: it is copied before put returns)"""
        # numpy's copy, not torch's copy_: the latter runs on torch's intra-op pool, and from a reader thread on a 256-core host every
        # 12 MB deposit paid ~29 ms of thread wake-ups (0.36 GB/s measured; 2.3 GB/s = the unpickling rate with the plain memcpy)
        src = rows.numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows)
        done = 0
        while done < len(src):
            if self._cur is None:
                self._cur, self._fill = self._free.get(), 0
                if self._cur < 0:                                   # abort(): the consumer failed, nobody will release a block again
                    self._cur = None
                    raise RuntimeError('BlockRing: the consumer stopped')
            k = min(len(src) - done, self.rows - self._fill)
            np.copyto(self._np[self._cur][self._fill:self._fill + k], src[done:done + k])
            self._fill += k
            done += k
            if self._fill == self.rows:
                self._full.put((self._cur, self._fill))
                self._cur = None

    def close(self, error=None):
        """no more rows: hand over the ragged last block, then the end marker (or the producer's exception)"""
        if error is None and self._cur is not None and self._fill > 0:
            self._full.put((self._cur, self._fill))
        self._cur = None
        self._full.put(error if error is not None else None)

    # ---- consumer side -----------------------------------------------------------------------------------------
    def get(self):
        """(block index, tensor view of its filled rows), or None at the end; re-raises a producer failure"""
        got = self._full.get()
        if got is None:
            return None
        if isinstance(got, BaseException):
            raise got
        return got[0], self.blocks[got[0]][:got[1]]

    def release(self, index):
        self._free.put(index)

    def abort(self):
        """consumer side, on failure: a producer blocked in put() (every block full) wakes up and raises instead of waiting forever"""
        self._free.put(-1)


def _block_rows(flags, row_bytes, batch):
    """observation rows embedded per block (bounds the host memory of a rank): --embed_block, else about 1 GiB of frames"""
    blk = int(getattr(flags, 'embed_block', 0) or 0)
    return max(1, blk) if blk > 0 else max(batch, min(8192, (1 << 30) // max(1, row_bytes)))


def run(flags):
    save_name = os.path.join(flags.data_path, flags.env + '_' + flags.embedding_name + '.pickle')
    if os.path.isfile(save_name):
        return                                                  # idempotent skip (:97-101)
    dist, rank, world = _dist()
    torch.manual_seed(flags.run_id)
    np.random.seed(flags.run_id)
    random.seed(flags.run_id)
    flags.device = torch.device('cuda') if torch.cuda.is_available() and not flags.disable_cuda else torch.device('cpu')
    index_job = None
    if flags.source != 'png' and os.path.isfile(os.path.join(flags.data_path, flags.env) + '.pickle'):
        # pass 1 over the scene (lengths + the small arrays: file I/O and unpickling, no GPU) runs while the encoder is being built
        from concurrent.futures import ThreadPoolExecutor
        from .scene_pickle import scene_index as _scene_index
        _pool = ThreadPoolExecutor(max_workers=1)
        index_job = _pool.submit(_scene_index, os.path.join(flags.data_path, flags.env) + '.pickle')
        _pool.shutdown(wait=False)
    embedding_model = EmbeddingNet(flags.embedding_name, in_channels=3, pretrained=flags.pretrained_embedding,
                                   train=flags.train_embedding, disable_cuda=flags.disable_cuda,
                                   compute_dtype=getattr(flags, 'compute_dtype', None),
                                   max_batch=getattr(flags, 'embed_batch', 256), crops=getattr(flags, 'crops', 1))
    if rank == 0:
        emb_path = os.path.join(flags.data_path, flags.embedding_name)
        if flags.embedding_name == 'random':
            emb_path += '_' + str(flags.run_id)
        torch.save({'embedding_model_state_dict': embedding_model.state_dict()}, emb_path + '.tar')
    print('=== Loading trajectories ===')
    batch = getattr(flags, 'embed_batch', 256)
    keys = ('obs', 'action', 'reward', 'done', 'true_state')
    cover = dict(rank=rank, world=world, source=flags.source, embedding=flags.embedding_name, crops=int(getattr(flags, 'crops', 1)),
                 run_id=int(flags.run_id) if flags.embedding_name == 'random' else None, weights=_weights_fingerprint(embedding_model))
    if flags.source == 'png':
        # the goal frame is per trajectory, so the png source shards on trajectory boundaries: rank r takes trajectories [t_lo, t_hi)
        png_dir = os.path.join(flags.data_path, flags.env)
        keys = keys + ('png',)          # (the reference dumps this dict as it is, 'png' file list included: save_embedded_obs.py:53,78,171-172)
        t_range = shard_bounds(count_png_trajectories(png_dir, flags.n_trajectories), rank, world) if world > 1 else None
        cover['range'] = tuple(t_range) if t_range is not None else (0, int(flags.n_trajectories))
        if load_complete_shard(save_name, rank, cover) is None:
            writer = ShardWriter(save_name, rank)
            data = read_habitat_data_from_png(png_dir, embedding_model, flags.n_trajectories, batch, t_range=t_range, row_sink=writer.append)
            writer.finish({k: data[k] for k in keys[1:]}, cover)
        else:
            print('  ', 'rank %d: shard %s is complete, nothing to embed' % (rank, os.path.basename(shard_name(save_name, rank))))
    else:
        # (the reference reads the whole scene whatever --n_trajectories says: save_embedded_obs.py:142-145 passes no count)
        from .scene_pickle import scene_index, scene_rows
        scene = os.path.join(flags.data_path, flags.env) + '.pickle'
        print('loading %s ...' % scene[:-len('.pickle')])
        # pass 1: trajectory lengths + the small arrays; no frame is kept (started above, beside the encoder's construction)
        lengths, frame_shape, small = index_job.result() if index_job is not None else scene_index(scene)
        small = {k: np.concatenate(v) for k, v in small.items()}
        n_samples = int(sum(lengths))
        print('  ', '%d trajectories for a total of %d samples' % (len(lengths), n_samples))
        print('  ', 'avg. return is', small['reward'].sum() / max(1, len(lengths)))
        lo, hi = shard_bounds(n_samples, rank, world)
        cover['range'] = (int(lo), int(hi))
        st_ = os.stat(scene)
        cover['scene'] = (int(st_.st_size), int(st_.st_mtime))      # a regenerated scene of equal length is not the scene this shard embedded
        if load_complete_shard(save_name, rank, cover) is None:
            print('  ', 'passing observations through embedding model')
            n_frames = max(frame_shape[2] // 3, 1) if frame_shape else 1
            direct = (world == 1 and hi > lo and frame_shape and hasattr(getattr(embedding_model, 'embedding', None), 'forward_into')
                      and os.environ.get('PVR_DIRECT_PICKLE', '1') != '0')
            if direct:
                # one rank: rows go straight into the output pickle (no shard row file, no stitch copy)
                writer = DirectPickleWriter(save_name, hi - lo, n_frames * embedding_model.out_size, small, keys)
                print('  ', 'total number of samples', hi - lo)
            else:
                writer = ShardWriter(save_name, rank)
            # (a host-backend encoder - disable_cuda / no GPU - takes the reference's own batch loop below: nothing to overlap on the CPU)
            hip = hasattr(getattr(embedding_model, 'embedding', None), 'forward_into') and not getattr(embedding_model, '_host', False)
            block = _block_rows(flags, int(np.prod(frame_shape)) if frame_shape else 1, batch)
            if hip and hi > lo:
                # HIP encoder: every frame is embedded independently (bit-exact batch-composition invariance is a GPU test), so the
                # per-batch split/stack/concat of save_embedded_obs.py:151-156 is reproduced by streaming whole (H,W,3F) rows through the
                # overlapped H2D / compute / D2H path, each 3-channel plane embedded into its column block (stream_embed, planes=F).
                # A reader thread unpickles the scene and deposits rows into a ring of pinned blocks one block ahead of the embedder.
                import threading
                block = min(block, hi - lo)
                ring = BlockRing(block, frame_shape, count=2)
                osz = embedding_model.out_size
                out_buf = torch.empty((block, n_frames * osz), dtype=torch.float32, pin_memory=torch.cuda.is_available())

                def reader():
                    try:
                        scene_rows(scene, lo, hi, ring.put)         # pass 2: only this rank's rows, straight into pinned blocks
                        ring.close()
                    except BaseException as exc:                    # noqa: BLE001 - re-raised by ring.get() in the embedding thread
                        ring.close(exc)
                th = threading.Thread(target=reader, name='pvr-scene-reader', daemon=True)
                th.start()
                try:
                    while True:
                        got = ring.get()
                        if got is None:
                            break
                        idx, rows = got
                        emb = stream_embed(embedding_model, rows, batch, out=out_buf[:len(rows)], planes=n_frames)
                        writer.append(emb.numpy())
                        ring.release(idx)
                finally:
                    ring.abort()                                    # (a no-op after a clean end: the reader has returned)
                    th.join(timeout=60)
            else:
                pending, n_pending = [], [0]

                def flush():
                    if not pending:
                        return
                    obs = pending[0] if len(pending) == 1 else np.concatenate(pending)
                    del pending[:]
                    n_pending[0] = 0
                    writer.append(embed_rows(embedding_model, obs, n_frames, max(1, batch // n_frames)))

                def take(rows):                                # rows: a view of one trajectory's buffer - keep a copy, embed per block
                    pending.append(np.array(rows))
                    n_pending[0] += len(rows)
                    if n_pending[0] >= block:
                        flush()
                if hi > lo:
                    scene_rows(scene, lo, hi, take)             # pass 2: only this rank's rows are kept, one block at a time
                flush()
            writer.finish({k: small[k][lo:hi] for k in keys[1:]}, cover)
            if direct:
                return                                              # the output pickle is complete
        else:
            print('  ', 'rank %d: shard %s is complete, nothing to embed' % (rank, os.path.basename(shard_name(save_name, rank))))
    # every rank has written its own shard files; rank 0 stitches them in rank order (= the reference's row order): nothing but a
    # barrier crosses ranks, and no rank ever holds another rank's rows (cfg 5: 125 GB of embeddings)
    if world > 1:
        dist.barrier()
    if rank == 0:
        stitch_shards(save_name, world, keys)


def shard_name(save_name, rank):
    """<data_path>/<env>_<embedding>.rank<r>.pickle: rank r's rows while a multi-GPU run is in progress"""
    return save_name[:-len('.pickle')] + '.rank%d.pickle' % rank


def main(argv=None):
    flags = make_parser().parse_args(argv)
    init_distributed()                                          # device + process group first, before any GPU call
    try:
        run(flags)
    finally:
        finalize_distributed()


if __name__ == '__main__':
    main()
