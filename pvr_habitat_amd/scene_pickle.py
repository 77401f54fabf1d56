"""Bounded-memory access to the per-scene trajectory pickle of reference save_opt_trajectories.py:100-106
    {obs: [(L,H,W,3n) uint8, ...], action: [...], reward: [...], done: [...], true_state: [(L,12), ...]}
which the reference loads whole and concatenates (save_embedded_obs.py:29-47, utils_bc.py read_habitat_data): 1 M frames of 256x256x6
bytes do not fit a host, and under torch.distributed every rank would hold all of them.

The file is ONE pickle stream, so it cannot be seeked per trajectory; but numpy arrays inside it are rebuilt through two module-level
functions (`numpy.core.numeric._frombuffer` for protocol 5, `numpy.core.multiarray._reconstruct` + `__setstate__` below it), which a
`pickle.Unpickler.find_class` override can replace.  The replacement sees every observation block (the 4-D uint8 arrays, in file
order = row order) the moment it has been read, hands the rows a caller asked for to a sink and drops the rest, so a pass holds one
trajectory at a time:
    index pass   `scene_index(path)`            -> trajectory lengths + the small per-row arrays (action, reward, done, true_state)
    row pass     `scene_rows(path, lo, hi, fn)`  -> fn(block) for the observation rows [lo, hi) in order, block by block
Both passes read the file sequentially (8 ranks share the page cache).

The one-trajectory bound holds for pickle protocol >= 3 (bytes payloads; save_opt_trajectories.py writes protocol 4 under Python >= 3.8,
and the bound is tested for 3, 4 and 5).  A protocol <= 2 scene still reads correctly, but its payload arrives as a latin-1 str through
_codecs.encode and stays in the unpickler's memo until the end of the pass (measured: 58 MB peak for 49 MB of frames at protocol 2):
both passes warn about it."""
import pickle
import warnings

import numpy as np

_SMALL_KEYS = ('action', 'reward', 'done', 'true_state')


def _is_obs_block(shape, dtype):
    return len(shape) == 4 and np.dtype(dtype) == np.uint8


class _Rows(object):
    """what to do with each observation block: rows [lo, hi) of the concatenated scene go to `sink`, everything else is dropped"""

    def __init__(self, lo, hi, sink):
        self.lo, self.hi, self.sink, self.row, self.lengths, self.frame_shape = lo, hi, sink, 0, [], None

    def take(self, arr):
        L = arr.shape[0]
        a, b = max(self.lo - self.row, 0), min(self.hi - self.row, L)
        if self.sink is not None and b > a:
            self.sink(arr[a:b])
        self.row += L
        self.lengths.append(L)
        if self.frame_shape is None:
            self.frame_shape = tuple(arr.shape[1:])
        return np.empty((0,) + tuple(arr.shape[1:]), np.uint8)          # placeholder left in the unpickled structure


class _SceneUnpickler(pickle._Unpickler):
    """The pure-Python unpickler (a scene is a few opcodes per trajectory around multi-megabyte payloads, which it reads with one
    file.read each): its memo is a plain dict, and the memo is what would otherwise keep every trajectory's bytes alive until the end
    of the load - each payload is MEMOIZEd when read.  Once a block has been handed on, its memo slots are cleared (nothing refers
    back to a raw payload: a repeated array refers to the array object)."""

    def __init__(self, f, rows):
        super().__init__(f)
        self._rows = rows

    def _forget(self, *objs):
        memo = self.memo
        for k in range(len(memo) - 1, max(len(memo) - 48, -1), -1):
            v = memo.get(k)
            # the payload itself, or a tuple that carries it (the argument tuple of _frombuffer / the __setstate__ state: memoized too)
            if any(v is o for o in objs) or (type(v) is tuple and any(x is o for x in v for o in objs)):
                memo[k] = None

    def find_class(self, module, name):
        rows, forget = self._rows, self._forget
        if name == '_frombuffer' and module in ('numpy.core.numeric', 'numpy._core.numeric'):
            def frombuffer(buf, dtype, shape, order):
                arr = np.frombuffer(buf, dtype=dtype).reshape(shape, order=order)
                if not _is_obs_block(shape, dtype):
                    return arr
                forget(buf)
                return rows.take(arr)
            return frombuffer
        if name == '_reconstruct' and module in ('numpy.core.multiarray', 'numpy._core.multiarray'):
            class Lazy(np.ndarray):
                def __setstate__(self, state):
                    shape, dtype = state[1], state[2]
                    if not _is_obs_block(shape, dtype):
                        return np.ndarray.__setstate__(self, state)
                    full = np.ndarray.__new__(np.ndarray, (0,), np.uint8)
                    full.__setstate__(state)
                    forget(state, state[4])
                    kept = rows.take(full)
                    np.ndarray.__setstate__(self, (state[0], kept.shape, state[2], False, b''))

            def reconstruct(subtype, shape, dtype):
                return np.ndarray.__new__(Lazy if subtype is np.ndarray else subtype, shape, dtype)
            return reconstruct
        return super().find_class(module, name)


def _pass(path, lo, hi, sink):
    rows = _Rows(lo, hi, sink)
    with open(path, 'rb') as f:
        head = f.read(2)
        f.seek(0)
        proto = head[1] if len(head) == 2 and head[0] == 0x80 else 0
        if proto < 3:
            warnings.warn('%s is a protocol-%d pickle: it reads correctly, but the one-trajectory memory bound of this reader needs protocol >= 3 '
                          '(re-save the scene with pickle.dump(..., protocol=4) for scenes that do not fit the host)' % (path, proto), RuntimeWarning)
        data = _SceneUnpickler(f, rows).load()
    return data, rows


def scene_index(path):
    """-> (lengths per trajectory, frame shape (H,W,C), {action, reward, done, true_state: lists of per-trajectory arrays});
    no observation is kept."""
    data, rows = _pass(path, 0, 0, None)
    if not rows.lengths and len(data.get('obs', [])):                   # observations that are not 4-D uint8 blocks (true_state-only scenes)
        rows.lengths = [len(o) for o in data['obs']]
    small = {k: [np.asarray(v) for v in data[k]] for k in _SMALL_KEYS if k in data}
    return rows.lengths, rows.frame_shape, small


def scene_rows(path, lo, hi, sink):
    """sink(block) for the observation rows [lo, hi) of the concatenated scene, in order; blocks are views of one trajectory's
    buffer (copy what must outlive the call).  Returns the number of rows delivered."""
    n = [0]

    def count(block):
        n[0] += len(block)
        sink(block)
    _pass(path, lo, hi, count)
    return n[0]
