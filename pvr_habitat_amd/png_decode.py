"""PNG decoding for the per-frame png source (reference behavioral_cloning/save_embedded_obs.py:50-93 reads one file per frame with
cv2.imread).  Decoding a 64x64 PNG costs ~0.2 ms and holds the GIL, so threads do not scale it (measured on the GPU box: 5.2 k
frames/s with one thread, 3.3 k with 32); worker PROCESSES do.  The workers are plain child processes running this module
(`python -m pvr_habitat_amd.png_decode`) that read pickled lists of paths on stdin and write pickled uint8 stacks to stdout - no
multiprocessing start method, so nothing re-imports the caller's __main__ and nothing forks a process that holds a GPU context.
This module imports nothing heavy and never touches the GPU."""
import os
import pickle
import struct
import subprocess
import sys

import numpy as np


def imread(path):
    """cv2.imread equivalent (the reference writes RGB arrays with cv2.imwrite and reads them back with cv2.imread, so the array
    round-trips; PIL returns the file's RGB, i.e. the array reversed).  None for a missing file, like cv2."""
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        if not os.path.isfile(path):
            return None
        try:
            return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[..., ::-1])
        except Exception as e:                                  # an undecodable file is an error that names the file
            raise ValueError('cannot decode %s: %s' % (path, e))


def decode_many(paths):
    """stack of the decoded frames of `paths` (all the same size): (n,H,W,3) uint8"""
    frames = [imread(p) for p in paths]
    for p, f in zip(paths, frames):
        if f is None:                                           # (cv2.imread's answer to a missing or undecodable file)
            raise ValueError('cannot decode %s' % p)
    return np.stack(frames)


def _send(f, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack('<Q', len(b)))
    f.write(b)
    f.flush()


def _recv(f):
    h = f.read(8)
    if len(h) < 8:
        raise EOFError('png decode worker closed its pipe')
    return pickle.loads(f.read(struct.unpack('<Q', h)[0]))


class _Pool(object):
    def __init__(self, workers):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1')
        self.procs = [subprocess.Popen([sys.executable, '-m', 'pvr_habitat_amd.png_decode'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for _ in range(workers)]

    def map(self, chunks):
        """decode the path lists in `chunks`, in order: chunk i goes to worker i mod n; a worker answers its chunks in the order sent"""
        n = len(self.procs)
        out = [None] * len(chunks)
        for base in range(0, len(chunks), n):                       # one round: every worker gets at most one chunk, then all are collected
            rnd = chunks[base:base + n]
            for w, c in enumerate(rnd):
                _send(self.procs[w].stdin, c)
            for w in range(len(rnd)):
                r = _recv(self.procs[w].stdout)
                if isinstance(r, Exception):
                    raise r
                out[base + w] = r
        return out

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            p.wait(timeout=10)


_POOL = None


def decode_parallel(paths, workers, chunk=None):
    """decode `paths` in order with `workers` processes (<= 1, or few files: in this process).  chunk: files per task; by default
    the list is cut so that every worker gets one task (a trajectory has at most 500 frames: with a fixed chunk of 32 only 8-16
    workers ever had work)"""
    global _POOL
    if workers <= 1 or len(paths) < 64:
        return decode_many(paths) if paths else None
    if chunk is None:
        chunk = max(4, (len(paths) + workers - 1) // workers)
    if _POOL is None or len(_POOL.procs) != workers:
        shutdown()
        _POOL = _Pool(workers)
    return np.concatenate(_POOL.map([paths[i:i + chunk] for i in range(0, len(paths), chunk)]))


def shutdown():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL = None


def _worker():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                                          # nothing but frames on the pipe
    while True:
        try:
            paths = _recv(inp)
        except EOFError:
            return
        try:
            _send(out, decode_many(paths))
        except Exception as e:                                       # noqa: BLE001 - reported to the parent
            _send(out, RuntimeError('png decode worker: %s: %s' % (type(e).__name__, e)))


if __name__ == '__main__':
    _worker()
