"""PNG decoding for the per-frame png source (reference behavioral_cloning/save_embedded_obs.py:50-93 reads one file per frame with
cv2.imread).  Decoding 64x64 PNGs costs ~0.2 ms each and holds the GIL, so threads do not scale it (measured on the GPU box:
5.2 k frames/s with one thread, 3.3 k with 32); worker PROCESSES do.  This module imports nothing heavy so that spawned workers
start fast; it never touches the GPU."""
import os

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
        return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[..., ::-1])


def decode_many(paths):
    """stack of the decoded frames of `paths` (all the same size): (n,H,W,3) uint8"""
    return np.stack([imread(p) for p in paths])


_POOL = None


def pool(workers):
    """process pool (spawn context: the parent may hold a GPU context; the children only run imread), created once"""
    global _POOL
    if _POOL is None:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        _POOL = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn'))
    return _POOL


def decode_parallel(paths, workers, chunk=32):
    """decode `paths` in order with `workers` processes (<= 1, or few files: in this process)"""
    if workers <= 1 or len(paths) < 2 * chunk:
        return decode_many(paths) if paths else None
    parts = list(pool(workers).map(decode_many, [paths[i:i + chunk] for i in range(0, len(paths), chunk)]))
    return np.concatenate(parts)


def shutdown():
    global _POOL
    if _POOL is not None:
        _POOL.shutdown()
        _POOL = None
