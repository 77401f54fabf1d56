"""PNG files -> frames in HBM (reference behavioral_cloning/save_embedded_obs.py:63-64,71-72 reads every frame with cv2.imread on
the host).  The host reads file BYTES only (threads: file I/O releases the GIL); one launch of csrc/png_decode.hip inflates and
unfilters all of them and leaves cv2.imread's (n, H, W, 3) uint8 B,G,R array on the device, where the encoder takes it without a
host round trip.  Files the kernel does not cover (palette, 16-bit, interlaced) are decoded on the host by png_decode.imread and
copied in; a corrupt file raises (cv2.imread would have returned None and the reference would have failed on it)."""
import ctypes as C
import os
import struct

import numpy as np
import torch

from . import _lib
from .png_decode import imread

_STATUS = {1: 'unsupported PNG kind', 2: 'not a PNG (signature / IHDR)', 3: 'truncated file', 4: 'bad zlib header', 5: 'bad deflate block',
           6: 'bad Huffman code', 7: 'distance before the start of the stream', 8: 'more data than the image holds',
           9: 'less data than the image holds', 10: 'Adler-32 mismatch', 11: 'unknown scanline filter', 12: 'image size differs'}


def read_files(paths, threads=16):
    """file bytes of `paths`, concatenated: (page-locked uint8 tensor, int64 offsets of n + 1 entries).  Read by native threads
    (pvr_read_files: the interpreter lock is not held; 16 Python threads managed ~25 k files/s on the GPU box)."""
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    off = np.zeros(n + 1, np.int64)
    _lib.check(_lib.lib().pvr_file_sizes(arr, n, C.c_void_p(off[1:].ctypes.data), threads))
    np.cumsum(off[1:], out=off[1:])
    buf = torch.empty((int(off[-1]) + 16,), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
    _lib.check(_lib.lib().pvr_read_files(arr, n, C.c_void_p(buf.data_ptr()), C.c_void_p(off.ctypes.data), threads))
    return buf, off


def png_size(blob):
    """(H, W) from the IHDR chunk of a file's first 33 bytes"""
    blob = bytes(blob)
    if len(blob) < 33 or blob[:8] != b'\x89PNG\r\n\x1a\n' or blob[12:16] != b'IHDR':
        raise ValueError('not a PNG file')
    w, h = struct.unpack('>II', blob[16:24])
    return h, w


def decode_files(paths, threads=16, size=None):
    """cv2.imread of every path, stacked, ON THE DEVICE: uint8 CUDA tensor (n, H, W, 3) in B,G,R order.  All files must have one size
    (np.stack in the reference's loader requires it too); `size` = (H, W) if known."""
    _lib.require_gpu()
    n = len(paths)
    buf, off = read_files(paths, threads)
    h, w = size if size is not None else png_size(buf[:min(33, int(off[1]))].numpy().tobytes())
    dev = torch.device('cuda', torch.cuda.current_device())
    files = buf.to(dev, non_blocking=True)
    offsets = torch.from_numpy(off).to(dev, non_blocking=True)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    sb = int(_lib.lib().pvr_png_scratch_bytes(n, h, w))
    scratch = torch.empty((sb,), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().pvr_png_decode(C.c_void_p(files.data_ptr()), C.c_void_p(offsets.data_ptr()), n, h, w, C.c_void_p(out.data_ptr()),
                                         C.c_void_p(scratch.data_ptr()), sb, C.c_void_p(status.data_ptr()), _lib.stream_ptr()))
    st = status.cpu().numpy()
    for i in np.nonzero(st)[0]:
        if st[i] == 1:                             # a PNG kind the kernel leaves to the host decoder
            a = imread(paths[i])
            if a is None or a.shape != (h, w, 3):
                raise ValueError('%s: cannot decode as a %dx%d image' % (paths[i], h, w))
            out[i].copy_(torch.from_numpy(a))
        else:
            raise ValueError('%s: %s' % (paths[i], _STATUS.get(int(st[i]), 'status %d' % st[i])))
    return out
