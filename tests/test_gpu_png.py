"""csrc/png_decode.hip against the host decoders (PIL here, cv2 where installed): the per-frame PNG source of reference
behavioral_cloning/save_embedded_obs.py:50-93 decoded on the GPU must give cv2.imread's array bit for bit - every deflate block
type, zlib strategy and window size, every scanline filter, IDAT payloads split at awkward places, every supported colour type -
and must reject what libpng rejects."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from pvr_habitat_amd import synth

pytestmark = pytest.mark.gpu


def _chunk(tag, data):
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, img, filters=(0, 1, 2, 3, 4), level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, splits=(), extra_chunks=True):
    """A PNG writer with every choice exposed: img (H,W) or (H,W,C) uint8 with C in 1..4 (grey, grey+alpha, RGB, RGBA); row y uses
    scanline filter filters[y % len]; the zlib stream is cut into IDAT chunks at the byte positions `splits`."""
    img = np.asarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = img.reshape(h, w * c).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur, up = rows[y], (rows[y - 1] if y else np.zeros(w * c, np.int32))
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), up[:-c]])
        if ft == 0: pred = 0
        elif ft == 1: pred = left
        elif ft == 2: pred = up
        elif ft == 3: pred = (left + up) >> 1
        else: pred = np.array([_paeth(int(a), int(b), int(cc)) for a, b, cc in zip(left, up, ul)], np.int32)
        raw.append(ft)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    z = co.compress(bytes(raw)) + co.flush()
    cuts = [0] + sorted(s for s in splits if 0 <= s <= len(z)) + [len(z)]
    out = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0))
    if extra_chunks:
        out += _chunk(b'pHYs', struct.pack('>IIB', 2835, 2835, 1)) + _chunk(b'tEXt', b'Comment\x00synthetic')
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += _chunk(b'IDAT', z[a:b])
    out += _chunk(b'IEND', b'')
    with open(path, 'wb') as f:
        f.write(out)
    return len(z)


def _bgr(img):
    """what cv2.imread(IMREAD_COLOR) returns for the array written above"""
    img = np.asarray(img, np.uint8)
    if img.ndim == 2 or img.shape[2] <= 2:
        g = img if img.ndim == 2 else img[..., 0]
        return np.repeat(g[..., None], 3, axis=2)
    return np.ascontiguousarray(img[..., 2::-1])


def _frames(n, h, w, seed=5):
    return synth.smooth_frames(seed, n, h, w)


def test_every_deflate_block_type_strategy_and_filter(tmp_path):
    from pvr_habitat_amd import png_gpu
    fr = _frames(40, 64, 64)
    rng = np.random.default_rng(0)
    paths, want = [], []
    cases = [dict(level=0), dict(level=1, strategy=zlib.Z_RLE), dict(level=1), dict(level=6), dict(level=9), dict(strategy=zlib.Z_FIXED),
             dict(strategy=zlib.Z_HUFFMAN_ONLY), dict(strategy=zlib.Z_FILTERED), dict(wbits=9), dict(wbits=12), dict(filters=(0,)), dict(filters=(1,)),
             dict(filters=(2,)), dict(filters=(3,)), dict(filters=(4,)), dict(filters=(4, 3, 2, 1, 0, 2, 4)), dict(splits=(1,)), dict(splits=(2, 2, 3)),
             dict(splits=(0, 5, 4000)), dict(splits=tuple(range(0, 9000, 257))), dict(extra_chunks=False)]
    for i, kw in enumerate(cases):
        img = fr[i]
        p = str(tmp_path / ('c%d.png' % i)); write_png(p, img, **kw); paths.append(p); want.append(_bgr(img))
    noise = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)            # incompressible: stored blocks / long literal runs
    for i, kw in enumerate([dict(level=6), dict(level=0, splits=(7, 8191, 8192))]):
        p = str(tmp_path / ('n%d.png' % i)); write_png(p, noise, **kw); paths.append(p); want.append(_bgr(noise))
    flat = np.full((64, 64, 3), 37, np.uint8)                              # one long run: maximal <length, distance 1..3> copies
    p = str(tmp_path / 'flat.png'); write_png(p, flat, filters=(0,)); paths.append(p); want.append(_bgr(flat))
    out = png_gpu.decode_files(paths)
    assert out.is_cuda and out.dtype == torch.uint8 and tuple(out.shape) == (len(paths), 64, 64, 3)
    got = out.cpu().numpy()
    for i, p in enumerate(paths):
        assert np.array_equal(got[i], want[i]), p
    from pvr_habitat_amd.png_decode import imread                         # and the host decoder agrees with the test's own writer
    assert all(np.array_equal(imread(p), w_) for p, w_ in zip(paths[:6], want[:6]))


@pytest.mark.parametrize('shape', [(1, 1, 3), (37, 53, 3), (5, 300, 3), (256, 256, 3), (64, 64, 4), (64, 64, 1), (31, 17, 2), (64, 64)])
def test_sizes_and_colour_types(tmp_path, shape):
    from pvr_habitat_amd import png_gpu
    rng = np.random.default_rng(sum(shape))
    n = 5
    base = _frames(n, max(shape[0], 2), max(shape[1], 2), seed=9)[:, :shape[0], :shape[1]]
    paths, want = [], []
    for i in range(n):
        if len(shape) == 2: img = base[i][..., 0]
        elif shape[2] == 3: img = base[i]
        elif shape[2] == 4: img = np.concatenate([base[i], rng.integers(0, 256, shape[:2] + (1,), dtype=np.uint8)], -1)
        else: img = np.concatenate([base[i][..., :1]] + [rng.integers(0, 256, shape[:2] + (1,), dtype=np.uint8)] * (shape[2] - 1), -1)
        p = str(tmp_path / ('s%d.png' % i))
        write_png(p, img, filters=(i % 5, 4, 1), level=(0, 1, 6, 9, 6)[i], splits=(() if i % 2 else (3, 100, 101)))
        paths.append(p); want.append(_bgr(img))
    got = png_gpu.decode_files(paths).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], want[i]), (shape, i)


def test_files_written_by_pil_and_a_large_batch(tmp_path):
    """The encoders of the image libraries themselves (adaptive filtering, their own zlib settings, 64 KB IDAT chunks), 700 files in
    one call (44 wavefronts of 16 files), against the host decoder the product used before."""
    from PIL import Image
    from pvr_habitat_amd import png_gpu
    from pvr_habitat_amd.png_decode import imread
    fr = _frames(700, 64, 64, seed=3)
    paths = []
    for i in range(700):
        p = str(tmp_path / ('%d_%d.png' % (i // 250, i % 250)))
        Image.fromarray(fr[i]).save(p, compress_level=(1, 6, 9, 0)[i % 4], optimize=bool(i % 3 == 0))
        paths.append(p)
    got = png_gpu.decode_files(paths, threads=8).cpu().numpy()
    want = np.stack([imread(p) for p in paths])
    assert np.array_equal(got, want)
    assert np.array_equal(got, fr[..., ::-1])
    big = _frames(2, 300, 500, seed=4)                                      # > 64 KB of compressed data: several IDAT chunks from PIL
    rng = np.random.default_rng(1)
    big[1] = rng.integers(0, 256, big[1].shape, dtype=np.uint8)
    bp = []
    for i in range(2):
        p = str(tmp_path / ('big%d.png' % i)); Image.fromarray(big[i]).save(p); bp.append(p)
    assert os.path.getsize(bp[1]) > 300000
    assert np.array_equal(png_gpu.decode_files(bp).cpu().numpy(), big[..., ::-1])


def test_unsupported_kinds_fall_back_to_the_host_decoder_and_corrupt_files_raise(tmp_path):
    from PIL import Image
    from pvr_habitat_amd import png_gpu
    from pvr_habitat_amd.png_decode import imread
    fr = _frames(4, 48, 40, seed=8)
    good = str(tmp_path / 'good.png'); write_png(good, fr[0])
    pal = str(tmp_path / 'palette.png'); Image.fromarray(fr[1]).convert('P', palette=Image.ADAPTIVE).save(pal)
    inter = str(tmp_path / 'deep.png'); Image.fromarray((fr[2][..., 0].astype(np.uint16) * 257)).save(inter)       # 16-bit grey
    out = png_gpu.decode_files([good, pal, inter, good]).cpu().numpy()
    assert np.array_equal(out[0], _bgr(fr[0])) and np.array_equal(out[3], out[0])
    assert np.array_equal(out[1], imread(pal)) and np.array_equal(out[2], imread(inter))
    blob = open(good, 'rb').read()
    idat = blob.index(b'IDAT')

    def bad(name, data):
        p = str(tmp_path / name)
        with open(p, 'wb') as f:
            f.write(data)
        return p
    flipped = bytearray(blob); flipped[idat + 4 + 40] ^= 0x55
    cases = [bad('flipped.png', bytes(flipped)), bad('cut.png', blob[:idat + 60]), bad('text.png', b'not a png at all' * 8),
             bad('noend.png', blob[:-12 - 4]), bad('hdr.png', blob[:idat + 4] + b'\x79' + blob[idat + 5:])]
    for p in cases:
        with pytest.raises(ValueError):
            png_gpu.decode_files([good, p, good])
    other = str(tmp_path / 'other.png'); write_png(other, fr[3][:20, :30])
    with pytest.raises(ValueError, match='size'):
        png_gpu.decode_files([good, other])


def test_png_source_through_the_encoder_matches_the_host_decoded_run(tmp_path):
    """save_embedded_obs.read_habitat_data_from_png with the frames decoded on the GPU == the same call with the host decoders
    (bit-identical embeddings: the encoder sees the same uint8 frames), incl. a trajectory boundary inside a decode group."""
    import pickle
    from PIL import Image
    from pvr_habitat_amd import save_embedded_obs as S
    from pvr_habitat_amd.embeddings import EmbeddingNet
    os.environ.setdefault('PVR_SYNTHETIC_WEIGHTS', '1')
    d = str(tmp_path / 'scene'); os.makedirs(d)
    fr = _frames(64, 64, 64, seed=12)
    lens = [7, 1, 12, 5, 9, 3]
    k = 0
    for t, L in enumerate(lens):
        for s in range(L):
            Image.fromarray(fr[k][..., ::-1]).save(os.path.join(d, '%d_%d.png' % (t, s))); k += 1
        Image.fromarray(fr[40 + t][..., ::-1]).save(os.path.join(d, '%d_goal.png' % t))
        pickle.dump(dict(action=np.arange(L), reward=np.ones(L), done=np.zeros(L, bool), true_state=np.zeros((L, 3))), open(os.path.join(d, '%d.pickle' % t), 'wb'))
    net = EmbeddingNet('resnet18', pretrained=False, max_batch=16)
    a = S.read_habitat_data_from_png(d, net, -1, batch=16, decode_workers=1, gpu_decode=False)
    b = S.read_habitat_data_from_png(d, net, -1, batch=16, decode_workers=4, gpu_decode=True)
    assert a['obs'].shape == (sum(lens), 2 * net.out_size) and np.array_equal(a['obs'], b['obs'])
    assert a['png'] == b['png'] and np.array_equal(a['action'], b['action'])


def test_mutated_files_never_hang_and_never_decode_to_wrong_pixels(tmp_path):
    """1500 files damaged at random (bit flips, byte substitutions, truncations, bytes inserted / deleted - in headers, chunk
    framing, Huffman tables and data alike), decoded one by one status-wise in a single launch: the kernel must come back, and
    whenever it reports success on a file the host decoder also accepts, the pixels must be the same."""
    import ctypes as C
    import io
    from PIL import Image
    from pvr_habitat_amd import _lib, png_gpu
    fr = _frames(6, 40, 56, seed=21)
    rng = np.random.default_rng(4)
    seeds = []
    for i in range(6):
        p = str(tmp_path / ('seed%d.png' % i))
        write_png(p, fr[i], level=(0, 1, 6, 9, 6, 6)[i], strategy=(zlib.Z_FIXED if i == 4 else zlib.Z_DEFAULT_STRATEGY), splits=(() if i % 2 else (5, 900)))
        seeds.append(open(p, 'rb').read())
    blobs = []
    for k in range(1500):
        b = bytearray(seeds[k % 6])
        for _ in range(int(rng.integers(1, 4))):
            kind, pos = int(rng.integers(0, 5)), int(rng.integers(8, len(b)))
            if kind == 0: b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1: b[pos] = int(rng.integers(0, 256))
            elif kind == 2: del b[pos:]
            elif kind == 3: b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            else: del b[pos:pos + int(rng.integers(1, 9))]
            if len(b) < 9:
                b = bytearray(seeds[k % 6][:9])
        blobs.append(bytes(b))
    off = np.zeros(len(blobs) + 1, np.int64); np.cumsum([len(b) for b in blobs], out=off[1:])
    files = torch.from_numpy(np.frombuffer(b''.join(blobs) + b'\0' * 16, np.uint8).copy()).cuda()
    n, h, w = len(blobs), 40, 56
    out = torch.zeros((n, h, w, 3), dtype=torch.uint8, device='cuda'); status = torch.full((n,), -1, dtype=torch.int32, device='cuda')
    sb = int(_lib.lib().pvr_png_scratch_bytes(n, h, w)); scratch = torch.empty((sb,), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().pvr_png_decode(C.c_void_p(files.data_ptr()), C.c_void_p(torch.from_numpy(off).cuda().data_ptr()), n, h, w,
                                         C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), sb, C.c_void_p(status.data_ptr()), _lib.stream_ptr()))
    torch.cuda.synchronize()
    st, got = status.cpu().numpy(), out.cpu().numpy()
    assert (st >= 0).all() and (st <= 12).all()
    ok = both = 0
    for k in range(n):
        if st[k] != 0:
            continue
        ok += 1
        try:
            im = Image.open(io.BytesIO(blobs[k])); im.load()
            if im.size != (w, h) or im.mode not in ('RGB', 'RGBA', 'L', 'LA'):
                continue
            host = np.asarray(im.convert('RGB'))[..., ::-1]
        except Exception:
            continue                                                # (the host decoder also checks chunk CRCs; the kernel checks Adler-32 only)
        both += 1
        assert np.array_equal(got[k], host), k
    print('\n%d of %d damaged files still decode on the GPU, %d of them also on the host (identical pixels); statuses %s'
          % (ok, n, both, np.bincount(st, minlength=13).tolist()))
    assert ok < n // 2 and both > 0
