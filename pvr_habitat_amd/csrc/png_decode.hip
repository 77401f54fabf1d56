// PNG source on the GPU: the per-frame files of reference behavioral_cloning/save_embedded_obs.py:50-93 (`cv2.imread` of
// <t>_goal.png and <t>_<s>.png, one zlib stream of 64x64x3 filtered scanlines each) are decoded where they are consumed.
// The host only reads file bytes; one launch inflates n files (RFC 1950 / 1951: stored, fixed and dynamic Huffman blocks,
// IDAT chunk boundaries crossed inside the byte reader), checks Adler-32, undoes the five PNG scanline filters (PNG spec 9.2) and
// a second, coalesced launch writes cv2.imread's layout (n, h, w, 3) uint8 B,G,R straight into HBM for the encoder.
//
// Parallelism is across files, not inside one: a DEFLATE stream is serial by construction (every symbol's position depends on
// all earlier code lengths), so one lane walks one file - bit reader, canonical-code decode, LZ77 copies from its own output -
// and 16 files share a wavefront (PNG_LANES; the other lanes idle) so that a few hundred files already spread over all CUs.
// Integer / byte work, HBM-resident: nothing here belongs on MFMA.  Per file the tables (2 x {count[16], symbol[]}) live in LDS,
// the output window is the file's own slice of a global scratch (n x h x (1 + w*bpp) filtered bytes).
//
// Supported: bit depth 8, non-interlaced, colour types 0 (grey), 2 (RGB), 4 (grey + alpha), 6 (RGBA) - what cv2.imwrite produces
// for uint8 arrays.  Anything else (palette, 16-bit, Adam7, a size other than the requested h x w) sets a per-file status and the
// caller decodes that file on the host; a corrupt stream sets an error status (cv2.imread would return None).
#include "common.h"

namespace pvr {

enum { PNG_OK = 0, PNG_UNSUPPORTED = 1, PNG_BAD_SIGNATURE = 2, PNG_TRUNCATED = 3, PNG_BAD_ZLIB_HEADER = 4, PNG_BAD_BLOCK = 5,
       PNG_BAD_CODE = 6, PNG_BAD_DISTANCE = 7, PNG_OVERRUN = 8, PNG_SHORT = 9, PNG_BAD_ADLER = 10, PNG_BAD_FILTER = 11,
       PNG_SIZE_MISMATCH = 12 };

constexpr int PNG_LANES = 16;                     // files per wavefront
constexpr int PNG_MAXL = 288, PNG_MAXD = 30, PNG_MAXBITS = 15;

struct PngTables {                                // canonical Huffman codes: per length the number of codes, symbols in code order
    unsigned short lcount[16], lsym[PNG_MAXL], dcount[16], dsym[PNG_MAXD];
    unsigned char lens[19 + PNG_MAXL + PNG_MAXD + 7];   // code-length code lengths [0, 19), then the literal/length + distance code lengths
};

__constant__ unsigned short png_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ unsigned char png_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ unsigned short png_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ unsigned char png_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ unsigned char png_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct PngReader {                                // bytes of the zlib stream = the payloads of consecutive IDAT chunks
    const unsigned char *f;
    long long pos, end, left;                      // next byte, end of file, bytes left in the current IDAT payload
    unsigned bitbuf;
    int bitcnt, err;
};

__device__ __forceinline__ unsigned png_be32(const unsigned char *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

__device__ __forceinline__ unsigned png_byte(PngReader &r) {
    while (r.left == 0) {                         // payload exhausted: skip this chunk's CRC, the next chunk must be another IDAT
        r.pos += 4;
        if (r.pos + 8 > r.end) { r.err = r.err ? r.err : PNG_TRUNCATED; return 0; }
        const unsigned len = png_be32(r.f + r.pos), type = png_be32(r.f + r.pos + 4);
        r.pos += 8;
        if (type != 0x49444154u || r.pos + (long long)len > r.end) { r.err = r.err ? r.err : PNG_TRUNCATED; r.left = 1ll << 40; return 0; }
        r.left = len;
    }
    if (r.err) return 0;
    --r.left;
    return r.f[r.pos++];
}

__device__ __forceinline__ unsigned png_bits(PngReader &r, int n) {       // n <= 16, LSB first (RFC 1951 3.1.1)
    while (r.bitcnt < n) { r.bitbuf |= png_byte(r) << r.bitcnt; r.bitcnt += 8; }
    const unsigned v = r.bitbuf & ((1u << n) - 1u);
    r.bitbuf >>= n; r.bitcnt -= n;
    return v;
}

// canonical-code decode, one bit at a time: codes of length L occupy [first_L, first_L + count_L) after L bits (RFC 1951 3.2.2)
__device__ __forceinline__ int png_symbol(PngReader &r, const unsigned short *count, const unsigned short *sym) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= PNG_MAXBITS; ++len) {
        code |= (int)png_bits(r, 1);
        const int c = count[len];
        if (code - c < first) return sym[index + (code - first)];
        index += c; first += c;
        first <<= 1; code <<= 1;
    }
    return -1;
}

// lengths[0..n) -> count / symbol tables; returns < 0 for an over-subscribed set, > 0 for an incomplete one, 0 for a complete code
__device__ int png_build(const unsigned char *lengths, int n, unsigned short *count, unsigned short *sym) {
    unsigned short offs[16];
    for (int l = 0; l <= PNG_MAXBITS; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) ++count[lengths[s]];
    if (count[0] == n) return 0;                  // no codes at all: complete, but decoding from it fails
    int left = 1;
    for (int l = 1; l <= PNG_MAXBITS; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int l = 1; l < PNG_MAXBITS; ++l) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < n; ++s)
        if (lengths[s]) sym[offs[lengths[s]]++] = (unsigned short)s;
    return left;
}

// one file: signature + IHDR checks, inflate into raw[0 .. raw_len), Adler-32, unfilter in place.  Returns a PNG_* status.
__device__ int png_one(const unsigned char *f, long long fbytes, int h, int w, unsigned char *raw, PngTables &tb, int *ctype_out) {
    if (fbytes < 8 + 25 + 12) return PNG_TRUNCATED;
    if (png_be32(f) != 0x89504e47u || png_be32(f + 4) != 0x0d0a1a0au) return PNG_BAD_SIGNATURE;
    if (png_be32(f + 8) != 13u || png_be32(f + 12) != 0x49484452u) return PNG_BAD_SIGNATURE;       // IHDR first
    const int fw = (int)png_be32(f + 16), fh = (int)png_be32(f + 20), depth = f[24], ctype = f[25], interlace = f[28];
    if (fw != w || fh != h) return PNG_SIZE_MISMATCH;
    if (depth != 8 || interlace != 0 || f[26] != 0 || f[27] != 0 || !(ctype == 0 || ctype == 2 || ctype == 4 || ctype == 6)) return PNG_UNSUPPORTED;
    *ctype_out = ctype;
    const int bpp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : 4;
    const int rowb = w * bpp, stride = rowb + 1;
    const long long raw_len = (long long)h * stride;
    // first IDAT (ancillary chunks in front of it are skipped)
    PngReader r;
    r.f = f; r.end = fbytes; r.pos = 8 + 25; r.left = 0; r.bitbuf = 0; r.bitcnt = 0; r.err = 0;
    for (;;) {
        if (r.pos + 8 > r.end) return PNG_TRUNCATED;
        const unsigned len = png_be32(f + r.pos), type = png_be32(f + r.pos + 4);
        if (r.pos + 12 + (long long)len > r.end) return PNG_TRUNCATED;
        if (type == 0x49444154u) { r.pos += 8; r.left = len; break; }
        if (type == 0x49454e44u) return PNG_TRUNCATED;       // IEND before any IDAT
        r.pos += 12 + (long long)len;
    }
    // zlib header (RFC 1950): deflate, window <= 32 KB, no preset dictionary, check bits
    const unsigned cmf = png_byte(r), flg = png_byte(r);
    if (r.err) return r.err;
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 32u) || ((cmf << 8) | flg) % 31u) return PNG_BAD_ZLIB_HEADER;

    long long out = 0;
    for (int last = 0; !last;) {
        last = (int)png_bits(r, 1);
        const int type = (int)png_bits(r, 2);
        if (r.err) return r.err;
        if (type == 0) {                          // stored
            r.bitbuf = 0; r.bitcnt = 0;
            unsigned len = png_byte(r); len |= png_byte(r) << 8;
            unsigned nlen = png_byte(r); nlen |= png_byte(r) << 8;
            if (r.err) return r.err;
            if ((len ^ 0xffffu) != nlen) return PNG_BAD_BLOCK;
            if (out + len > raw_len) return PNG_OVERRUN;
            for (unsigned i = 0; i < len; ++i) raw[out++] = (unsigned char)png_byte(r);
            if (r.err) return r.err;
            continue;
        }
        if (type == 3) return PNG_BAD_BLOCK;
        if (type == 1) {                          // fixed codes (RFC 1951 3.2.6)
            for (int s = 0; s < 144; ++s) tb.lens[s] = 8;
            for (int s = 144; s < 256; ++s) tb.lens[s] = 9;
            for (int s = 256; s < 280; ++s) tb.lens[s] = 7;
            for (int s = 280; s < 288; ++s) tb.lens[s] = 8;
            png_build(tb.lens, 288, tb.lcount, tb.lsym);
            for (int s = 0; s < 30; ++s) tb.lens[s] = 5;
            png_build(tb.lens, 30, tb.dcount, tb.dsym);
        } else {                                  // dynamic codes (3.2.7)
            const int nlen = (int)png_bits(r, 5) + 257, ndist = (int)png_bits(r, 5) + 1, ncode = (int)png_bits(r, 4) + 4;
            if (r.err) return r.err;
            if (nlen > 286 || ndist > 30) return PNG_BAD_BLOCK;
            for (int i = 0; i < 19; ++i) tb.lens[i] = 0;
            for (int i = 0; i < ncode; ++i) tb.lens[png_clorder[i]] = (unsigned char)png_bits(r, 3);
            if (png_build(tb.lens, 19, tb.lcount, tb.lsym) != 0) return PNG_BAD_CODE;        // the code-length code must be complete
            int idx = 0;
            while (idx < nlen + ndist) {
                const int s = png_symbol(r, tb.lcount, tb.lsym);
                if (s < 0 || r.err) return r.err ? r.err : PNG_BAD_CODE;
                if (s < 16) { tb.lens[19 + idx++] = (unsigned char)s; continue; }
                int prev = 0, rep;
                if (s == 16) {
                    if (idx == 0) return PNG_BAD_CODE;
                    prev = tb.lens[19 + idx - 1]; rep = 3 + (int)png_bits(r, 2);
                } else if (s == 17) rep = 3 + (int)png_bits(r, 3);
                else rep = 11 + (int)png_bits(r, 7);
                if (idx + rep > nlen + ndist) return PNG_BAD_CODE;
                while (rep--) tb.lens[19 + idx++] = (unsigned char)prev;
            }
            if (r.err) return r.err;
            if (tb.lens[19 + 256] == 0) return PNG_BAD_CODE;                                  // no end-of-block code
            // (the code-length tables in lcount / lsym are dead now; lens[19 ..] holds the nlen + ndist lengths)
            int e = png_build(tb.lens + 19, nlen, tb.lcount, tb.lsym);
            if (e < 0 || (e > 0 && nlen - tb.lcount[0] != 1)) return PNG_BAD_CODE;            // incomplete only if a single code
            e = png_build(tb.lens + 19 + nlen, ndist, tb.dcount, tb.dsym);
            if (e < 0 || (e > 0 && ndist - tb.dcount[0] != 1)) return PNG_BAD_CODE;
        }
        for (;;) {                                // literals and <length, distance> pairs until end-of-block
            int s = png_symbol(r, tb.lcount, tb.lsym);
            if (s < 0 || r.err) return r.err ? r.err : PNG_BAD_CODE;
            if (s < 256) {
                if (out >= raw_len) return PNG_OVERRUN;
                raw[out++] = (unsigned char)s;
                continue;
            }
            if (s == 256) break;
            s -= 257;
            if (s >= 29) return PNG_BAD_CODE;
            const int len = png_lbase[s] + (int)png_bits(r, png_lext[s]);
            const int ds = png_symbol(r, tb.dcount, tb.dsym);
            if (ds < 0 || ds >= 30 || r.err) return r.err ? r.err : PNG_BAD_CODE;
            const long long dist = png_dbase[ds] + (long long)png_bits(r, png_dext[ds]);
            if (r.err) return r.err;
            if (dist > out) return PNG_BAD_DISTANCE;
            if (out + len > raw_len) return PNG_OVERRUN;
            for (int i = 0; i < len; ++i, ++out) raw[out] = raw[out - dist];                  // byte by byte: ranges may overlap
        }
    }
    if (out != raw_len) return PNG_SHORT;
    // Adler-32 of the inflated bytes (RFC 1950), stored big-endian after the last block
    r.bitbuf = 0; r.bitcnt = 0;
    unsigned want = png_byte(r) << 24; want |= png_byte(r) << 16; want |= png_byte(r) << 8; want |= png_byte(r);
    if (r.err) return r.err;
    unsigned a = 1, b = 0;
    // unfilter in place (PNG spec 9.2; bytes left of the first pixel / above the first row count as 0), Adler over the filtered bytes
    for (int y = 0; y < h; ++y) {
        unsigned char *row = raw + (long long)y * stride;
        const unsigned char *up = y ? row - stride : nullptr;
        const int ft = row[0];
        if (ft > 4) return PNG_BAD_FILTER;
        a += ft; b += a;
        for (int x = 1; x <= rowb; ++x) {
            const int v = row[x];
            a += v; b += a;
            const int L = x > bpp ? row[x - bpp] : 0, U = up ? up[x] : 0, UL = (up && x > bpp) ? up[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = L;
            else if (ft == 2) pred = U;
            else if (ft == 3) pred = (L + U) >> 1;
            else if (ft == 4) {
                const int p = L + U - UL, pa = abs(p - L), pb = abs(p - U), pc = abs(p - UL);
                pred = (pa <= pb && pa <= pc) ? L : (pb <= pc ? U : UL);
            }
            row[x] = (unsigned char)(v + pred);
        }
        a %= 65521u; b %= 65521u;                 // deferred modulo: a scanline is below zlib's 5552-byte bound (w <= 1387, checked by the launcher)
    }
    if (((b << 16) | a) != want) return PNG_BAD_ADLER;
    return PNG_OK;
}

__global__ __launch_bounds__(64) void png_inflate_kernel(const unsigned char *__restrict__ files, const long long *__restrict__ offsets, int n,
                                                        int h, int w, unsigned char *__restrict__ scratch, long long raw_stride,
                                                        int *__restrict__ status, unsigned char *__restrict__ ctypes) {
    __shared__ PngTables tb[PNG_LANES];
    const int lane = threadIdx.x;
    if (lane >= PNG_LANES) return;
    const int i = blockIdx.x * PNG_LANES + lane;
    if (i >= n) return;
    int ctype = 2;
    const int st = png_one(files + offsets[i], offsets[i + 1] - offsets[i], h, w, scratch + (long long)i * raw_stride, tb[lane], &ctype);
    status[i] = st;
    ctypes[i] = (unsigned char)ctype;
}

// unfiltered scanlines -> cv2.imread(IMREAD_COLOR) layout: (n, h, w, 3) uint8 in B, G, R order; alpha dropped, grey replicated
__global__ __launch_bounds__(256) void png_pack_kernel(const unsigned char *__restrict__ scratch, long long raw_stride, const int *__restrict__ status,
                                                      const unsigned char *__restrict__ ctypes, int n, int h, int w, unsigned char *__restrict__ out) {
    const long long total = (long long)n * h * w;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int x = (int)(p % w);
        const long long t = p / w;
        const int y = (int)(t % h), i = (int)(t / h);
        unsigned char bgr[3] = {0, 0, 0};
        if (status[i] == PNG_OK) {
            const int ct = ctypes[i], bpp = ct == 0 ? 1 : ct == 2 ? 3 : ct == 4 ? 2 : 4;
            const unsigned char *px = scratch + (long long)i * raw_stride + (long long)y * (w * bpp + 1) + 1 + (long long)x * bpp;
            if (ct == 0 || ct == 4) { bgr[0] = bgr[1] = bgr[2] = px[0]; }
            else { bgr[0] = px[2]; bgr[1] = px[1]; bgr[2] = px[0]; }
        }
        unsigned char *o = out + p * 3;
        o[0] = bgr[0]; o[1] = bgr[1]; o[2] = bgr[2];
    }
}

}  // namespace pvr

extern "C" int64_t pvr_png_scratch_bytes(int32_t n, int32_t h, int32_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return (int64_t)n * ((int64_t)h * (4 * (int64_t)w + 1)) + (int64_t)n;      // filtered scanlines at 4 bytes per pixel + one colour-type byte per file
}

extern "C" pvr_status pvr_png_decode(const uint8_t *files_dev, const int64_t *offsets_dev, int32_t n, int32_t h, int32_t w, uint8_t *out_dev,
                                     uint8_t *scratch_dev, int64_t scratch_bytes, int32_t *status_dev, void *hip_stream) {
    PVR_REQUIRE(files_dev && offsets_dev && out_dev && scratch_dev && status_dev, "pvr_png_decode: null argument");
    PVR_REQUIRE(n > 0 && h > 0 && w > 0 && w <= 1387, "pvr_png_decode: n=%d h=%d w=%d (width limit 1387: one scanline per deferred Adler modulo)", n, h, w);
    PVR_REQUIRE(scratch_bytes >= pvr_png_scratch_bytes(n, h, w), "pvr_png_decode: scratch of %lld bytes, %lld needed", (long long)scratch_bytes,
                (long long)pvr_png_scratch_bytes(n, h, w));
    hipStream_t st = (hipStream_t)hip_stream;
    const long long raw_stride = (long long)h * (4 * (long long)w + 1);
    unsigned char *ctypes = scratch_dev + (long long)n * raw_stride;
    hipLaunchKernelGGL(pvr::png_inflate_kernel, dim3((unsigned)((n + pvr::PNG_LANES - 1) / pvr::PNG_LANES)), dim3(64), 0, st, files_dev,
                       (const long long *)offsets_dev, n, h, w, scratch_dev, raw_stride, status_dev, ctypes);
    PVR_LAUNCH_CHECK();
    const long long total = (long long)n * h * w;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pvr::png_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, scratch_dev, raw_stride, status_dev, ctypes, n, h, w, out_dev);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}
