// PNG source on the GPU: the per-frame files of reference behavioral_cloning/save_embedded_obs.py:50-93 (`cv2.imread` of
// <t>_goal.png and <t>_<s>.png, one zlib stream of 64x64x3 filtered scanlines each) are decoded where they are consumed.
// The host only reads file bytes; one launch inflates n files (RFC 1950 / 1951: stored, fixed and dynamic Huffman blocks,
// IDAT chunk boundaries crossed inside the byte reader), checks Adler-32, undoes the five PNG scanline filters (PNG spec 9.2) and
// a second, coalesced launch writes cv2.imread's layout (n, h, w, 3) uint8 B,G,R straight into HBM for the encoder.
//
// Parallelism is across files, not inside one: a DEFLATE stream is serial by construction (every symbol's position depends on
// all earlier code lengths), so one lane walks one file - bit reader, table-driven code decode, LZ77 copies from its own output -
// and 16 files share a wavefront (PNG_LANES; the other lanes idle) so that a few hundred files already spread over all CUs.
// Integer / byte work, HBM-resident: nothing here belongs on MFMA.  Per file the tables (2 x {count[16], symbol[]}) live in LDS,
// the output window is the file's own slice of a global scratch (n x h x (1 + w*bpp) filtered bytes); a second launch undoes the filters
// with one lane per (file, colour channel) - the only independent chains a filtered image has.
//
// Supported: bit depth 8, non-interlaced, colour types 0 (grey), 2 (RGB), 4 (grey + alpha), 6 (RGBA) - what cv2.imwrite produces
// for uint8 arrays.  Anything else (palette, 16-bit, Adam7, a size other than the requested h x w) sets a per-file status and the
// caller decodes that file on the host; a corrupt stream sets an error status (cv2.imread would return None).
#include "common.h"
#include <atomic>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pvr {

enum { PNG_OK = 0, PNG_UNSUPPORTED = 1, PNG_BAD_SIGNATURE = 2, PNG_TRUNCATED = 3, PNG_BAD_ZLIB_HEADER = 4, PNG_BAD_BLOCK = 5,
       PNG_BAD_CODE = 6, PNG_BAD_DISTANCE = 7, PNG_OVERRUN = 8, PNG_SHORT = 9, PNG_BAD_ADLER = 10, PNG_BAD_FILTER = 11,
       PNG_SIZE_MISMATCH = 12 };

constexpr int PNG_LANES = 16;                     // files per wavefront
constexpr int PNG_MAXL = 288, PNG_MAXD = 30, PNG_MAXBITS = 15;
constexpr int PNG_FASTL = 9, PNG_FASTD = 7;       // bits of the one-read lookup tables (longer codes take the bit-by-bit path)

struct PngTables {                                // canonical Huffman codes: per length the number of codes, symbols in code order,
    unsigned short lcount[16], lsym[PNG_MAXL], dcount[16], dsym[PNG_MAXD];
    unsigned short lfast[1 << PNG_FASTL], dfast[1 << PNG_FASTD];         // next bits -> (symbol << 4) | code length, 0 = not in the table
    unsigned char lens[19 + PNG_MAXL + PNG_MAXD + 7];   // code-length code lengths [0, 19), then the literal/length + distance code lengths
};

// RFC 1951 3.2.5 in closed form (a per-lane index into a __constant__ table is a vector memory load on this hardware):
//   length code s = 0..28:  3..10 | 11,13,15,17 (+1 bit) | 19,23,27,31 (+2) | ... | 131,163,195,227 (+5) | 258
//   distance code d = 0..29: 1..4 | 5,7 (+1 bit) | 9,13 (+2) | ... | 16385,24577 (+13)
__device__ __forceinline__ int png_len_extra(int s) { return (s < 8 || s == 28) ? 0 : (s - 4) >> 2; }
__device__ __forceinline__ int png_len_base(int s) { return s < 8 ? 3 + s : (s == 28 ? 258 : 3 + ((4 + (s & 3)) << ((s - 4) >> 2))); }
__device__ __forceinline__ int png_dist_extra(int d) { return d < 4 ? 0 : (d >> 1) - 1; }
__device__ __forceinline__ int png_dist_base(int d) { return d < 4 ? 1 + d : 1 + ((2 + (d & 1)) << ((d >> 1) - 1)); }
// order in which the code-length code's own lengths are stored (3.2.7), 5 bits each, first entry in the low bits
__device__ __forceinline__ int png_clorder(int i) {
    const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                                  (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    return i < 12 ? (int)((lo >> (5 * i)) & 31u) : (int)((hi >> (5 * (i - 12))) & 31u);
}

// Bytes of the zlib stream = the payloads of consecutive IDAT chunks.  The bit buffer is filled ahead of need (table lookups peek
// PNG_FASTL bits), so running out of IDAT data is not an error by itself: zero bits are supplied and counted (`fake`), and only
// CONSUMING one of them is (PNG_TRUNCATED).
struct PngReader {
    const unsigned char *f;
    long long pos, end, left;                      // next byte, end of file, bytes left in the current IDAT payload
    unsigned long long bitbuf;
    int bitcnt, fake, err;                         // bits held, how many of them (at the top) are padding, first error
};

__device__ __forceinline__ unsigned png_be32(const unsigned char *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

__device__ __forceinline__ void png_fill(PngReader &r) {
    if (r.bitcnt <= 32 && r.left >= 4) {          // the common case: four payload bytes in one (unaligned) load
        unsigned v;
        __builtin_memcpy(&v, r.f + r.pos, 4);
        r.bitbuf |= (unsigned long long)v << r.bitcnt;
        r.bitcnt += 32; r.pos += 4; r.left -= 4;
        return;
    }
    while (r.bitcnt <= 56) {
        unsigned b = 0;
        if (r.fake == 0) {
            while (r.left == 0) {                 // payload exhausted: skip this chunk's CRC; the stream continues only in another IDAT
                if (r.pos + 12 > r.end) { r.left = -1; break; }
                const unsigned len = png_be32(r.f + r.pos + 4), type = png_be32(r.f + r.pos + 8);
                if (type != 0x49444154u || r.pos + 12 + (long long)len > r.end) { r.left = -1; break; }
                r.pos += 12; r.left = len;
            }
        }
        if (r.left > 0) { --r.left; b = r.f[r.pos++]; }
        else r.fake += 8;
        r.bitbuf |= (unsigned long long)b << r.bitcnt;
        r.bitcnt += 8;
    }
}

__device__ __forceinline__ void png_drop(PngReader &r, int n) {
    r.bitbuf >>= n; r.bitcnt -= n;
    if (r.bitcnt < r.fake && !r.err) r.err = PNG_TRUNCATED;            // consumed a padding bit
}

__device__ __forceinline__ unsigned png_bits(PngReader &r, int n) {       // n <= 16, LSB first (RFC 1951 3.1.1)
    if (r.bitcnt < n) png_fill(r);
    const unsigned v = (unsigned)r.bitbuf & ((1u << n) - 1u);
    png_drop(r, n);
    return v;
}

// one symbol: the next FAST bits index a table that resolves every code of <= FAST bits in one LDS read; longer codes are walked
// one bit at a time (codes of length L occupy [first_L, first_L + count_L) after L bits, RFC 1951 3.2.2)
template <int FAST>
__device__ __forceinline__ int png_symbol(PngReader &r, const unsigned short *fast, const unsigned short *count, const unsigned short *sym) {
    if (r.bitcnt < PNG_MAXBITS) png_fill(r);
    const unsigned e = fast[(unsigned)r.bitbuf & ((1u << FAST) - 1u)];
    if (e) { png_drop(r, (int)(e & 15u)); return (int)(e >> 4); }
    int code = 0, first = 0, index = 0;
    unsigned long long bb = r.bitbuf;
    for (int len = 1; len <= PNG_MAXBITS; ++len) {
        code |= (int)(bb & 1u); bb >>= 1;
        const int c = count[len];
        if (code - c < first) { png_drop(r, len); return sym[index + (code - first)]; }
        index += c; first += c;
        first <<= 1; code <<= 1;
    }
    return -1;
}

// lengths[0..n) -> count / symbol (/ lookup) tables; returns < 0 for an over-subscribed set, > 0 for an incomplete one, 0 for a complete code
template <int FAST>
__device__ int png_build(const unsigned char *lengths, int n, unsigned short *count, unsigned short *sym, unsigned short *fast) {
    unsigned short offs[16];
    for (int l = 0; l <= PNG_MAXBITS; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) ++count[lengths[s]];
    if (fast) for (int k = 0; k < (1 << FAST); ++k) fast[k] = 0;
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
    if (fast) {                                   // code values in symbol order per length (3.2.2), bit-reversed: the stream carries codes MSB first
        int code = 0, idx = 0;
        for (int l = 1; l <= FAST; ++l) {
            for (int k = 0; k < count[l]; ++k, ++code, ++idx) {
                unsigned rev = __brev((unsigned)code) >> (32 - l);
                const unsigned short e = (unsigned short)((sym[idx] << 4) | l);
                for (; rev < (1u << FAST); rev += 1u << l) fast[rev] = e;
            }
            code <<= 1;
        }
    }
    return left;
}

// one file: signature + IHDR checks, inflate into raw[0 .. raw_len) with the Adler-32 of the produced bytes.  Returns a PNG_* status.
__device__ int png_one(const unsigned char *f, long long fbytes, int h, int w, unsigned char *raw, PngTables &tb, int *ctype_out) {
    if (fbytes < 8 + 25 + 12) return PNG_TRUNCATED;
    if (png_be32(f) != 0x89504e47u || png_be32(f + 4) != 0x0d0a1a0au) return PNG_BAD_SIGNATURE;
    if (png_be32(f + 8) != 13u || png_be32(f + 12) != 0x49484452u) return PNG_BAD_SIGNATURE;       // IHDR first
    const int fw = (int)png_be32(f + 16), fh = (int)png_be32(f + 20), depth = f[24], ctype = f[25], interlace = f[28];
    if (fw != w || fh != h) return PNG_SIZE_MISMATCH;
    if (depth != 8 || interlace != 0 || f[26] != 0 || f[27] != 0 || !(ctype == 0 || ctype == 2 || ctype == 4 || ctype == 6)) return PNG_UNSUPPORTED;
    *ctype_out = ctype;
    const int bpp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : 4;
    const long long raw_len = (long long)h * (w * bpp + 1);
    // first IDAT (ancillary chunks in front of it are skipped); the reader then sits on the last byte before the payload's CRC rule:
    // pos = start of a chunk's CRC whenever left == 0
    PngReader r;
    r.f = f; r.end = fbytes; r.pos = 8 + 25; r.left = 0; r.bitbuf = 0; r.bitcnt = 0; r.fake = 0; r.err = 0;
    for (;;) {
        if (r.pos + 8 > r.end) return PNG_TRUNCATED;
        const unsigned len = png_be32(f + r.pos), type = png_be32(f + r.pos + 4);
        if (r.pos + 12 + (long long)len > r.end) return PNG_TRUNCATED;
        if (type == 0x49444154u) { r.pos += 8; r.left = len; break; }
        if (type == 0x49454e44u) return PNG_TRUNCATED;       // IEND before any IDAT
        r.pos += 12 + (long long)len;
    }
    // zlib header (RFC 1950): deflate, window <= 32 KB, no preset dictionary, check bits
    const unsigned cmf = png_bits(r, 8), flg = png_bits(r, 8);
    if (r.err) return r.err;
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 32u) || ((cmf << 8) | flg) % 31u) return PNG_BAD_ZLIB_HEADER;

    long long out = 0;
    unsigned ad_a = 1, ad_b = 0;                  // Adler-32 of the produced bytes, modulo deferred (zlib's bound: 5552 bytes)
    int ad_n = 0;
#define PNG_EMIT(v_) { const unsigned vv_ = (v_); raw[out++] = (unsigned char)vv_; ad_a += vv_; ad_b += ad_a; \
                       if (++ad_n == 5552) { ad_a %= 65521u; ad_b %= 65521u; ad_n = 0; } }
    for (int last = 0; !last;) {
        last = (int)png_bits(r, 1);
        const int type = (int)png_bits(r, 2);
        if (r.err) return r.err;
        if (type == 0) {                          // stored: to the next byte boundary, LEN, ~LEN, LEN bytes
            png_drop(r, r.bitcnt & 7);
            const unsigned len = png_bits(r, 16), nlen = png_bits(r, 16);
            if (r.err) return r.err;
            if ((len ^ 0xffffu) != nlen) return PNG_BAD_BLOCK;
            if (out + len > raw_len) return PNG_OVERRUN;
            for (unsigned i = 0; i < len; ++i) PNG_EMIT(png_bits(r, 8));
            if (r.err) return r.err;
            continue;
        }
        if (type == 3) return PNG_BAD_BLOCK;
        if (type == 1) {                          // fixed codes (RFC 1951 3.2.6)
            for (int s = 0; s < 144; ++s) tb.lens[s] = 8;
            for (int s = 144; s < 256; ++s) tb.lens[s] = 9;
            for (int s = 256; s < 280; ++s) tb.lens[s] = 7;
            for (int s = 280; s < 288; ++s) tb.lens[s] = 8;
            png_build<PNG_FASTL>(tb.lens, 288, tb.lcount, tb.lsym, tb.lfast);
            for (int s = 0; s < 30; ++s) tb.lens[s] = 5;
            png_build<PNG_FASTD>(tb.lens, 30, tb.dcount, tb.dsym, tb.dfast);
        } else {                                  // dynamic codes (3.2.7)
            const int nlen = (int)png_bits(r, 5) + 257, ndist = (int)png_bits(r, 5) + 1, ncode = (int)png_bits(r, 4) + 4;
            if (r.err) return r.err;
            if (nlen > 286 || ndist > 30) return PNG_BAD_BLOCK;
            for (int i = 0; i < 19; ++i) tb.lens[i] = 0;
            for (int i = 0; i < ncode; ++i) tb.lens[png_clorder(i)] = (unsigned char)png_bits(r, 3);
            if (png_build<PNG_FASTD>(tb.lens, 19, tb.dcount, tb.dsym, tb.dfast) != 0) return PNG_BAD_CODE;   // the code-length code must be complete
            int idx = 0;
            while (idx < nlen + ndist) {
                const int s = png_symbol<PNG_FASTD>(r, tb.dfast, tb.dcount, tb.dsym);
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
            int e = png_build<PNG_FASTL>(tb.lens + 19, nlen, tb.lcount, tb.lsym, tb.lfast);
            if (e < 0 || (e > 0 && nlen - tb.lcount[0] != 1)) return PNG_BAD_CODE;            // incomplete only if a single code
            e = png_build<PNG_FASTD>(tb.lens + 19 + nlen, ndist, tb.dcount, tb.dsym, tb.dfast);
            if (e < 0 || (e > 0 && ndist - tb.dcount[0] != 1)) return PNG_BAD_CODE;
        }
        for (;;) {                                // literals and <length, distance> pairs until end-of-block
            int s = png_symbol<PNG_FASTL>(r, tb.lfast, tb.lcount, tb.lsym);
            if (s < 0 || r.err) return r.err ? r.err : PNG_BAD_CODE;
            if (s < 256) {
                if (out >= raw_len) return PNG_OVERRUN;
                PNG_EMIT((unsigned)s);
                continue;
            }
            if (s == 256) break;
            s -= 257;
            if (s >= 29) return PNG_BAD_CODE;
            const int len = png_len_base(s) + (int)png_bits(r, png_len_extra(s));
            const int ds = png_symbol<PNG_FASTD>(r, tb.dfast, tb.dcount, tb.dsym);
            if (ds < 0 || ds >= 30 || r.err) return r.err ? r.err : PNG_BAD_CODE;
            const long long dist = png_dist_base(ds) + (long long)png_bits(r, png_dist_extra(ds));
            if (r.err) return r.err;
            if (dist > out) return PNG_BAD_DISTANCE;
            if (out + len > raw_len) return PNG_OVERRUN;
            // the copy runs from registers: ONE load of the 8 bytes at the source (the scratch has 8 bytes of slack behind the last file)
            // per 8 output bytes; a source closer than 8 bytes is a repeating pattern of `dist` bytes, loaded once and cycled
            if (dist >= 8) {
                for (int i = 0; i < len; i += 8) {
                    unsigned long long q;
                    __builtin_memcpy(&q, raw + out - dist, 8);
                    const int m = len - i < 8 ? len - i : 8;
                    for (int k = 0; k < m; ++k) { PNG_EMIT((unsigned)(q & 255u)); q >>= 8; }
                }
            } else {
                unsigned long long q;
                __builtin_memcpy(&q, raw + out - dist, 8);
                const int d = (int)dist;
                for (int i = 0, j = 0; i < len; ++i) {
                    PNG_EMIT((unsigned)((q >> (8 * j)) & 255u));
                    j = j + 1 == d ? 0 : j + 1;
                }
            }
        }
    }
#undef PNG_EMIT
    if (out != raw_len) return PNG_SHORT;
    // Adler-32 of the inflated bytes (RFC 1950), stored big-endian after the last block's byte boundary
    png_drop(r, r.bitcnt & 7);
    unsigned want = png_bits(r, 8) << 24; want |= png_bits(r, 8) << 16; want |= png_bits(r, 8) << 8; want |= png_bits(r, 8);
    if (r.err) return r.err;
    ad_a %= 65521u; ad_b %= 65521u;
    if (((ad_b << 16) | ad_a) != want) return PNG_BAD_ADLER;
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

// Scanline filters undone (PNG spec 9.2; bytes left of the first pixel / above the first row count as 0) and cv2.imread(IMREAD_COLOR)'s
// layout written in one pass: (n, h, w, 3) uint8 in B, G, R order; alpha dropped, grey replicated.  A filtered byte depends on the
// reconstructed byte bpp positions to its left and on the row above, so the dependency chains of a file are its colour channels: one
// lane per (file, channel), rows in order, 8 pixels per step - the 8 filtered bytes and the 8 bytes above are loaded first (they do not
// depend on this step), the chain itself runs in registers (left and upper-left are carried), then 8 stores.
__global__ __launch_bounds__(64) void png_unfilter_kernel(const unsigned char *__restrict__ scratch, long long raw_stride, int *__restrict__ status,
                                                         const unsigned char *__restrict__ ctypes, int n, int h, int w, unsigned char *out) {
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
    const int i = (int)(tid / 3), c = (int)(tid % 3);
    if (i >= n) return;
    unsigned char *o = out + (long long)i * h * w * 3;
    if (status[i] != PNG_OK) {                    // (the caller replaces or rejects this file; keep the output defined)
        for (long long p = c; p < (long long)h * w * 3; p += 3) o[p] = 0;
        return;
    }
    const int ct = ctypes[i], bpp = ct == 0 ? 1 : ct == 2 ? 3 : ct == 4 ? 2 : 4;
    const bool grey = ct == 0 || ct == 4;
    if (grey && c != 0) return;                   // one chain feeds all three outputs
    const int oc = grey ? 0 : 2 - c;              // output channel of this chain (file order R,G,B -> B,G,R)
    const unsigned char *raw = scratch + (long long)i * raw_stride;
    const int stride = w * bpp + 1;
    bool bad = false;
    for (int y = 0; y < h; ++y) {
        const unsigned char *row = raw + (long long)y * stride + 1 + c;
        const int ft = row[-1 - c];
        bad |= ft > 4;
        unsigned char *orow = o + (long long)y * w * 3, *oup = orow - (long long)w * 3;
        int left = 0, ul = 0;
        for (int x0 = 0; x0 < w; x0 += 8) {
            int v[8], up[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = x0 + k < w ? x0 + k : w - 1;
                v[k] = row[(long long)x * bpp];
                up[k] = y ? oup[x * 3 + oc] : 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = left + up[k] - ul, pa = abs(p - left), pb = abs(p - up[k]), pc = abs(p - ul);
                const int paeth = (pa <= pb && pa <= pc) ? left : (pb <= pc ? up[k] : ul);
                const int pred = ft == 1 ? left : ft == 2 ? up[k] : ft == 3 ? ((left + up[k]) >> 1) : ft == 4 ? paeth : 0;
                ul = up[k];
                left = (v[k] + pred) & 255;
                v[k] = left;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (x0 + k < w) {
                    if (grey) { orow[(x0 + k) * 3] = (unsigned char)v[k]; orow[(x0 + k) * 3 + 1] = (unsigned char)v[k]; orow[(x0 + k) * 3 + 2] = (unsigned char)v[k]; }
                    else orow[(x0 + k) * 3 + oc] = (unsigned char)v[k];
                }
            }
        }
    }
    if (bad && c == 0) status[i] = PNG_BAD_FILTER;      // (every chain of the file saw the same filter bytes: one writer)
}

}  // namespace pvr

extern "C" int64_t pvr_png_scratch_bytes(int32_t n, int32_t h, int32_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return (int64_t)n * ((int64_t)h * (4 * (int64_t)w + 1)) + (int64_t)n + 16;  // filtered scanlines at 4 bytes per pixel, slack, one colour-type byte per file
}

extern "C" pvr_status pvr_png_decode(const uint8_t *files_dev, const int64_t *offsets_dev, int32_t n, int32_t h, int32_t w, uint8_t *out_dev,
                                     uint8_t *scratch_dev, int64_t scratch_bytes, int32_t *status_dev, void *hip_stream) {
    PVR_REQUIRE(files_dev && offsets_dev && out_dev && scratch_dev && status_dev, "pvr_png_decode: null argument");
    pvr::TraceScope trace("pvr_png_decode");
    PVR_REQUIRE(n > 0 && h > 0 && w > 0 && (int64_t)h * w <= (1 << 26), "pvr_png_decode: n=%d h=%d w=%d", n, h, w);
    PVR_REQUIRE(scratch_bytes >= pvr_png_scratch_bytes(n, h, w), "pvr_png_decode: scratch of %lld bytes, %lld needed", (long long)scratch_bytes,
                (long long)pvr_png_scratch_bytes(n, h, w));
    hipStream_t st = (hipStream_t)hip_stream;
    const long long raw_stride = (long long)h * (4 * (long long)w + 1);
    unsigned char *ctypes = scratch_dev + (long long)n * raw_stride + 16;
    hipLaunchKernelGGL(pvr::png_inflate_kernel, dim3((unsigned)((n + pvr::PNG_LANES - 1) / pvr::PNG_LANES)), dim3(64), 0, st, files_dev,
                       (const long long *)offsets_dev, n, h, w, scratch_dev, raw_stride, status_dev, ctypes);
    PVR_LAUNCH_CHECK();
    const long long lanes = 3ll * n;
    hipLaunchKernelGGL(pvr::png_unfilter_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, scratch_dev, raw_stride, status_dev, ctypes, n, h, w, out_dev);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// ---- host side: file bytes -> one staging buffer -----------------------------------------------------------------------------------
// The reference opens one file per frame (save_embedded_obs.py:71).  At 64x64 a file is ~8 KB and the cost is the system calls, so the
// files are read by a few native threads (Python threads hold the GIL through open / read / close: ~25 k files/s measured; worker
// processes would have to ship the bytes back through pipes).  Two passes: sizes (stat), then reads into place.
namespace pvr {
template <class F>
static void png_parallel_for(int n, int threads, F f) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n > 0 ? n : 1;
    std::atomic<int> next(0);
    auto work = [&] { for (int i = next.fetch_add(16); i < n; i = next.fetch_add(16)) for (int k = i; k < n && k < i + 16; ++k) f(k); };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}
}  // namespace pvr

extern "C" pvr_status pvr_file_sizes(const char *const *paths, int32_t n, int64_t *sizes, int32_t threads) {
    PVR_REQUIRE(paths && sizes && n >= 0, "pvr_file_sizes: null argument");
    std::atomic<int> bad(-1);
    pvr::png_parallel_for(n, threads, [&](int i) {
        struct stat sb;
        if (stat(paths[i], &sb) != 0 || !S_ISREG(sb.st_mode)) { sizes[i] = -1; bad.store(i); }
        else sizes[i] = (int64_t)sb.st_size;
    });
    if (bad.load() >= 0) { pvr::set_error("pvr_file_sizes: cannot stat %s", paths[bad.load()]); return PVR_ERR_INVALID; }
    return PVR_OK;
}

// file i -> dst[offsets[i] .. offsets[i+1]) (offsets from the sizes above; a file that changed size in between is an error)
extern "C" pvr_status pvr_read_files(const char *const *paths, int32_t n, uint8_t *dst, const int64_t *offsets, int32_t threads) {
    PVR_REQUIRE(paths && dst && offsets && n >= 0, "pvr_read_files: null argument");
    std::atomic<int> bad(-1);
    pvr::png_parallel_for(n, threads, [&](int i) {
        const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
        if (fd < 0) { bad.store(i); return; }
        int64_t want = offsets[i + 1] - offsets[i], got = 0;
        while (got < want) {
            const ssize_t r = read(fd, dst + offsets[i] + got, (size_t)(want - got));
            if (r <= 0) break;
            got += r;
        }
        close(fd);
        if (got != want) bad.store(i);
    });
    if (bad.load() >= 0) { pvr::set_error("pvr_read_files: cannot read %s", paths[bad.load()]); return PVR_ERR_INVALID; }
    return PVR_OK;
}
