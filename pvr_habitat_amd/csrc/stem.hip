// K3/K4/K8: ResNet50 stem conv1 7x7/2 + folded BN + ReLU on MFMA, maxpool 3x3/2,
// global average pool, and the C-major flatten of the compression heads.
//
// Replaces torchvision ResNet.conv1/bn1/relu/maxpool/avgpool reached from reference
// src/embeddings.py:118-120 and src/vision_models/moco.py:11-12.
//
// Stem as an implicit GEMM with NO im2col and NO bounds checks: the preprocess kernel writes a
// zero-bordered (230 x 232 x 4) image whose 4th channel is a validity flag, so
//   * K = 7 rows x 8 taps x 4 channels = 224 = 7 MFMA k-steps of 32 (tap 7 of each row has zero weight)
//   * lane (px = lane&15, g = lane>>4) needs taps 2g,2g+1 of row s for output pixel px:
//     image[2*ho+s][2*(wo0+px)+2g .. +1][0..3] = ONE aligned 16-byte load
//   * Normalize((x/255-mean)/std) lives in the weights; the validity channel carries -sum(w*mean/std)
//     per tap, so zero padding in the NORMALISED domain (what the reference pads) stays exact at the border.
// Weights (64 x 224, 28 KB) stay in registers: 28 fragments per wave, reused over 7 pixel tiles.
#include "common.h"

namespace pvr {

constexpr int STEM_K = 224;     // 7 * 8 * 4
constexpr int STEM_CO = 64;
constexpr int STEM_IPB = 4;     // images per block of the fused stem+pool kernel

template <bool F16>
__global__ __launch_bounds__(256) void stem_kernel(const u16 *__restrict__ img, const u16 *__restrict__ wgt,
                                                   const float *__restrict__ bias, u16 *__restrict__ out,
                                                   int crop) {
    typedef typename HT<F16>::V8 V8;
    const int PW = crop + 8, PH = crop + 6, OW = crop / 2;      // 232, 230, 112
    const int n = blockIdx.y;
    const int ho0 = blockIdx.x * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = lane & 15, g = lane >> 4;

    // A operand = weights: lane holds W[co = i*16 + px][k = s*32 + g*8 .. +7]
    V8 wf[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 7; ++s)
            wf[i][s] = *reinterpret_cast<const V8 *>(wgt + (size_t)(i * 16 + px) * STEM_K + s * 32 + g * 8);
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(bias + i * 16 + g * 4);

    const int tiles_per_row = OW / 16;                           // 7
    const int ntile = 4 * tiles_per_row;                         // 28 tiles per block
    const u16 *imgn = img + (size_t)n * PH * PW * 4;
    for (int t = wave; t < ntile; t += 4) {
        const int ho = ho0 + t / tiles_per_row;
        const int wo0 = (t % tiles_per_row) * 16;
        if (ho >= OW) break;
        // B operand = pixels: 7 aligned 16-B loads (2 pixels x 4 channels)
        V8 xf[7];
        const u16 *base = imgn + ((size_t)(2 * ho) * PW + 2 * (wo0 + px) + 2 * g) * 4;
#pragma unroll
        for (int s = 0; s < 7; ++s) xf[s] = *reinterpret_cast<const V8 *>(base + (size_t)s * PW * 4);
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma16<F16>(wf[i][s], xf[s], acc[i]);
        // D: row (cout in tile) = 4*g + reg, col = pixel px  ->  4 consecutive channels per lane
        u16 *o = out + (((size_t)n * OW + ho) * OW + wo0 + px) * STEM_CO + g * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2 r = {pack2_h<F16>(fmaxf(acc[i][0] + bv[i].x, 0.f), fmaxf(acc[i][1] + bv[i].y, 0.f)), pack2_h<F16>(fmaxf(acc[i][2] + bv[i].z, 0.f), fmaxf(acc[i][3] + bv[i].w, 0.f))};   // (one v_cvt_pk per dword)
            *reinterpret_cast<uint2 *>(o + i * 16) = r;
        }
    }
}

// Fused conv1 + bn1 + relu + maxpool(3x3/2, pad 1): a block produces 2 pooled rows (x 56 cols x 64 ch) of one image.
// It computes the 5 conv rows they need (25 % recompute of a 3 %-of-FLOPs layer) into an LDS tile
// [5][112][64] 16-bit (70 KB, 16-byte chunks XOR-swizzled with the column), then pools from LDS and writes
// 16-byte coalesced NHWC rows.  Replaces stem_kernel + maxpool_kernel on the encoder path: the 1.6 MB/frame
// conv1 activation never goes to HBM (write 411 MB + read 411 MB per 256 frames saved).
template <bool F16>
__global__ __launch_bounds__(256) void stem_pool_kernel(const u16 *__restrict__ img, const u16 *__restrict__ wgt,
                                                        const float *__restrict__ bias, u16 *__restrict__ out, int nimg, int ipb) {
    typedef typename HT<F16>::V8 V8;
    constexpr int PW = 232, PH = 230, OW = 112, PO = 56;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [5][112] pixels x 128 B
    const int pr0 = blockIdx.x * 2;                                  // pooled rows pr0, pr0+1
    const int cr0 = 2 * pr0 - 1;                                     // first conv row held (may be -1: above the image)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 15, g = lane >> 4;
    V8 wf[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 7; ++s)
            wf[i][s] = *reinterpret_cast<const V8 *>(wgt + (size_t)(i * 16 + px) * STEM_K + s * 32 + g * 8);
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(bias + i * 16 + g * 4);
    // the 28 weight fragments (112 VGPRs) are loaded once and reused over the ipb images of this row pair
    for (int n = blockIdx.y * ipb; n < nimg && n < (int)(blockIdx.y + 1) * ipb; ++n) {
    const u16 *imgn = img + (size_t)n * PH * PW * 4;
    // software pipeline: the 7 pixel fragments of this wave's NEXT tile are in flight while the current tile's 28
    // MFMAs run (the kernel is latency-bound: 9 dependent load->MFMA rounds per wave otherwise)
    auto tile_base = [&](int t) -> const u16 * {
        const int lr = t / 7, ho = cr0 + lr, wo0 = (t % 7) * 16;
        const int hc = ho < 0 ? 0 : (ho >= OW ? OW - 1 : ho);      // clamped: rows outside the image are loaded but unused
        return imgn + ((size_t)(2 * hc) * PW + 2 * (wo0 + px) + 2 * g) * 4;
    };
    V8 xn[7];
    {
        const u16 *b0 = tile_base(wave);
#pragma unroll
        for (int s = 0; s < 7; ++s) xn[s] = *reinterpret_cast<const V8 *>(b0 + (size_t)s * PW * 4);
    }
    for (int t = wave; t < 35; t += 4) {                             // 5 rows x 7 tiles of 16 columns
        const int lr = t / 7, ho = cr0 + lr, wo0 = (t % 7) * 16;
        V8 xf[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) xf[s] = xn[s];
        if (t + 4 < 35) {
            const u16 *b1 = tile_base(t + 4);
#pragma unroll
            for (int s = 0; s < 7; ++s) xn[s] = *reinterpret_cast<const V8 *>(b1 + (size_t)s * PW * 4);
        }
        if (ho < 0 || ho >= OW) continue;                            // rows outside the image are never pooled
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma16<F16>(wf[i][s], xf[s], acc[i]);
        const int col = wo0 + px;
        char *prow = smem + ((size_t)lr * OW + col) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2 r = {pack2_h<F16>(fmaxf(acc[i][0] + bv[i].x, 0.f), fmaxf(acc[i][1] + bv[i].y, 0.f)), pack2_h<F16>(fmaxf(acc[i][2] + bv[i].z, 0.f), fmaxf(acc[i][3] + bv[i].w, 0.f))};   // (one v_cvt_pk per dword)
            const int c16 = i * 2 + (g >> 1);                        // 16-byte chunk of channels 16i+4g .. +3
            *reinterpret_cast<uint2 *>(prow + ((c16 ^ (col & 7)) << 4) + (g & 1) * 8) = r;
        }
    }
    __syncthreads();
    // pooling: 2 rows x 56 cols x 8 channel-chunks = 896 outputs of 16 B
    for (int o = tid; o < 2 * PO * 8; o += 256) {
        const int k = o & 7, pc = (o >> 3) % PO, pr = o / (8 * PO);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = 0.f;                      // inputs are post-ReLU (>= 0): 0 is the identity
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int lr = 2 * pr + dy, ho = cr0 + lr;
            if (ho < 0 || ho >= OW) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int col = 2 * pc - 1 + dx;
                if (col < 0 || col >= OW) continue;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + ((size_t)lr * OW + col) * 128 + ((k ^ (col & 7)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], from_h<F16>((u16)(v[e] & 0xffffu)));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], from_h<F16>((u16)(v[e] >> 16)));
                }
            }
        }
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (unsigned)to_h<F16>(m[2 * e]) | ((unsigned)to_h<F16>(m[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4 *>(out + (((size_t)n * PO + pr0 + pr) * PO + pc) * STEM_CO + k * 8) = r;
    }
    __syncthreads();                                                 // LDS tile is rewritten for the next image
    }
}

// Round 3 form of the fused stem (stem_pool_kernel above stays as the A/B, PVR_STEM_LDS=0).  rocprofv3 / SQ counters of the round-2 kernel:
// 199 us per 256 frames, 1.6 TB/s, MFMA 27 % busy - a latency chain: each wave walked 9 tiles, every tile waited for 7 global 16-byte
// loads (a 27.8 KB working set per image and block, re-read 8.8 x through L2) with one tile of prefetch and two waves per SIMD.
// Here the 15 image rows a block needs per image are CONTIGUOUS in the zero-bordered image (rows 2*cr0 .. 2*cr0 + 14, 1856 B each), so
// they are copied to LDS once by LDS-DMA (32 wave instructions of 1 KB, double-buffered across the block's images: the rows of image
// n + 2 are requested when image n's MFMA phase has ended, i.e. a full image period before they are read) and the MFMA fragments
// become ds_read_b128 (lanes px + g share / neighbour 16-byte pieces: conflict-free).  512 threads: seven waves own one 16-column tile
// each for the five conv rows, pooling runs over all 512 threads.  LDS: conv tile 70 KB + 2 x 32 KB of rows = 134 KB, one block per CU.  Same fragments, same MFMA order, same
// rounding points as stem_kernel + maxpool_kernel: bit-identical (test_fused_stem_pool_is_bit_identical_to_stem_then_maxpool).
// Rows above the image (cr0 = -1 for the first row pair) have a negative source offset: the buffer range check returns zeros; the conv
// rows they would feed are outside the image and never pooled, as before.
#ifdef STEM_STAMP
__device__ long long stem_stamps[8][64][6];
#define ST_T(i_) { if (blockIdx.x == 13 && blockIdx.y == 1 && lane == 0 && n - n0 < 64) stem_stamps[wave][n - n0][i_] = __builtin_amdgcn_s_memtime(); }
#else
#define ST_T(i_)
#endif
// U8 (round 3, last build): the kernel reads the uint8 frames itself - no preprocess launch, no padded 16-bit image in HBM (110 MB
// written and 234 MB read per 256 frames).  Valid when the reference's Resize is the identity (frame edge == resize) - the bench
// configuration: the crop window's rows, 672 B each, are DMA'd RAW into one of three small LDS buffers three images ahead and
// converted (x - 128 exactly, 4th channel 1, zero border: what preprocess_kernel writes) into the two row buffers by all 512 threads
// at the start of the previous image's pooling phase.  u8s: the frames, geometry in U8Geo.
struct U8Geo { const uint8_t *src; int pitch, img_bytes, off0; };     // bytes per source row, bytes per frame, byte offset of the crop's first pixel
// layer1.0.conv1 inside the fused stem (stem_pool_reg_kernel): w = the 64 x 64 weights as eight MFMA A fragments (launch_stem_c1_pack), b = bias, t1 = output
// (n, 56, 56, 64) NHWC or in the blocked layout of the wave-form tails (blk)
struct StemC1 { const u16 *w = nullptr; const float *b = nullptr; u16 *t1 = nullptr; int blk = 0; };
template <bool F16, bool U8 = false>
__global__ __launch_bounds__(512, 1) void stem_pool_lds_kernel(const u16 *__restrict__ img, const u16 *__restrict__ wgt,
                                                               const float *__restrict__ bias, u16 *__restrict__ out, int nimg, int ipb, U8Geo u8g = U8Geo{}) {
    typedef typename HT<F16>::V8 V8;
    constexpr int PW = 232, PH = 230, OW = 112, PO = 56;
    constexpr int ROWB = PW * 8;                                     // bytes of one image row (4 channels x 16 bit)
    constexpr int CT = 5 * OW * 128;                                 // conv tile [5][112] pixels x 128 B
    constexpr int INB = U8 ? 28672 : 32768;                          // one input-row buffer (15 rows = 27 840 B, copied as 32 x 1 KB)
    constexpr int RAWROW = 224 * 3, RAWB = 10240, RAW0 = CT + 2 * INB; // U8: raw crop rows [15][672] uint8 (10 x 1 KB of DMA), three buffers
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [conv tile | rows buffer 0 | rows buffer 1 (| raw 0 | raw 1 | raw 2)]
    const int pr0 = blockIdx.x * 2;                                  // pooled rows pr0, pr0+1
    const int cr0 = 2 * pr0 - 1;                                     // first conv row held (may be -1: above the image)
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    V8 wf[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 7; ++s)
            wf[i][s] = *reinterpret_cast<const V8 *>(wgt + (size_t)(i * 16 + px) * STEM_K + s * 32 + g * 8);
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(bias + i * 16 + g * 4);
    const int n0 = blockIdx.y * ipb, n1 = n0 + ipb < nimg ? n0 + ipb : nimg;
    // rows of image n_ -> buffer b_: wave w issues instructions 4 w .. 4 w + 3 of the flat 32 KB copy
    auto stage_rows = [&](int n_, int b_) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(img) + (size_t)n_ * PH * PW * 4, 0, (unsigned)(PH * ROWB), 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int flat = (wave * 4 + k) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + CT + b_ * INB + flat), 16,
                                                     2 * cr0 * ROWB + flat + lane * 16, 0, 0, 0);
        }
    };
    // U8: raw rows of image n_ -> raw buffer (n_ - n0) % 3: 630 sixteen-byte chunks [row r][42], instruction i = chunks 64 i .. 64 i + 63
    auto stage_raw = [&](int n_) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(u8g.src) + (size_t)n_ * u8g.img_bytes, 0, (unsigned)u8g.img_bytes, 0x00020000);
        char *rb = smem + RAW0 + ((n_ - n0) % 3) * RAWB;
        for (int i = wave; i < 10; i += 8) {
            const int q = i * 64 + lane, r = q / 42, c = q % 42;
            const int yi = 2 * cr0 - 3 + r;                               // crop row of padded row 2 cr0 + r
            const bool ok = q < 630 && yi >= 0 && yi < 224;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(rb + i * 1024), 16,
                                                     ok ? u8g.off0 + yi * u8g.pitch + c * 16 : 0x7ffffff0, 0, 0, 0);
        }
    };
    // U8: raw buffer of image n_ -> rows buffer zb_ (what preprocess_kernel would have written for these 15 padded rows)
    auto convert_rows = [&](int n_, int zb_) {
        const char *rb = smem + RAW0 + ((n_ - n0) % 3) * RAWB;
        char *zb = smem + CT + zb_ * INB;
        // groups of four crop pixels = three aligned dwords of raw bytes -> four padded pixels of 8 bytes (v_cvt_f32_ubyte0..3: one
        // instruction per byte; byte-wise LDS reads made the first build of this path as slow as the preprocess launch it replaces)
        for (int i = tid; i < 15 * 56; i += 512) {
            const int r = i / 56, k = i - r * 56, yi = 2 * cr0 - 3 + r;
            typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
            u32x3 w = u32x3{0u, 0u, 0u};
            const bool ok = yi >= 0 && yi < 224;
            if (ok) w = *reinterpret_cast<const u32x3 *>(rb + r * RAWROW + k * 12);
            const float one = ok ? 1.0f : 0.f, off = ok ? 128.f : 0.f;
            float f[12];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                f[4 * d] = (float)(w[d] & 0xffu); f[4 * d + 1] = (float)((w[d] >> 8) & 0xffu);
                f[4 * d + 2] = (float)((w[d] >> 16) & 0xffu); f[4 * d + 3] = (float)(w[d] >> 24);
            }
            char *dst = zb + r * ROWB + (4 * k + 3) * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 o = {pack2_h<F16>(f[3 * q] - off, f[3 * q + 1] - off), pack2_h<F16>(f[3 * q + 2] - off, one)};
                *reinterpret_cast<uint2 *>(dst + q * 8) = o;
            }
        }
        if (tid < 15 * 8) {                                           // the zero border: padded columns 0..2 and 227..231 of every row
            const int r = tid >> 3, e = tid & 7, pc = e < 3 ? e : 224 + e;
            *reinterpret_cast<ushort4 *>(zb + r * ROWB + pc * 8) = make_ushort4(0, 0, 0, 0);
        }
    };
    if constexpr (U8) {
        for (int k = 0; k < 3; ++k) if (n0 + k < n1) stage_raw(n0 + k);
    } else {
        if (n0 < n1) stage_rows(n0, 0);
        if (n0 + 1 < n1) stage_rows(n0 + 1, 1);
    }
    // LDS-DMA data may be read one barrier after the barrier that follows the s_waitcnt.  Per image that costs nothing extra here: the
    // rows of image n + 1 are waited for at the END of image n's MFMA phase (they were requested a whole image earlier), in front of
    // the barrier that phase ends with anyway, and the barrier at the top of the next image - needed so that nobody rewrites the conv
    // tile while another wave still pools from it - is the second one.  (Until round 3's last build the wait and two extra barriers sat
    // at the top of every image: s_memtime stamps, scripts/stem_stamps.hip, priced them at 1400 of the 11 000 cycles per image.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (U8) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (n0 < n1) convert_rows(n0, 0);
    }
    for (int n = n0; n < n1; ++n) {
        const int b = (n - n0) & 1;
        ST_T(0);
        if constexpr (U8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // convert_rows' LDS stores (previous pooling phase / prologue)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ST_T(1);
        ST_T(2);
        const char *rows = smem + CT + b * INB;
        // Wave w < 7 owns column tile w (16 conv columns) for all five conv rows: lr is a compile-time index, so every fragment address
        // is one base register + an immediate and the two fragment sets alternate without register copies (the round-2 loop spent
        // ~30 VALU instructions per tile on t / 7, t % 7 and v_mov: the kernel is VALU-bound, 78 M wave instructions per launch against
        // 7 M MFMAs - profiles/r03_sq_counters_conv.txt).  Lane (px, g), filter row s: pixels 2 (wo0 + px) + 2 g, + 1 of LDS row 2 lr + s.
        if (wave < 7) {
            const int wo0 = wave * 16, col = wo0 + px;
            const char *fb = rows + (2 * (wo0 + px) + 2 * g) * 8;
            V8 xa[7], xb[7];
#pragma unroll
            for (int s = 0; s < 7; ++s) xa[s] = *reinterpret_cast<const V8 *>(fb + s * ROWB);
#pragma unroll
            for (int lr = 0; lr < 5; ++lr) {
                V8 (&xc)[7] = (lr & 1) ? xb : xa;
                V8 (&xnx)[7] = (lr & 1) ? xa : xb;
                if (lr + 1 < 5) {
#pragma unroll
                    for (int s = 0; s < 7; ++s) xnx[s] = *reinterpret_cast<const V8 *>(fb + (2 * (lr + 1) + s) * ROWB);
                }
                const int ho = cr0 + lr;
                if (ho < 0 || ho >= OW) continue;                        // rows outside the image are never pooled (wave-uniform)
                f32x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma16<F16>(wf[i][s], xc[s], acc[i]);
                char *prow = smem + ((size_t)lr * OW + col) * 128;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint2 r = {pack2_h<F16>(fmaxf(acc[i][0] + bv[i].x, 0.f), fmaxf(acc[i][1] + bv[i].y, 0.f)), pack2_h<F16>(fmaxf(acc[i][2] + bv[i].z, 0.f), fmaxf(acc[i][3] + bv[i].w, 0.f))};   // (one v_cvt_pk per dword)
                    const int c16 = i * 2 + (g >> 1);                    // 16-byte chunk of channels 16i+4g .. +3
                    *reinterpret_cast<uint2 *>(prow + ((c16 ^ (col & 7)) << 4) + (g & 1) * 8) = r;
                }
            }
        }
        ST_T(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the rows of image n + 1 (and this wave's older pooling stores) have landed
        __syncthreads();                                                 // conv tile complete; every wave has finished with rows buffer b
        ST_T(4);
        if constexpr (U8) {
            // raw rows of image n + 3 into the buffer image n's came from (converted one pooling phase ago); image n + 1's raw rows -
            // waited for before the barrier that ended image n - 1, two barriers back - become the rows buffer MFMA(n - 1) is done with
            if (n + 3 < n1) stage_raw(n + 3);
            if (n + 1 < n1) convert_rows(n + 1, 1 - b);
        } else {
            if (n + 2 < n1) stage_rows(n + 2, b);                        // (travels under the pooling phase and the next image's MFMAs)
        }
        // pooling: 2 rows x 56 cols x 8 channel-chunks = 896 outputs of 16 B.  The inputs are post-ReLU, i.e. non-negative, and for
        // non-negative bf16 / f16 values the numeric order IS the order of their bit patterns read as unsigned integers: the maximum is
        // four packed v_pk_max_u16 per tap instead of 8 unpack-convert-fmax chains (sign bit masked first: a -0 would read as 0x8000).
        // The selected element is one of the inputs, bit for bit, as with fmaxf.
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        for (int o = tid; o < 2 * PO * 8; o += 512) {
            const int k = o & 7, pc = (o >> 3) % PO, pr = o / (8 * PO);
            unsigned mx[4] = {0u, 0u, 0u, 0u};               // (scalars: hipcc mis-folds a bit_cast of a vector ELEMENT lvalue to element 0)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int lr = 2 * pr + dy, ho = cr0 + lr;
                if (ho < 0 || ho >= OW) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int col = 2 * pc - 1 + dx;
                    if (col < 0 || col >= OW) continue;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + ((size_t)lr * OW + col) * 128 + ((k ^ (col & 7)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w = v[e] & 0x7fff7fffu;
                        mx[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, mx[e]), __builtin_bit_cast(us2, w)));
                    }
                }
            }
            u32x4 *const dst = reinterpret_cast<u32x4 *>(out + (((size_t)n * PO + pr0 + pr) * PO + pc) * STEM_CO + k * 8);
            if constexpr (PVR_NT & 128) __builtin_nontemporal_store(u32x4{mx[0], mx[1], mx[2], mx[3]}, dst);
            else *dst = u32x4{mx[0], mx[1], mx[2], mx[3]};
        }
        // (the conv tile is rewritten only after the barriers at the top of the next image)
    }
}

// Round 6 form of the fused stem: the 3 x 3 / 2 max pool runs IN REGISTERS, straight from the MFMA accumulators.
//
// stem_pool_lds_kernel above writes a block's five conv rows (5 x 112 pixels x 64 channels, 70 KB) to LDS, meets a barrier and pools them with all 512 threads:
// 182 us per 256 frames at 0.13 of BOTH roofs, LDS bank conflicts 0.40 of the LDS cycles, 41 M vector instructions against 7 M MFMAs, one of eight waves idle
// in the matrix phase (seven 16-column tiles), the matrix phase and the pooling phase strictly one after the other (profiles/r05_sq_counters_conv.txt).
// Here a wave's 16-column conv tile is cut so that it holds every conv column its pooled columns need: wave w owns pooled columns 7 w .. 7 w + 6 = conv columns
// 14 w - 1 .. 14 w + 13 (15 of the tile's 16 columns; 8 waves x 7 = the row's 56 pooled columns, all eight waves in the matrix phase).  Then
//   * vertical max: the three conv rows of a pooled row are the same lane's accumulators of three consecutive row iterations (running max, fp32);
//   * horizontal max: the neighbouring conv columns are the neighbouring LANES of the 16-lane MFMA row: two DPP row shifts (row_shr:1 / row_shl:1, lanes
//     without a source keep -inf) - the pooled column 7 w + k sits in lane 2 k + 1;
//   * bias, ReLU and the 16-bit rounding are applied ONCE, to the maximum: x -> round(relu(x + b)) is monotone, so round(relu(max(acc) + b)) is the element
//     the old kernel picks among the rounded taps, bit for bit (a -0 is cleared as there); columns / rows outside the image are -inf / skipped;
//   * the 14 KB of pooled values cross a small LDS tile only to leave as whole 128-byte lines.
// No conv tile, no pooling pass over it, one quarter of the epilogue conversions.  Same fragments and MFMA order per output pixel as stem_kernel: bit-identical
// (test_fused_stem_pool_is_bit_identical_to_stem_then_maxpool, test_stem_reading_uint8_frames_...).  PVR_STEM_REGPOOL=0 keeps the LDS-tile form (A/B).
template <bool F16, bool U8 = false>
__global__ __launch_bounds__(512, 1) void stem_pool_reg_kernel(const u16 *__restrict__ img, const u16 *__restrict__ wgt,
                                                               const float *__restrict__ bias, u16 *__restrict__ out, int nimg, int ipb, U8Geo u8g = U8Geo{},
                                                               StemC1 c1 = StemC1{}) {
    typedef typename HT<F16>::V8 V8;
    constexpr int PW = 232, PH = 230, OW = 112, PO = 56;
    constexpr int ROWB = PW * 8;                                     // bytes of one image row (4 channels x 16 bit)
    constexpr int INB = U8 ? 28672 : 32768;                          // one input-row buffer (15 rows = 27 840 B, copied as 32 x 1 KB)
    constexpr int RAWROW = 224 * 3, RAWB = 10240;                    // U8: raw crop rows [15][672] uint8 (10 x 1 KB of DMA), three buffers
    constexpr int PAD = 1024;                                        // (conv column -1 of wave 0 reads 16 bytes in front of a row: keep that inside the allocation)
    constexpr int IN0 = PAD, RAW0 = IN0 + 2 * INB, OT0 = RAW0 + (U8 ? 3 * RAWB : 0);   // [pad | rows buffer 0 | rows buffer 1 (| raw 0 | raw 1 | raw 2) | pooled tile [2][56][128 B] | W1 image]
    constexpr int OTB = 2 * PO * 128;                                // one pooled tile; TWO of them: image n's is copied out (and convolved) during image n + 1's matrix phase
    constexpr int W1L = OT0 + 2 * OTB;                               // layer1.0.conv1's weights (8 KB: eight MFMA A fragments), when that convolution runs here
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pr0 = blockIdx.x * 2;                                  // pooled rows pr0, pr0+1
    const int cr0 = 2 * pr0 - 1;                                     // first conv row held (may be -1: above the image)
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    V8 wf[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 7; ++s)
            wf[i][s] = *reinterpret_cast<const V8 *>(wgt + (size_t)(i * 16 + px) * STEM_K + s * 32 + g * 8);
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(bias + i * 16 + g * 4);
    const int n0 = blockIdx.y * ipb, n1 = n0 + ipb < nimg ? n0 + ipb : nimg;
    auto stage_rows = [&](int n_, int b_) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(img) + (size_t)n_ * PH * PW * 4, 0, (unsigned)(PH * ROWB), 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int flat = (wave * 4 + k) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + IN0 + b_ * INB + flat), 16,
                                                     2 * cr0 * ROWB + flat + lane * 16, 0, 0, 0);
        }
    };
    auto stage_raw = [&](int n_) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(u8g.src) + (size_t)n_ * u8g.img_bytes, 0, (unsigned)u8g.img_bytes, 0x00020000);
        char *rb = smem + RAW0 + ((n_ - n0) % 3) * RAWB;
        for (int i = wave; i < 10; i += 8) {
            const int q = i * 64 + lane, r = q / 42, c = q % 42;
            const int yi = 2 * cr0 - 3 + r;                               // crop row of padded row 2 cr0 + r
            const bool ok = q < 630 && yi >= 0 && yi < 224;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(rb + i * 1024), 16,
                                                     ok ? u8g.off0 + yi * u8g.pitch + c * 16 : 0x7ffffff0, 0, 0, PVR_NT_AUX(1024));
        }
    };
    auto convert_rows = [&](int n_, int zb_) {
        const char *rb = smem + RAW0 + ((n_ - n0) % 3) * RAWB;
        char *zb = smem + IN0 + zb_ * INB;
        for (int i = tid; i < 15 * 56; i += 512) {
            const int r = i / 56, k = i - r * 56, yi = 2 * cr0 - 3 + r;
            typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
            u32x3 w = u32x3{0u, 0u, 0u};
            const bool ok = yi >= 0 && yi < 224;
            if (ok) w = *reinterpret_cast<const u32x3 *>(rb + r * RAWROW + k * 12);
            const float one = ok ? 1.0f : 0.f, off = ok ? 128.f : 0.f;
            float f[12];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                f[4 * d] = (float)(w[d] & 0xffu); f[4 * d + 1] = (float)((w[d] >> 8) & 0xffu);
                f[4 * d + 2] = (float)((w[d] >> 16) & 0xffu); f[4 * d + 3] = (float)(w[d] >> 24);
            }
            char *dst = zb + r * ROWB + (4 * k + 3) * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 o = {pack2_h<F16>(f[3 * q] - off, f[3 * q + 1] - off), pack2_h<F16>(f[3 * q + 2] - off, one)};
                *reinterpret_cast<uint2 *>(dst + q * 8) = o;
            }
        }
        if (tid < 15 * 8) {                                           // the zero border: padded columns 0..2 and 227..231 of every row
            const int r = tid >> 3, e = tid & 7, pc = e < 3 ? e : 224 + e;
            *reinterpret_cast<ushort4 *>(zb + r * ROWB + pc * 8) = make_ushort4(0, 0, 0, 0);
        }
    };
    if (c1.w) {                                                      // (visible after the prologue's barrier)
        *reinterpret_cast<u32x4 *>(smem + W1L + tid * 16) = *reinterpret_cast<const u32x4 *>(c1.w + tid * 8);
        if (tid < 64) *reinterpret_cast<float *>(smem + W1L + 8192 + tid * 4) = c1.b[tid];
    }
    if constexpr (U8) {
        for (int k = 0; k < 3; ++k) if (n0 + k < n1) stage_raw(n0 + k);
    } else {
        if (n0 < n1) stage_rows(n0, 0);
        if (n0 + 1 < n1) stage_rows(n0 + 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (U8) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (n0 < n1) convert_rows(n0, 0);
    }
    // this lane's conv column and its pooled slot
    const int col = 14 * wave - 1 + px;                              // conv column of MFMA column px of this wave's tile
    const bool col_ok = (unsigned)col < (unsigned)OW && px < 15;
    const bool owner = (px & 1) && px < 15;                          // lane 2 k + 1 holds pooled column 7 w + k
    const int pc = 7 * wave + (px >> 1);
    const float NEG = -__builtin_huge_valf();
    const bool border = wave == 0 || wave == 7;                       // the only tiles with a conv column outside the image (-1 / 112)
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // v_max_f32 as it is: fmaxf() makes hipcc canonicalise both operands first (a v_max x, x each: 96 more vector instructions per image in a kernel that
    // is bound by vector issue); MFMA results and -inf need no quieting
    auto vmax = [](float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto vmax3 = [](float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    // What happens to a finished pooled tile (image m, tile m & 1), INSIDE the next image's matrix phase - or, for the block's last image, behind the loop:
    //   copy-out (waves 0-3, in front of their MFMAs): the tile leaves as whole 128-byte lines, 2 rows x 56 columns x 8 chunks of 16 bytes;
    //   layer1.0.conv1 (waves 4-7, behind their MFMAs; 1 x 1, 64 -> 64, + bn1 + ReLU: torchvision Bottleneck.conv1 of the first block) on the tile's 112 pixels:
    //   seven 16-pixel MFMA tiles (the block's two pooled rows are seven ALIGNED 16-pixel blocks of the tensor), two per wave, weights as A fragments from the LDS
    //   image, rows permuted so that a tile pair gives a lane 8 consecutive output channels (chain_row_source) = 16-byte stores in either layout.  The launch of
    //   that convolution and its read of the pooled tensor (103 MB per 256 frames) are gone; same K order and rounding.
    // so that each SIMD's two waves (w, w + 4) have vector work and matrix work at different times.
    auto copy_out = [&](int m, int first, int step) {
        const char *ot = smem + OT0 + ((m - n0) & 1) * OTB;
        for (int o = first; o < 2 * PO * 8; o += step) {
            const int k = o & 7, pcc = (o >> 3) % PO, pr = o / (8 * PO);
            const u32x4 vv = *reinterpret_cast<const u32x4 *>(ot + (pr * PO + pcc) * 128 + ((k ^ (pcc & 7)) << 4));
            u32x4 *const dst = reinterpret_cast<u32x4 *>(out + (((size_t)m * PO + pr0 + pr) * PO + pcc) * STEM_CO + k * 8);
            if constexpr (PVR_NT & 128) __builtin_nontemporal_store(vv, dst);
            else *dst = vv;
        }
    };
    auto conv1_tile = [&](int m, int tile) {
        const char *ot = smem + OT0 + ((m - n0) & 1) * OTB;
        const int pixl = tile * 16 + px, pr = pixl / PO, pcc = pixl - pr * PO;
        f32x4 a1[4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const V8 xf = *reinterpret_cast<const V8 *>(ot + (pr * PO + pcc) * 128 + (((ks * 4 + g) ^ (pcc & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a1[i] = mfma16<F16>(*reinterpret_cast<const V8 *>(smem + W1L + (i * 2 + ks) * 1024 + lane * 16), xf, ks == 0 ? zero : a1[i]);
        }
        const long long m0 = ((long long)m * PO + pr0) * PO;                                  // first pixel of the block's two rows: a multiple of 16
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 bA = *reinterpret_cast<const float4 *>(smem + W1L + 8192 + (32 * q + 8 * g) * 4), bB = *reinterpret_cast<const float4 *>(smem + W1L + 8192 + (32 * q + 8 * g + 4) * 4);
            const f32x4 lo = a1[2 * q], hi = a1[2 * q + 1];
            u32x4 o;
            o[0] = pack2_h<F16>(vmax(lo[0] + bA.x, 0.f), vmax(lo[1] + bA.y, 0.f)); o[1] = pack2_h<F16>(vmax(lo[2] + bA.z, 0.f), vmax(lo[3] + bA.w, 0.f));
            o[2] = pack2_h<F16>(vmax(hi[0] + bB.x, 0.f), vmax(hi[1] + bB.y, 0.f)); o[3] = pack2_h<F16>(vmax(hi[2] + bB.z, 0.f), vmax(hi[3] + bB.w, 0.f));
            u16 *dst = c1.blk ? c1.t1 + ((m0 >> 4) + tile) * 1024 + (q * 4 + g) * 128 + px * 8                 // [pixel >> 4][channel >> 3][pixel & 15][8]
                              : c1.t1 + (m0 + pixl) * 64 + q * 32 + g * 8;
            *reinterpret_cast<u32x4 *>(dst) = o;
        }
    };
    for (int n = n0; n < n1; ++n) {
        const int b = (n - n0) & 1;
        ST_T(0);
        if constexpr (U8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // convert_rows' LDS stores (previous copy-out phase / prologue)
        __builtin_amdgcn_s_barrier();                                    // rows buffer b complete
        __builtin_amdgcn_sched_barrier(0);
        ST_T(1);
        // U8: the next image's rows are converted inside this image's matrix phase - by waves 0-3 in front of their MFMAs, by waves 4-7 behind them, so that
        // each SIMD's two waves (w, w + 4) have vector work and matrix work at different times (raw rows of image n + 1: waited for before barrier 2 of image
        // n - 1; rows buffer 1 - b: free since that barrier)
        if constexpr (U8) { if (wave < 4 && n + 1 < n1) convert_rows(n + 1, 1 - b); }
        if (wave < 4 && n > n0) copy_out(n - 1, tid, 256);
        const char *rows = smem + IN0 + b * INB;
        const char *fb = rows + (2 * col + 2 * g) * 8;                   // lane (px, g), filter row s: pixels 2 col + 2 g, + 1 of LDS row 2 lr + s
        V8 xa[7], xb[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) xa[s] = *reinterpret_cast<const V8 *>(fb + s * ROWB);
        // The five conv rows of the block: rows 0-2 make pooled row pr0, rows 2-4 pooled row pr0 + 1 (one v_max3 per value and pooled row).  Rows 1-3 are
        // always inside the image; row 0 is above it in the first block row, row 4 below it in the last (block-uniform).
        const bool top_ok = cr0 >= 0, bot_ok = cr0 + 4 < OW;
        f32x4 ra[4], rb[4], rc[4], v0[4], v1[4];
#define STEM_ROW(dst_, lr_)                                                                                             \
        {                                                                                                               \
            V8 (&xc)[7] = ((lr_) & 1) ? xb : xa;                                                                        \
            V8 (&xnx)[7] = ((lr_) & 1) ? xa : xb;                                                                       \
            if ((lr_) + 1 < 5) {                                                                                        \
                _Pragma("unroll") for (int s = 0; s < 7; ++s) xnx[s] = *reinterpret_cast<const V8 *>(fb + (2 * ((lr_) + 1) + s) * ROWB); \
            }                                                                                                           \
            _Pragma("unroll") for (int s = 0; s < 7; ++s)                                                               \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) dst_[i] = mfma16<F16>(wf[i][s], xc[s], s == 0 ? zero : dst_[i]); \
            if (border) {                              /* (wave-uniform) columns outside the image never win */         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) dst_[i][e] = col_ok ? dst_[i][e] : NEG;               \
            }                                                                                                           \
        }
        if (top_ok) STEM_ROW(ra, 0)
        else {                                                           // (keeps the fragment double-buffering in step: row 1 reads the other set)
#pragma unroll
            for (int s = 0; s < 7; ++s) xb[s] = *reinterpret_cast<const V8 *>(fb + (2 + s) * ROWB);
        }
        STEM_ROW(rb, 1)
        STEM_ROW(rc, 2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v0[i][e] = top_ok ? vmax3(ra[i][e], rb[i][e], rc[i][e]) : vmax(rb[i][e], rc[i][e]);
        STEM_ROW(ra, 3)
        if (bot_ok) {
            STEM_ROW(rb, 4)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v1[i][e] = vmax3(rc[i][e], ra[i][e], rb[i][e]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v1[i][e] = vmax(rc[i][e], ra[i][e]);
        }
#undef STEM_ROW
        // horizontal max over conv columns col - 1, col, col + 1 = lanes px - 1, px, px + 1 of the 16-lane row (the DPP shift rides in the max instruction);
        // then bias, ReLU, rounding - once
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ve = pr ? v1[i][e] : v0[i][e];
                    float t;
                    // (lanes 0 / 15 of a row have no left / right source: the instruction leaves their result undefined - the owners of a pooled column are lanes 1, 3 .. 13)
                    asm("v_max_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(ve), "v"(ve));      // max(lane px - 1, lane px)
                    asm("v_max_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(h[e]) : "v"(ve), "v"(t));    // max(lane px + 1, that)
                }
                if (owner) {
                    const uint2 o = {pack2_h<F16>(vmax(h[0] + bv[i].x, 0.f), vmax(h[1] + bv[i].y, 0.f)) & 0x7fff7fffu,
                                     pack2_h<F16>(vmax(h[2] + bv[i].z, 0.f), vmax(h[3] + bv[i].w, 0.f)) & 0x7fff7fffu};
                    const int c16 = i * 2 + (g >> 1);                    // 16-byte chunk of channels 16 i + 4 g .. + 3
                    *reinterpret_cast<uint2 *>(smem + OT0 + b * OTB + (pr * PO + pc) * 128 + ((c16 ^ (pc & 7)) << 4) + (g & 1) * 8) = o;
                }
            }
        }
        if constexpr (U8) { if (wave >= 4 && n + 1 < n1) convert_rows(n + 1, 1 - b); }
        if (c1.w && wave >= 4 && n > n0) {
            conv1_tile(n - 1, 2 * (wave - 4));
            if (wave < 7) conv1_tile(n - 1, 2 * (wave - 4) + 1);
        }
        ST_T(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the rows of image n + 1 (and this wave's older stores) have landed; the pooled tile is written
        ST_T(3);
        __builtin_amdgcn_s_barrier();                                    // pooled tile complete; every wave has finished with rows buffer b
        __builtin_amdgcn_sched_barrier(0);
        ST_T(4);
        if constexpr (U8) {
            if (n + 3 < n1) stage_raw(n + 3);
        } else {
            if (n + 2 < n1) stage_rows(n + 2, b);                        // (travels under the copy-out and the next image's MFMAs)
        }
        ST_T(5);
    }
    // the block's last image: its pooled tile is complete behind the loop's last barrier
    if (n1 > n0) {
        copy_out(n1 - 1, tid, 512);
        if (c1.w && wave < 7) conv1_tile(n1 - 1, wave);
    }
}
#ifdef STEM_STAMP
}  // namespace pvr
extern "C" int pvr_debug_stem_stamps(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pvr::stem_stamps), sizeof(pvr::stem_stamps)); }
namespace pvr {
#endif

// maxpool 3x3 stride 2 pad 1, NHWC, 8 channels (16 B) per thread.  Inputs are post-ReLU but the
// kernel is general: out-of-range taps are skipped (-inf padding as torch does).
template <bool F16>
__global__ __launch_bounds__(256) void maxpool_kernel(const u16 *__restrict__ in, u16 *__restrict__ out, int n,
                                                      int h, int w, int c, int ho, int wo) {
    const int cg = c / 8;
    const size_t total = (size_t)n * ho * wo * cg;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ch = (int)(idx % cg) * 8;
        size_t r = idx / cg;
        const int x = (int)(r % wo); r /= wo;
        const int y = (int)(r % ho);
        const int b = (int)(r / ho);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * y - 1 + dy;
            if (yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * x - 1 + dx;
                if (xx < 0 || xx >= w) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(in + (((size_t)b * h + yy) * w + xx) * c + ch);
                const u16 *e = reinterpret_cast<const u16 *>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], from_h<F16>(e[j]));
            }
        }
        uint4 o;
        u16 *oe = reinterpret_cast<u16 *>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) oe[j] = to_h<F16>(m[j]);
        *reinterpret_cast<uint4 *>(out + (((size_t)b * ho + y) * wo + x) * c + ch) = o;
    }
}

// AdaptiveAvgPool2d(1)+flatten: NHWC (fp32 or 16-bit) -> fp32 row at out + b*out_stride
template <bool F16, bool IN_F32>
__global__ __launch_bounds__(256) void avgpool_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                      int64_t out_stride, int hw, int c) {
    const int b = blockIdx.y;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    if constexpr (IN_F32) {
        const float *p = (const float *)in + (size_t)b * hw * c + ch;
        if (hw == 49) {
            // the ResNet50 trunk: all 49 loads in flight at once.  Summation order (round 5) = the order conv_wfrag's pooled epilogue sums in, so that
            // a plan with the pool inside the last convolution and a plan with this launch agree bit for bit: sixteen partial sums over
            // q = 16 j + l (j ascending), then the shift-and-add tree l += l - 1, l - 2, l - 4, l - 8 whose last element is the total
            float v[49];
#pragma unroll
            for (int i = 0; i < 49; ++i) v[i] = p[(size_t)i * c];
            float a[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                a[l] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (16 * j + l < 49) a[l] += v[16 * j + l];
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh *= 2)
#pragma unroll
                for (int l = 15; l >= sh; --l) a[l] += a[l - sh];
            s = a[15];
        } else {
#pragma unroll 7
            for (int i = 0; i < hw; ++i) s += p[(size_t)i * c];  // (same summation order; the unroll only lets the loads go out together)
        }
    } else {
        const u16 *p = (const u16 *)in + (size_t)b * hw * c + ch;
        for (int i = 0; i < hw; ++i) s += from_h<F16>(p[(size_t)i * c]);
    }
    out[(size_t)b * out_stride + ch] = s / (float)hw;
}

// compression-head flatten (moco.py:57-60: avgpool/fc are empty Sequentials, so the reference
// flattens (N,c,H,W) C-major): fp32 NHWC with padded channels -> out[b][ch*hw + i]
__global__ __launch_bounds__(256) void nhwc_to_chw_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          int64_t out_stride, int hw, int cpad, int creal) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw * creal) return;
    const int ch = idx / hw, i = idx % hw;
    out[(size_t)b * out_stride + idx] = in[((size_t)b * hw + i) * cpad + ch];
}

// fp32 tap of a 16-bit activation buffer (parity debugging)
template <bool F16>
__global__ __launch_bounds__(256) void h_to_f32_kernel(const u16 *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = from_h<F16>(in[i]);
}

// 16-bit copy of an fp32 activation (the conv-operand view of the fp32 residual stream of the compressed PVRs' parity plan):
// 8 elements per thread, 2 x 16-byte loads -> one 16-byte store
template <bool F16>
__global__ __launch_bounds__(256) void f32_to_h_kernel(const float *__restrict__ in, u16 *__restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(in + i * 8), b = *reinterpret_cast<const f32x4 *>(in + i * 8 + 4);
        u32x4 r;
        r[0] = (unsigned)to_h<F16>(a[0]) | ((unsigned)to_h<F16>(a[1]) << 16);
        r[1] = (unsigned)to_h<F16>(a[2]) | ((unsigned)to_h<F16>(a[3]) << 16);
        r[2] = (unsigned)to_h<F16>(b[0]) | ((unsigned)to_h<F16>(b[1]) << 16);
        r[3] = (unsigned)to_h<F16>(b[2]) | ((unsigned)to_h<F16>(b[3]) << 16);
        *reinterpret_cast<u32x4 *>(out + i * 8) = r;
    }
}

pvr_status launch_f32_to_h(const float *in, void *out, size_t n, int dtype, hipStream_t stream) {
    PVR_REQUIRE(n % 8 == 0, "f32_to_h: element count %zu not a multiple of 8", n);
    const size_t n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (dtype == PVR_F16) hipLaunchKernelGGL(f32_to_h_kernel<true>, dim3(blocks), dim3(256), 0, stream, in, (u16 *)out, n8);
    else hipLaunchKernelGGL(f32_to_h_kernel<false>, dim3(blocks), dim3(256), 0, stream, in, (u16 *)out, n8);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_stem(const void *img, const void *wgt, const float *bias, void *out, int n, int crop, int dtype,
                       hipStream_t stream) {
    PVR_REQUIRE(crop == 224, "stem: crop must be 224 (got %d)", crop);
    dim3 grid((crop / 2 + 3) / 4, n);
    if (dtype == PVR_F16)
        hipLaunchKernelGGL(stem_kernel<true>, grid, dim3(256), 0, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, crop);
    else
        hipLaunchKernelGGL(stem_kernel<false>, grid, dim3(256), 0, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, crop);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// Fused form for frames that need no resize: uint8 NHWC frames [n][h][w][3] in, pooled stem output out; (top, left) = crop origin.
// stem_pool_u8_ok: the geometry fits the kernel's 16-byte row DMA (the PVR_STEM_U8 / PVR_STEM_LDS switches are the encoder's: PlanSwitches).
// PVR_STEM_REGPOOL=0: the LDS-tile pooling of rounds 3-5 (A/B; both forms are bit-identical).  Process-wide, read once.
static bool stem_regpool() {
    static const bool on = [] { const char *e = getenv("PVR_STEM_REGPOOL"); return !e || atoi(e) != 0; }();
    return on;
}
static int g_stem_regpool_override = -1;                         // pvr_debug_set_stem_regpool: -1 environment, 0 / 1 forced (the A/B test flips it inside one process)
void set_stem_regpool(int v) { g_stem_regpool_override = v; }
static bool stem_regpool_now() { return g_stem_regpool_override >= 0 ? g_stem_regpool_override != 0 : stem_regpool(); }
bool stem_pool_u8_ok(const void *frames, int h, int w, int top, int left) {
    // 16-byte row DMA: every source chunk aligned; the crop window inside the frame (an out-of-frame window would read wrong rows, not fail)
    return ((uintptr_t)frames & 15) == 0 && ((long long)h * w * 3) % 16 == 0 && (w * 3) % 16 == 0 && (left * 3) % 16 == 0 &&
           top >= 0 && left >= 0 && top + 224 <= h && left + 224 <= w && (long long)h * w * 3 < 0x7ffffff0ll;
}
// The stem can run layer1.0.conv1 itself (register-pooling form only): callers ask first
bool stem_conv1_capable() { return stem_regpool_now(); }

// (64, 64) 16-bit weights of a 1 x 1 convolution in pvr_op_conv2d's layout -> the 8 KB image stem_pool_reg_kernel reads: fragment (cout tile i, K step ks) =
// [k chunk][row & 15][8], rows permuted inside the two 32-row blocks (row 16 t + 4 a + c holds cout 8 a + 4 t + c); host side, once per plan
void stem_c1_pack(const u16 *w, u16 *img) {
    for (int i = 0; i < 4; ++i)
        for (int ks = 0; ks < 2; ++ks)
            for (int c = 0; c < 4; ++c)
                for (int r = 0; r < 16; ++r) {
                    const int row = 16 * i + r, src = (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);
                    for (int e = 0; e < 8; ++e) img[(((i * 2 + ks) * 4 + c) * 16 + r) * 8 + e] = w[src * 64 + ks * 32 + c * 8 + e];
                }
}

pvr_status launch_stem_pool_u8(const uint8_t *frames, int n, int h, int w, int top, int left, const void *wgt, const float *bias, void *out,
                               int dtype, hipStream_t stream, const void *c1_w, const float *c1_b, void *c1_t1, int c1_blk) {
    PVR_REQUIRE(stem_pool_u8_ok(frames, h, w, top, left), "stem (uint8 form): geometry h=%d w=%d top=%d left=%d not supported", h, w, top, left);
    static const int cus = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
    const int yb = cus / 28 > 0 ? cus / 28 : 1;
    const int ipb_fit = (n + yb - 1) / yb;
    const int ipb = n <= 8 ? 1 : (ipb_fit > STEM_IPB ? ipb_fit : STEM_IPB);
    dim3 grid(28, (n + ipb - 1) / ipb);
    const size_t lds = 5 * 112 * 128 + 2 * 28672 + 3 * 10240, lds_reg = 1024 + 2 * 28672 + 3 * 10240 + 2 * 2 * 56 * 128 + 8192 + 256;
    PVR_REQUIRE(!c1_w || (stem_regpool_now() && c1_b && c1_t1), "stem: layer1.0.conv1 inside the stem needs the register-pooling form");
    StemC1 c1; c1.w = (const u16 *)c1_w; c1.b = c1_b; c1.t1 = (u16 *)c1_t1; c1.blk = c1_blk;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_lds_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_lds_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_reg_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_reg_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg));
        attr_done.mark();
    }
    U8Geo g;
    g.src = frames; g.pitch = w * 3; g.img_bytes = h * w * 3; g.off0 = (top * w + left) * 3;
    if (stem_regpool_now()) {                                    // round 6: the max pool in registers (stem_pool_reg_kernel)
        if (dtype == PVR_F16)
            hipLaunchKernelGGL((stem_pool_reg_kernel<true, true>), grid, dim3(512), lds_reg, stream, (const u16 *)nullptr, (const u16 *)wgt, bias, (u16 *)out, n, ipb, g, c1);
        else
            hipLaunchKernelGGL((stem_pool_reg_kernel<false, true>), grid, dim3(512), lds_reg, stream, (const u16 *)nullptr, (const u16 *)wgt, bias, (u16 *)out, n, ipb, g, c1);
        PVR_LAUNCH_CHECK();
        return PVR_OK;
    }
    if (dtype == PVR_F16)
        hipLaunchKernelGGL((stem_pool_lds_kernel<true, true>), grid, dim3(512), lds, stream, (const u16 *)nullptr, (const u16 *)wgt, bias, (u16 *)out, n, ipb, g);
    else
        hipLaunchKernelGGL((stem_pool_lds_kernel<false, true>), grid, dim3(512), lds, stream, (const u16 *)nullptr, (const u16 *)wgt, bias, (u16 *)out, n, ipb, g);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_stem_pool(const void *img, const void *wgt, const float *bias, void *out, int n, int crop, int dtype,
                            hipStream_t stream, const void *c1_w, const float *c1_b, void *c1_t1, int c1_blk) {
    PVR_REQUIRE(crop == 224, "stem: crop must be 224 (got %d)", crop);
    PVR_REQUIRE(!c1_w || (stem_regpool_now() && c1_b && c1_t1), "stem: layer1.0.conv1 inside the stem needs the register-pooling form");
    StemC1 c1; c1.w = (const u16 *)c1_w; c1.b = c1_b; c1.t1 = (u16 *)c1_t1; c1.blk = c1_blk;
    static const bool use_lds = [] { const char *e = getenv("PVR_STEM_LDS"); return !e || atoi(e) != 0; }();
    // images per block: a handful of frames (online embedding) -> one image per block (28 n blocks); else as many as it takes for the
    // 28 x ceil(n / ipb) blocks to fit the chip in ONE round (n = 256 on 256 CUs: 9 block rows of 29 images = 252 blocks), never fewer
    // than STEM_IPB: a block's prologue (weights, first rows) costs about what 0.7 images do
    static const int cus = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
    const int yb = cus / 28 > 0 ? cus / 28 : 1;
    const int ipb_fit = (n + yb - 1) / yb;
    const int ipb = n <= 8 ? 1 : (ipb_fit > STEM_IPB && use_lds ? ipb_fit : STEM_IPB);
    dim3 grid(28, (n + ipb - 1) / ipb);
    if (use_lds && stem_regpool_now()) {                         // round 6: the max pool in registers (stem_pool_reg_kernel), padded 16-bit image in
        const size_t lds_reg = 1024 + 2 * 32768 + 2 * 2 * 56 * 128 + 8192 + 256;
        static DeviceOnce attr3_done;
        if (attr3_done.needed()) {
            PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_reg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg));
            PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_reg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg));
            attr3_done.mark();
        }
        if (dtype == PVR_F16)
            hipLaunchKernelGGL(stem_pool_reg_kernel<true>, grid, dim3(512), lds_reg, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb, U8Geo{}, c1);
        else
            hipLaunchKernelGGL(stem_pool_reg_kernel<false>, grid, dim3(512), lds_reg, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb, U8Geo{}, c1);
        PVR_LAUNCH_CHECK();
        return PVR_OK;
    }
    if (use_lds) {
        const size_t lds = 5 * 112 * 128 + 2 * 32768;
        static bool attr2_done = false;
        if (!attr2_done) {
            PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr2_done = true;
        }
        if (dtype == PVR_F16)
            hipLaunchKernelGGL(stem_pool_lds_kernel<true>, grid, dim3(512), lds, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb);
        else
            hipLaunchKernelGGL(stem_pool_lds_kernel<false>, grid, dim3(512), lds, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb);
        PVR_LAUNCH_CHECK();
        return PVR_OK;
    }
    const size_t lds = 5 * 112 * 128;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)stem_pool_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    if (dtype == PVR_F16)
        hipLaunchKernelGGL(stem_pool_kernel<true>, grid, dim3(256), lds, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb);
    else
        hipLaunchKernelGGL(stem_pool_kernel<false>, grid, dim3(256), lds, stream, (const u16 *)img, (const u16 *)wgt, bias, (u16 *)out, n, ipb);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_maxpool(const void *in, void *out, int n, int h, int w, int c, int dtype, hipStream_t stream) {
    PVR_REQUIRE(c % 8 == 0, "maxpool: channels %d not a multiple of 8", c);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)n * ho * wo * (c / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == PVR_F16)
        hipLaunchKernelGGL(maxpool_kernel<true>, dim3(blocks), dim3(256), 0, stream, (const u16 *)in, (u16 *)out, n, h, w, c, ho, wo);
    else
        hipLaunchKernelGGL(maxpool_kernel<false>, dim3(blocks), dim3(256), 0, stream, (const u16 *)in, (u16 *)out, n, h, w, c, ho, wo);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_avgpool(const void *in, float *out, int64_t out_stride, int n, int hw, int c, int in_f32, int dtype,
                          hipStream_t stream) {
    dim3 grid((c + 255) / 256, n);
    if (in_f32)
        hipLaunchKernelGGL((avgpool_kernel<false, true>), grid, dim3(256), 0, stream, in, out, out_stride, hw, c);
    else if (dtype == PVR_F16)
        hipLaunchKernelGGL((avgpool_kernel<true, false>), grid, dim3(256), 0, stream, in, out, out_stride, hw, c);
    else
        hipLaunchKernelGGL((avgpool_kernel<false, false>), grid, dim3(256), 0, stream, in, out, out_stride, hw, c);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_nhwc_to_chw(const float *in, float *out, int64_t out_stride, int n, int hw, int cpad, int creal,
                              hipStream_t stream) {
    dim3 grid((hw * creal + 255) / 256, n);
    hipLaunchKernelGGL(nhwc_to_chw_kernel, grid, dim3(256), 0, stream, in, out, out_stride, hw, cpad, creal);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_h_to_f32(const void *in, float *out, size_t n, int dtype, hipStream_t stream) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == PVR_F16) hipLaunchKernelGGL(h_to_f32_kernel<true>, dim3(blocks), dim3(256), 0, stream, (const u16 *)in, out, n);
    else hipLaunchKernelGGL(h_to_f32_kernel<false>, dim3(blocks), dim3(256), 0, stream, (const u16 *)in, out, n);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
