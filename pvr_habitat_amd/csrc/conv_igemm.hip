// K5/K6/K7/K9: NHWC implicit-GEMM convolution on MFMA with fused bias (folded BN),
// residual add and ReLU.  Serves every 1x1 and 3x3 convolution of the ResNet50 family
// (torchvision Bottleneck / BasicBlock reached from reference src/embeddings.py:118-120,
// src/vision_models/moco.py:11,34-50,78-94).
//
// GEMM view:  out[m][co] = sum_k  X[m][k] * W[co][k],   m = (n,ho,wo),  k = (kh,kw,c)
//   * Cin % 64 == 0, so a BK=64 K-slice is ONE filter tap and 64 contiguous channels = 128 B
//     contiguous in NHWC: no im2col, every staging load is a full 16-B lane load.
//   * zero padding / tile tails read a 256-B zero page instead of branching.
//   * LDS tiles are [rows][64] 16-bit (128-B rows), XOR-swizzled on the 16-B chunk index
//     (chunk ^= (row>>1)&7) so the ds_read_b128 fragment reads are bank-conflict-free.
//   * MFMA 16x16x32 (bf16 or f16 inputs, fp32 accumulate) with the WEIGHTS as the A operand, so a
//     lane's 4 accumulator registers are 4 consecutive output channels of one pixel -> 8-B NHWC stores
//     and 8-B residual loads.
//   * 2-stage software pipeline: global loads for K-slice t+1 are issued before the MFMAs of slice t
//     and written to the other LDS buffer after them (one barrier per slice).
//   * block -> tile map is XCD-aware: consecutive tiles on one XCD share the activation rows (L2 reuse).
#include "common.h"

namespace pvr {

// conv_expand.hip / conv3x3_halo.hip
bool conv_expand_supported(int64_t M, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32, bool has_res);
pvr_status launch_conv_expand(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                              int cout, int stride, int relu, int dtype, hipStream_t stream, int out_blk);
bool conv3x3_halo_supported(int64_t M, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32, int64_t in_bytes);
pvr_status launch_conv3x3_halo(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int c, int relu,
                               int dtype, hipStream_t stream);
// conv_pp256.hip
bool pp256_supported(int64_t M, int cin, int cout, int kh, int kw, int64_t in_bytes, int64_t w_bytes, int64_t out_bytes, int64_t res_bytes);
pvr_status launch_conv_w4(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                          int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, hipStream_t stream);
pvr_status launch_conv_pp256(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                             int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, int bm, hipStream_t stream,
                             const void *in2 = nullptr, int h2 = 0, int w2 = 0, int cin2 = 0, int stride2 = 1);

struct ConvP {
    const u16 *in;
    const u16 *wgt;
    const float *bias;
    const u16 *res;
    void *out;
    const u16 *zero;      // >= 256 B of zeros
    int N, H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad;
    int M, K;
    unsigned in_bytes, w_bytes;   // buffer sizes for the raw-buffer range check
    int res_f32;          // residual buffer is fp32 (out_f32 layout) instead of 16-bit
    int act, out_f32;     // act: 0 none, 1 relu, 2 QuickGELU x*sigmoid(1.702x), 3 GELU (erf)
    int m_tiles, n_tiles;
    int nk_split;         // > 0: split-K launch, blockIdx.y sums K-slices [y*nk_split, (y+1)*nk_split) into out + y*M*Cout (fp32, no bias)
};

// Diagnostic build only (scripts/igemm_stamps.hip defines IGEMM_STAMP): s_memrealtime (100 MHz) stamps per block
#ifdef IGEMM_STAMP
__device__ unsigned long long igemm_stamps[8192][6];
#define IG_T(i_) { if (threadIdx.x == 0) ig_tt[i_] = __builtin_amdgcn_s_memrealtime(); }
#else
#define IG_T(i_)
#endif

// STAGES = 2: double-buffered K loop.  STAGES = 1: single K-slice convolutions (1x1, Cin = 64): half the LDS, so
// three blocks per CU overlap each other's load / MFMA / store phases (there is no K loop to pipeline).
// RES: 0 no residual, 1 16-bit residual (ResNet), 2 fp32 residual (transformer residual stream)
// NK4: the K loop has exactly four slices (K = 256, the 1x1 expand convolutions of layer3): ALL their loads are issued in the
// prologue through three register stages, so a block pays one memory latency instead of four (a 4-slice loop has nothing to hide
// the next slice's latency behind: 32 MFMAs per wave per slice against ~1 us)
template <int BM, int BN, bool F16, int STAGES, int RES, bool NK4 = false>
__global__ __launch_bounds__(256, (STAGES == 1 || BM == 64) ? 3 : 2) void conv_igemm_kernel(ConvP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int BK = 64;
    constexpr int A_CH = BM / 32;                 // 16-B chunks per thread, activation tile
    constexpr int B_CH = BN / 32;                 // 16-B chunks per thread, weight tile
    constexpr int TM = BM / 32;                   // 16-pixel tiles per wave (2x2 waves)
    constexpr int TN = BN / 32;                   // 16-cout tiles per wave
    constexpr int STAGE = (BM + BN) * 128;        // bytes per pipeline stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef IGEMM_STAMP
    unsigned long long ig_tt[6] = {0, 0, 0, 0, 0, 0};
#endif
    IG_T(0);
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = swz % p.n_tiles, tm = swz / p.n_tiles;
    const int m0 = tm * BM, co0 = tn * BN;

    // ---- staging assignment: chunk q = tid + 256*i -> LDS (row = q>>3, physical chunk = q&7) -------
    // Address arithmetic is kept OUT of the K loop (a first version spent ~480 VALU instructions per K-slice per
    // wave on 64-bit pointers, bounds tests and swizzles against 512 MFMA cycles - rocprofv3 SQ_INSTS_VALU):
    //   * loads are raw buffer loads: one 32-bit byte offset per chunk, precomputed at tap (0,0); per slice it costs
    //     one add of the wave-uniform tap offset and one select; out-of-image taps and tile tails take an offset past
    //     num_records, for which the hardware returns zeros (no zero page, no branches)
    //   * per-row 9-bit masks say which filter taps fall inside the image
    //   * LDS addresses are per-thread constants plus compile-time stage offsets (the loop is unrolled by two)
    const int srow = tid >> 3;                    // 0..31 (+32*i)
    const int pch = tid & 7;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wgt), 0, p.w_bytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;               // >= num_records for every buffer (< 2 GiB, checked by the launcher)
    int a_off[A_CH], a_mask[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = srow + 32 * i;
        const int lch = pch ^ ((row >> 1) & 7);   // logical chunk stored at this physical slot
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.Wo;
        const int t = mm / p.Wo;
        const int ho = t % p.Ho;
        const int n = t / p.Ho;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + lch * 8) * 2;
        // branch-free tap mask (KH, KW <= 3): bit th*KW+tw set when tap (th,tw) of this row falls inside the image
        int hb = 0, wb = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && t3 < p.KH && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
            wb |= (int)(t3 < p.KW && (unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
        }
        int mask = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * p.KW)) : 0;
        a_mask[i] = mask;
    }
    int b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int row = srow + 32 * i;
        const int lch = pch ^ ((row >> 1) & 7);
        // NK4: LDS row 32b + 16t + 4a + c holds cout 32b + 8a + 4t + c, so that a lane's accumulators of an MFMA tile PAIR are
        // 8 consecutive output channels of one pixel and the epilogue runs straight from registers (16-B residual loads / stores)
        const int co = co0 + (NK4 ? (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3) : row);
        b_off[i] = co < p.CoutPad ? (co * p.K + lch * 8) * 2 : OOB;
    }
    const int lds_st = srow * 128 + pch * 16;     // + 32*i*128 ; B tile after A tile

    u32x4 ra[A_CH], rb[B_CH];                     // native vectors: HIP's uint4 struct went to scratch
    const int cpt = p.Cin / BK;                   // K-slices per filter tap
    const int nk_all = p.KH * p.KW * cpt;
    const int kt0 = p.nk_split ? (int)blockIdx.y * p.nk_split : 0;               // split-K: this block's first slice
    const int nk = p.nk_split ? (nk_all - kt0 < p.nk_split ? nk_all - kt0 : p.nk_split) : nk_all;

    int tap = kt0 / cpt, cs = kt0 % cpt, kh = tap / p.KW, kw = tap % p.KW;       // position of the slice being LOADED (wave-uniform)
#define PVR_LOAD_SLICE(kt_)                                                                             \
    {                                                                                                   \
        const int tap_off = ((kh * p.W + kw) * p.Cin + cs * BK) * 2;                                    \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                              \
            const int vo = ((a_mask[i] >> tap) & 1) ? a_off[i] + tap_off : OOB;                         \
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0));  \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], (kt0 + (kt_)) * (BK * 2), 0)); \
        if (++cs == cpt) { cs = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }                         \
    }
#define PVR_STORE_SLICE(buf_)                                                                           \
    {                                                                                                   \
        char *base = smem + (buf_) * STAGE + lds_st;                                                    \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + i * 32 * 128) = ra[i];                                    \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + BM * 128 + i * 32 * 128) = rb[i];                         \
    }

    // ---- wave tiling: 2x2 waves; wave (wm, wn) owns pixels [wm*BM/2,+BM/2) x couts [wn*BN/2,+BN/2) ----
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int a_rd[2][TM], b_rd[2][TN];                 // byte offsets of this lane's fragments inside a stage, per k-step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = wm * (BM / 2) + j * 16 + fr;
            a_rd[ks][j] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int row = wn * (BN / 2) + i * 16 + fr;
            b_rd[ks][i] = BM * 128 + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 qa[NK4 ? 3 : 1][A_CH], qb[NK4 ? 3 : 1][B_CH];       // NK4: register stages of whole K-slices
#define PVR_LOADQ(q_, kt_)                                                                              \
    {                                                                                                   \
        const int tap_off = ((kh * p.W + kw) * p.Cin + cs * BK) * 2;                                    \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                              \
            const int vo = ((a_mask[i] >> tap) & 1) ? a_off[i] + tap_off : OOB;                         \
            qa[q_][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0)); \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            qb[q_][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], (kt0 + (kt_)) * (BK * 2), 0)); \
        if (++cs == cpt) { cs = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }                         \
    }
#define PVR_STOREQ(q_, buf_)                                                                            \
    {                                                                                                   \
        char *base = smem + (buf_) * STAGE + lds_st;                                                    \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + i * 32 * 128) = qa[q_][i];                                \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + BM * 128 + i * 32 * 128) = qb[q_][i];                     \
    }
    if constexpr (NK4) { PVR_LOADQ(0, 0); PVR_LOADQ(1, 1); PVR_LOADQ(2, 2); }
    else PVR_LOAD_SLICE(0);
    // residual prefetch: issue the epilogue's 16-B residual loads now so their latency hides under the K loop
    constexpr int UPR = BN / 8;                                  // 8-cout units per pixel row
    constexpr int EP_IT = BM * UPR / 256;
    u32x4 rres[RES == 0 ? 1 : (RES == 1 ? EP_IT : 2 * EP_IT)];
    if constexpr (NK4) {                                         // (pair bp, pixel tile j) -> the lane's 8 couts of its pixel
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int bp = it / TM, j = it % TM;
            const int m = m0 + wm * (BM / 2) + j * 16 + fr, co = co0 + wn * (BN / 2) + bp * 32 + fq * 8;
            const bool ok = m < p.M && co < p.Cout;
            rres[it] = *reinterpret_cast<const u32x4 *>(ok ? p.res + (size_t)m * p.Cout + co : p.zero);
        }
    } else if constexpr (RES == 1) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int u = tid + it * 256;
            const int m = m0 + u / UPR, co = co0 + (u % UPR) * 8;
            const bool ok = m < p.M && co < p.Cout;
            rres[it] = *reinterpret_cast<const u32x4 *>(ok ? p.res + (size_t)m * p.Cout + co : p.zero);
        }
    } else if constexpr (RES == 2) {
        const float *resf = reinterpret_cast<const float *>(p.res);
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int u = tid + it * 256;
            const int m = m0 + u / UPR, co = co0 + (u % UPR) * 8;
            const bool ok = m < p.M && co < p.Cout;
            const float *src = ok ? resf + (size_t)m * p.Cout + co : reinterpret_cast<const float *>(p.zero);
            rres[2 * it] = *reinterpret_cast<const u32x4 *>(src);
            rres[2 * it + 1] = *reinterpret_cast<const u32x4 *>(src + 4);
        }
    }
    if constexpr (NK4) { PVR_STOREQ(0, 0); PVR_LOADQ(0, 3); }  // slice 0 -> LDS; its register stage takes slice 3
    else PVR_STORE_SLICE(0);
    __syncthreads();
    IG_T(1);
    // one K-slice: loads of slice kt+1 are issued first, the MFMAs of slice kt run, then slice kt+1 is written to the
    // other stage.  CUR_ is a literal so every LDS access is base register + immediate.
#define PVR_K_STEP(kt_, CUR_)                                                                           \
    {                                                                                                   \
        const bool more = STAGES > 1 && (kt_) + 1 < nk;                                                 \
        if (more) PVR_LOAD_SLICE((kt_) + 1);                                                            \
        const char *sb = smem + (CUR_) * STAGE;                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
            V8 xa[TM], wb[TN];                                                                          \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(sb + a_rd[ks][j]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i) wb[i] = *reinterpret_cast<const V8 *>(sb + b_rd[ks][i]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                              \
                _Pragma("unroll") for (int j = 0; j < TM; ++j) acc[i][j] = mfma16<F16>(wb[i], xa[j], acc[i][j]); \
        }                                                                                               \
        if (more) PVR_STORE_SLICE((STAGES > 1 ? 1 - (CUR_) : 0));                                       \
        __syncthreads();                                                                                \
    }
#define PVR_MATH(CUR_)                                                                                  \
    {                                                                                                   \
        const char *sb = smem + (CUR_) * STAGE;                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
            V8 xa[TM], wb[TN];                                                                          \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(sb + a_rd[ks][j]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i) wb[i] = *reinterpret_cast<const V8 *>(sb + b_rd[ks][i]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                              \
                _Pragma("unroll") for (int j = 0; j < TM; ++j) acc[i][j] = mfma16<F16>(wb[i], xa[j], acc[i][j]); \
        }                                                                                               \
    }
    if constexpr (NK4) {                          // same slice order, same MFMA order as the loop below: bit-identical
        PVR_MATH(0); PVR_STOREQ(1, 1); __syncthreads();
        PVR_MATH(1); PVR_STOREQ(2, 0); __syncthreads();
        PVR_MATH(0); PVR_STOREQ(0, 1); __syncthreads();
        PVR_MATH(1);
    } else {
    for (int kt = 0; kt < nk; kt += 2) {
        PVR_K_STEP(kt, 0);
        if (kt + 1 < nk) PVR_K_STEP(kt + 1, 1);
    }
    }
    IG_T(2);
#undef PVR_MATH
#undef PVR_LOADQ
#undef PVR_STOREQ
#undef PVR_K_STEP
#undef PVR_LOAD_SLICE
#undef PVR_STORE_SLICE

    // ---- epilogue ------------------------------------------------------------------------------------------
    // (1) acc + bias -> LDS as an fp32 [pixels][BN couts] tile (reuses the pipeline buffers; every wave is
    //     past the loop's last barrier).  D row = 4*fq + reg = cout, D col = fr = pixel, so a lane owns one
    //     16-B chunk per MFMA tile; chunks are XOR-swizzled with (pixel & 7) against ds_write bank conflicts.
    // (2) the whole block walks the tile in 8-cout units: coalesced 16-B stores (a pixel's BN couts are
    //     contiguous in NHWC) with the prefetched residual, instead of 8-B stores at a 2*Cout-byte stride.
    // The tile is staged in EP_PASS passes of EP_ROWS pixel rows when it exceeds the pipeline buffers.
    if constexpr (NK4) {
        // epilogue straight from the accumulators (TN = 4 tiles = 2 pairs): no LDS staging, no barrier
        static_assert(!NK4 || (TN == 4 && EP_IT == 2 * TM && RES == 1), "NK4 epilogue layout");
        IG_T(3);
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            const int co = co0 + wn * (BN / 2) + bp * 32 + fq * 8;
            float4 bA = make_float4(0.f, 0.f, 0.f, 0.f), bB = bA;
            if (co < p.Cout) { bA = *reinterpret_cast<const float4 *>(p.bias + co); bB = *reinterpret_cast<const float4 *>(p.bias + co + 4); }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * (BM / 2) + j * 16 + fr;
                const f32x4 lo = acc[2 * bp][j], hi = acc[2 * bp + 1][j];
                float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                const u32x4 r = rres[bp * TM + j];
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v0 = v[2 * e] + from_h<F16>((u16)(r[e] & 0xffffu)), v1 = v[2 * e + 1] + from_h<F16>((u16)(r[e] >> 16));
                    if (p.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    o[e] = pack2_h<F16>(v0, v1);
                }
                if (m < p.M && co < p.Cout) *reinterpret_cast<u32x4 *>((u16 *)p.out + (size_t)m * p.Cout + co) = o;
            }
        }
    } else {
    constexpr int EP_FIT = STAGES * STAGE / (BN * 4);           // pixel rows the pipeline buffers can hold as fp32
    constexpr int EP_ROWS = EP_FIT >= BM ? BM : (EP_FIT >= BM / 2 ? BM / 2 : BM / 4);
    constexpr int EP_PASS = BM / EP_ROWS;
    static_assert(EP_ROWS % 16 == 0 && EP_ROWS <= EP_FIT, "epilogue pass must align to MFMA tiles and fit LDS");
    float *ep = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int pass = 0; pass < EP_PASS; ++pass) {
        if (pass > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int cl = wn * (BN / 2) + i * 16 + fq * 4;      // cout within the tile
            const int co = co0 + cl;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < p.Cout) bv = *reinterpret_cast<const float4 *>(p.bias + co);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int pl = wm * (BM / 2) + j * 16 + fr;      // pixel within the tile
                if (pl / EP_ROWS != pass) continue;              // wave-uniform (EP_ROWS % 16 == 0)
                const int pr = pl - pass * EP_ROWS;
                f32x4 v = acc[i][j];
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                *reinterpret_cast<f32x4 *>(ep + pr * BN + (((cl >> 2) ^ (pr & 7)) << 2)) = v;
            }
        }
        __syncthreads();
        if (pass == 0) IG_T(3);
#pragma unroll
        for (int it = 0; it < EP_IT / EP_PASS; ++it) {
            const int itg = pass * (EP_IT / EP_PASS) + it;       // index into the prefetched residual
            const int u = tid + itg * 256;
            const int pl = u / UPR, cu = u % UPR;
            const int pr = pl - pass * EP_ROWS;
            const int m = m0 + pl, co = co0 + cu * 8;
            if (m >= p.M || co >= p.Cout) continue;
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(ep + pr * BN + (((2 * cu) ^ (pr & 7)) << 2));
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(ep + pr * BN + (((2 * cu + 1) ^ (pr & 7)) << 2));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const size_t o = (size_t)m * p.Cout + co;
            if constexpr (RES == 1) {
                const u32x4 r = rres[itg];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += from_h<F16>((u16)(r[e] & 0xffffu));
                    v[2 * e + 1] += from_h<F16>((u16)(r[e] >> 16));
                }
            } else if constexpr (RES == 2) {
                // (bit_cast of a single vector ELEMENT lvalue is mis-folded to element 0 by hipcc: cast whole vectors)
                const f32x4 r0 = __builtin_bit_cast(f32x4, rres[2 * itg]), r1 = __builtin_bit_cast(f32x4, rres[2 * itg + 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += r0[e];
                    v[4 + e] += r1[e];
                }
            }
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));   // v_rcp_f32 (1 ulp): an IEEE division here cost 15 % of the FC1 launch
            } else if (p.act == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
            }
            if (p.out_f32) {
                float *op = (float *)p.out + o + (p.nk_split ? (size_t)blockIdx.y * p.M * p.Cout : 0);
                *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                u32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    r[e] = pack2_h<F16>(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4 *>((u16 *)p.out + o) = r;
            }
        }
    }
    }
#ifdef IGEMM_STAMP
    IG_T(4);
    if (threadIdx.x == 0 && blockIdx.x < 8192) { _Pragma("unroll") for (int k = 0; k < 6; ++k) igemm_stamps[blockIdx.x][k] = ig_tt[k]; }
#endif
}

template <int BM, int BN, bool F16, int STAGES, int RES, bool NK4 = false>
static pvr_status launch_inst2(ConvP &p, hipStream_t stream) {
    const int grid = p.m_tiles * p.n_tiles;
    const size_t lds = STAGES * (BM + BN) * 128;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_igemm_kernel<BM, BN, F16, STAGES, RES, NK4>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    const int ksplit = p.nk_split ? (p.KH * p.KW * (p.Cin / 64) + p.nk_split - 1) / p.nk_split : 1;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, F16, STAGES, RES, NK4>), dim3(grid, ksplit), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// PVR_IGEMM_NK4=0: the four-slice launches keep the generic double-buffered loop (A/B runs; bit-identical)
static bool nk4_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_IGEMM_NK4"); v = e ? atoi(e) : 1; }
    return v != 0;
}

template <int BM, int BN, bool F16, int STAGES>
static pvr_status launch_inst(ConvP &p, hipStream_t stream) {
    if constexpr (STAGES == 2 && BM == 128 && BN == 128) {
        if (p.KH == 1 && p.KW == 1 && p.K == 256 && p.res && !p.res_f32 && !p.out_f32 && p.act <= 1 && nk4_enabled()) return launch_inst2<BM, BN, F16, 2, 1, true>(p, stream);
    }
    if (!p.res) return launch_inst2<BM, BN, F16, STAGES, 0>(p, stream);
    return p.res_f32 ? launch_inst2<BM, BN, F16, STAGES, 2>(p, stream) : launch_inst2<BM, BN, F16, STAGES, 1>(p, stream);
}

template <int BM, int BN>
static pvr_status launch_cfg(ConvP &p, int dtype, hipStream_t stream) {
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    const bool single = p.K == 64;
    if (dtype == PVR_F16)
        return single ? launch_inst<BM, BN, true, 1>(p, stream) : launch_inst<BM, BN, true, 2>(p, stream);
    return single ? launch_inst<BM, BN, false, 1>(p, stream) : launch_inst<BM, BN, false, 2>(p, stream);
}

// kernel choice: -1 auto (measured crossover), 0 conv_igemm only, 1 / 2 conv_pp256 with 256- / 128-pixel tiles
// whenever it accepts the shape
static int g_conv_algo = -2;
int conv_algo() {
    if (g_conv_algo == -2) { const char *e = getenv("PVR_CONV_ALGO"); g_conv_algo = e ? atoi(e) : -1; }
    return g_conv_algo;
}
void set_conv_algo(int a) { g_conv_algo = a; }

pvr_status launch_conv(const void *in, const void *wgt, const float *bias, const void *res, void *out, const void *zero,
                       int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                       int out_f32, int dtype, hipStream_t stream) {
    PVR_REQUIRE(cin % 64 == 0, "conv: cin %d not a multiple of 64", cin);
    PVR_REQUIRE(cout % 8 == 0, "conv: cout %d not a multiple of 8", cout);
    PVR_REQUIRE(zero != nullptr, "conv: zero page missing");
    if (conv_algo() != 0 && conv3x3_halo_supported((int64_t)n * h * w, h, w, cin, cout, kh, kw, stride, pad, relu, out_f32, (int64_t)n * h * w * cin * 2))
        return launch_conv3x3_halo(in, wgt, bias, res, out, n, h, w, cin, relu, dtype, stream);
    if (conv_algo() == -1 && kh == 1 && kw == 1 && (stride == 1 || stride == 2) &&
        conv_expand_supported((int64_t)n * (h / stride) * (w / stride), h, w, cin, cout, kh, kw, stride, pad, relu, out_f32, res != nullptr))
        return launch_conv_expand(in, wgt, bias, res, out, n, h, w, cin, cout, stride, relu, dtype, stream, 0);
    {
        const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
        const int64_t M = (int64_t)n * ho * wo, K = (int64_t)kh * kw * cin;
        const int of32 = out_f32 & 1, rf32 = (out_f32 >> 1) & 1;
        const int algo = conv_algo();
        const bool ok = pp256_supported(M, cin, cout, kh, kw, (int64_t)n * h * w * cin * 2, (int64_t)((cout + 63) / 64 * 64) * K * 2,
                                        M * cout * (of32 ? 4 : 2), res ? M * cout * (rf32 ? 4 : 2) : 0);
        // auto (measured per launch at batch 256, profiles/experiments/r01_pp256_vs_igemm_per_op.txt): the 256x256 kernel runs one
        // block per CU, so its epilogue is not hidden behind another block's main loop and a grid of fewer tiles than ~2/3 of the
        // CUs leaves the chip idle.  It wins for K >= 512, and for K = 256 when there is no residual to read in the epilogue.
        const int64_t nt = (cout + 255) / 256, tiles = ((M + 255) / 256) * nt, tiles128 = ((M + 127) / 128) * nt;
        const bool deep = cout >= 256 && (K >= 512 || (K >= 256 && !res));
        // algo 1 / 2 / 3: force the 256- / 128- / 224-pixel tile; auto: 256 when that grid fills ~2/3 of the CUs, else 128 when that one does;
        // 224 instead of 256 when it needs fewer CU-rounds x rows (batch 256 at 14 x 14: 196 tiles of 256 on 256 CUs vs 224 tiles of 224)
        const int64_t tiles224 = ((M + 223) / 224) * nt;
        static const bool use224 = [] { const char *e = getenv("PVR_PP_BM224"); return !e || atoi(e) != 0; }();
        // Rounds first: a tile iteration of the persistent form costs about the same for 224 and 256 rows (its cadence is set by the feed
        // half-phases and ~12 k cycles of epilogue / setup, scripts/pp256_tile_stamps.hip), so more, smaller tiles only pay when they do not
        // add a round (fc1 of ViT-B/16: 10 rounds of 256 rows instead of 11 of 224).  PVR_PP_BM224=2 restores round 2's rows-only rule.
        static const int rule224 = [] { const char *e = getenv("PVR_PP_BM224"); return e ? atoi(e) : 1; }();
        const int64_t r224 = (tiles224 + 255) / 256, r256 = (tiles + 255) / 256;
        const bool better224 = use224 && (rule224 == 2 ? r224 * 224 < r256 * 256 : (r224 <= r256 && r224 * 224 < r256 * 256));
        const int bm = algo == 1 ? 256 : algo == 2 ? 128 : algo == 3 ? 224
                     : (algo == -1 && deep) ? (tiles >= 160 ? (better224 ? 224 : 256) : (tiles128 >= 160 ? 128 : 0)) : 0;
        // algo 4 (round 3): the four-wave 128 x 128-per-wave kernel (conv_w4.hip) wherever it accepts the shape
#ifdef PVR_EXPERIMENTS
        if (ok && algo == 4)
            return launch_conv_w4(in, wgt, bias, res, out, n, h, w, cin, cout, kh, kw, stride, pad, relu, of32, rf32, dtype, stream);
#endif
        if (ok && bm)
            return launch_conv_pp256(in, wgt, bias, res, out, n, h, w, cin, cout, kh, kw, stride, pad, relu, of32, rf32, dtype, bm, stream);
    }
    ConvP p;
    p.nk_split = 0;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.bias = bias; p.res = (const u16 *)res; p.out = out;
    p.zero = (const u16 *)zero;
    static int bm64 = -1;
    if (bm64 < 0) { const char *e = getenv("PVR_IGEMM_BM64"); bm64 = e ? atoi(e) : 0; }
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64;
    p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - kh) / stride + 1;
    p.Wo = (w + 2 * pad - kw) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    PVR_REQUIRE(M < (1ll << 31) && (int64_t)M * cout < (1ll << 40), "conv: problem too large");
    p.M = (int)M; p.K = kh * kw * cin;
    const int64_t inb = (int64_t)n * h * w * cin * 2, wb = (int64_t)p.CoutPad * p.K * 2;
    PVR_REQUIRE(inb < 0x7ffffff0ll && wb < 0x7ffffff0ll && kh <= 3 && kw <= 3, "conv: operand larger than 2 GiB or filter larger than 3x3");
    p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb;
    p.act = relu; p.out_f32 = out_f32 & 1; p.res_f32 = (out_f32 >> 1) & 1;   // out_f32 bit1: residual is fp32
    if (cout <= 64) return launch_cfg<128, 64>(p, dtype, stream);
    // experiment (PVR_IGEMM_BM64=1): 64-pixel tiles for the K = 256 expand convolutions with residual -> three blocks per CU
    if (bm64 && kh == 1 && kw == 1 && p.K == 256 && res && !p.res_f32 && !p.out_f32 && relu <= 1 && nk4_enabled()) {
        p.m_tiles = (p.M + 63) / 64; p.n_tiles = (p.Cout + 127) / 128;
        return dtype == PVR_F16 ? launch_inst2<64, 128, true, 2, 1, true>(p, stream) : launch_inst2<64, 128, false, 2, 1, true>(p, stream);
    }
    return launch_cfg<128, 128>(p, dtype, stream);
}

// ---- split-K for long, narrow convolutions ---------------------------------------------------------------------------------
// The compression heads of the *_l3 / *_l4 PVRs (3x3 on 1024 / 2048 channels down to <= 64): 196 / 98 tiles of 128 pixels, each
// walking 144 / 288 K-slices on its own - a quarter of the CUs busy for 0.14-0.23 ms.  Smaller pixel tiles only multiply the weight
// traffic (every block streams the whole [Cout][K] matrix) and a deeper load pipeline changes nothing (both measured).
// Here `ksplit` blocks share a tile: block y sums K-slices [y*nk/ksplit, ...) into an fp32 partial plane, and one elementwise
// launch adds the planes IN PLANE ORDER, then bias, residual and ReLU.  ksplit depends on the layer's shape only, never on the
// batch size, so embeddings stay independent of how the frames are batched.
template <bool F16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int ksplit, size_t plane, const float *__restrict__ bias,
                                                            const u16 *__restrict__ res, void *__restrict__ out, int cout, int relu, int out_f32) {
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u * 8 < plane; u += (size_t)gridDim.x * 256) {
        const size_t o = u * 8;
        const int co = (int)(o % cout);
        float v[8];
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(part + o), a1 = *reinterpret_cast<const f32x4 *>(part + o + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
        int s = 1;
        for (; s + 4 <= ksplit; s += 4) {             // four planes' loads in flight per round trip; the additions stay in plane order
            f32x4 b0[4], b1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b0[q] = *reinterpret_cast<const f32x4 *>(part + (size_t)(s + q) * plane + o);
                b1[q] = *reinterpret_cast<const f32x4 *>(part + (size_t)(s + q) * plane + o + 4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[q][e]; v[4 + e] += b1[q][e]; }
        }
        for (; s < ksplit; ++s) {
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(part + s * plane + o), b1 = *reinterpret_cast<const f32x4 *>(part + s * plane + o + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        const float4 bA = *reinterpret_cast<const float4 *>(bias + co), bB = *reinterpret_cast<const float4 *>(bias + co + 4);
        v[0] += bA.x; v[1] += bA.y; v[2] += bA.z; v[3] += bA.w; v[4] += bB.x; v[5] += bB.y; v[6] += bB.z; v[7] += bB.w;
        if (res) {
            const u32x4 r = *reinterpret_cast<const u32x4 *>(res + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += from_h<F16>((u16)(r[e] & 0xffffu)); v[2 * e + 1] += from_h<F16>((u16)(r[e] >> 16)); }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (out_f32) {
            *reinterpret_cast<f32x4 *>((float *)out + o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>((float *)out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            u32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = pack2_h<F16>(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<u32x4 *>((u16 *)out + o) = r;
        }
    }
}

// same operands as launch_conv (16-bit residual only, ReLU or no activation) + the fp32 scratch of ksplit planes of M*cout floats
pvr_status launch_conv_splitk(const void *in, const void *wgt, const float *bias, const void *res, void *out, const void *zero, float *scratch,
                              int ksplit, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32,
                              int dtype, hipStream_t stream) {
    PVR_REQUIRE(cin % 64 == 0 && cout % 8 == 0 && zero && scratch && ksplit > 1, "split-K conv: unsupported shape");
    // the K-range blocks read their (all-zero) bias from the zero page: 64 floats fit any zero page, wider outputs need the encoder's
    PVR_REQUIRE(cout <= 64 || (size_t)cout * sizeof(float) <= PVR_ZERO_BYTES, "split-K conv: cout %d wider than the zero page", cout);
    PVR_REQUIRE(relu <= 1 && (out_f32 & 2) == 0 && kh <= 3 && kw <= 3, "split-K conv: ReLU / 16-bit residual / filters up to 3x3 only");
    ConvP p;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.bias = (const float *)zero; p.res = nullptr; p.out = scratch;
    p.zero = (const u16 *)zero;
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64;
    p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - kh) / stride + 1;
    p.Wo = (w + 2 * pad - kw) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    PVR_REQUIRE(M < (1ll << 31), "conv: problem too large");
    p.M = (int)M; p.K = kh * kw * cin;
    const int64_t inb = (int64_t)n * h * w * cin * 2, wb = (int64_t)p.CoutPad * p.K * 2;
    PVR_REQUIRE(inb < 0x7ffffff0ll && wb < 0x7ffffff0ll, "conv: operand larger than 2 GiB");
    p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb;
    p.act = 0; p.out_f32 = 1; p.res_f32 = 0;
    const int nk = p.K / 64;
    p.nk_split = (nk + ksplit - 1) / ksplit;
    p.m_tiles = (p.M + 127) / 128;
    pvr_status s;
    if (cout <= 64) {
        p.n_tiles = 1;
        s = dtype == PVR_F16 ? launch_inst2<128, 64, true, 2, 0>(p, stream) : launch_inst2<128, 64, false, 2, 0>(p, stream);
    } else {                                                  // wide outputs (small-batch plan): 128-cout tiles
        p.n_tiles = (cout + 127) / 128;
        s = dtype == PVR_F16 ? launch_inst2<128, 128, true, 2, 0>(p, stream) : launch_inst2<128, 128, false, 2, 0>(p, stream);
    }
    if (s) return s;
    const size_t plane = (size_t)M * cout;
    const int used = (nk + p.nk_split - 1) / p.nk_split;            // planes actually written
    const int grid = (int)((plane / 8 + 255) / 256 > 4096 ? 4096 : (plane / 8 + 255) / 256);
    if (dtype == PVR_F16)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(grid), dim3(256), 0, stream, scratch, used, plane, bias, (const u16 *)res, out, cout, relu, out_f32 & 1);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(grid), dim3(256), 0, stream, scratch, used, plane, bias, (const u16 *)res, out, cout, relu, out_f32 & 1);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
