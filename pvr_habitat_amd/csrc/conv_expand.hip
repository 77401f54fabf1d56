// HBM-bound 1x1 convolutions, K <= 512: conv3 (+ identity) of the torchvision Bottlenecks of layer3 / layer4 (reference
// src/embeddings.py:118-120 -> torchvision resnet50: out = relu(bn3(conv3(t2)) + identity), K = 256 / 512 -> Cout = 4K), the
// downsample convolutions of layer1 - layer3 (1x1, stride 1 / 2, no activation) and layer1.0.conv1.
//
// These launches are HBM-bound (AI ~ 110 FLOP/B: 103 MB residual in + 103 MB out + 26 MB in for layer3 at batch 256) and were the
// weakest significant launches of the plan: 80 us = 2.9 TB/s on conv_igemm's all-loads-up-front instantiation AND on conv_pp256,
// with two or three blocks per CU alike (profiles/experiments/r02_stream_access_patterns.txt) - every block paid one full memory
// latency (4.7 us until its first slice is in LDS) before 2.4 us of MFMA work and a 2.5 us epilogue, and the blocks of a launch
// move through those phases together, so HBM idles half of the time.
//
// This kernel keeps every CU's demand continuous instead:
//   * persistent, WEIGHT-STATIONARY blocks: a block owns one tile of BN output channels, loads its [BN][K] weights into LDS once
//     (64 KB) and walks a list of 64-pixel tiles; per tile only X (64 x K) and the residual / output (64 x BN) move
//     (conv_igemm re-read the 64 KB weight tile for every pixel tile: 57 % of the bytes a block requested);
//   * the X slices of the NEXT pixel tile are requested while the current tile is still in its MFMA steps (three register stages,
//     refilled as soon as a stage has been written to LDS), and the next tile's residual before the current tile's epilogue, so a
//     block always has loads in flight;
//   * the n-tiles of one pixel-tile group sit on one XCD (blocks b, b+8, ... share an L2), so X is fetched into that L2 once.
// Same GEMM view, K order (64-wide slices ascending, two 32-deep MFMA steps each), operand layouts and epilogue arithmetic as
// conv_igemm.hip / conv_pp256.hip: results are bit-identical (tests/test_gpu_encoder.py).
#include "common.h"

namespace pvr {

struct ExpP {
    const u16 *in, *wgt, *res;
    const float *bias;
    u16 *out;
    int M, K, Cout, CoutPad, n_tiles, n_mgroups, m_tiles, act;
    int H, W, Ho, Wo;              // STRIDE 2: input / output geometry (output pixel (n,ho,wo) reads input pixel (n,2ho,2wo))
    unsigned in_bytes, w_bytes, out_bytes;
    int out_blk;                   // output in the blocked layout [pixel >> 4][cout >> 3][pixel & 15][8] (chain_wave.hip reads layer1.0's t1 so)
};

// KS = K / 64 slices; BN couts per block (KS * BN * 128 B of weights stay in LDS: 64 KB); BM = 64 pixels per tile
// RES: add a 16-bit residual in the epilogue; STRIDE: 1 or 2
template <int KS, int BN, bool F16, bool RES, int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_expand_kernel(ExpP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int BM = 64, TM = 2, TN = BN / 32, NP = TN / 2;          // 2x2 waves: 32 pixels x BN/2 couts each; NP cout pairs per lane
    constexpr int WSL = BN * 128;                                      // bytes of one weight slice [BN][64]
    constexpr int X_OFF = KS * WSL, XST = BM * 128;                    // two X stages [64][64] behind the weights
    constexpr int W_CH = KS * BN / 32;                                 // 16-B weight chunks per thread
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    // blocks b, b + 8, ... share an XCD: give one XCD all n-tiles of its pixel-tile groups
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, gpx = p.n_mgroups >> 3;
    const int tn = q % p.n_tiles, mgroup = xcd * gpx + q / p.n_tiles;
    const int co0 = tn * BN;

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wgt), 0, p.w_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.res), 0, p.out_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

    // ---- weights -> LDS, once.  LDS row 32b + 16t + 4a + c holds cout 32b + 8a + 4t + c (a lane's accumulators of an MFMA tile
    //      pair are then 8 consecutive output channels of one pixel: 16-byte residual loads / stores straight from registers)
    {
        const int srow = tid >> 3, pch = tid & 7;
#pragma unroll
        for (int b4 = 0; b4 < W_CH; b4 += 4) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = b4 + u, s = i / (BN / 32), row = srow + 32 * (i % (BN / 32));
                if (i >= W_CH) continue;
                const int lch = pch ^ ((row >> 1) & 7);
                const int co = co0 + (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);
                const int vo = co < p.CoutPad ? (co * p.K + s * 64 + lch * 8) * 2 : OOB;
                w[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = b4 + u, s = i / (BN / 32), row = srow + 32 * (i % (BN / 32));
                if (i >= W_CH) continue;
                *reinterpret_cast<u32x4 *>(smem + s * WSL + row * 128 + pch * 16) = w[u];
            }
        }
    }

    // ---- per-thread constants ------------------------------------------------------------------------------------------
    int a_row[2], a_ch[2];                                             // X staging: chunk tid + 256 i -> tile row, byte offset of its logical chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qq = tid + 256 * i, row = qq >> 3, pch = qq & 7;
        a_row[i] = row;
        a_ch[i] = (pch ^ ((row >> 1) & 7)) * 16;
    }
    // byte offset of the input pixel that output pixel m reads (slice 0, this thread's chunk); OOB past the last pixel
    auto x_off = [&](int m, int i) -> int {
        if (m >= p.M) return OOB;
        if constexpr (STRIDE == 1) return m * p.K * 2 + a_ch[i];
        const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, n = t / p.Ho;
        return ((n * p.H + ho * STRIDE) * p.W + wo * STRIDE) * p.K * 2 + a_ch[i];
    };
    const int lds_st = X_OFF + (tid >> 3) * 128 + (tid & 7) * 16;      // + 32 * 128 for the second chunk, + XST for stage 1
    int a_rd[2][TM], b_rd[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = wm * 32 + j * 16 + fr;
            a_rd[ks][j] = X_OFF + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int row = wn * (BN / 2) + i * 16 + fr;
            b_rd[ks][i] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
    }
    float4 bA[NP], bB[NP];
    int cbase[NP];                                                     // the lane's first cout of pair bp (element offset inside a pixel row)
#pragma unroll
    for (int bp = 0; bp < NP; ++bp) {
        const int co = co0 + wn * (BN / 2) + bp * 32 + fq * 8;
        cbase[bp] = co < p.Cout ? co : -1;
        bA[bp] = co < p.Cout ? *reinterpret_cast<const float4 *>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        bB[bp] = co < p.Cout ? *reinterpret_cast<const float4 *>(p.bias + co + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // FULL: 128-couts-per-block instances with NHWC output - the wave's 64 couts of a pixel are one 128-byte line (see the epilogue)
    constexpr bool FULL = NP == 2 && KS % 2 == 0;
    const int fl_c = ((lane & 7) ^ ((lane >> 3) & 7)) << 4;            // line form: lane l holds rows (l >> 3), (l >> 3) + 8, this 16-byte chunk of the 128
    const int fl_f0 = fr * 128 + ((fq ^ (fr & 7)) << 4), fl_f1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);   // fragment (fr, fq) of cout pair 0 / 1 in the slot image
    u32x4 qx[3][2];                                                    // register stages of X slices
    u32x4 rA[RES ? NP * TM : 1], rB[RES ? NP * TM : 1];                // residual of the current / next tile (alternating)
    int ac[2], an[2];                                                  // X byte offsets of the current / next pixel tile
    // X slice `s_` of the tile whose offsets are in a_ -> register stage q_ (tiles past the end: OOB -> zeros, never used)
#define EX_LOADX(q_, a_, s_)                                                                                       \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
            qx[q_][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_[i], (s_) * 128, 0)); \
    }
#define EX_STOREX(q_, st_)                                                                                         \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
            *reinterpret_cast<u32x4 *>(smem + lds_st + (st_) * XST + i * 32 * 128) = qx[q_][i];                    \
    }
#define EX_LOADRES(r_, mt_)                                                                                        \
    if constexpr (RES) {                                                                                           \
        if constexpr (FULL) {                              /* whole 128-byte lines: rows (lane >> 3) and + 8 of pixel tile j */ \
            _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                         \
                _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                 \
                    const int m = (mt_) * BM + wm * 32 + j * 16 + (lane >> 3) + 8 * hf;                            \
                    const int vo = m < p.M ? (m * p.Cout + co0 + wn * 64) * 2 + fl_c : OOB;                        \
                    r_[j * 2 + hf] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, vo, 0, (RES ? PVR_NT_AUX(64) : 0))); \
                }                                                                                                  \
        } else {                                                                                                   \
        _Pragma("unroll") for (int bp = 0; bp < NP; ++bp)                                                          \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) {                                                       \
                const int m = (mt_) * BM + wm * 32 + j * 16 + fr;                                                  \
                const int vo = (m < p.M && cbase[bp] >= 0) ? (m * p.Cout + cbase[bp]) * 2 : OOB;                   \
                r_[bp * TM + j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, vo, 0, (RES ? PVR_NT_AUX(64) : 0))); \
            }                                                                                                      \
        }                                                                                                          \
    }
#define EX_MATH(s_, st_)                                                                                           \
    {                                                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
            V8 xa[TM], wb[TN];                                                                                     \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(smem + (st_) * XST + a_rd[ks][j]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i) wb[i] = *reinterpret_cast<const V8 *>(smem + (s_) * WSL + b_rd[ks][i]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < TM; ++j) acc[i][j] = mfma16<F16>(wb[i], xa[j], acc[i][j]);   \
        }                                                                                                          \
    }
    // one pixel tile: slices s = 0 .. KS-1 live in register stage s % 3 and LDS stage (s + PAR_) & 1; a register stage is refilled right
    // after it has been written to LDS - with slice s + 3 of this tile, or with slice s % 3 of the NEXT tile once this tile has no more.
    // LDS hazards: a stage written at step s was last read by the MFMAs of slice s - 2, one barrier earlier.  Across the tile boundary
    // there is NO barrier between the last slice's MFMAs and the next tile's first LDS write, so the two must use different stages:
    // with an odd slice count (K = 64) the tiles alternate the stage parity (PAR_ = 1 for every second tile).  (Found by
    // test_second_process_loading_the_gpu_does_not_change_results: 3 of 1096 forwards differed under load before this.)
#define EX_TILE(RC_, RN_, PAR_)                                                                                    \
    {                                                                                                              \
        f32x4 acc[TN][TM];                                                                                         \
        _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};                  \
        const int mtn = mt + p.n_mgroups;                                                                          \
        an[0] = x_off(mtn * BM + a_row[0], 0); an[1] = x_off(mtn * BM + a_row[1], 1);                              \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                           \
            if (s > 0) EX_MATH(s - 1, (s - 1 + (PAR_)) & 1);                                                       \
            EX_STOREX(s % 3, (s + (PAR_)) & 1);                                                                    \
            if (s + 3 < KS) { EX_LOADX(s % 3, ac, s + 3); } else if (s % 3 < KS) { EX_LOADX(s % 3, an, s % 3); }   \
            if (s == KS - 1) EX_LOADRES(RN_, mtn);                                                                 \
            __syncthreads();                                                                                       \
        }                                                                                                          \
        EX_MATH(KS - 1, (KS - 1 + (PAR_)) & 1);                                                                    \
        if constexpr (FULL) {                                                                                      \
            /* The wave's 64 couts are one whole 128-byte line per pixel: residual and output cross the texture-address path as full   */ \
            /* lines (8 rows x 128 B per instruction) and change to / from the fragment layout through 2 KB of the X stage that is idle */ \
            /* here (its last readers passed the barrier of the last slice); the barrier below keeps the next tile's first X store,    */ \
            /* which goes to that stage, behind every wave's last use of it.                                                           */ \
            char *slot = smem + X_OFF + ((((KS - 1 + (PAR_)) & 1) ^ 1) * XST) + wave * 2048;                       \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) {                                                       \
                u32x4 r[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};                                       \
                if constexpr (RES) {                                                                               \
                    *reinterpret_cast<u32x4 *>(slot + lane * 16) = RC_[j * 2];                                     \
                    *reinterpret_cast<u32x4 *>(slot + 1024 + lane * 16) = RC_[j * 2 + 1];                          \
                    r[0] = *reinterpret_cast<const u32x4 *>(slot + fl_f0);                                         \
                    r[1] = *reinterpret_cast<const u32x4 *>(slot + fl_f1);                                         \
                }                                                                                                  \
                u32x4 o[2];                                                                                        \
                _Pragma("unroll") for (int bp = 0; bp < 2; ++bp) {                                                 \
                    const f32x4 lo = acc[2 * bp][j], hi = acc[2 * bp + 1][j];                                      \
                    const float v[8] = {lo[0] + bA[bp].x, lo[1] + bA[bp].y, lo[2] + bA[bp].z, lo[3] + bA[bp].w,    \
                                        hi[0] + bB[bp].x, hi[1] + bB[bp].y, hi[2] + bB[bp].z, hi[3] + bB[bp].w};   \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
                        float v0 = v[2 * e], v1 = v[2 * e + 1];                                                    \
                        if constexpr (RES) { v0 += from_h<F16>((u16)(r[bp][e] & 0xffffu)); v1 += from_h<F16>((u16)(r[bp][e] >> 16)); } \
                        if (p.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }                              \
                        o[bp][e] = pack2_h<F16>(v0, v1);                      \
                    }                                                                                              \
                }                                                                                                  \
                *reinterpret_cast<u32x4 *>(slot + fl_f0) = o[0];                                                   \
                *reinterpret_cast<u32x4 *>(slot + fl_f1) = o[1];                                                   \
                _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                 \
                    const u32x4 ln = *reinterpret_cast<const u32x4 *>(slot + hf * 1024 + lane * 16);               \
                    const int m = mt * BM + wm * 32 + j * 16 + (lane >> 3) + 8 * hf;                               \
                    int vo = (m * p.Cout + co0 + wn * 64) * 2 + fl_c;                                              \
                    vo = m < p.M ? vo : OOB;                                                                       \
                    __builtin_amdgcn_raw_buffer_store_b128(ln, rs_out, vo, 0, (RES ? PVR_NT_AUX(32) : 0));                                  \
                }                                                                                                  \
            }                                                                                                      \
            __syncthreads();                                                                                       \
        } else {                                                                                                   \
        _Pragma("unroll") for (int bp = 0; bp < NP; ++bp)                                                          \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) {                                                       \
                const int m = mt * BM + wm * 32 + j * 16 + fr;                                                     \
                const f32x4 lo = acc[2 * bp][j], hi = acc[2 * bp + 1][j];                                          \
                const float v[8] = {lo[0] + bA[bp].x, lo[1] + bA[bp].y, lo[2] + bA[bp].z, lo[3] + bA[bp].w,        \
                                    hi[0] + bB[bp].x, hi[1] + bB[bp].y, hi[2] + bB[bp].z, hi[3] + bB[bp].w};       \
                u32x4 r = u32x4{0u, 0u, 0u, 0u};                                                                   \
                if constexpr (RES) r = RC_[bp * TM + j];                                                           \
                u32x4 o;                                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                    float v0 = v[2 * e], v1 = v[2 * e + 1];                                                        \
                    if constexpr (RES) { v0 += from_h<F16>((u16)(r[e] & 0xffffu)); v1 += from_h<F16>((u16)(r[e] >> 16)); } \
                    if (p.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }                                  \
                    o[e] = pack2_h<F16>(v0, v1);                              \
                }                                                                                                  \
                int vo = (m * p.Cout + cbase[bp]) * 2;                                                             \
                if (p.out_blk) vo = ((m >> 4) * (p.Cout >> 3) + (cbase[bp] >> 3)) * 256 + (m & 15) * 16;           \
                vo = (m < p.M && cbase[bp] >= 0) ? vo : OOB;                                                       \
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo, 0, 0);   /* 32-byte pieces: nt would give up write combining (PMC: +30 % written) */ \
            }                                                                                                      \
        }                                                                                                          \
        mt = mtn; ac[0] = an[0]; ac[1] = an[1];                                                                    \
    }

    int mt = mgroup;
    ac[0] = x_off(mt * BM + a_row[0], 0); ac[1] = x_off(mt * BM + a_row[1], 1);
    EX_LOADX(0, ac, 0);
    if constexpr (KS > 1) EX_LOADX(1, ac, 1);
    if constexpr (KS > 2) EX_LOADX(2, ac, 2);
    EX_LOADRES(rA, mt);
    __syncthreads();                                                   // weights visible
    while (mt < p.m_tiles) {
        EX_TILE(rA, rB, 0);
        if (mt >= p.m_tiles) break;
        EX_TILE(rB, rA, (KS & 1));
    }
#undef EX_TILE
#undef EX_MATH
#undef EX_LOADRES
#undef EX_STOREX
#undef EX_LOADX
}

template <int KS, int BN, bool F16, bool RES, int STRIDE>
static pvr_status launch_expand_inst(ExpP &p, hipStream_t stream) {
    constexpr int lds = KS * BN * 128 + 2 * 64 * 128;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_expand_kernel<KS, BN, F16, RES, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    hipLaunchKernelGGL((conv_expand_kernel<KS, BN, F16, RES, STRIDE>), dim3(p.n_mgroups * p.n_tiles), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

template <int KS, int BN>
static pvr_status launch_expand_shape(ExpP &p, bool res, int stride, int dtype, hipStream_t stream) {
    if (dtype == PVR_F16) {
        if (res) return launch_expand_inst<KS, BN, true, true, 1>(p, stream);
        return stride == 1 ? launch_expand_inst<KS, BN, true, false, 1>(p, stream) : launch_expand_inst<KS, BN, true, false, 2>(p, stream);
    }
    if (res) return launch_expand_inst<KS, BN, false, true, 1>(p, stream);
    return stride == 1 ? launch_expand_inst<KS, BN, false, false, 1>(p, stream) : launch_expand_inst<KS, BN, false, false, 2>(p, stream);
}

// couts per block for a given K: the [BN][K] weight tile stays in LDS (<= 64 KB)
static int expand_bn(int cin, int cout) {
    if (cin == 64) return cout % 128 == 0 ? 128 : (cout == 64 ? 64 : 0);
    if (cin == 256) return cout % 128 == 0 ? 128 : 0;
    if (cin == 512) return cout % 64 == 0 ? 64 : 0;
    return 0;
}

// PVR_CONV_EXPAND=0 keeps these convolutions on conv_igemm / conv_pp256 (A/B runs; bit-identical)
bool conv_expand_supported(int64_t M, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32, bool has_res) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("PVR_CONV_EXPAND"); on = e ? atoi(e) : 1; }
    if (!on || out_f32 != 0 || relu > 1 || kh != 1 || kw != 1 || pad != 0 || (stride != 1 && stride != 2)) return false;
    if (has_res && stride != 1) return false;
    if (cin == 512 && !has_res) return false;        // (layer3.0.downsample: 52.6 GFLOP on 154 MB is MFMA-bound; conv_pp256 is faster, measured)
    if (stride == 2 && ((h | w) & 1)) return false;
    const int bn = expand_bn(cin, cout);
    if (!bn) return false;
    const int64_t tiles = (M + 63) / 64, n_tiles = cout / bn;
    // persistent grid of 512 blocks (two per CU): worth it only when every block gets several pixel tiles
    return n_tiles <= 64 && 512 % n_tiles == 0 && (512 / n_tiles) % 8 == 0 && tiles >= 4 * (512 / n_tiles) && M * (int64_t)cout * 2 < 0x7ffffff0ll &&
           M * (int64_t)cin * 2 * stride * stride < 0x7ffffff0ll;
}

static long long g_expand_launches = 0;
long long conv_expand_launches() { return g_expand_launches; }

// M = output pixels (n * ho * wo); h, w = input height / width (used for stride 2)
pvr_status launch_conv_expand(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                              int cout, int stride, int relu, int dtype, hipStream_t stream, int out_blk) {
    // the blocked (P16C8) output layout exists in the per-pair epilogue only; the full-line epilogue (cin 256 / 512 instances) writes NHWC
    PVR_REQUIRE(!out_blk || cin == 64, "conv_expand: blocked output is built for the cin = 64 instances only (cin = %d)", cin);
    ExpP p;
    p.out_blk = out_blk;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.res = (const u16 *)res; p.bias = bias; p.out = (u16 *)out;
    p.H = h; p.W = w; p.Ho = h / stride; p.Wo = w / stride;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    p.M = (int)M; p.K = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64; p.act = relu;
    const int bn = expand_bn(cin, cout);
    ++g_expand_launches;
    p.n_tiles = cout / bn;
    p.n_mgroups = 512 / p.n_tiles;
    p.m_tiles = (int)((M + 63) / 64);
    p.in_bytes = (unsigned)((int64_t)n * h * w * cin * 2); p.w_bytes = (unsigned)((int64_t)p.CoutPad * cin * 2); p.out_bytes = (unsigned)(M * cout * 2);
    if (cin == 64) return bn == 128 ? launch_expand_shape<1, 128>(p, res != nullptr, stride, dtype, stream) : launch_expand_shape<1, 64>(p, res != nullptr, stride, dtype, stream);
    if (cin == 256) return launch_expand_shape<4, 128>(p, res != nullptr, stride, dtype, stream);
    return launch_expand_shape<8, 64>(p, res != nullptr, stride, dtype, stream);
}

}  // namespace pvr
