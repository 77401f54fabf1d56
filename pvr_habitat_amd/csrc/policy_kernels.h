// Device kernels of the BC policy path (fp32).  Included only by policy.hip.
//
// Reference arithmetic: src/models.py:57-89 (PolicyNet.forward), main_bc_2.py:211-227 (loss, clip, RMSprop).
// Everything here is deterministic: no float atomics; every reduction has a fixed order, so two runs give
// bit-identical parameters (the reference's own run-to-run behaviour with cudnn.deterministic=True).
//
// Matrix work uses the f32-input MFMA v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, MI355X_MICROARCH.md
// "Matrix cores"): lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]; D col = l&15, row = 4*(l>>4)+reg.
#pragma once
#include "common.h"
#include "sample_rng.h"

namespace pvr {

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------
// fp32 GEMM, 64x64x32 tiles, 2x2 waves, 2-stage LDS pipeline.
//   C[m][n] = sum_k A'(m,k) * B'(n,k)  (+ bias[n]) (relu) (zero where mask[m][n] <= 0)
//   ATR: A stored [K][M] (lda = row stride) else [M][K];  BTR: B stored [K][N] else [N][K].
// Used for: FC layers and LSTM input projections (NT), dX = dY W (NN), dW = dY^T X (TN).
// ---------------------------------------------------------------------------------------------------------
struct GemmP {
    const float *A, *B, *bias, *mask;
    float *C;
    int M, N, K, lda, ldb, ldc, relu;
    // split-K (blockIdx.y = slice): slice s multiplies k in [s*K, (s+1)*K) of the full problem into C + s*c_stride (K is the slice length);
    // used when M*N alone gives too few blocks (BPTT chunk GEMMs: M = 400, K = 4096); the slices are summed in a fixed order afterwards
    long long a_kstride, b_kstride, c_stride;
    unsigned a_bytes = 0, b_bytes = 0;      // extent of A / B from the (slice's) base pointer: gemm_bf16x3_kernel's buffer resources
};

// BM x BN in {64, 128}: the wave grid stays 2 x 2, a wave owns (BM/2) x (BN/2) outputs = (BM/32) x (BN/32) MFMA tiles.  Every output
// is the same ascending-k fma chain whatever the tile (the choice never changes a result), so the host picks per shape
// (policy.hip gemm(): tile quantisation on 256 CUs against operand re-reads; PVR_GEMM_TILE forces one).
template <bool ATR, bool BTR, int BM = 64, int BN = 64>
static __global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    constexpr int BK = 32;
    constexpr int LDN = 36;                 // [rows][k] stride (floats) of a k-contiguous operand tile
    constexpr int LDTA = BM + 16, LDTB = BN + 16;   // [k][rows] stride of a row-contiguous operand tile (== 16 mod 32: conflict-free)
    constexpr int TA = ATR ? BK * LDTA : BM * LDN, TB = BTR ? BK * LDTB : BN * LDN;
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;      // float4 loads per thread and K tile
    constexpr int TM = BM / 32, TN = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float gsm[];       // [2 stages][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (swz / n_tiles) * BM, n0 = (swz % n_tiles) * BN;
    const int nk = (p.K + BK - 1) / BK;
    p.A += (long long)blockIdx.y * p.a_kstride; p.B += (long long)blockIdx.y * p.b_kstride; p.C += (long long)blockIdx.y * p.c_stride;

    f32x4 ra[NA], rb[NB];
    auto g_load = [&](int kt_) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i;
            if (!ATR) {
                const int gm = m0 + (idx >> 3), gk = kt_ * BK + (idx & 7) * 4;
                ra[i] = (gm < p.M && gk < p.K) ? *reinterpret_cast<const f32x4 *>(p.A + (size_t)gm * p.lda + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int gk = kt_ * BK + idx / (BM / 4), gm = m0 + (idx % (BM / 4)) * 4;
                ra[i] = (gm < p.M && gk < p.K) ? *reinterpret_cast<const f32x4 *>(p.A + (size_t)gk * p.lda + gm) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            if (!BTR) {
                const int gn = n0 + (idx >> 3), gk = kt_ * BK + (idx & 7) * 4;
                rb[i] = (gn < p.N && gk < p.K) ? *reinterpret_cast<const f32x4 *>(p.B + (size_t)gn * p.ldb + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int gk = kt_ * BK + idx / (BN / 4), gn = n0 + (idx % (BN / 4)) * 4;
                rb[i] = (gn < p.N && gk < p.K) ? *reinterpret_cast<const f32x4 *>(p.B + (size_t)gk * p.ldb + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto g_store = [&](int buf_) {
        float *sa = gsm + buf_ * (TA + TB), *sb = sa + TA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i;
            float *da = !ATR ? &sa[(idx >> 3) * LDN + (idx & 7) * 4] : &sa[(idx / (BM / 4)) * LDTA + (idx % (BM / 4)) * 4];
            *reinterpret_cast<f32x4 *>(da) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            float *db = !BTR ? &sb[(idx >> 3) * LDN + (idx & 7) * 4] : &sb[(idx / (BN / 4)) * LDTB + (idx % (BN / 4)) * 4];
            *reinterpret_cast<f32x4 *>(db) = rb[i];
        }
    };

    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    g_load(0);
    g_store(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) g_load(kt + 1);
        const float *As = gsm + cur * (TA + TB), *Bs = As + TA;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int k = ks * 4 + fq;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 16 + fr;
                a[i] = !ATR ? As[row * LDN + k] : As[k * LDTA + row];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / 2) + j * 16 + fr;
                b[j] = !BTR ? Bs[row * LDN + k] : Bs[k * LDTB + row];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f32(a[i], b[j], acc[i][j]);
        }
        if (more) g_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.relu) v = v < 0.f ? 0.f : v;                      // (not fmaxf: torch's ReLU hands a NaN on, fmaxf(NaN, 0) = 0 would hide it)
                if (p.mask && p.mask[(size_t)m * p.ldc + n] <= 0.f) v = 0.f;
                p.C[(size_t)m * p.ldc + n] = v;
            }
    }
}
template <int BM, int BN, bool ATR, bool BTR> constexpr size_t gemm_f32_lds() {
    return (size_t)2 * ((ATR ? 32 * (BM + 16) : BM * 36) + (BTR ? 32 * (BN + 16) : BN * 36)) * sizeof(float);
}

#ifdef PVR_EXPERIMENTS   // round-3 experiment, as accurate and no faster than gemm_f32_kernel (profiles/experiments/r03_gemm_split_bf16.txt)
// ---------------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix pipe (round 3): every fp32 operand value is split EXACTLY into three bf16 terms
//     a = a1 + a2 + a3,   a1 = top 16 bits of a,  a2 = top 16 bits of (a - a1),  a3 = a - a1 - a2   (8 + 8 + 8 significand bits)
// and C accumulates the six products of combined order <= 4 in fp32:  a1 b1  |  a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1.
// bf16 x bf16 products are exact in fp32; the dropped terms (a2 b3, a3 b2, a3 b3) are <= 2^-22 |a b| each, the size of the fp32
// rounding of one product in gemm_f32_kernel's fma chain.  The leading products go to their own accumulator and the corrections
// (2^-8 of them) to a second one, added once at the end - the small terms are not rounded away against a large running sum.
// Measured against an fp64 product the result is as accurate as gemm_f32_kernel's (tests/test_gpu_policy.py).  Six bf16 MFMAs
// (16 cycles per 16x16x32) replace eight fp32 MFMAs (32 cycles per 16x16x4): 96 against 256 matrix-pipe cycles per 16x16x32 block.
// Non-finite inputs: a NaN stays a NaN; +-Inf becomes a NaN (Inf - Inf in the split) where gemm_f32_kernel hands on the Inf.
//
// Tile BM x BN x 32, 2 x 2 waves.  A thread converts what it loaded (f32x4: four consecutive k of a row, or, for an operand stored
// [K][rows], the same four k of four consecutive rows gathered from four loads) and writes 8-byte pieces of the three planes;
// LDS planes are [rows][32 k] bf16 = 64-byte rows with the chunk swizzle c ^ 2*((row>>2)&1) (conflict-free ds_read_b128 fragment reads,
// see conv_w4.hip).  Two LDS stages, two register sets: a tile's loads are requested two tiles ahead.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN> constexpr size_t gemm_bf16x3_lds() { return (size_t)2 * 3 * (BM + BN) * 64; }

__device__ __forceinline__ void split3(float a, unsigned &h1, unsigned &h2, unsigned &h3) {
    h1 = __builtin_bit_cast(unsigned, a);
    const float r1 = a - __builtin_bit_cast(float, h1 & 0xffff0000u);
    h2 = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, h2 & 0xffff0000u);
    h3 = __builtin_bit_cast(unsigned, r2);
}
// (hi16(x1) << 16) | hi16(x0)
__device__ __forceinline__ unsigned pack_hi(unsigned x0, unsigned x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

template <bool ATR, bool BTR, int BM, int BN>
static __global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmP p) {
    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
    constexpr int BK = 32;
    constexpr int PA = BM * 64, PB = BN * 64, STG = 3 * (PA + PB);          // bytes per plane / per stage
    constexpr int NA = ATR ? 4 : BM / 32, NB = BTR ? 4 : BN / 32;           // f32x4 loads per thread and K tile
    constexpr int TM = BM / 32, TN = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (swz / n_tiles) * BM, n0 = (swz % n_tiles) * BN;
    const int nk = (p.K + BK - 1) / BK;
    p.A += (long long)blockIdx.y * p.a_kstride; p.B += (long long)blockIdx.y * p.b_kstride; p.C += (long long)blockIdx.y * p.c_stride;

    f32x4 ra[2][NA], rb[2][NB];         // two register sets: tile kt + 2 is requested while tile kt is multiplied (one iteration of latency cover)
    // k-contiguous operand: load i of a thread = row (tid + 256 i) / 8, k quad (tid + 256 i) % 8
    // row-contiguous ([K][rows]) operand: the thread owns k quad kq = tid / (R/4) and rows 4 rq .. 4 rq + 3, rq = tid % (R/4) (rows fastest:
    //   coalesced 512-byte runs per k row; its 8-byte LDS writes conflict 8-way, and the k-fastest mapping that avoids that was 35 % SLOWER
    //   on the weight-gradient shapes: the loads matter more); load j = k row 4 kq + j (four loads; at
    //   R = 64 the threads 128.. repeat what threads 0..127 do: identical values to identical addresses)
    // Branch-free: 16-byte buffer loads; a row or k quad outside the matrix goes to an out-of-range offset and reads zeros (a
    // predicated load makes hipcc carry each register set through the loop as one tuple and copy it at every join, see conv_w4.hip).
    constexpr int OOB = 0x7ffffff0;
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, p.a_bytes, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.B), 0, p.b_bytes, 0x00020000);
    auto g_load_one = [&](__amdgpu_buffer_rsrc_t rs, int ld, int rows, int r0, bool tr, int R, f32x4 *dst, int n_ld, int kt_) {
        if (!tr) {
            const int gk = kt_ * BK + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < n_ld; ++i) {
                const int gr = r0 + ((tid + 256 * i) >> 3);
                dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (gr < rows && gk < p.K) ? (gr * ld + gk) * 4 : OOB, 0, 0));
            }
        } else {
            const int kq = (tid / (R / 4)) & 7, gr = r0 + (tid % (R / 4)) * 4;      // (R = 64: threads 128.. repeat the work of threads 0..127)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = kt_ * BK + kq * 4 + j;
                dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (gr < rows && gk < p.K) ? (gk * ld + gr) * 4 : OOB, 0, 0));
            }
        }
    };
    auto cvt_store_quad = [&](char *plane0, int plane_bytes, int row, int kq, float x0, float x1, float x2, float x3) {
        unsigned a1[4], a2[4], a3[4];
        split3(x0, a1[0], a2[0], a3[0]); split3(x1, a1[1], a2[1], a3[1]); split3(x2, a1[2], a2[2], a3[2]); split3(x3, a1[3], a2[3], a3[3]);
        char *d = plane0 + row * 64 + ((((kq >> 1) ^ (((row >> 2) & 1) << 1))) << 4) + (kq & 1) * 8;
        *reinterpret_cast<u32x2_ *>(d) = u32x2_{pack_hi(a1[0], a1[1]), pack_hi(a1[2], a1[3])};
        *reinterpret_cast<u32x2_ *>(d + plane_bytes) = u32x2_{pack_hi(a2[0], a2[1]), pack_hi(a2[2], a2[3])};
        *reinterpret_cast<u32x2_ *>(d + 2 * plane_bytes) = u32x2_{pack_hi(a3[0], a3[1]), pack_hi(a3[2], a3[3])};
    };
    auto g_store_one = [&](char *plane0, int plane_bytes, bool tr, int R, const f32x4 *src, int n_ld) {
        if (!tr) {
#pragma unroll
            for (int i = 0; i < n_ld; ++i) {
                const int idx = tid + 256 * i;
                cvt_store_quad(plane0, plane_bytes, idx >> 3, idx & 7, src[i][0], src[i][1], src[i][2], src[i][3]);
            }
        } else {
            const int kq = (tid / (R / 4)) & 7, rq = tid % (R / 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) cvt_store_quad(plane0, plane_bytes, rq * 4 + rr, kq, src[0][rr], src[1][rr], src[2][rr], src[3][rr]);
        }
    };
    auto g_load = [&](int kt_, int set) {
        g_load_one(rs_a, p.lda, p.M, m0, ATR, BM, ra[set], NA, kt_);
        g_load_one(rs_b, p.ldb, p.N, n0, BTR, BN, rb[set], NB, kt_);
    };
    auto g_store = [&](int buf_, int set) {
        char *st = bsm + buf_ * STG;
        g_store_one(st, PA, ATR, BM, ra[set], NA);
        g_store_one(st + 3 * PA, PB, BTR, BN, rb[set], NB);
    };

    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int frag = fr * 64 + ((fq ^ (((fr >> 2) & 1) << 1)) << 4);
    f32x4 hi[TM][TN], lo[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto k_tile = [&](int cur) {
        const char *As = bsm + cur * STG + (wm * (BM / 2)) * 64 + frag, *Bs = bsm + cur * STG + 3 * PA + (wn * (BN / 2)) * 64 + frag;
        bf16x8 b1[TN], b2[TN], b3[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            b1[j] = *reinterpret_cast<const bf16x8 *>(Bs + j * 1024);
            b2[j] = *reinterpret_cast<const bf16x8 *>(Bs + PB + j * 1024);
            b3[j] = *reinterpret_cast<const bf16x8 *>(Bs + 2 * PB + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(As + i * 1024);
            const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(As + PA + i * 1024);
            const bf16x8 a3 = *reinterpret_cast<const bf16x8 *>(As + 2 * PA + i * 1024);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                hi[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[j], hi[i][j], 0, 0, 0);
                f32x4 l = lo[i][j];
                l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1[j], l, 0, 0, 0);      // smallest first
                l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2[j], l, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3[j], l, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1[j], l, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2[j], l, 0, 0, 0);
                lo[i][j] = l;
            }
        }
    };
    // tile kt is multiplied from LDS stage kt & 1 while tile kt + 1 waits in register set (kt + 1) & 1 (requested one iteration ago)
    // and tile kt + 2 is requested into set kt & 1 (its previous content, tile kt, went to LDS one iteration ago)
    // (requests past the last tile read zeros and the stores of those zeros go to a stage nobody reads again: no conditions)
    g_load(0, 0);
    g_store(0, 0);
    g_load(1, 1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        g_load(kt + 2, 0);
        k_tile(0);
        g_store(1, 1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        g_load(kt + 3, 1);
        k_tile(1);
        g_store(0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = (hi[i][j][r] + lo[i][j][r]) + bv;
                if (p.relu) v = v < 0.f ? 0.f : v;
                if (p.mask && p.mask[(size_t)m * p.ldc + n] <= 0.f) v = 0.f;
                p.C[(size_t)m * p.ldc + n] = v;
            }
    }
}
#endif  // PVR_EXPERIMENTS

// out = sum over the split-K slices (ascending), then the epilogue the GEMM kernel would have applied
static __global__ __launch_bounds__(256) void splitk_sum_kernel(const float *__restrict__ part, float *__restrict__ out, const float *__restrict__ bias,
                                                         const float *__restrict__ mask, int S, size_t n, int N, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s_ = 0; s_ < S; ++s_) v += part[(size_t)s_ * n + i];
        if (bias) v += bias[i % N];
        if (relu) v = v < 0.f ? 0.f : v;
        if (mask && mask[i] <= 0.f) v = 0.f;
        out[i] = v;
    }
}

// [R][C] -> [C][R]
static __global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int R, int C) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) t[i][tx] = in[(size_t)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) out[(size_t)(c0 + i) * R + r0 + tx] = t[tx][i];
}

// ---------------------------------------------------------------------------------------------------------
// column reductions over the N = T*B rows (fixed order: 8 row groups per column, then an LDS tree)
//   MODE 0: out[c] = sum_r X[r][c]
//   MODE 1: BN statistics (models.py:31-34, training): mean, 1/sqrt(var_biased+eps), running-stat update
//   MODE 2: BN affine grads: dgamma[c] = sum_r dY[r][c]*(X[r][c]-mean[c])*invstd[c], dbeta[c] = sum_r dY[r][c]
// ---------------------------------------------------------------------------------------------------------
struct ColP {
    const float *X, *dY, *mean_in, *invstd_in;
    float *out0, *out1, *running_mean, *running_var;
    long long *nbt;
    int R, C;
    int out_stride;   // MODE 0 with two outputs (bias_ih and bias_hh get the same sum): out1 optional
};

template <int MODE>
static __global__ __launch_bounds__(256) void colreduce_kernel(ColP p) {
    __shared__ float s0[8][33], s1[8][33];
    const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const bool ok = c < p.C;
    float a0 = 0.f, a1 = 0.f;
    if (MODE == 0) {
        if (ok) for (int r = g; r < p.R; r += 8) a0 += p.X[(size_t)r * p.C + c];
    } else if (MODE == 1) {
        if (ok) for (int r = g; r < p.R; r += 8) a0 += p.X[(size_t)r * p.C + c];
    } else if (MODE == 3) {          // centred sum of squares around a given mean (SyncBN second pass)
        if (ok) {
            const float mu = p.mean_in[c];
            for (int r = g; r < p.R; r += 8) { const float d = p.X[(size_t)r * p.C + c] - mu; a0 += d * d; }
        }
    } else {
        if (ok) {
            const float mu = p.mean_in[c], is = p.invstd_in[c];
            for (int r = g; r < p.R; r += 8) {
                const float dy = p.dY[(size_t)r * p.C + c];
                a0 += dy * ((p.X[(size_t)r * p.C + c] - mu) * is);
                a1 += dy;
            }
        }
    }
    s0[g][cx] = a0; s1[g][cx] = a1;
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t0 += s0[i][cx]; t1 += s1[i][cx]; }
    if (MODE == 0 || MODE == 3) {
        if (ok && g == 0) { p.out0[c] = t0; if (p.out1) p.out1[c] = t0; }
    } else if (MODE == 2) {
        if (ok && g == 0) { p.out0[c] = t0; p.out1[c] = t1; }
    } else {
        const float mean = t0 / (float)p.R;
        __syncthreads();
        float v = 0.f;
        if (ok) for (int r = g; r < p.R; r += 8) { const float d = p.X[(size_t)r * p.C + c] - mean; v += d * d; }
        s0[g][cx] = v;
        __syncthreads();
        if (ok && g == 0) {
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) vs += s0[i][cx];
            const float var_b = vs / (float)p.R;
            p.out0[c] = mean;
            p.out1[c] = 1.0f / sqrtf(var_b + 1e-5f);
            // momentum 0.1, unbiased variance in the running estimate (torch BatchNorm1d)
            p.running_mean[c] = 0.9f * p.running_mean[c] + 0.1f * mean;
            p.running_var[c] = 0.9f * p.running_var[c] + 0.1f * (var_b * (float)p.R / (float)(p.R - 1));
            if (c == 0 && p.nbt) *p.nbt += 1;
        }
    }
}

// Two-stage column reductions for tall matrices (R >= 512): colreduce_kernel has only C/32 blocks, each walking all R rows with
// 128-byte row segments (55-100 us for 1600 x 4096 fp32, rocprofv3) - here a (column block, row group) grid reads 16 bytes per lane
// and a second launch adds the row-group partials in ascending order (fixed order: deterministic).
//   MODE 0: sum x     MODE 3: sum (x - mean[c])^2     MODE 2: out0 = sum dy * (x - mean[c]) * invstd[c], out1 = sum dy
template <int MODE>
static __global__ __launch_bounds__(256) void colpart_kernel(ColP p, float *__restrict__ part, int rpg) {
    __shared__ f32x4 s0[8][33], s1[8][33];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cq) * 4, g = blockIdx.y;
    const bool ok = c < p.C;
    const int r0 = g * rpg, r1 = r0 + rpg < p.R ? r0 + rpg : p.R;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
        f32x4 mu = f32x4{0.f, 0.f, 0.f, 0.f}, is = f32x4{1.f, 1.f, 1.f, 1.f};
        if (MODE != 0) mu = *reinterpret_cast<const f32x4 *>(p.mean_in + c);
        if (MODE == 2) is = *reinterpret_cast<const f32x4 *>(p.invstd_in + c);
        for (int r = r0 + rl; r < r1; r += 8) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(p.X + (size_t)r * p.C + c);
            if (MODE == 0) a0 += x;
            else if (MODE == 3) { const f32x4 d = x - mu; a0 += d * d; }
            else { const f32x4 dy = *reinterpret_cast<const f32x4 *>(p.dY + (size_t)r * p.C + c); a0 += dy * ((x - mu) * is); a1 += dy; }
        }
    }
    s0[rl][cq] = a0; if (MODE == 2) s1[rl][cq] = a1;
    __syncthreads();
    if (rl == 0 && ok) {
        f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0 += s0[i][cq]; if (MODE == 2) t1 += s1[i][cq]; }
        *reinterpret_cast<f32x4 *>(part + (size_t)g * p.C + c) = t0;
        if (MODE == 2) *reinterpret_cast<f32x4 *>(part + (size_t)(gridDim.y + g) * p.C + c) = t1;
    }
}
// out0[c] (= out1[c] when TWO == 0 and out1 given) = sum_g part[g][c];  TWO: out1[c] = sum_g part[G + g][c]
template <int TWO>
static __global__ __launch_bounds__(256) void colfinal_kernel(const float *__restrict__ part, float *__restrict__ out0, float *__restrict__ out1, int G, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float t0 = 0.f, t1 = 0.f;
    for (int g = 0; g < G; ++g) { t0 += part[(size_t)g * C + c]; if (TWO) t1 += part[(size_t)(G + g) * C + c]; }
    out0[c] = t0;
    if (TWO) out1[c] = t1; else if (out1) out1[c] = t0;
}

// BatchNorm1d affine gradients without d(loss)/d(a0) (PolicyNet: nothing below the BatchNorm needs a gradient, so the [N][O] product
// dz1 W1 - 13.4 GFLOP at obs 4096, more than 6 % of the iteration - only ever fed two column sums).  With a0 = gamma * xhat + beta:
//     S      = dz1^T xhat                       (the fc1 weight-gradient GEMM, on xhat instead of a0: same cost)
//     dW1    = gamma * S + beta * db1^T         (db1 = column sums of dz1 = the fc1 bias gradient)
//     dgamma = sum_j W1[j][c] * S[j][c]         dbeta = sum_j W1[j][c] * db1[j]
// (sum_r (dz1 W1)[r][c] xhat[r][c] = sum_j W1[j][c] sum_r dz1[r][j] xhat[r][c]; no division by gamma anywhere.)
// bn_xhat_kernel writes xhat; bn_fold_grads_kernel turns S into dW1 in place and emits the per-row-group partials of the two sums
// (row groups summed in ascending order by colfinal_kernel<1>: fixed order).
static __global__ __launch_bounds__(256) void bn_xhat_kernel(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                      float *__restrict__ xhat, size_t total4, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % (size_t)C);
        const f32x4 xv = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + c), is = *reinterpret_cast<const f32x4 *>(invstd + c);
        reinterpret_cast<f32x4 *>(xhat)[i] = (xv - mu) * is;
    }
}
// grid (C / 128, G row groups of rpg rows); S_dW [R][C] in: S, out: dW1
static __global__ __launch_bounds__(256) void bn_fold_grads_kernel(float *__restrict__ S_dW, const float *__restrict__ W1, const float *__restrict__ db1,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ part,
                                                            int R, int C, int rpg) {
    __shared__ f32x4 s0[8][33], s1[8][33];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cq) * 4, g = blockIdx.y;
    const bool ok = c < C;
    const int r0 = g * rpg, r1 = r0 + rpg < R ? r0 + rpg : R;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c), be = *reinterpret_cast<const f32x4 *>(beta + c);
        for (int r = r0 + rl; r < r1; r += 8) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(S_dW + (size_t)r * C + c);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(W1 + (size_t)r * C + c);
            const float d = db1[r];
            a0 += wv * sv;
            a1 += wv * d;
            *reinterpret_cast<f32x4 *>(S_dW + (size_t)r * C + c) = ga * sv + be * d;
        }
    }
    s0[rl][cq] = a0; s1[rl][cq] = a1;
    __syncthreads();
    if (rl == 0 && ok) {
        f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0 += s0[i][cq]; t1 += s1[i][cq]; }
        *reinterpret_cast<f32x4 *>(part + (size_t)g * C + c) = t0;
        *reinterpret_cast<f32x4 *>(part + (size_t)(gridDim.y + g) * C + c) = t1;
    }
}

// y = (x - mean) * invstd * gamma + beta  (training: batch stats; eval: running stats, invstd computed here)
static __global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd_or_var, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ y, size_t total4,
                                                       int C, int is_var) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % (size_t)C);
        const f32x4 xv = reinterpret_cast<const f32x4 *>(x)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float is = is_var ? 1.0f / sqrtf(invstd_or_var[c + e] + 1e-5f) : invstd_or_var[c + e];
            o[e] = (xv[e] - mean[c + e]) * is * gamma[c + e] + beta[c + e];
        }
        reinterpret_cast<f32x4 *>(y)[i] = o;
    }
}

static __global__ __launch_bounds__(256) void notdone_kernel(const uint8_t *__restrict__ done, float *__restrict__ nd, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) nd[i] = fabsf(1.0f - (done[i] ? 1.0f : 0.0f));      // models.py:66
}

// ---------------------------------------------------------------------------------------------------------
// LSTM forward, one launch per (layer, timestep): gates = Gx[t] + (nd*h_prev) W_hh^T, cell update.
// Block = 4 hidden units (16 gate rows i,f,g,o); wave w contracts k in [w*H/4, (w+1)*H/4) on the f32 MFMA.
// G is overwritten in place with the activated gates (saved for BPTT).
// ---------------------------------------------------------------------------------------------------------
struct LstmFwdP {
    float *G;                  // [B][4H] at t: in = input projection (+ both biases), out = i,f,g,o
    const float *h_prev, *c_prev, *nd, *W, *bhh;   // [B][H], [B][H], [B], [4H][H], [4H]
    float *h_out, *c_out;      // [B][H]
    int B, H;
};

static __device__ __forceinline__ void lstm_fwd_step_body(const LstmFwdP &p) {
    __shared__ float part[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int u0 = blockIdx.x * 4, H = p.H;
    const float *wrow = p.W + (size_t)((fr >> 2) * H + u0 + (fr & 3)) * H;
    const int kbeg = wave * (H / 4);
    for (int b0 = 0; b0 < p.B; b0 += 16) {
        const int b = b0 + fr;
        const bool bok = b < p.B;
        const float ndb = bok ? p.nd[b] : 0.f;
        const float *hrow = p.h_prev + (size_t)(bok ? b : 0) * H;
        // operands of the cell update (threads < 64): issued now so their latency overlaps the weight stream
        const int tb = tid >> 2, tj = tid & 3, tbi = b0 + tb;
        const bool tok = tid < 64 && tbi < p.B;
        float gx[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f, ndt = 0.f;
        if (tok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gx[g] = p.G[(size_t)tbi * 4 * H + g * H + u0 + tj] + p.bhh[g * H + u0 + tj];
            cprev = p.c_prev[(size_t)tbi * H + u0 + tj];
            ndt = p.nd[tbi];
        }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        // every load of the wave's K range is issued before the MFMA chain, so the L2 / Infinity-Cache latency of
        // the weight stream is paid once, not once per 16-k chunk (the launch is latency-bound)
        for (int c0 = 0; c0 < H / 64; c0 += 16) {
            f32x4 hv[16], wv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int k = kbeg + (c0 + c) * 16 + fq * 4;
                hv[c] = *reinterpret_cast<const f32x4 *>(hrow + k);
                wv[c] = *reinterpret_cast<const f32x4 *>(wrow + k);
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the 32 loads ahead of the MFMA chain (hipcc otherwise
                                                         // re-serialises them: load, vmcnt(0), 2 MFMAs, load, ...)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                hv[c] *= ndb;                            // state * notdone (models.py:69); 0 for padded batch rows
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma_f32(hv[c][e], wv[c][e], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];
        __syncthreads();
        if (tok) {
            float s[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                s[g] = ((part[0][tb][g * 4 + tj] + part[1][tb][g * 4 + tj]) + (part[2][tb][g * 4 + tj] + part[3][tb][g * 4 + tj])) + gx[g];
            const float ig = sigmoidf_(s[0]), fg = sigmoidf_(s[1]), gg = tanhf(s[2]), og = sigmoidf_(s[3]);
            const float cm = ndt * cprev;
            const float c = fg * cm + ig * gg;
            const float h = og * tanhf(c);
            float *g = p.G + (size_t)tbi * 4 * H + u0 + tj;
            g[0] = ig; g[H] = fg; g[2 * H] = gg; g[3 * H] = og;
            p.c_out[(size_t)tbi * H + u0 + tj] = c;
            p.h_out[(size_t)tbi * H + u0 + tj] = h;
        }
        __syncthreads();
    }
}

static __global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmFwdP p) { lstm_fwd_step_body(p); }

// Chunked layer wavefront: blockIdx.y selects one of two independent (layer, step) jobs - layer 0 at step t and layer 1 one chunk
// behind - so the two recurrences share launches instead of alternating them (same per-job arithmetic, bit-identical).
struct LstmFwd2P { LstmFwdP j[2]; int active[2]; };
static __global__ __launch_bounds__(256) void lstm_fwd_step2_kernel(LstmFwd2P pp) {
    if (!pp.active[blockIdx.y]) return;
    lstm_fwd_step_body(pp.j[blockIdx.y]);
}

// ---------------------------------------------------------------------------------------------------------
// Persistent forward recurrence: ONE launch runs steps [t0, t1) of one layer.  Same block <-> hidden-unit map, same MFMA
// chain and same reduction order as lstm_fwd_step_kernel (bit-identical results); what changes is what a step costs:
//   * the block's 16 W_hh rows (4 gates x 4 units, 64 KB) stay in registers for the whole sequence;
//   * the cell state of the block's units stays in registers;
//   * a step ends with a grid-wide hand-off instead of a kernel boundary (~11 us of launch + ramp per step before): h_t is
//     stored write-through (sc1), every storing wave drains its stores, one lane adds to an agent-scope counter; consumers
//     poll that counter (relaxed, one lane) and read h_t with sc1 loads only (cdna_hip_programming.md section 6 G16,
//     publish/consume recipe R1 with the all-sc1-loads form, so no acquire fence is needed).
// The grid (H/4 blocks) must be co-resident: 256 blocks of 256 threads on 256 CUs, <= 2 per CU when both layers' lanes run.
// Spins are bounded: on a timeout the block poisons its outputs with NaN (the loss shows it) instead of hanging the GPU.
// ---------------------------------------------------------------------------------------------------------
struct LstmSeqP {
    float *G;                          // layer base [T][B][4H]: in = input projection (+ b_ih), out = activated gates
    const float *h_init, *c_init;      // [B][H]: state before step t0
    const float *nd, *W, *bhh;         // [T][B], [4H][H], [4H]
    float *Hs, *Cs;                    // layer bases [T][B][H]
    unsigned *counter;                 // zeroed before the launch
    int t0, t1, B, H;
    int data_flag;                     // 1: no counter - Hs[t0 .. t1) is pre-filled with 0xFFFFFFFF words and a consumer re-reads its slice of
                                       // h_{t-1} until none of its words is that pattern (the data is its own flag: no atomics, no drain, no block-wide poll)
    int drop_block;                    // -1; test hook (pvr_policy_debug_drop_block): this block leaves at once, so its peers' waits run out
    unsigned *status;                  // host-visible (pinned, fine-grained) status word: a wave whose bounded spin runs out stores LSTM_SEQ_TIMEOUT
                                       // there (a plain system-scope store: PCIe atomics offer no OR); pvr_policy_forward / _step / _status turn it into PVR_ERR_TIMEOUT (policy.hip)
};
constexpr unsigned LSTM_SEQ_TIMEOUT = 1u;
constexpr unsigned LSTM_SEQ_SPINS = 1u << 20;   // ~1-2 s of re-reads: far beyond any scheduling delay of a co-resident grid

static __global__ __launch_bounds__(256) void lstm_fwd_seq_kernel(LstmSeqP p) {
    __shared__ float part[4][16][17];
    __shared__ __attribute__((aligned(16))) float hstage[16][4];
    __shared__ int dead_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int u0 = blockIdx.x * 4, H = p.H, B = p.B;
    if ((int)blockIdx.x == p.drop_block) return;
    if (tid == 0) dead_s = 0;
    __syncthreads();
    const float *wrow = p.W + (size_t)((fr >> 2) * H + u0 + (fr & 3)) * H;
    const int kbeg = wave * (H / 4);
    // weights of this lane for the whole sequence (H = 1024: 16 x float4)
    f32x4 wv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wv[c] = *reinterpret_cast<const f32x4 *>(wrow + kbeg + c * 16 + fq * 4);
    const int tb = tid >> 2, tj = tid & 3;
    float bh[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < 64) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bh[g] = p.bhh[g * H + u0 + tj];
    }
    float creg[4] = {0.f, 0.f, 0.f, 0.f};           // cell state of (batch row b0 + tb, unit u0 + tj) per 16-row batch group
    const unsigned nblk = gridDim.x;
    const size_t hbytes = (size_t)B * H * 4;
    bool dead = false;                               // this wave's spin ran out once (wave-uniform)
    for (int t = p.t0; t < p.t1; ++t) {
        const int s = t - p.t0;
        if (s > 0 && !p.data_flag) {
            if (tid == 0 && !dead_s) {
                unsigned spins = 0;
                while (__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk * (unsigned)s) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4 * LSTM_SEQ_SPINS) {                  // bounded: never hang the GPU on a lost block
                        dead_s = 1;
                        __hip_atomic_store(p.status, LSTM_SEQ_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the sc1 loads below the poll)
        }
        const float *hprev = s == 0 ? p.h_init : p.Hs + (size_t)(t - 1) * B * H;
        const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(hprev), 0, (unsigned)hbytes, 0x00020000);
        float *Gt = p.G + (size_t)t * B * 4 * H;
        int grp = 0;
        for (int b0 = 0; b0 < B; b0 += 16, ++grp) {
            const int b = b0 + fr;
            const bool bok = b < B;
            const float ndb = bok ? p.nd[(size_t)t * B + b] : 0.f;
            const int hoff = ((bok ? b : 0) * H + kbeg + fq * 4) * 4;
            const int tbi = b0 + tb;
            const bool tok = tid < 64 && tbi < B;
            float gx[4] = {0.f, 0.f, 0.f, 0.f}, ndt = 0.f;
            if (tok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gx[g] = Gt[(size_t)tbi * 4 * H + g * H + u0 + tj] + bh[g];
                ndt = p.nd[(size_t)t * B + tbi];
                if (s == 0) creg[grp & 3] = p.c_init[(size_t)tbi * H + u0 + tj];
            }
            f32x4 hv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)                                  // sc1 loads: h_{t-1} was published by other CUs in this launch
                hv[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, hoff, c * 64, 16));
            if (p.data_flag && s > 0 && !dead) {
                // every word of this lane's slice must have been written (0xFFFFFFFF = the pre-fill; a stored h is never that
                // pattern: the store below canonicalises NaNs); the wave re-reads until all of its lanes see data.  Bounded like the
                // counter poll; a wave that ran out once never waits again (`dead`), so a broken launch drains in ~one timeout.
                unsigned spins = 0;
                for (;;) {
                    unsigned m = 0xffffffffu;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const u32x4 w = __builtin_bit_cast(u32x4, hv[c]);
                        const bool written = (w[0] != 0xffffffffu) & (w[1] != 0xffffffffu) & (w[2] != 0xffffffffu) & (w[3] != 0xffffffffu);
                        m &= written ? 0xffffffffu : 0u;
                    }
                    if (__all(m != 0u || !bok)) break;
                    if (++spins > LSTM_SEQ_SPINS) {
                        dead = true; dead_s = 1;
                        if (lane == 0) __hip_atomic_store(p.status, LSTM_SEQ_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        hv[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, hoff, c * 64, 16));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                hv[c] *= ndb;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma_f32(hv[c][e], wv[c][e], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];
            __syncthreads();
            if (tok) {
                float sg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    sg[g] = ((part[0][tb][g * 4 + tj] + part[1][tb][g * 4 + tj]) + (part[2][tb][g * 4 + tj] + part[3][tb][g * 4 + tj])) + gx[g];
                const float ig = sigmoidf_(sg[0]), fg = sigmoidf_(sg[1]), gg = tanhf(sg[2]), og = sigmoidf_(sg[3]);
                const float cm = ndt * creg[grp & 3];
                const float c = fg * cm + ig * gg;
                const float h = og * tanhf(c);
                creg[grp & 3] = c;
                float *g = Gt + (size_t)tbi * 4 * H + u0 + tj;
                g[0] = ig; g[H] = fg; g[2 * H] = gg; g[3 * H] = og;
                p.Cs[((size_t)t * B + tbi) * H + u0 + tj] = c;
                // (h != h: an input NaN may carry any payload, 0xFFFFFFFF included - the stored word must never look like the pre-fill)
                hstage[tb][tj] = (dead_s || h != h) ? __builtin_nanf("") : h;
            }
            __syncthreads();
            if (tid < 16 && b0 + tid < B) {                              // one 16-byte write-through store per batch row
                const f32x4 hv4 = *reinterpret_cast<const f32x4 *>(&hstage[tid][0]);
                const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(p.Hs + (size_t)t * B * H, 0, (unsigned)hbytes, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv4), rs_o, ((b0 + tid) * H + u0) * 4, 0, 16);
            }
        }
        // publish h_t: the storing wave drains its write-through stores, then one lane signals (data_flag: the stores are the signal)
        if (!p.data_flag) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) (void)__hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LSTM backward (BPTT), two launches per (layer, timestep), t descending, 2-D partition of the recurrent product:
//   phase A  lstm_bwd_rec_kernel : partial[kg][b][u] = sum_{k in group kg} dG[t+1][b][k] * W_hh[k][u]
//            grid 16 k-groups x 16 u-groups = 256 blocks; a block reads a 256 x 64 slab of the ORIGINAL [4H][H]
//            weights (64 KB, 256-B row segments) and 16 KB of gate gradients - every CU streams 1/256 of W_hh,
//            instead of 64 blocks each re-reading all 256 KB of dG[t+1]
//   phase B  lstm_bwd_cell_kernel: dh = dh_ext + nd[t+1] * sum_kg partial (fixed order), gate gradients, dc carry
// MFMA roles in phase A: A[i = batch][k] = dG, B[k][j] = W with j <-> u = u0 + 4*fr + e (float4 along u).
// ---------------------------------------------------------------------------------------------------------
struct LstmRecP {
    const float *dG_next, *W;      // [B][4H], [4H][H]
    float *partial;                // [16][B][H]
    int B, H;
};

static __device__ __forceinline__ void lstm_bwd_rec_body(const LstmRecP &p) {
    __shared__ float part[4][4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int H = p.H, K = 4 * p.H;
    const int kg = blockIdx.x >> 4, ug = blockIdx.x & 15;
    const int kper = K / 16, uper = H / 16;                     // 256 and 64 for H = 1024
    const int u0 = ug * uper, kbeg = kg * kper + wave * (kper / 4);
    for (int b0 = 0; b0 < p.B; b0 += 16) {
        const int b = b0 + fr;
        const bool bok = b < p.B;
        const float *drow = p.dG_next + (size_t)(bok ? b : 0) * K;
        for (int ub = 0; ub < uper; ub += 64) {
            f32x4 acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            // 16-k chunks of this wave, four at a time: all 20 loads of a group are issued before its 64 MFMAs, so the L2 latency of
            // the weight slab is paid once per group, not once per chunk (the ISA of the rolled loop waited vmcnt(0) per chunk)
            for (int c0 = 0; c0 < kper / 64; c0 += 4) {
                f32x4 a[4], w[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k0 = kbeg + (c0 + c) * 16;
                    a[c] = *reinterpret_cast<const f32x4 *>(drow + k0 + fq * 4);
#pragma unroll
                    for (int e1 = 0; e1 < 4; ++e1)
                        w[c][e1] = *reinterpret_cast<const f32x4 *>(p.W + (size_t)(k0 + fq * 4 + e1) * H + u0 + ub + fr * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float bmask = bok ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c] *= bmask;
#pragma unroll
                    for (int e1 = 0; e1 < 4; ++e1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = mfma_f32(a[c][e1], w[c][e1][e], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][e][fq * 4 + r][fr] = acc[e][r];
            __syncthreads();
            // 16 batch x 64 u outputs, 4 per thread: u = u0 + ub + 4*j + e
            for (int o = tid; o < 16 * 64; o += 256) {
                const int bb = o >> 6, uu = o & 63, j = uu >> 2, e = uu & 3;
                if (b0 + bb < p.B)
                    p.partial[((size_t)kg * p.B + b0 + bb) * H + u0 + ub + uu] =
                        (part[0][e][bb][j] + part[1][e][bb][j]) + (part[2][e][bb][j] + part[3][e][bb][j]);
            }
            __syncthreads();
        }
    }
}

static __global__ __launch_bounds__(256) void lstm_bwd_rec_kernel(LstmRecP p) { lstm_bwd_rec_body(p); }
struct LstmRec2P { LstmRecP j[2]; int active[2]; };
static __global__ __launch_bounds__(256) void lstm_bwd_rec2_kernel(LstmRec2P pp) {      // chunked layer wavefront (blockIdx.y = job)
    if (!pp.active[blockIdx.y]) return;
    lstm_bwd_rec_body(pp.j[blockIdx.y]);
}

// Gate gradients of one (batch row, unit) at one step - the arithmetic of autograd's LSTM cell backward, written ONCE for the per-step
// kernel and the persistent one: contraction is switched off and the fused multiply-adds are spelled out, so both kernels (and every
// future one) round identically whatever code surrounds the inlined body (hipcc contracts a*b+c opportunistically per context: the
// first persistent build differed from the launches in the last bit of a handful of products).
struct CellGrad { float d0, d1, d2, d3, dc; };
static __device__ __forceinline__ CellGrad lstm_cell_grad(float ig, float fg, float gg, float og, float ct, float cprev, float ndt, float dh, float dc_in) {
#pragma clang fp contract(off)
    const float tc = tanhf(ct);
    const float cm = ndt * cprev;
    const float dog = dh * tc * og * (1.f - og);
    const float dcc = __builtin_fmaf(dh * og, __builtin_fmaf(-tc, tc, 1.f), dc_in);
    CellGrad r;
    r.d0 = dcc * gg * ig * (1.f - ig);
    r.d1 = dcc * cm * fg * (1.f - fg);
    r.d2 = dcc * ig * __builtin_fmaf(-gg, gg, 1.f);
    r.d3 = dog;
    r.dc = ndt * dcc * fg;
    return r;
}

struct LstmCellBP {
    const float *partial, *nd_next, *dh_ext;      // [16][B][H] (nullptr at t = T-1), [B], [B][H]
    float *dc_carry, *G;                          // [B][H] in/out, [B][4H] gates in / dG out
    const float *c_t, *c_prev, *nd;
    int B, H;
};

static __device__ __forceinline__ void lstm_bwd_cell_body(const LstmCellBP &p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int H = p.H;
    if (i >= p.B * H) return;
    const int bi = i / H, u = i % H;
    float dh = p.dh_ext[i], dc_in = 0.f;
    if (p.partial) {
        float s = 0.f;
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) s += p.partial[((size_t)kg * p.B + bi) * H + u];
        dh = __builtin_fmaf(p.nd_next[bi], s, dh);
        dc_in = p.dc_carry[i];
    }
    float *g = p.G + (size_t)bi * 4 * H + u;
    const CellGrad r = lstm_cell_grad(g[0], g[H], g[2 * H], g[3 * H], p.c_t[i], p.c_prev[i], p.nd[bi], dh, dc_in);
    g[0] = r.d0;
    g[H] = r.d1;
    g[2 * H] = r.d2;
    g[3 * H] = r.d3;
    p.dc_carry[i] = r.dc;
}

static __global__ __launch_bounds__(256) void lstm_bwd_cell_kernel(LstmCellBP p) { lstm_bwd_cell_body(p); }
struct LstmCellB2P { LstmCellBP j[2]; int active[2]; };
static __global__ __launch_bounds__(256) void lstm_bwd_cell2_kernel(LstmCellB2P pp) {
    if (!pp.active[blockIdx.y]) return;
    lstm_bwd_cell_body(pp.j[blockIdx.y]);
}

#ifdef PVR_EXPERIMENTS   // round-3 experiment, slower than the two launches it replaces (profiles/experiments/r03_bc_fused_bptt_step.txt)
// ---------------------------------------------------------------------------------------------------------
// BPTT step as ONE launch (round 3, opt-in PVR_POLICY_BWD_FUSED=1: correct and 2.5 x SLOWER per step than the two launches it replaces -
// 128 blocks stream W_hh^T with 16 loads in flight per wave and every block re-reads all of dG[t+1]; the two-launch form's rec kernel
// already runs at the ~7 us it takes to stream 32 MB of W_hh from the Infinity Cache, so only the 4.8 us cell launch was ever to be
// had): the recurrent product and the cell update of a (layer, timestep) in the same kernel.
// The two-launch form above splits the product over 16 k-groups x 16 unit-groups, so the cell update has to wait for a grid-wide
// reduction - a second dependent launch per step (6.9 + 4.8 us per wavefronted pair, 124 pairs per iteration).  Here a block owns 16
// units for ALL 4096 gate rows: its four waves take a quarter of k each, their tiles are summed in LDS in a fixed order, and each of
// the 256 threads then finishes one (batch row, unit) of the cell update.  64 blocks per layer, both layers of the chunk wavefront
// as blockIdx.y.  The weights are read from a TRANSPOSED copy W_hh^T [H][4H] (made once per backward pass): a lane streams 16 bytes
// of its unit's row per four MFMAs; dG[t+1] (256 KB, shared by every block) comes from L2.
// Arithmetic per element: the MFMA chain over one quarter of k per wave, ((w0 + w1) + (w2 + w3)), then lstm_cell_grad - a different
// summation ORDER from the two-launch form (16 k-groups), same fp32 accuracy; deterministic.
// MFMA roles: A[i = batch][k] = dG[t+1], B[k][j = unit] = W_hh^T[u0 + j][k]; lane (fr, fq) holds k0 + 4 fq .. + 3 of both.
// ---------------------------------------------------------------------------------------------------------
struct LstmStepP {
    const float *dG_next, *WT;                    // [B][4H] (nullptr at t = T-1), W_hh^T [H][4H]
    const float *nd_next, *dh_ext;                // [B], [B][H]
    float *dc_carry, *G;                          // [B][H] in/out, [B][4H] gates in / dG out
    const float *c_t, *c_prev, *nd;
    int B, H;
};

static __device__ __forceinline__ void lstm_bwd_step_body(const LstmStepP &p) {
    __shared__ float part[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int H = p.H, K = 4 * p.H;
    const int u0 = blockIdx.x * 16;
    const int kq = K / 4;                                             // k per wave
    const float *wrow = p.WT + (size_t)(u0 + fr) * K + wave * kq + fq * 4;
    for (int b0 = 0; b0 < p.B; b0 += 16) {
        float s = 0.f;
        if (p.dG_next) {
            const int b = b0 + fr;
            const bool bok = b < p.B;
            const float bmask = bok ? 1.f : 0.f;
            const float *drow = p.dG_next + (size_t)(bok ? b : 0) * K + wave * kq + fq * 4;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            // 16-k chunks, eight at a time: all 16 loads of a group are in flight before its 32 MFMAs
            for (int c0 = 0; c0 < kq; c0 += 128) {
                f32x4 a[8], w[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    a[c] = *reinterpret_cast<const f32x4 *>(drow + c0 + c * 16);
                    w[c] = *reinterpret_cast<const f32x4 *>(wrow + c0 + c * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    a[c] *= bmask;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma_f32(a[c][e], w[c][e], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][fq * 4 + r][fr] = acc[r];      // D: row = batch 4 fq + r, col = unit fr
            __syncthreads();
            s = (part[0][tid >> 4][tid & 15] + part[1][tid >> 4][tid & 15]) + (part[2][tid >> 4][tid & 15] + part[3][tid >> 4][tid & 15]);
        }
        const int bi = b0 + (tid >> 4), u = u0 + (tid & 15);
        if (bi < p.B) {
            const size_t i = (size_t)bi * H + u;
            float dh = p.dh_ext[i], dc_in = 0.f;
            if (p.dG_next) {
                dh = __builtin_fmaf(p.nd_next[bi], s, dh);
                dc_in = p.dc_carry[i];
            }
            float *g = p.G + (size_t)bi * 4 * H + u;
            const CellGrad r = lstm_cell_grad(g[0], g[H], g[2 * H], g[3 * H], p.c_t[i], p.c_prev[i], p.nd[bi], dh, dc_in);
            g[0] = r.d0;
            g[H] = r.d1;
            g[2 * H] = r.d2;
            g[3 * H] = r.d3;
            p.dc_carry[i] = r.dc;
        }
        if (b0 + 16 < p.B) __syncthreads();                            // part is rewritten by the next batch tile
    }
}
struct LstmStep2P { LstmStepP j[2]; int active[2]; };
static __global__ __launch_bounds__(256) void lstm_bwd_step2_kernel(LstmStep2P pp) {      // chunked layer wavefront (blockIdx.y = job)
    if (!pp.active[blockIdx.y]) return;
    lstm_bwd_step_body(pp.j[blockIdx.y]);
}
#endif  // PVR_EXPERIMENTS

#ifdef PVR_EXPERIMENTS   // round-3 experiment, slower than the per-step launches (profiles/experiments/r03_bc_persistent_bptt.txt)
// ---------------------------------------------------------------------------------------------------------
// Persistent BPTT (round 3): ONE launch runs steps t_hi-1 ... t_lo of up to two independent jobs (blockIdx.y: layer 1 on one chunk of
// the sequence, layer 0 on the chunk behind it - the chunked layer wavefront of the per-step launches, 2 x 256 blocks, co-resident).
// Per job the 256 blocks keep the 2-D partition of lstm_bwd_rec_kernel (kg = 16 k-groups x ug = 16 unit-groups; the block's 256 x 64
// slab of W_hh stays in registers for the whole range) and a step is two grid-wide hand-offs inside the kernel instead of two
// launches (~11 us per wavefronted pair of launches before):
//   phase 1  partial[kg][b][u0..u0+63] = sum_{k in group kg} dG[t+1][b][k] W_hh[k][u]    (same MFMA chain / LDS reduce as the launch)
//   phase 2  the block ALSO owns the cell update of units u0 + 4 kg .. +3 (all batch rows): dh = dh_ext + nd[t+1] * sum_kg partial
//            (fixed kg order), gate gradients, dc carry in a register
// Same arithmetic per element as lstm_bwd_rec_kernel + lstm_bwd_cell_kernel: bit-identical.  Hand-offs use the data-as-flag scheme of
// lstm_fwd_seq_kernel (sc1 write-through stores, sc1 polling loads, 0xFFFFFFFF = "not written yet", NaNs canonicalised on publish):
//   * dG[t] goes to G in place (plain stores: the weight-gradient GEMMs of later launches read it) AND to dGx[t], a hand-off copy whose
//     range [t_lo, t_hi) the host pre-fills before the launch; phase 1 of step t polls dGx[t+1] (16 consumers per word: no re-arming);
//     the first step of a launch reads G[t+1], which an earlier launch finished;
//   * the partials live in a two-slot ring Px[t & 1]; every word has exactly ONE consumer, which re-arms it (stores 0xFFFFFFFF) right
//     after reading.  vmcnt retires vector-memory operations in issue order, so by the time the consumer has USED the polling loads of
//     step t-1 (issued after the re-arming stores of step t) those stores have completed; dG[t-1] is published after that, and the
//     producer's next write to the slot (step t-2) causally follows that publication: a re-arm never lands on top of new data.
//     The ring is fully armed whenever no launch is in flight (armed at create; every word written is consumed).
// Bounded spins, status word and NaN poisoning as in the forward kernel.
// ---------------------------------------------------------------------------------------------------------
struct LstmBwdJob {
    float *G;                  // [T][B][4H]: activated gates in, dG out (in place)
    float *dGx;                // [T][B][4H]: hand-off copy of dG
    float *Px;                 // [2][16][B][H]: ring of the partial products of phase 1
    const float *W;            // [4H][H]
    const float *dh_ext;       // [T][B][H]: d(loss)/d(h_t) arriving from above
    const float *Cs, *c0;      // [T][B][H] cell states, [B][H] state before t = 0 (zeros)
    float *dc_carry;           // [B][H]: carry across launches (read unless the range starts at T-1, written at the end)
    int t_lo, t_hi, active;
};
struct LstmBwdSeqP {
    LstmBwdJob j[2];
    const float *nd;           // [T][B]
    int T, B, H, drop_block;
    unsigned *status;
};

static __global__ __launch_bounds__(256) void lstm_bwd_seq_kernel(LstmBwdSeqP pp) {
    const LstmBwdJob &p = pp.j[blockIdx.y];
    if (!p.active || (int)blockIdx.x == pp.drop_block) return;
    __shared__ float part[4][4][16][17];
    __shared__ __attribute__((aligned(16))) float gstage[64][4][4];        // [batch row][gate][unit]
    __shared__ int dead_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int H = pp.H, K = 4 * pp.H, B = pp.B, T = pp.T;
    const int kg = blockIdx.x >> 4, ug = blockIdx.x & 15;
    const int u0 = ug * (H / 16), kbeg = kg * (K / 16) + wave * (K / 64);         // 64 units, 256 k per block, 64 k per wave
    if (tid == 0) dead_s = 0;
    __syncthreads();
    // the wave's 64 x 64 weight slab, for the whole range: lane (fr, fq) holds W[kbeg + 16 c + 4 fq + e1][u0 + 4 fr .. + 3]
    f32x4 w[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) w[c][e1] = *reinterpret_cast<const f32x4 *>(p.W + (size_t)(kbeg + c * 16 + fq * 4 + e1) * H + u0 + fr * 4);
    // cell ownership: thread (cb, cj) updates unit cu of batch row cb at every step
    const int cb = tid >> 2, cj = tid & 3, cu = u0 + kg * 4 + cj;
    const bool cok = cb < B;
    float dc = (cok && p.t_hi - 1 < T - 1) ? p.dc_carry[(size_t)cb * H + cu] : 0.f;
    bool dead = false;                                                            // this wave's spin ran out once (wave-uniform)
    const unsigned ARM = 0xffffffffu;
    for (int t = p.t_hi - 1; t >= p.t_lo; --t) {
        const bool has_next = t < T - 1;
        // operands of the cell update: plain loads of data earlier launches finished.  vmcnt is ONE in-order queue: a polling load issued
        // behind them cannot return before they have, so they are requested right after phase 1's poll has succeeded (they then travel
        // under the MFMA chain and the second hand-off), not in front of it (an HBM miss per step on the critical path, measured)
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cprev = 0.f, ndt = 0.f, dhe = 0.f, ndn = 0.f;
        float *gp = p.G + ((size_t)t * B + (cok ? cb : 0)) * K + cu;
        auto fetch_cell = [&]() {
            if (cok) {
                ig = gp[0]; fg = gp[H]; gg = gp[2 * H]; og = gp[3 * H];
                ct = p.Cs[((size_t)t * B + cb) * H + cu];
                cprev = t == 0 ? p.c0[(size_t)cb * H + cu] : p.Cs[((size_t)(t - 1) * B + cb) * H + cu];
                ndt = pp.nd[(size_t)t * B + cb];
                dhe = p.dh_ext[((size_t)t * B + cb) * H + cu];
                if (has_next) ndn = pp.nd[(size_t)(t + 1) * B + cb];
            }
        };
        if (!has_next) fetch_cell();
        float s = 0.f;
        if (has_next) {
            const bool polled = t + 1 < p.t_hi;                                   // produced inside this launch: hand-off copy, sc1
            const float *src = (polled ? p.dGx : p.G) + (size_t)(t + 1) * B * K;
            const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (unsigned)((size_t)B * K * 4), 0x00020000);
            float *slot = p.Px + (size_t)(t & 1) * 16 * B * H;
            const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(slot, 0, (unsigned)((size_t)16 * B * H * 4), 0x00020000);
            // ---- phase 1 ----
            for (int b0 = 0; b0 < B; b0 += 16) {
                const int b = b0 + fr;
                const bool bok = b < B;
                const int goff = ((bok ? b : 0) * K + kbeg + fq * 4) * 4;
                f32x4 a[4];
                if (polled) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, goff, c * 64, 16));
                    if (!dead) {
                        unsigned spins = 0;
                        for (;;) {
                            bool written = true;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const u32x4 q = __builtin_bit_cast(u32x4, a[c]);
                                written = written & (q[0] != ARM) & (q[1] != ARM) & (q[2] != ARM) & (q[3] != ARM);
                            }
                            if (__all(written || !bok)) break;
                            if (++spins > LSTM_SEQ_SPINS) {
                                dead = true; dead_s = 1;
                                if (lane == 0) __hip_atomic_store(pp.status, LSTM_SEQ_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
#pragma unroll
                            for (int c = 0; c < 4; ++c) a[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, goff, c * 64, 16));
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, goff, c * 64, 0));
                }
                if (b0 == 0) fetch_cell();
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float bmask = bok ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c] *= bmask;
#pragma unroll
                    for (int e1 = 0; e1 < 4; ++e1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = mfma_f32(a[c][e1], w[c][e1][e], acc[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[wave][e][fq * 4 + r][fr] = acc[e][r];
                __syncthreads();
                // 16 batch rows x 64 units; thread: row bb = tid >> 4, four consecutive units 4 jq .. + 3 -> one 16-byte store
                {
                    const int bb = tid >> 4, jq = tid & 15;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = (part[0][e][bb][jq] + part[1][e][bb][jq]) + (part[2][e][bb][jq] + part[3][e][bb][jq]);
                        v[e] = (dead_s || x != x) ? __builtin_nanf("") : x;
                    }
                    if (b0 + bb < B)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_p, ((kg * B + b0 + bb) * H + u0 + jq * 4) * 4, 0, 16);
                }
                __syncthreads();
            }
            // ---- phase 2: the 16 partials of (cb, cu), summed in kg order; every word is re-armed by its one reader ----
            float pv[16];
            const int poff = ((cok ? cb : 0) * H + cu) * 4;
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_p, poff, q * B * H * 4, 16));
            if (!dead) {
                unsigned spins = 0;
                for (;;) {
                    bool written = true;
#pragma unroll
                    for (int q = 0; q < 16; ++q) written = written & (__builtin_bit_cast(unsigned, pv[q]) != ARM);
                    if (__all(written || !cok)) break;
                    if (++spins > LSTM_SEQ_SPINS) {
                        dead = true; dead_s = 1;
                        if (lane == 0) __hip_atomic_store(pp.status, LSTM_SEQ_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int q = 0; q < 16; ++q) pv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_p, poff, q * B * H * 4, 16));
                }
            }
            if (cok) {
#pragma unroll
                for (int q = 0; q < 16; ++q) __builtin_amdgcn_raw_buffer_store_b32(ARM, rs_p, poff + q * B * H * 4, 0, 16);      // (offset in voffset: soffset stays 0 for stores)
#pragma unroll
                for (int q = 0; q < 16; ++q) s += pv[q];
            }
        }
        // ---- cell update (lstm_cell_grad: the per-step kernel's arithmetic, element for element) ----
        if (cok) {
            const float dh = has_next ? __builtin_fmaf(ndn, s, dhe) : dhe;
            const CellGrad r = lstm_cell_grad(ig, fg, gg, og, ct, cprev, ndt, dh, has_next ? dc : 0.f);
            dc = r.dc;
            gp[0] = r.d0; gp[H] = r.d1; gp[2 * H] = r.d2; gp[3 * H] = r.d3;
            const bool bad = dead_s != 0;
            gstage[cb][0][cj] = (bad || r.d0 != r.d0) ? __builtin_nanf("") : r.d0;
            gstage[cb][1][cj] = (bad || r.d1 != r.d1) ? __builtin_nanf("") : r.d1;
            gstage[cb][2][cj] = (bad || r.d2 != r.d2) ? __builtin_nanf("") : r.d2;
            gstage[cb][3][cj] = (bad || r.d3 != r.d3) ? __builtin_nanf("") : r.d3;
        }
        // No drain is needed for the re-arming stores: vmcnt retires in issue order, and the NEXT step's polling loads - issued after
        // them - have returned before that step's dG is published, which is what the producer's next write to the slot (two steps on)
        // causally follows.  (The slot re-armed at the launch's last step is next written by a later launch: kernel boundary.)
        __syncthreads();
        if (t > p.t_lo && tid < 4 * B) {                                           // (nobody in this launch reads dGx[t_lo])
            const int pb = tid >> 2, pg = tid & 3;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&gstage[pb][pg][0]);
            const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(p.dGx + (size_t)t * B * K, 0, (unsigned)((size_t)B * K * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, (pb * K + pg * H + u0 + kg * 4) * 4, 0, 16);
        }
        __syncthreads();                                                           // gstage / part are reused by the next step
    }
    if (cok) p.dc_carry[(size_t)cb * H + cu] = dc;
}
#endif  // PVR_EXPERIMENTS

// Hprev_m[t][b][:] = nd[t][b] * h[t-1][b][:]  (h[-1] = h_init): the recurrent operand of dW_hh
static __global__ __launch_bounds__(256) void hprev_kernel(const float *__restrict__ hs, const float *__restrict__ h_init,
                                                    const float *__restrict__ nd, float *__restrict__ out, int T, int B, int H) {
    const size_t total4 = (size_t)T * B * H / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 4;
        const int row = (int)(e / H);                    // t*B + b
        const int t = row / B, b = row % B;
        const float *src = t == 0 ? h_init + (size_t)b * H + (e % H) : hs + (size_t)(row - B) * H + (e % H);
        f32x4 v = *reinterpret_cast<const f32x4 *>(src);
        v *= nd[row];
        reinterpret_cast<f32x4 *>(out)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// heads (models.py:75-82) + BC loss (main_bc_2.py:211-214): one wave per row
//   logits = out W_p^T + b_p, baseline = out W_b^T + b_b, action = argmax (first max on ties, as torch.argmax)
//   with targets: loss_row = -log_softmax(logits)[target], dlogits = (softmax - onehot) / N   (mean reduction)
// dlogits rows are padded to 16 floats.
// ---------------------------------------------------------------------------------------------------------
struct HeadP {
    const float *out, *Wp, *bp, *Wb, *bb;
    float *logits, *baseline, *dlogits, *loss_row;
    long long *action;
    const long long *target;
    int N, H, A;
    int sample;                                     // training-mode forward: action = one sample of softmax(logits) (sample_rng.h)
    unsigned long long seed, call;
};

static __global__ __launch_bounds__(256) void heads_kernel(HeadP p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const float *x = p.out + (size_t)n * p.H;
    float l[17];
    for (int a = 0; a <= p.A; ++a) {
        const float *w = a < p.A ? p.Wp + (size_t)a * p.H : p.Wb;
        float s = 0.f;
        for (int k = lane * 4; k < p.H; k += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + k), wv = *reinterpret_cast<const f32x4 *>(w + k);
            s += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        l[a] = s + (a < p.A ? p.bp[a] : p.bb[0]);
    }
    if (lane != 0) return;
    int best = 0;
    float mx = l[0];
    for (int a = 1; a < p.A; ++a) if (l[a] > mx) { mx = l[a]; best = a; }
    for (int a = 0; a < p.A; ++a) p.logits[(size_t)n * p.A + a] = l[a];
    p.baseline[n] = l[p.A];
    p.action[n] = p.sample ? sample_softmax_row(l, p.A, p.seed, p.call, (unsigned long long)n) : best;
    if (p.target) {
        float se = 0.f;
        for (int a = 0; a < p.A; ++a) se += expf(l[a] - mx);
        const float lse = mx + logf(se);
        const long long tgl = p.target[n];
        const bool tg_ok = tgl >= 0 && tgl < p.A;                 // torch's nll_loss raises for a target outside [0, A): fail loudly here too
        const int tg = tg_ok ? (int)tgl : 0;
        p.loss_row[n] = tg_ok ? lse - l[tg] : __builtin_nanf("");   // NaN loss (and NaN gradient norm downstream) instead of an out-of-bounds read
        const float inv = 1.0f / (float)p.N;
        for (int a = 0; a < 16; ++a)
            p.dlogits[(size_t)n * 16 + a] = a < p.A ? (expf(l[a] - lse) - (a == tg ? 1.f : 0.f)) * inv : 0.f;
    }
}

// fixed-order sum of n values scaled by `scale` -> out[0]  (loss mean; 1 block)
static __global__ __launch_bounds__(256) void sum_kernel(const float *__restrict__ x, int n, float scale, float *__restrict__ out) {
    __shared__ float s[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += x[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0] * scale;
}

// dout[n][k] = sum_a dlogits[n][a] * Wp[a][k]
static __global__ __launch_bounds__(256) void head_dx_kernel(const float *__restrict__ dl, const float *__restrict__ Wp,
                                                      float *__restrict__ dout, int N, int H, int A) {
    const size_t total = (size_t)N * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / H), k = (int)(i % H);
        float s = 0.f;
        for (int a = 0; a < A; ++a) s += dl[(size_t)n * 16 + a] * Wp[(size_t)a * H + k];
        dout[i] = s;
    }
}

// dWp[a][k] = sum_n dlogits[n][a] * out[n][k];  dbp[a] = sum_n dlogits[n][a]   (32 columns x 8 row groups / block)
static __global__ __launch_bounds__(256) void head_dw_kernel(const float *__restrict__ dl, const float *__restrict__ out,
                                                      float *__restrict__ dWp, float *__restrict__ dbp, int N, int H, int A) {
    __shared__ float s[8][33];
    const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + cx;                  // k == H -> bias column (out == 1)
    for (int a = 0; a < A; ++a) {
        float acc = 0.f;
        if (k <= H)
            for (int n = g; n < N; n += 8) acc += dl[(size_t)n * 16 + a] * (k < H ? out[(size_t)n * H + k] : 1.f);
        s[g][cx] = acc;
        __syncthreads();
        if (g == 0 && k <= H) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += s[i][cx];
            if (k < H) dWp[(size_t)a * H + k] = t; else dbp[a] = t;
        }
        __syncthreads();
    }
}

// two-stage version of head_dw_kernel for tall batches (33 blocks x all N rows took 107 us at N = 1600): (column block, row group)
// partials, then a fixed-order sum over the row groups.  part layout: [G][A][H + 4] (column H = bias sum)
static __global__ __launch_bounds__(256) void head_dw_part_kernel(const float *__restrict__ dl, const float *__restrict__ out,
                                                           float *__restrict__ part, int N, int H, int A, int rpg) {
    __shared__ f32x4 s[8][33];
    __shared__ float sb[8];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int k = (blockIdx.x * 32 + cq) * 4, g = blockIdx.y;
    const int r0 = g * rpg, r1 = r0 + rpg < N ? r0 + rpg : N;
    const bool ok = k < H;
    for (int a = 0; a < A; ++a) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        float accb = 0.f;
        for (int n = r0 + rl; n < r1; n += 8) {
            const float d = dl[(size_t)n * 16 + a];
            if (ok) acc += d * *reinterpret_cast<const f32x4 *>(out + (size_t)n * H + k);
            accb += d;
        }
        s[rl][cq] = acc;
        if (cq == 0) sb[rl] = accb;
        __syncthreads();
        if (rl == 0) {
            float *prow = part + ((size_t)g * A + a) * (H + 4);
            if (ok) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) t += s[i][cq];
                *reinterpret_cast<f32x4 *>(prow + k) = t;
            }
            if (cq == 0 && blockIdx.x == 0) {
                float tb = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) tb += sb[i];
                prow[H] = tb;
            }
        }
        __syncthreads();
    }
}
static __global__ __launch_bounds__(256) void head_dw_final_kernel(const float *__restrict__ part, float *__restrict__ dWp, float *__restrict__ dbp,
                                                            int G, int H, int A) {
    const int i = blockIdx.x * 256 + threadIdx.x;            // over A * (H + 1)
    if (i >= A * (H + 1)) return;
    const int a = i / (H + 1), k = i % (H + 1);
    float t = 0.f;
    for (int g = 0; g < G; ++g) t += part[((size_t)g * A + a) * (H + 4) + k];
    if (k < H) dWp[(size_t)a * H + k] = t; else dbp[a] = t;
}

// ---------------------------------------------------------------------------------------------------------
// grad norm + clip + RMSprop (main_bc_2.py:220-227; torch.optim.RMSprop momentum=0, centered=False)
// ---------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ g, size_t n, float *__restrict__ partial) {
    __shared__ float s[256];
    // a block's range is a whole number of float4s (the flat gradient is 16-byte aligned); 16-byte loads, four independent partial sums per
    // thread (the scalar one-accumulator loop ran at 1.7 TB/s: one dependent FMA per 4-byte load), fixed order -> deterministic
    const size_t nv = n / 4, per = (nv + gridDim.x - 1) / gridDim.x;
    const size_t beg = (size_t)blockIdx.x * per, end = beg + per < nv ? beg + per : nv;
    const f32x4 *gv = reinterpret_cast<const f32x4 *>(g);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    size_t i = beg + threadIdx.x;
    for (; i + 768 < end; i += 1024) {
        const f32x4 v0 = gv[i], v1 = gv[i + 256], v2 = gv[i + 512], v3 = gv[i + 768];
        a0 += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
        a1 += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
        a2 += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
        a3 += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
    }
    for (; i < end; i += 256) { const f32x4 v = gv[i]; a0 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < (n & 3)) { const float t = g[nv * 4 + threadIdx.x]; a1 += t * t; }   // the last 1-3 floats
    float a = (a0 + a1) + (a2 + a3);
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

// stats[1] = ||g||, stats[2] = clip coefficient min(1, max_norm / (norm + 1e-6))
static __global__ __launch_bounds__(256) void norm_final_kernel(const float *__restrict__ partial, int n, float max_norm, float *__restrict__ stats) {
    __shared__ float s[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(s[0]);
        stats[1] = nrm;
        const float coef = max_norm / (nrm + 1e-6f);
        stats[2] = coef < 1.f ? coef : 1.f;
    }
}

static __global__ __launch_bounds__(256) void rmsprop_kernel(float *__restrict__ p, float *__restrict__ v, const float *__restrict__ g,
                                                      const float *__restrict__ stats, size_t n4, float alpha, float eps) {
    const float coef = stats[2], lr = stats[3];   // lr lives in device memory so that a captured graph can be replayed with a new value
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
        f32x4 vv = reinterpret_cast<f32x4 *>(v)[i];
        f32x4 pv = reinterpret_cast<f32x4 *>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * coef;
            vv[e] = vv[e] * alpha + (1.f - alpha) * (ge * ge);
            pv[e] = pv[e] - lr * (ge / (sqrtf(vv[e]) + eps));
        }
        reinterpret_cast<f32x4 *>(v)[i] = vv;
        reinterpret_cast<f32x4 *>(p)[i] = pv;
    }
}

// torch.optim.RMSprop with momentum > 0 (not centred): buf = momentum * buf + g / (sqrt(v) + eps);  p -= lr * buf
static __global__ __launch_bounds__(256) void rmsprop_momentum_kernel(float *__restrict__ p, float *__restrict__ v, float *__restrict__ buf,
                                                               const float *__restrict__ g, const float *__restrict__ stats, size_t n4,
                                                               float alpha, float eps, float momentum) {
    const float coef = stats[2], lr = stats[3];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
        f32x4 vv = reinterpret_cast<f32x4 *>(v)[i], bv = reinterpret_cast<f32x4 *>(buf)[i], pv = reinterpret_cast<f32x4 *>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * coef;
            vv[e] = vv[e] * alpha + (1.f - alpha) * (ge * ge);
            bv[e] = bv[e] * momentum + ge / (sqrtf(vv[e]) + eps);
            pv[e] = pv[e] - lr * bv[e];
        }
        reinterpret_cast<f32x4 *>(v)[i] = vv;
        reinterpret_cast<f32x4 *>(buf)[i] = bv;
        reinterpret_cast<f32x4 *>(p)[i] = pv;
    }
}

// torch.optim.Adam (amsgrad off, weight_decay 0): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)       (bias corrections bc1, bc2 computed on the host from the step count)
static __global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                                   const float *__restrict__ g, const float *__restrict__ stats, size_t n4, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt) {
    const float coef = stats[2], lr = stats[3];
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4 *>(m)[i], vv = reinterpret_cast<f32x4 *>(v)[i], pv = reinterpret_cast<f32x4 *>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * coef;
            mv[e] = mv[e] * b1 + (1.f - b1) * ge;
            vv[e] = vv[e] * b2 + (1.f - b2) * (ge * ge);
            pv[e] = pv[e] - step_size * (mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
        }
        reinterpret_cast<f32x4 *>(m)[i] = mv;
        reinterpret_cast<f32x4 *>(v)[i] = vv;
        reinterpret_cast<f32x4 *>(p)[i] = pv;
    }
}

// upstream gradient of the logits (autograd bridge): [N][A] contiguous -> the workspace's zero-padded [N][16]
static __global__ __launch_bounds__(256) void dlogits_pad_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * 16) return;
    const int n = i >> 4, a = i & 15;
    out[i] = a < A ? in[(size_t)n * A + a] : 0.f;
}

static __global__ void set_scalar_kernel(float *dst, float v) { *dst = v; }

// SyncBN (data-parallel finetune): statistics over the GLOBAL batch of n_global rows, from all-reduced column sums
static __global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ out, const float *__restrict__ in, float s, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * s;
}
static __global__ __launch_bounds__(256) void scale_inplace_kernel(float *x, float s, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= s;
}
static __global__ __launch_bounds__(256) void bn_sync_final_kernel(const float *__restrict__ sumsq, const float *__restrict__ mean,
                                                            float n_global, float *__restrict__ invstd, float *__restrict__ running_mean,
                                                            float *__restrict__ running_var, long long *nbt, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float var_b = sumsq[c] / n_global;
    invstd[c] = 1.0f / sqrtf(var_b + 1e-5f);
    running_mean[c] = 0.9f * running_mean[c] + 0.1f * mean[c];
    running_var[c] = 0.9f * running_var[c] + 0.1f * (var_b * n_global / (n_global - 1.f));
    if (c == 0 && nbt) *nbt += 1;
}

}  // namespace pvr
