// Fused tail of a torchvision Bottleneck (reference src/embeddings.py:118-120 -> torchvision resnet50), wave form for the stride-1
// blocks of layer2 (Cm = 128, 28 x 28):
//
//     t2  = relu(conv2_3x3(t1) + b2)                       (128 -> 128)
//     y   = relu(conv3_1x1(t2) + b3 + residual)            (128 -> 512)
//     t1' = relu(conv1_1x1_of_the_NEXT_block(y) + b1')     (512 -> 128, optional)
//
// Round 6.  chain_wave.hip (layer1, Cm = 64) moves its bytes at 4.7-5.4 TB/s because a WAVE owns 32 pixels through all three GEMMs - t2 and y
// go from accumulators to MFMA operands in registers, no wave ever waits for another - while its weights (136 KB) stay in LDS.  At Cm = 128
// the weights are 544 KB (W2 288, W3 128, W1' 128): the block form (bottleneck_chain.hip) therefore stayed for layer2 - 128-pixel blocks,
// four waves exchanging t2 and the y groups through LDS behind barriers, 3.5-4.1 TB/s, the furthest-from-roof kernel of rounds 3-5.
// This kernel keeps the wave-owned pixels and STREAMS the weights: the eight waves of a 512-thread workgroup walk the same sequence of 17
// weight units of 32 KB in step - nine taps of W2, then eight pairs of 32-cout half-groups (their W3 rows + the matching K slice of W1') -
// each unit copied global -> registers -> LDS one unit ahead into a two-slot ring, one barrier per unit (64 MFMAs per wave).  Every weight
// byte crosses L2 -> LDS once per 256 pixels (0.8 x the launch's HBM bytes); pixels, residual, y and t1' stay wave-private requests in the
// blocked P16C8 layout ([pixel >> 4][channel >> 3][pixel & 15][8]: a fragment column is 256 contiguous bytes), in flight across the barriers.
//
// Weight image (launch_chain_wave128_pack, once per plan): every unit is the LDS image itself, MFMA A fragments of 1 KB
// [k chunk][row & 15][8] - a lane reads its 16 bytes at fragment * 1024 + lane * 16, conflict-free - with the rows permuted inside every
// 32-row block (chain_row_source) so that a lane's accumulators of a tile pair are 8 consecutive output channels of one pixel = the next
// GEMM's B fragment.
//
// Numerics: the same rounding points (t2, y, t1') and the same K order per accumulator as bottleneck_chain.hip and the unfused launches:
// bit-identical (tests/test_gpu_encoder.py::test_layer2_wave_form_equals_block_form).
#include "chain_params.h"

namespace pvr {

#ifndef CW8_KNOCK
#define CW8_KNOCK 0     // timing experiments: 1 no y / t1' stores, 2 residual loads out of range, 4 conv2 pixel loads out of range
#endif

__device__ __forceinline__ int cw8_row_source(int row) { return (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

template <int AUX>
__device__ __forceinline__ void cw8_store(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int imm) {
    if constexpr (!(CW8_KNOCK & 1)) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, AUX);
}

// two consecutive 32-channel fragments (64 channels = 128 B per pixel) of a 16-pixel tile -> two registers of FULL 128-byte lines (chain_wave.hip,
// cw_f2m_pair): lane l gets rows (l >> 3) and (l >> 3) + 8, chunk (l & 7) ^ (l >> 3)
__device__ __forceinline__ void cw8_f2m_pair(char *slot, int lane, u32x4 even, u32x4 odd, u32x4 &lo, u32x4 &hi) {
    const int fr = lane & 15, fq = lane >> 4, base = fr * 128, sw = fr & 7;
    *reinterpret_cast<u32x4 *>(slot + base + ((fq ^ sw) << 4)) = even;
    *reinterpret_cast<u32x4 *>(slot + base + (((4 + fq) ^ sw) << 4)) = odd;
    lo = *reinterpret_cast<const u32x4 *>(slot + lane * 16);
    hi = *reinterpret_cast<const u32x4 *>(slot + 1024 + lane * 16);
}

struct Cw8Tile {
    int xb[2];      // conv2 input: the pixel index m0 + 16 j + fr this lane's fragment column belongs to
    int mk[2];      // 9-bit "tap inside the image" mask of that pixel
    int yi[2];      // residual: byte offset of the lane's 16 bytes of half-group 0 (blocked)
    int yo[2];      // y: likewise for the store (blocked or NHWC full lines)
    int to[2];      // t1': likewise (tile pair 0; blocked)
};

template <bool OUTB>
__device__ __forceinline__ void cw8_setup(Cw8Tile &a, int m0, int lane, int M, int H, int W) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 16 * j + fr;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int wo = mm % W, ho = (mm / W) % H;
        int hb = 0, wb = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && (unsigned)(ho - 1 + t3) < (unsigned)H) << t3;
            wb |= (int)((unsigned)(wo - 1 + t3) < (unsigned)W) << t3;
        }
        int mask = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * 3)) : 0;
        a.mk[j] = mask;
        a.xb[j] = m;
        const int blk = (m0 >> 4) + j;                      // (m0 is a multiple of 32)
        a.yi[j] = blk * 16384 + fq * 256 + fr * 16;
        a.yo[j] = OUTB ? blk * 16384 + fq * 256 + fr * 16 : (m0 + 16 * j + (lane >> 3)) * 1024 + (((lane & 7) ^ (lane >> 3)) << 4);
        a.to[j] = blk * 4096 + fq * 256 + fr * 16;
    }
}

// CMN: width of the next block's conv1 (128, or 0: none); OUTB: y (and t1') leave in the blocked layout (else y as NHWC full lines; t1' is always blocked)
// XD: conv2 K-steps of pixel fragments in flight (8 VGPRs each); RD: residual half-groups in flight (8 VGPRs each)
template <int CMN, bool F16, bool OUTB, int XD, int RD>
__global__ __launch_bounds__(512, 2) void chain_wave128_kernel(ChainP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int NK = 36, NH = 16, NUA = 9, NUB = 8, NU = NUA + NUB;   // conv2 K-steps of 32 channels; half-groups of 32 couts; weight units
    constexpr int UNIT = 32768, NQ = 4, NQB = CMN ? 4 : 2;                // bytes per unit slot; 16-byte staging pieces per thread (phase B without W1': 16 KB)
    constexpr int TN1 = CMN / 16;
    constexpr int B2L = 2 * UNIT, B3L = B2L + 512, B1L = B3L + 2048, SCR = B1L + 512;
    constexpr int OOB = 0x7ffffff0;
    static_assert(XD >= 1 && XD <= NK && RD >= 1 && NH % RD == 0, "prefetch depths");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fq = lane >> 4;
    char *const slot0 = smem + SCR + wave * 2048;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wpk), 0, p.wpk_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.res), 0, p.y_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);

    // this workgroup's 32-pixel tiles: a contiguous range (an XCD's workgroups cover a contiguous run: halo rows meet in that XCD's L2)
    const int T = (p.M + 31) >> 5, G = gridDim.x, bx = xcd_remap(blockIdx.x, G);   // (M is a multiple of 16: the last tile may hold one 16-pixel block)
    const int t_lo = (int)((long long)bx * T / G), t_hi = (int)((long long)(bx + 1) * T / G);
    const int rounds = (t_hi - t_lo + 7) >> 3;
    if (rounds <= 0) return;
    const int m_pad = (p.M + 63) & ~31;                     // a tile past the tensor: every load reads zeros, every store is dropped (range check)

    // ---- prologue: biases -> LDS, weight units 0 (-> ring slot 0) and 1 (-> staging registers) ----------------------------------
    if (tid < 128) *reinterpret_cast<float *>(smem + B2L + tid * 4) = p.b2[tid];
    *reinterpret_cast<float *>(smem + B3L + tid * 4) = p.b3[tid];
    if constexpr (CMN > 0) { if (tid < CMN) *reinterpret_cast<float *>(smem + B1L + tid * 4) = p.b1n[tid]; }
    u32x4 wst[NQ];
    const int st_off = tid * 16;
#define CW8_W_LOAD(u_, nq_)                                                                                            \
    {                                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < (nq_); ++q)                                                               \
            wst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, st_off + q * 8192, (u_) * UNIT, 0)); \
    }
#define CW8_W_STORE(slot_, nq_)                                                                                         \
    {                                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < (nq_); ++q)                                                               \
            *reinterpret_cast<u32x4 *>(smem + (slot_) + st_off + q * 8192) = wst[q];                                    \
    }
    CW8_W_LOAD(0, NQ);
    CW8_W_STORE(0, NQ);
    CW8_W_LOAD(1, NQ);

    int tile = t_lo + wave;
    Cw8Tile cur, nxt;
    cw8_setup<OUTB>(cur, tile < t_hi ? tile * 32 : m_pad, lane, p.M, p.H, p.W);

    u32x4 xr[XD][2];                                        // conv2 pixel fragments, K-steps kt .. kt + XD - 1
    u32x4 rres[RD][2];                                      // residual fragments, half-groups h .. h + RD - 1
    // conv2 fragment of K-step kt_ (tap kt_ / 4, channels 32 (kt_ & 3) ..) of tile A_: pixel pm's 16 bytes of chunk 4 (kt_ & 3) + fq
#define CW8_ISSUE_X(slot_, kt_, A_)                                                                                     \
    {                                                                                                                   \
        const int tp_ = (kt_) >> 2;                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
            const int pm_ = A_.xb[j] + (tp_ / 3 - 1) * p.W + (tp_ % 3 - 1);                                             \
            int vo_ = (pm_ >> 4) * 4096 + (pm_ & 15) * 16 + ((((kt_) & 3) * 4 + fq) << 8);                              \
            vo_ = ((A_.mk[j] >> tp_) & 1) ? vo_ : OOB;                                                                  \
            if constexpr (CW8_KNOCK & 4) vo_ = OOB;                                                                     \
            xr[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo_, 0, 0));          \
        }                                                                                                               \
    }
#define CW8_ISSUE_RES(slot_, h_, A_)                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            rres[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CW8_KNOCK & 2) ? OOB : A_.yi[j], (h_) * 1024, PVR_NT_AUX(2))); \
    }
#pragma unroll
    for (int k = 0; k < XD; ++k) CW8_ISSUE_X(k, k, cur);
#pragma unroll
    for (int d = 0; d < RD; ++d) CW8_ISSUE_RES(d, d, cur);
    __syncthreads();                                        // biases and weight unit 0 visible

    const char *const frag = smem + lane * 16;              // + slot * UNIT + fragment * 1024
    for (int r = 0; r < rounds; ++r) {
        const bool more = r + 1 < rounds;
        const int tile_n = tile + 8;
        // ring slot of unit u of this round: consecutive units alternate slots, and a round has an ODD number of units (17), so the parity flips per round
        const int s_even = (r & 1) * UNIT, s_odd = UNIT - s_even;
#define CW8_SLOT(u_) (((u_) & 1) ? s_odd : s_even)

        // ---- conv2 3x3: 32 pixels x 128 couts, K = 9 taps x 128 channels; one weight unit (tap) per barrier ---------------------
        f32x4 acc2[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NUA; ++u) {
            // unit u + 1 (in the staging registers since the previous unit) -> the ring slot unit u - 1 has left; unit u + 2 -> registers
            CW8_W_STORE(CW8_SLOT(u + 1), (u + 1 < NUA) ? NQ : NQB);
            CW8_W_LOAD((u + 2) % NU, (u + 2 >= NUA && u + 2 < NU) ? NQB : NQ);
            const char *const wu = frag + CW8_SLOT(u);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kt = 4 * u + ks;
                const V8 x0 = __builtin_bit_cast(V8, xr[kt % XD][0]), x1 = __builtin_bit_cast(V8, xr[kt % XD][1]);
                if (kt + XD < NK) CW8_ISSUE_X(kt % XD, kt + XD, cur);
                V8 wb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wb[i] = *reinterpret_cast<const V8 *>(wu + (i * 4 + ks) * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc2[i][0] = mfma16<F16>(wb[i], x0, acc2[i][0]);
                    acc2[i][1] = mfma16<F16>(wb[i], x1, acc2[i][1]);
                }
                __builtin_amdgcn_sched_barrier(0);          // (bounds hipcc's hoisting of later steps' LDS reads: register pressure)
            }
            __syncthreads();                                // unit u + 1 visible; every wave is done with unit u's slot
        }
        // t2 = relu(acc2 + b2) -> 16 bit: tile pair q of pixel tile j IS conv3's B fragment of K step q
        u32x4 t2[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bA = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq) * 4);
            const float4 bB = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq + 4) * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 lo = acc2[2 * q][j], hi = acc2[2 * q + 1][j];
                const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                t2[q][j] = o;
            }
        }

        // ---- the next tile's addresses and its first conv2 fragments: requested before this tile's stores are issued ---------------
        cw8_setup<OUTB>(nxt, (more && tile_n < t_hi) ? tile_n * 32 : m_pad, lane, p.M, p.H, p.W);
#pragma unroll
        for (int k = 0; k < XD; ++k) CW8_ISSUE_X(k, k, nxt);

        // ---- conv3 (+ residual) and conv1', one 32-cout half-group at a time; one weight unit = two half-groups ------------------------
        f32x4 acc1[CMN ? TN1 : 1][2];
        if constexpr (CMN > 0) {
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        u32x4 oe[2];                                        // NHWC out: y of the even half-group, held until its odd partner completes the 128-byte line
#pragma unroll
        for (int ub = 0; ub < NUB; ++ub) {
            const int u = NUA + ub;
            CW8_W_STORE(CW8_SLOT(u + 1), (u + 1 < NU) ? NQB : NQ);
            CW8_W_LOAD((u + 2) % NU, (u + 2 < NU) ? NQB : NQ);
            const char *const wu = frag + CW8_SLOT(u);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int h = 2 * ub + hh;
                u32x4 rp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) rp[j] = rres[h % RD][j];
                if (h + RD < NH) CW8_ISSUE_RES(h % RD, h + RD, cur)
                else CW8_ISSUE_RES(h % RD, h + RD - NH, nxt)
                f32x4 acc3[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc3[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    V8 wb[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) wb[t] = *reinterpret_cast<const V8 *>(wu + ((hh * 2 + t) * 4 + ks) * 1024);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc3[t][j] = mfma16<F16>(wb[t], __builtin_bit_cast(V8, t2[ks][j]), acc3[t][j]);
                }
                // y = relu(acc3 + b3 + residual): 8 consecutive couts per lane = conv1''s B fragment of K step h
                const float4 bA = *reinterpret_cast<const float4 *>(smem + B3L + (32 * h + 8 * fq) * 4);
                const float4 bB = *reinterpret_cast<const float4 *>(smem + B3L + (32 * h + 8 * fq + 4) * 4);
                u32x4 o[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 lo = acc3[0][j], hi = acc3[1][j];
                    const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                    const u32x4 rr = rp[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = fmaxf(v[2 * e] + from_h<F16>((u16)(rr[e] & 0xffffu)), 0.f);
                        const float v1 = fmaxf(v[2 * e + 1] + from_h<F16>((u16)(rr[e] >> 16)), 0.f);
                        o[j][e] = pack2_h<F16>(v0, v1);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (OUTB) cw8_store<PVR_NT_AUX(1)>(o[j], rs_y, cur.yo[j], h * 1024);
                    else if (h & 1) {                       // NHWC: the pair (h - 1, h) = 128 bytes per pixel leaves as full lines
                        u32x4 lo, hi;
                        cw8_f2m_pair(slot0, lane, oe[j], o[j], lo, hi);
                        cw8_store<PVR_NT_AUX(1)>(lo, rs_y, cur.yo[j], (h >> 1) * 128);
                        cw8_store<PVR_NT_AUX(1)>(hi, rs_y, cur.yo[j], (h >> 1) * 128 + 8 * 1024);
                    } else oe[j] = o[j];
                }
                if constexpr (CMN > 0) {
#pragma unroll
                    for (int i = 0; i < TN1; ++i) {
                        const V8 wb = *reinterpret_cast<const V8 *>(wu + 16384 + (i * 2 + hh) * 1024);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16<F16>(wb, __builtin_bit_cast(V8, o[j]), acc1[i][j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }

        // ---- t1' = relu(acc1 + b1'): tile pair q = 8 consecutive couts per lane; always blocked (the next launch is this kernel) --------
        if constexpr (CMN > 0) {
#pragma unroll
            for (int q = 0; q < TN1 / 2; ++q) {
                const float4 bA = *reinterpret_cast<const float4 *>(smem + B1L + (32 * q + 8 * fq) * 4);
                const float4 bB = *reinterpret_cast<const float4 *>(smem + B1L + (32 * q + 8 * fq + 4) * 4);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 lo = acc1[2 * q][j], hi = acc1[2 * q + 1][j];
                    const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                    cw8_store<0>(o, rs_t, cur.to[j], q * 1024);
                }
            }
        }
        cur = nxt;
        tile = tile_n;
    }
#undef CW8_SLOT
#undef CW8_W_LOAD
#undef CW8_W_STORE
#undef CW8_ISSUE_X
#undef CW8_ISSUE_RES
}

// The 17 weight units of a launch, each the 32 KB LDS image the kernel reads (see the header comment).  w2: conv2 (128, 9 * 128) in pvr_op_conv2d's
// layout; w3p: conv3 (512, 128) and w1np: the next conv1 (128, 512; may be null), both already row-permuted (chain_row_source)
__global__ void chain_wave128_pack_kernel(const u16 *__restrict__ w2, const u16 *__restrict__ w3p, const u16 *__restrict__ w1np, u16 *__restrict__ out) {
    const int total = 17 * 2048;                            // 16-byte pieces
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int u = idx >> 11, f = (idx >> 6) & 31, c = (idx >> 4) & 3, r = idx & 15;
        const u16 *src = nullptr;
        if (u < 9) {                                        // tap u: fragment (row tile i, K step ks)
            const int i = f >> 2, ks = f & 3;
            src = w2 + (size_t)cw8_row_source(16 * i + r) * 1152 + u * 128 + ks * 32 + c * 8;
        } else {
            const int ub = u - 9;
            if (f < 16) {                                   // W3 rows 64 ub .. + 63: fragment (row tile rt, K step ks)
                const int rt = f >> 2, ks = f & 3;
                src = w3p + (size_t)(64 * ub + 16 * rt + r) * 128 + ks * 32 + c * 8;
            } else if (w1np) {                              // W1' K steps 2 ub, 2 ub + 1: fragment (row tile i, hh)
                const int g = f - 16, i = g >> 1, hh = g & 1;
                src = w1np + (size_t)(16 * i + r) * 512 + (2 * ub + hh) * 32 + c * 8;
            }
        }
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (src) v = *reinterpret_cast<const u32x4 *>(src);
        *reinterpret_cast<u32x4 *>(out + (size_t)idx * 8) = v;
    }
}

constexpr size_t CW8_PACK_BYTES = (size_t)17 * 32768;
size_t chain_wave128_pack_bytes() { return CW8_PACK_BYTES; }

pvr_status launch_chain_wave128_pack(const void *w2, const void *w3p, const void *w1np, void *out, hipStream_t stream) {
    PVR_REQUIRE(w2 && w3p && out, "chain_wave128_pack: null argument");
    hipLaunchKernelGGL(chain_wave128_pack_kernel, dim3(136), dim3(256), 0, stream, (const u16 *)w2, (const u16 *)w3p, (const u16 *)w1np, (u16 *)out);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

static long long g_cw8_launches = 0;
long long chain_wave128_launches() { return g_cw8_launches; }

// PVR_CHAIN_WAVE_L2=0 keeps layer2's stride-1 tails on the block form (A/B runs; bit-identical); read when a plan is built
bool chain_wave128_supported(int cm, int cmn, int stride, int64_t M) {
    const char *e = getenv("PVR_CHAIN_WAVE_L2");              // (plan time only: plans built under different settings coexist in the tests)
    const int on = e ? atoi(e) : 1;
    return on && cm == 128 && (cmn == 128 || cmn == 0) && stride == 1 && M % 16 == 0;
}

static int cw8_num_cus() {
    static const int v = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    return v;
}

template <int CMN, bool F16, bool OUTB>
static pvr_status launch_cw8_one(ChainP &p, hipStream_t stream) {
    constexpr int XD = 4, RD = 2;
    const size_t lds = 2 * 32768 + 512 + 2048 + 512 + 8 * 2048;
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)chain_wave128_kernel<CMN, F16, OUTB, XD, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    const int tiles = (p.M + 31) >> 5;
    int grid = cw8_num_cus() & ~7;                          // one persistent workgroup per CU; a multiple of 8 (workgroups b and b + 8 share an XCD)
    if (grid < 8) grid = 8;
    const int need = ((tiles + 7) / 8 + 7) / 8 * 8;         // small launches: one round (eight tiles) per workgroup
    if (grid > need) grid = need;
    ++g_cw8_launches;
    hipLaunchKernelGGL((chain_wave128_kernel<CMN, F16, OUTB, XD, RD>), dim3(grid), dim3(512), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// t1 and the residual arrive blocked, t1' leaves blocked; y blocked (out_blk) or NHWC
pvr_status launch_chain_wave128(ChainP &p, int cmn, int dtype, hipStream_t stream) {
    PVR_REQUIRE(p.wpk && p.M % 16 == 0 && p.stride == 1, "bottleneck chain (layer2 wave form): needs the packed weights, stride 1 and a multiple of 16 pixels");
    PVR_REQUIRE((int64_t)(p.M + 64) * 1024 < 0x7ffffff0ll, "bottleneck chain (layer2 wave form): operand larger than 2 GiB (use a smaller chunk)");
    PVR_REQUIRE(p.in_blk, "bottleneck chain (layer2 wave form): t1 and the residual must be in the blocked layout");
    p.wpk_bytes = (unsigned)CW8_PACK_BYTES;
    const bool f16 = dtype == PVR_F16, ob = p.out_blk != 0;
    if (cmn == 128 && ob) return f16 ? launch_cw8_one<128, true, true>(p, stream) : launch_cw8_one<128, false, true>(p, stream);
    if (cmn == 128) return f16 ? launch_cw8_one<128, true, false>(p, stream) : launch_cw8_one<128, false, false>(p, stream);
    if (cmn == 0 && !ob) return f16 ? launch_cw8_one<0, true, false>(p, stream) : launch_cw8_one<0, false, false>(p, stream);
    if (cmn == 0) return f16 ? launch_cw8_one<0, true, true>(p, stream) : launch_cw8_one<0, false, true>(p, stream);
    set_error("bottleneck chain (layer2 wave form): no instance for next Cm=%d", cmn);
    return PVR_ERR_INVALID;
}

}  // namespace pvr
