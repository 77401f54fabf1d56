// Fused tail of a torchvision Bottleneck (reference src/embeddings.py:118-120 -> torchvision resnet50; the
// blocks of layer1 / layer2, whose 1x1 convolutions are HBM-bound as separate launches):
//
//     t2  = relu(conv2_3x3(t1) + b2)                       (Cm -> Cm, stride 1 or 2)   phase A
//     y   = relu(conv3_1x1(t2) + b3 + residual)            (Cm -> 4Cm)                 phase B, per 64-cout group
//     t1' = relu(conv1_1x1_of_the_NEXT_block(y) + b1')     (4Cm -> Cmn, optional)      phase B, accumulated over groups
//
// conv2 is the only spatial operator of the chain and it comes first, so one block of 128 output pixels carries
// its tile through all three GEMMs with no halo recompute: t2 never leaves LDS, y is written once and never
// re-read by the next block's conv1, and the residual is read once.  HBM bytes per pixel of a layer1 block drop
// from 128 (t2 w) + 128 (t2 r) + 512 (res) + 512 (y w) + 512 (y r) + 128 (t1' w) to 512 + 512 + 128.
//
// Numerics are those of the unfused launches bit for bit: the same 16-bit rounding points (t2, y, t1'), the same
// K order in every accumulation (conv1' walks y's channels in ascending 64-channel groups = conv_igemm's K slices).
//
// Layout tricks:
//   * weights are the MFMA A operand (D rows = couts).  W3 and W1' are stored with their rows permuted inside every
//     32-row block (row 16t+4a+c holds cout 8a+4t+c) so that a lane's accumulators of an MFMA tile PAIR are 8
//     CONSECUTIVE output channels of one pixel: residual loads, y / t1' stores and the LDS write of y are all 16-byte
//     lane accesses straight from the accumulator layout - no fp32 LDS staging pass.
//   * LDS (<= 80 KB, two blocks per CU): the phase-A pipeline buffers are re-used in phase B for t2, the y group and
//     the W3 group; only the W1' slice lives beside them.
//   * phase-B weights and the next group's residual are prefetched into registers one group ahead; plain loads, so
//     hipcc's counted vmcnt keeps them in flight across the barriers.
#include "chain_params.h"

namespace pvr {


// 16-byte buffer store with a compile-time byte offset.  The offset goes into the instruction's immediate field
// (voffset + constant, soffset = 0), NEVER into soffset: with an SGPR soffset hipcc (ROCm 7.2) omits the wait states
// between a buffer_store_dwordx4 and the next VALU write of its data registers (its hazard rule exempts that form),
// and on gfx950 the store then reads overwritten data - measured here as ~6000 wrong y elements per batch-256 launch
// in the fully unrolled group loop, where the next pixel tile's packing reuses the registers immediately
// (scripts/debug_determinism.py; the data dword clobbered was exactly the first VALU destination after the store).
// 16-byte global -> LDS DMA of one wave: lane l lands at lds + 16*l (the LDS address must be wave-uniform), range misses write zeros.
// (A __device__ helper, not a direct builtin call in the kernel template: hipcc's host pass silently drops the kernel stub otherwise.)
__device__ __forceinline__ void ch_dma16(__amdgpu_buffer_rsrc_t rs, char *lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

// Diagnostic build only (scripts/chain_stamps.hip defines CHAIN_STAMP): s_memrealtime (100 MHz) stamps per block
#ifdef CHAIN_STAMP
__device__ unsigned long long chain_stamps[8192][12];
#define CH_T(i_) { if (threadIdx.x == 0) ch_tt[i_] = __builtin_amdgcn_s_memrealtime(); }
#else
#define CH_T(i_)
#endif

#ifndef CH_KNOCK
#define CH_KNOCK 0      // timing experiments (scripts/chain_stamps.hip -DCH_KNOCK=bits): 1 no y / t1' stores, 2 residual loads out of range (zeros, no traffic);
                        // phase A of the halo form: 4 no W2 ring stores, 8 no W2 loads, 16 no per-step barrier, 32 no fragment reads
#endif
template <int AUX = 0>
__device__ __forceinline__ void store_b128_imm(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int imm) {
    if constexpr (!(CH_KNOCK & 1)) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, AUX);
}

// RD: residual prefetch depth in 64-cout groups (RD x 16 VGPRs); OCC: blocks per CU the register budget is capped for
// HALO (stride-1 blocks): phase A reads its pixels from ONE contiguous halo run of t1 held in LDS (see "phase A, halo form")
// DS: the block's downsample convolution (64 input channels, stride 1) runs inside conv3's accumulation; no residual tensor is read
template <int CM, int CMN, bool F16, int RD, int OCC, bool HALO, bool DS = false, int PFK = -1>
__global__ __launch_bounds__(256, OCC) void bottleneck_chain_kernel(ChainP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int BM = 128, BK = 64, BN = CM;
    constexpr int A_CH = BM / 32, B_CH = BN / 32, TM = 4, TN = BN / 32;
    // halo form: KS channel slices x [HROWS][64] halo rows (the last row all zeros), then two W2 stages [CM][64]; RL register
    // stages of W2 slices in flight ahead of them
    constexpr int KS = CM / 64, HROWS = CM == 64 ? 256 : 192, RL = CM == 64 ? 3 : 2;
    constexpr int HSL = HROWS * 128, RING_OFF = KS * HSL, SLICE = CM * 128, ZERO_OFF = (HROWS - 1) * 128;
    constexpr int HOPS = KS * HROWS / 32;                      // 16-byte LDS-DMA operations per thread for the halo
    constexpr int STAGE = HALO ? (RING_OFF + 2 * SLICE) / 2 : (BM + BN) * 128;        // 2*STAGE = bytes of the phase-A area
    constexpr int C4 = 4 * CM, G = C4 / 64, KS3 = CM / 64;     // conv3: G groups of 64 couts over KS3 K-slices
    constexpr int TN1 = CMN / 32;                              // conv1': 16-cout tiles per wave
    constexpr int W3_CH = CM / 32, W1_CH = CMN / 32;           // 16-B staging chunks per thread
    // phase-B LDS map (bytes); [rows][64] 16-bit tiles, 128-B rows, chunk ^= (row>>1)&7
    constexpr int T2_OFF = 0;                                  // KS3 x [128][64]
    constexpr int YG_OFF = KS3 * 16384;                        // [128][64]
    constexpr int W3_OFF = YG_OFF + 16384;                     // KS3 x [64][64]
    constexpr int W1_OFF = W3_OFF + KS3 * 8192;                // [CMN][64]; the block's LDS is max(phase-A area, W1_OFF + CMN * 128)
    static_assert(W3_OFF + KS3 * 8192 <= 2 * STAGE, "phase-B tiles must fit the phase-A area");
    static_assert(RD >= 1 && RD <= G, "residual prefetch depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CHAIN_STAMP
    unsigned long long ch_tt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    CH_T(0);
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int srow = tid >> 3, pch = tid & 7;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w2), 0, p.w2_bytes, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w3), 0, p.w3_bytes, 0x00020000);
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1n), 0, p.w1n_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.xds : p.res), 0, DS ? p.xds_bytes : p.y_bytes, 0x00020000);
    const auto rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.wds : p.w3), 0, DS ? p.wds_bytes : p.w3_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;

    // ---- phase A, halo form: the pixels phase A needs are requested before anything else ---------------------------------
    // Stride 1: output pixel m (flattened n,h,w) at tap (kh,kw) reads input pixel m + (kh-1)*W + (kw-1), so the 128 pixels of
    // the tile need the CONTIGUOUS run of 128 + 2W + 2 rows of t1 starting at m0 - W - 1, whatever image borders it crosses;
    // taps outside their image are redirected to a zero row at fragment-read time.  The run is fetched ONCE by LDS-DMA (t1 is
    // read 1.9x instead of 9x through L2) and the nine taps then run from LDS with no pixel load on the critical path.
    if constexpr (HALO) {
        const int HR = 128 + 2 * p.W + 2, hbase = m0 - p.W - 1;
#pragma unroll
        for (int i = 0; i < HOPS; ++i) {
            const int o = i * 4 + wave, sl = o / (HROWS / 8), rb = o % (HROWS / 8);
            const int r = rb * 8 + (lane >> 3), c = (lane & 7) ^ (r & 6);         // halo swizzle: see the tap reads below
            const int vo = (r < HR && hbase + r >= 0) ? ((hbase + r) * CM + sl * 64 + c * 8) * 2 : OOB;   // past the tensor: range miss -> zeros
            ch_dma16(rs_in, smem + sl * HSL + rb * 1024, vo, 0);
        }
    }

    // ---- phase A staging (conv_igemm.hip's scheme: one byte offset per chunk + 9-bit tap masks) ----------------
    int a_off[A_CH], a_mask[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = srow + 32 * i;
        const int lch = pch ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
        const int hi0 = ho * p.stride - 1, wi0 = wo * p.stride - 1;
        a_off[i] = (((n * p.H + hi0) * p.W + wi0) * CM + lch * 8) * 2;
        int hb = 0, wb = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
            wb |= (int)((unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
        }
        int mask = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * 3)) : 0;
        a_mask[i] = mask;
    }
    int b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int row = srow + 32 * i;
        b_off[i] = (row * (9 * CM) + (pch ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const int lds_st = srow * 128 + pch * 16;
    u32x4 ra[A_CH], rb[B_CH];
    constexpr int cpt = CM / BK, nk = 9 * cpt;
    int kh = 0, kw = 0, cs = 0, tap = 0;
#define PVR_LOAD_SLICE(kt_)                                                                             \
    {                                                                                                   \
        const int tap_off = ((kh * p.W + kw) * CM + cs * BK) * 2;                                       \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                              \
            const int vo = ((a_mask[i] >> tap) & 1) ? a_off[i] + tap_off : OOB;                         \
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0));  \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, b_off[i], (kt_) * (BK * 2), 0)); \
        if (++cs == cpt) { cs = 0; ++tap; if (++kw == 3) { kw = 0; ++kh; } }                            \
    }
#define PVR_STORE_SLICE(buf_)                                                                           \
    {                                                                                                   \
        char *base = smem + (buf_) * STAGE + lds_st;                                                    \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + i * 32 * 128) = ra[i];                                    \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                \
            *reinterpret_cast<u32x4 *>(base + BM * 128 + i * 32 * 128) = rb[i];                         \
    }

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int x_rd[2][TM];                              // pixel-operand fragment offsets inside a [128][64] tile (phases A and B)
    int b_rd[2][TN];                              // conv2 weight fragments
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = wm * 64 + j * 16 + fr;
            x_rd[ks][j] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int row = wn * (BN / 2) + i * 16 + fr;
            b_rd[ks][i] = BM * 128 + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
    }

    // ---- phase-B addressing and the first group's prefetches (their latency hides under phase A) ----------------
    // lane's 8 consecutive couts of a 64-cout group start at wn*32 + fq*8; pixel of tile j is wm*64 + 16j + fr
    // byte offset of (pixel, group 0) in res / y and of the pixel row in t1': tile j adds 16 rows.  Rows past M lie past the end of
    // the buffers (num_records = M rows): the hardware range check drops those stores and returns zeros for those loads
    // y / the residual in the BLOCKED layout ([pixel >> 4][cout >> 3][pixel & 15][8], chain_params.h: between two fused tails of a
    // layer): the lane's 8 couts of pixel tile j are 16 contiguous bytes of a 256-byte run, a wave instruction covers 2 x 512 contiguous
    // bytes instead of 16 x 64-byte pieces (a quarter wave touches 2 lines instead of 16)
    const int y_nhwc = ((m0 + wm * 64 + fr) * C4 + wn * 32 + fq * 8) * 2;
    const int y_blkd = ((m0 >> 4) + wm * 4) * (C4 * 32) + (wn * 4 + fq) * 256 + fr * 16;
    const int y_off0 = p.y_blk ? y_blkd : y_nhwc, y_js = p.y_blk ? C4 * 32 : 16 * C4 * 2, y_gs = p.y_blk ? 2048 : 128;
    const int r_off0 = p.res_blk ? y_blkd : y_nhwc, r_js = p.res_blk ? C4 * 32 : 16 * C4 * 2, r_gs = p.res_blk ? 2048 : 128;
    // t1' NHWC, or blocked for the layer2 wave form that follows (chain_wave128.hip): pixel tile j = 16 rows = one block of 16 x CMN x 2 bytes, the lane's
    // 8 couts of tile pair q are chunk wn * CMN / 16 + 4 q + fq of it
    const int t_nhwc = ((m0 + wm * 64 + fr) * CMN + wn * (CMN / 2) + fq * 8) * 2;
    const int t_blkd = ((m0 >> 4) + wm * 4) * (CMN * 32) + (wn * (CMN / 16) + fq) * 256 + fr * 16;
    const int t_off0 = p.t_blk ? t_blkd : t_nhwc, t_js = p.t_blk ? CMN * 32 : 16 * CMN * 2;
#define Y_OFF(j_) (y_off0 + (j_) * y_js)
#define R_OFF(j_) (r_off0 + (j_) * r_js)
#define T_OFF(j_) (t_off0 + (j_) * t_js)
    // W3 group / W1' slice staging: thread q = tid + 256 i moves chunk (q & 7) of row r; r advances by 32 per i (the swizzle term
    // (r >> 1) & 7 does not change), so every offset is ONE per-thread base + a compile-time constant (keeps the arrays out of VGPRs)
    const int st_r = tid >> 3, st_c = ((tid & 7) ^ ((st_r >> 1) & 7)) * 16;
    const int w3_g0 = st_r * CM * 2 + st_c, w3_l0 = W3_OFF + st_r * 128 + (tid & 7) * 16;
    const int w1_g0 = st_r * C4 * 2 + st_c, w1_l0 = W1_OFF + st_r * 128 + (tid & 7) * 16;
#define W3_G(i_) (w3_g0 + ((i_) & 1) * (32 * CM * 2) + ((i_) >> 1) * 128)
#define W3_L(i_) (w3_l0 + ((i_) & 1) * 4096 + ((i_) >> 1) * 8192)
#define W1_G(i_) (w1_g0 + (i_) * (32 * C4 * 2))
#define W1_L(i_) (w1_l0 + (i_) * 4096)
    u32x4 rres[RD][TM], w3r[W3_CH], w1r[CMN ? W1_CH : 1];
    // the first group's residual / W3 / W1' prefetch: before phase A (PFK < 0), or inside its loop at slice PFK (halo form) so that
    // their RD*16 + 4*(W3_CH + W1_CH) VGPRs are free for fragment prefetch during most of phase A
#define PVR_PHASE_B_PREFETCH()                                                                                          \
    {                                                                                                                   \
        if constexpr (!DS) {                                                                                            \
            _Pragma("unroll") for (int d = 0; d < RD; ++d)                                                              \
                _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                          \
                    rres[d][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CH_KNOCK & 2) ? OOB : R_OFF(j), d * r_gs, DS ? 0 : PVR_NT_AUX(16))); \
        }                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < W3_CH; ++i)                                                               \
            w3r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, W3_G(i), 0, 0));            \
        if constexpr (CMN > 0) {                                                                                        \
            _Pragma("unroll") for (int i = 0; i < W1_CH; ++i)                                                           \
                w1r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, W1_G(i), 0, 0));        \
        }                                                                                                               \
    }
    if constexpr (!HALO || PFK < 0) PVR_PHASE_B_PREFETCH();

    // ---- phase A: conv2 3x3 as implicit GEMM, 128 pixels x CM couts, K = 9*CM --------------------------------
    f32x4 acc2[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HALO) {
        // W2 slices: global -> registers RL slices ahead -> LDS stage one slice ahead (plain loads: hipcc counts their vmcnt; no
        // store is in flight in phase A, so they retire in order).  The halo DMA is waited for ONCE, with vmcnt(0), a barrier and
        // the address set-up below between that wait and the first fragment read.
        u32x4 w2r[RL][B_CH];
#pragma unroll
        for (int q = 0; q < RL; ++q)
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                w2r[q][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, b_off[i], q * (BK * 2), 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4 *>(smem + RING_OFF + lds_st + i * 32 * 128) = w2r[0][i];
        __syncthreads();                          // halo (every wave waited for its own DMA operations) and W2 slice 0 visible
        CH_T(1);
        // fragment rows inside the halo run and the 9-bit "tap inside the image" mask of the lane's four pixels
        int h_row[TM], h_mask[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int pr = wm * 64 + j * 16 + fr, m = m0 + pr;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int wo = mm % p.W, ho = (mm / p.W) % p.H;
            int hb = 0, wb_ = 0;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                hb |= (int)(ok && (unsigned)(ho - 1 + t3) < (unsigned)p.H) << t3;
                wb_ |= (int)((unsigned)(wo - 1 + t3) < (unsigned)p.W) << t3;
            }
            int mask = 0;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb_ << (t3 * 3)) : 0;
            h_row[j] = pr; h_mask[j] = mask;
        }
#pragma unroll
        for (int kt = 0; kt < nk; ++kt) {
            if (kt == PFK) PVR_PHASE_B_PREFETCH();
            if (kt + RL < nk) {                   // register stage kt % RL held slice kt, which reached LDS during step kt - 1
#pragma unroll
                for (int i = 0; i < B_CH; ++i)
                    w2r[kt % RL][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, (CH_KNOCK & 8) ? OOB : b_off[i], (kt + RL) * (BK * 2), 0));
            }
            const int tp = kt / KS, csl = kt % KS, shift = (tp / 3) * p.W + tp % 3;
            int xo[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = h_row[j] + shift;
                const int a = csl * HSL + r * 128 + ((fq ^ (r & 6)) << 4);
                xo[j] = ((h_mask[j] >> tp) & 1) ? a : ZERO_OFF;
            }
            const char *ring = smem + RING_OFF + (kt & 1) * SLICE - BM * 128;     // (b_rd carries the per-tap form's BM*128 base)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 xa[TM], wb[TN];
#pragma unroll
                for (int j = 0; j < TM; ++j) if constexpr (!(CH_KNOCK & 32)) xa[j] = *reinterpret_cast<const V8 *>(smem + (xo[j] ^ (ks * 64)));
#pragma unroll
                for (int i = 0; i < TN; ++i) if constexpr (!(CH_KNOCK & 32)) wb[i] = *reinterpret_cast<const V8 *>(ring + b_rd[ks][i]);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc2[i][j] = mfma16<F16>(wb[i], xa[j], acc2[i][j]);
            }
            if (kt + 1 < nk) {
#pragma unroll
                for (int i = 0; i < B_CH; ++i)
                    if constexpr (!(CH_KNOCK & 4)) *reinterpret_cast<u32x4 *>(smem + RING_OFF + ((kt + 1) & 1) * SLICE + lds_st + i * 32 * 128) = w2r[(kt + 1) % RL][i];
            }
            if constexpr (!(CH_KNOCK & 16)) __syncthreads();                      // slice kt + 1 visible; every wave is done with stage kt & 1 (and, at the end, the halo)
        }
    } else {
    PVR_LOAD_SLICE(0);
    PVR_STORE_SLICE(0);
    __syncthreads();
    CH_T(1);
#define PVR_K_STEP(kt_, CUR_)                                                                           \
    {                                                                                                   \
        const bool more = (kt_) + 1 < nk;                                                               \
        if (more) PVR_LOAD_SLICE((kt_) + 1);                                                            \
        const char *sb = smem + (CUR_) * STAGE;                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
            V8 xa[TM], wb[TN];                                                                          \
            _Pragma("unroll") for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(sb + x_rd[ks][j]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i) wb[i] = *reinterpret_cast<const V8 *>(sb + b_rd[ks][i]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                              \
                _Pragma("unroll") for (int j = 0; j < TM; ++j) acc2[i][j] = mfma16<F16>(wb[i], xa[j], acc2[i][j]); \
        }                                                                                               \
        if (more) PVR_STORE_SLICE(1 - (CUR_));                                                          \
        __syncthreads();                                                                                \
    }
    for (int kt = 0; kt < nk; kt += 2) {
        PVR_K_STEP(kt, 0);
        if (kt + 1 < nk) PVR_K_STEP(kt + 1, 1);
    }
    }
    CH_T(2);
    // DS form: the lane's MFMA fragments of x (pixel = wm*64 + 16j + fr, channels (4ks + fq)*8 ..) and of Wd's first 64-cout group come
    // straight from global memory in operand layout (x is 128 B per pixel, Wd 32 KB in all: no LDS stage); requested here, where phase
    // A's accumulators and W2 stages are dead, and first used after group 0's conv3 MFMAs
    V8 xd[2][TM], wdr[2][2];
    int wd_off[2];
    if constexpr (DS) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int xo_ = (m0 + wm * 64 + j * 16 + fr) * 128 + fq * 16;      // rows past M: range miss -> zeros
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xd[ks][j] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_res, xo_, ks * 64, 0));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            wd_off[t] = (wn * 32 + t * 16 + fr) * 128 + fq * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wdr[t][ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_wd, wd_off[t] + ks * 64, 0, 0));
        }
    }
#undef PVR_K_STEP
#undef PVR_LOAD_SLICE
#undef PVR_STORE_SLICE

    // t2 = relu(acc2 + b2) -> 16-bit -> LDS [pixel][cout] (every wave is past the loop's last barrier)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int c = wn * (BN / 2) + i * 16 + fq * 4;           // D row 4*fq + reg = cout
        const float4 bv = *reinterpret_cast<const float4 *>(p.b2 + c);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = wm * 64 + j * 16 + fr;
            const f32x4 a = acc2[i][j];
            const unsigned lo = pack2_h<F16>(fmaxf(a[0] + bv.x, 0.f), fmaxf(a[1] + bv.y, 0.f));
            const unsigned hi = pack2_h<F16>(fmaxf(a[2] + bv.z, 0.f), fmaxf(a[3] + bv.w, 0.f));
            const int cc = c & 63;
            char *dst = smem + T2_OFF + (c >> 6) * 16384 + row * 128 + ((((cc >> 3) ^ ((row >> 1) & 7))) << 4) + (cc & 4) * 2;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
        }
    }
    // first W3 group / W1' slice -> LDS
#pragma unroll
    for (int i = 0; i < W3_CH; ++i) *reinterpret_cast<u32x4 *>(smem + W3_L(i)) = w3r[i];
    if constexpr (CMN > 0) {
#pragma unroll
        for (int i = 0; i < W1_CH; ++i) *reinterpret_cast<u32x4 *>(smem + W1_L(i)) = w1r[i];
    }
    if (G > 1) {
#pragma unroll
        for (int i = 0; i < W3_CH; ++i)
            w3r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, W3_G(i), 64 * CM * 2, 0));
        if constexpr (CMN > 0) {
#pragma unroll
            for (int i = 0; i < W1_CH; ++i) w1r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, W1_G(i), 128, 0));
        }
    }
    __syncthreads();
    CH_T(3);

    // ---- phase B ---------------------------------------------------------------------------------------------
    int w3_rd[2][2], w1_rd[2][CMN ? TN1 : 1], yg_wr[TM];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = wn * 32 + t * 16 + fr;
            w3_rd[ks][t] = W3_OFF + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
        if constexpr (CMN > 0) {
#pragma unroll
            for (int i = 0; i < TN1; ++i) {
                const int row = wn * (CMN / 2) + i * 16 + fr;
                w1_rd[ks][i] = W1_OFF + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * 64 + j * 16 + fr;
        yg_wr[j] = YG_OFF + row * 128 + (((wn * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }
    f32x4 acc1[CMN ? TN1 : 1][TM];
    if constexpr (CMN > 0) {
#pragma unroll
        for (int i = 0; i < TN1; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g == 1) CH_T(6);
        // conv3 group: 128 pixels x 64 couts, K = CM
        f32x4 acc3[2][TM];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc3[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS3; ++s)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 xa[TM], wb[2];
#pragma unroll
                for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(smem + T2_OFF + s * 16384 + x_rd[ks][j]);
#pragma unroll
                for (int t = 0; t < 2; ++t) wb[t] = *reinterpret_cast<const V8 *>(smem + s * 8192 + w3_rd[ks][t]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc3[t][j] = mfma16<F16>(wb[t], xa[j], acc3[t][j]);
            }
        if constexpr (DS) {                       // + Wd[group] . x  (K = 64), then the next group's Wd fragments
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc3[t][j] = mfma16<F16>(wdr[t][ks], xd[ks][j], acc3[t][j]);
            if (g + 1 < G) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        wdr[t][ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_wd, wd_off[t] + ks * 64, (g + 1) * 8192, 0));
            }
        }
        if (g == 1) CH_T(7);
        // y = relu(acc3 + b3 + residual): 8 consecutive couts per lane -> 16-B global store + 16-B LDS write
        const int c0 = g * 64 + wn * 32 + fq * 8;
        const float4 bA = *reinterpret_cast<const float4 *>(p.b3 + c0), bB = *reinterpret_cast<const float4 *>(p.b3 + c0 + 4);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const f32x4 lo = acc3[0][j], hi = acc3[1][j];
            float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
            u32x4 o;
            if constexpr (DS) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
            } else {
                const u32x4 r = rres[g % RD][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = fmaxf(v[2 * e] + from_h<F16>((u16)(r[e] & 0xffffu)), 0.f);
                    const float v1 = fmaxf(v[2 * e + 1] + from_h<F16>((u16)(r[e] >> 16)), 0.f);
                    o[e] = pack2_h<F16>(v0, v1);
                }
            }
            // nt only for the blocked layout's 1 KB runs: on NHWC 16-byte pieces it gives up write combining (PMC: 266 MB written for 205)
            if (p.y_blk) store_b128_imm<PVR_NT_AUX(4)>(o, rs_y, Y_OFF(j) + g * y_gs, 0);
            else store_b128_imm(o, rs_y, Y_OFF(j) + g * y_gs, 0);
            if constexpr (CMN > 0) *reinterpret_cast<u32x4 *>(smem + yg_wr[j]) = o;
        }
        if (!DS && g + RD < G) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
                rres[g % RD][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CH_KNOCK & 2) ? OOB : R_OFF(j), (g + RD) * r_gs, DS ? 0 : PVR_NT_AUX(16)));
        }
        if (g == 1) CH_T(8);
        __syncthreads();                          // y group visible; every wave is done with this W3 group
        if (g == 1) CH_T(9);
        if (g + 1 < G) {
#pragma unroll
            for (int i = 0; i < W3_CH; ++i) *reinterpret_cast<u32x4 *>(smem + W3_L(i)) = w3r[i];
            if (g + 2 < G) {
#pragma unroll
                for (int i = 0; i < W3_CH; ++i)
                    w3r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, W3_G(i), (g + 2) * (64 * CM * 2), 0));
            }
        }
        if (g == 1) CH_T(10);
        if constexpr (CMN == 0) __syncthreads();  // next W3 group visible (the conv1' path has its own barrier below)
        if constexpr (CMN > 0) {
            // t1' += y_group x W1'[:, group]  (K = 64)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 xa[TM], wb[TN1];
#pragma unroll
                for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(smem + YG_OFF + x_rd[ks][j]);
#pragma unroll
                for (int i = 0; i < TN1; ++i) wb[i] = *reinterpret_cast<const V8 *>(smem + w1_rd[ks][i]);
#pragma unroll
                for (int i = 0; i < TN1; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc1[i][j] = mfma16<F16>(wb[i], xa[j], acc1[i][j]);
            }
            __syncthreads();                      // every wave is done with the y group and this W1' slice
            if (g == 1) CH_T(11);
            if (g + 1 < G) {
#pragma unroll
                for (int i = 0; i < W1_CH; ++i) *reinterpret_cast<u32x4 *>(smem + W1_L(i)) = w1r[i];
                if (g + 2 < G) {
#pragma unroll
                    for (int i = 0; i < W1_CH; ++i)
                        w1r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, W1_G(i), (g + 2) * 128, 0));
                }
            }
        }
    }

    CH_T(4);
    if constexpr (CMN > 0) {
        // t1' = relu(acc1 + b1'): tile pair (2q, 2q+1) = 8 consecutive couts per lane
#pragma unroll
        for (int q = 0; q < TN1 / 2; ++q) {
            const int c0 = wn * (CMN / 2) + q * 32 + fq * 8;
            const float4 bA = *reinterpret_cast<const float4 *>(p.b1n + c0), bB = *reinterpret_cast<const float4 *>(p.b1n + c0 + 4);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const f32x4 lo = acc1[2 * q][j], hi = acc1[2 * q + 1][j];
                const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                if (p.t_blk) store_b128_imm<0>(o, rs_t, T_OFF(j) + q * 1024, 0);
                else store_b128_imm<PVR_NT_AUX(8)>(o, rs_t, T_OFF(j), q * 64);
            }
        }
    }
#undef PVR_PHASE_B_PREFETCH
#undef W3_G
#undef W3_L
#undef W1_G
#undef W1_L
#undef Y_OFF
#undef R_OFF
#undef T_OFF
#ifdef CHAIN_STAMP
    CH_T(5);
    if (threadIdx.x == 0 && blockIdx.x < 8192) { _Pragma("unroll") for (int k = 0; k < 12; ++k) chain_stamps[blockIdx.x][k] = ch_tt[k]; }
#endif
}

template <int CM, int CMN, bool F16, int RD, int OCC, bool HALO, bool DS = false, int PFK = -1>
static pvr_status launch_chain_one(ChainP &p, hipStream_t stream) {
    const int grid = (p.M + 127) / 128;
    const size_t pipe = HALO ? (size_t)(CM / 64) * (CM == 64 ? 256 : 192) * 128 + (size_t)2 * CM * 128 : (size_t)2 * (128 + CM) * 128;
    const size_t phase_b = (CM / 64) * 16384 + 16384 + (CM / 64) * 8192 + CMN * 128;      // t2, y group, W3 group, W1' slice
    const size_t lds = phase_b <= pipe ? pipe : phase_b;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bottleneck_chain_kernel<CM, CMN, F16, RD, OCC, HALO, DS, PFK>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    hipLaunchKernelGGL((bottleneck_chain_kernel<CM, CMN, F16, RD, OCC, HALO, DS, PFK>), dim3(grid), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// PVR_CHAIN_HALO=0 keeps every block on the per-tap global-load form of phase A (A/B runs; both forms are bit-identical)
static bool chain_halo_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_CHAIN_HALO"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// DS instance: 184 VGPRs uncapped (two blocks per CU) or capped at 168 with 8 spilled (three); PVR_CHAIN_DS_OCC selects, default 3
static int chain_ds_occ() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_CHAIN_DS_OCC"); v = e ? atoi(e) : 3; }
    return v;
}

static int chain_pfk() {
    static int v = -2;
    if (v < -1) { const char *e = getenv("PVR_CHAIN_PFK"); v = e ? atoi(e) : 12; }
    return v;
}

template <int CM, int CMN, bool F16, int RD, int OCC, bool DS = false>
static pvr_status launch_chain_inst(ChainP &p, hipStream_t stream) {
    // halo form: stride 1 and the 128 + 2W + 2 halo rows (+ the zero row) fit the LDS tile
    const bool halo = p.stride == 1 && 128 + 2 * p.W + 2 <= (CM == 64 ? 256 : 192) - 1 && chain_halo_enabled();
    if constexpr (DS) {
        if (halo) return chain_ds_occ() == 3 ? launch_chain_one<CM, CMN, F16, RD, 3, true, true>(p, stream) : launch_chain_one<CM, CMN, F16, RD, 2, true, true>(p, stream);
        return launch_chain_one<CM, CMN, F16, RD, 2, false, true>(p, stream);
    } else {
        if (halo) {
            // Cm = 128: phase B's first prefetch is issued at slice 12 of phase A's 18 (its 48 VGPRs are free for fragment reads until
            // then): 0.184 -> 0.178 ms per layer2 tail; PVR_CHAIN_PFK=-1 keeps it in front of phase A
            if constexpr (CM == 128) {
                if (chain_pfk() == 12) return launch_chain_one<CM, CMN, F16, RD, 2, true, false, 12>(p, stream);
            }
            return launch_chain_one<CM, CMN, F16, RD, (CM == 64 ? OCC : 2), true>(p, stream);
        }
        return launch_chain_one<CM, CMN, F16, RD, OCC, false>(p, stream);
    }
}

// tuning knob for A/B runs: PVR_CHAIN_CFG = 10*RD + OCC for the Cm = 64 instances (default 12: measured best, profiles/experiments)
static int chain_cfg() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_CHAIN_CFG"); v = e ? atoi(e) : 12; }
    return v;
}

template <bool F16>
static pvr_status launch_chain_dt(ChainP &p, int cm, int cmn, hipStream_t stream) {
    if (p.xds) {                                  // downsample inside the chain: layer1's block 0 (Cm = 64, next block's conv1 64 wide)
        if (cm == 64 && cmn == 64) return launch_chain_inst<64, 64, F16, 1, 2, true>(p, stream);
        set_error("bottleneck chain with downsample: no instance for Cm=%d, next Cm=%d", cm, cmn);
        return PVR_ERR_INVALID;
    }
    const int cfg = chain_cfg();
#define PVR_CHAIN64(CMN_)                                                                 \
    switch (cfg) {                                                                        \
    case 12: return launch_chain_inst<64, CMN_, F16, 1, 2>(p, stream);                    \
    case 13: return launch_chain_inst<64, CMN_, F16, 1, 3>(p, stream);                    \
    case 22: return launch_chain_inst<64, CMN_, F16, 2, 2>(p, stream);                    \
    case 23: return launch_chain_inst<64, CMN_, F16, 2, 3>(p, stream);                    \
    default: return launch_chain_inst<64, CMN_, F16, 4, 2>(p, stream);                    \
    }
    if (cm == 64 && cmn == 64) PVR_CHAIN64(64)
    if (cm == 64 && cmn == 128) {                 // 64 KB of LDS: two blocks per CU whatever the register cap
        if (cfg / 10 == 1) return launch_chain_inst<64, 128, F16, 1, 2>(p, stream);
        if (cfg / 10 == 2) return launch_chain_inst<64, 128, F16, 2, 2>(p, stream);
        return launch_chain_inst<64, 128, F16, 4, 2>(p, stream);
    }
    if (cm == 64 && cmn == 0) PVR_CHAIN64(0)
#undef PVR_CHAIN64
    if (cm == 128 && cmn == 128) return launch_chain_inst<128, 128, F16, 1, 2>(p, stream);
    if (cm == 128 && cmn == 0) return launch_chain_inst<128, 0, F16, 1, 2>(p, stream);
    set_error("bottleneck chain: no instance for Cm=%d, next Cm=%d", cm, cmn);
    return PVR_ERR_INVALID;
}

// PVR_CHAIN_WAVE=0 keeps the stride-1 Cm = 64 tails on the block form above (A/B runs; both forms are bit-identical); read per plan
static bool chain_wave_enabled() {
    const char *e = getenv("PVR_CHAIN_WAVE");
    return !e || atoi(e) != 0;
}

bool chain_uses_wave_form(int cm, int cmn, int stride, bool ds) { return chain_wave_enabled() && chain_wave_supported(cm, cmn, stride, ds); }

bool chain_uses_wave128(int cm, int cmn, int stride, int64_t M) { return chain_wave_enabled() && chain_wave128_supported(cm, cmn, stride, M); }

bool chain_ds_supported(int cm, int cmn, int cin, int stride) { return cm == 64 && cmn == 64 && cin == 64 && stride == 1; }
bool chain_supported(int cm, int cmn) { return (cm == 64 && (cmn == 0 || cmn == 64 || cmn == 128)) || (cm == 128 && (cmn == 0 || cmn == 128)); }

// row permutation of the chain's 1x1 weights: inside every 32-row block, row 16t + 4a + c holds cout 8a + 4t + c
int chain_row_source(int row) { return (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

pvr_status launch_bottleneck_chain(const void *t1, const void *w2, const float *b2, const void *w3p, const float *b3, const void *res,
                                   void *y, const void *w1np, const float *b1n, void *t1n, int n, int h, int w, int cm, int cmn,
                                   int stride, int dtype, hipStream_t stream, const void *xds, const void *wdsp, const void *w3pb, const void *wdspb,
                                   int wave, int in_blk, int out_blk, const void *wpk) {
    // wave: 1 run the wave form (chain_wave.hip; the plan decided with chain_uses_wave_form), 2 the layer2 wave form (chain_wave128.hip; wpk: its packed
    // weights); w3pb / wdspb: blocked copies of w3p / wdsp for the former; in_blk / out_blk: blocked activations between two wave-form launches
    // (block form: out_blk bit 1 = t1' blocked too, for a layer2 wave-form launch that follows)
    // xds != null: `res` is unused; the identity branch is Wd . x (x = xds: [pixels][64], wdsp: [4Cm][64] row-permuted) and b3 = b3 + bd
    PVR_REQUIRE(chain_supported(cm, cmn), "bottleneck chain: unsupported widths Cm=%d next=%d", cm, cmn);
    PVR_REQUIRE(t1 && w2 && b2 && w3p && b3 && (res || xds) && y && (cmn == 0 || (w1np && b1n && t1n)), "bottleneck chain: null argument");
    PVR_REQUIRE(!xds || (wdsp && chain_ds_supported(cm, cmn, 64, stride)), "bottleneck chain with downsample: unsupported shape");
    ChainP p;
    p.in = (const u16 *)t1; p.w2 = (const u16 *)w2; p.w3 = (const u16 *)w3p; p.w1n = (const u16 *)w1np; p.res = (const u16 *)res;
    p.b2 = b2; p.b3 = b3; p.b1n = b1n; p.y = (u16 *)y; p.t1n = (u16 *)t1n;
    p.N = n; p.H = h; p.W = w; p.stride = stride;
    p.Ho = (h + 2 - 3) / stride + 1; p.Wo = (w + 2 - 3) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    const int64_t inb = (int64_t)n * h * w * cm * 2, yb = M * 4 * cm * 2, tb = M * (cmn ? cmn : 1) * 2;
    PVR_REQUIRE(inb < 0x7ffffff0ll && yb < 0x7ffffff0ll && tb < 0x7ffffff0ll, "bottleneck chain: operand larger than 2 GiB (use a smaller chunk)");
    p.M = (int)M; p.in_bytes = (unsigned)inb; p.y_bytes = (unsigned)yb; p.t1n_bytes = (unsigned)tb;
    p.xds = (const u16 *)xds; p.wds = (const u16 *)wdsp;
    p.xds_bytes = xds ? (unsigned)(M * 64 * 2) : 0; p.wds_bytes = xds ? (unsigned)(4 * cm * 64 * 2) : 0;
    p.w2_bytes = (unsigned)(cm * 9 * cm * 2); p.w3_bytes = (unsigned)(4 * cm * cm * 2); p.w1n_bytes = (unsigned)(cmn * 4 * cm * 2);
    if (wave == 2) {
        PVR_REQUIRE(chain_wave128_supported(cm, cmn, stride, M) && !xds, "bottleneck chain: no layer2 wave form for Cm=%d next=%d stride=%d", cm, cmn, stride);
        p.wpk = (const u16 *)wpk; p.in_blk = in_blk; p.out_blk = out_blk & 1;
        return launch_chain_wave128(p, cmn, dtype, stream);
    }
    if (wave) {
        PVR_REQUIRE(chain_wave_supported(cm, cmn, stride, xds != nullptr), "bottleneck chain: no wave form for Cm=%d next=%d stride=%d", cm, cmn, stride);
        p.w3b = (const u16 *)w3pb; p.wdsb = (const u16 *)wdspb; p.in_blk = in_blk; p.out_blk = out_blk;
        return launch_chain_wave(p, cmn, dtype, stream);
    }
    PVR_REQUIRE(!(in_blk || out_blk) || M % 16 == 0, "bottleneck chain: the blocked layout needs a multiple of 16 pixels");
    p.res_blk = xds ? 0 : in_blk; p.y_blk = out_blk & 1;       // block form: y / the residual travel blocked; t1 stays NHWC, t1' too unless a layer2 wave form follows
    p.t_blk = (out_blk & 2) && cmn > 0;
    return dtype == PVR_F16 ? launch_chain_dt<true>(p, cm, cmn, stream) : launch_chain_dt<false>(p, cm, cmn, stream);
}

}  // namespace pvr
