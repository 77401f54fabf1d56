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
//
// STATUS (end of round 6): correct and bit-identical, but NOT faster than the block form (0.154-0.167 ms against 0.156 per launch at batch 256) - LDS bandwidth
// sets the pace once the weights stream through it (every wave re-reads 544 KB of fragments per 32 pixels).  Opt-in (PVR_CHAIN_WAVE_L2=1); the measurements, the
// s_memtime stamps of both schedules and the knock-outs are in profiles/experiments/r06_chain_wave128.txt.
#include "chain_params.h"

namespace pvr {

#ifndef CW8_XD
#define CW8_XD 2
#endif
#ifndef CW8_RD
#define CW8_RD 2
#endif
#ifndef CW8_KNOCK
#define CW8_KNOCK 0     // timing experiments: 1 no y / t1' stores, 2 residual loads out of range, 4 conv2 pixel loads out of range, 8 no per-unit barrier,
                        // 16 no weight staging (wrong results; build variants with -DCW8_KNOCK=bits, scripts/r06_cw8_variants.sh)
#endif

__device__ __forceinline__ int cw8_row_source(int row) { return (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

// Diagnostic build only (-DCW8_STAMP, scripts/r06_cw8_variants.sh): s_memtime stamps of workgroup 13, waves 0 and 4: per round and weight unit [top, end of the
// unit's arithmetic, past its barrier]
#ifdef CW8_STAMP
__device__ unsigned long long cw8_stamps[2][9][9][3];
#define CW8_T(u_, k_) { if (blockIdx.x == 13 && (wave & 3) == 0 && lane == 0 && hr < 9) cw8_stamps[wave >> 2][hr][u_][k_] = __builtin_amdgcn_s_memtime(); }
#else
#define CW8_T(u_, k_)
#endif

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
//
// Schedule (second form of the round; the first, profiles/experiments/r06_chain_wave128.txt, ran all eight waves through conv2 and then through conv3 /
// conv1' together: s_memtime stamps showed a matrix phase that asks HBM for nothing followed by a phase that moves the round's whole residual / y / t1'
// traffic at 5 500 cycles per unit against 2 000 uncontended - every workgroup of the chip in the same phase at the same time).  Here the workgroup's two
// wave quartets run HALF A ROUND APART: in every 9-step half-round one quartet walks the nine taps of conv2 on its tiles while the other walks the eight
// conv3 / conv1' half-group pairs (+ one step for the t1' epilogue and the next tile's set-up) of its own, then they swap.  Waves w and w + 4 share a SIMD,
// so each SIMD always holds one MFMA-dense wave and one wave that issues the epilogues' vector work and the memory requests, and HBM sees a steady stream.
// A step needs both a tap of W2 and a half-group pair of W3 / W1': the ring slots hold [tap | pair] (2 x 64 KB), staged in two batches per step through the
// same four staging registers.
template <int CMN, bool F16, bool OUTB, int XD, int RD>
__global__ __launch_bounds__(512, 2) void chain_wave128_kernel(ChainP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int NK = 36, NH = 16, NS = 9;                 // conv2 K-steps of 32 channels; half-groups of 32 couts; steps per half-round
    constexpr int PART = 32768, SLOT = 2 * PART, NQ = 4, NQB = CMN ? 4 : 2;   // a ring slot = [tap unit | half-group-pair unit]; 16-byte staging pieces per thread
    constexpr int TN1 = CMN / 16;
    constexpr int B2L = 2 * SLOT, B3L = B2L + 512, B1L = B3L + 2048, SCR = B1L + 512;
    constexpr int OOB = 0x7ffffff0;
    static_assert((XD == 2 || XD == 4) && RD >= 1 && NH % RD == 0, "prefetch depths (the pixel ring is filled tap-wise: XD slices of ONE tap)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fq = lane >> 4;
    const int grp = wave >> 2;                              // quartet 0 starts with conv2 in half-round 0, quartet 1 in half-round 1
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
    const int n_half = 2 * rounds + 1;                      // quartet 1 runs one half-round behind
    const int m_pad = (p.M + 63) & ~31;                     // a tile past the tensor: every load reads zeros, every store is dropped (range check)

    // ---- prologue: biases -> LDS; step 0's units -> ring slot 0; tap 1 -> staging registers -------------------------------------------
    if (tid < 128) *reinterpret_cast<float *>(smem + B2L + tid * 4) = p.b2[tid];
    *reinterpret_cast<float *>(smem + B3L + tid * 4) = p.b3[tid];
    if constexpr (CMN > 0) { if (tid < CMN) *reinterpret_cast<float *>(smem + B1L + tid * 4) = p.b1n[tid]; }
    u32x4 wst[NQ];
    const int st_off = tid * 16;
#define CW8_BARRIER() { if constexpr (!(CW8_KNOCK & 8)) __syncthreads(); }
#define CW8_W_LOAD(u_, nq_)                                                                                            \
    {                                                                                                                   \
        if constexpr (!(CW8_KNOCK & 16))                                                                                \
        _Pragma("unroll") for (int q = 0; q < (nq_); ++q)                                                               \
            wst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, st_off + q * 8192, (u_) * PART, 0)); \
    }
#define CW8_W_STORE(off_, nq_)                                                                                          \
    {                                                                                                                   \
        if constexpr (!(CW8_KNOCK & 16))                                                                                \
        _Pragma("unroll") for (int q = 0; q < (nq_); ++q)                                                               \
            *reinterpret_cast<u32x4 *>(smem + (off_) + st_off + q * 8192) = wst[q];                                     \
    }
    CW8_W_LOAD(0, NQ);
    CW8_W_STORE(0, NQ);
    CW8_W_LOAD(NS, NQB);
    CW8_W_STORE(PART, NQB);
    CW8_W_LOAD(1, NQ);

    int tile = t_lo + wave;                                 // the tile this wave works on (conv2 half-round, then the conv3 / conv1' half-round)
    Cw8Tile cur;
    cw8_setup<OUTB>(cur, tile < t_hi ? tile * 32 : m_pad, lane, p.M, p.H, p.W);

    u32x4 xr[XD][2];                                        // conv2 pixel fragments, K-steps kt .. kt + XD - 1
    u32x4 rres[RD][2];                                      // residual fragments, half-groups h .. h + RD - 1
    // conv2 fragments: tap tp_'s byte offsets of the lane's pixel column (tile j) -> xa[j]; the tap's four 32-channel K-steps are that offset + 1024 * slice.
    // (One address per TAP: the blocked layout makes the pixel term non-linear in the tap shift - 12 VALU instructions per load when recomputed per K-step.)
#define CW8_TAP_ADDR(xa_, tp_, A_)                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
            const int pm_ = A_.xb[j] + ((tp_) / 3 - 1) * p.W + ((tp_) % 3 - 1);                                         \
            int vo_ = (pm_ >> 4) * 4096 + (pm_ & 15) * 16 + (fq << 8);                                                  \
            vo_ = ((A_.mk[j] >> (tp_)) & 1) ? vo_ : OOB;                                                                \
            if constexpr (CW8_KNOCK & 4) vo_ = OOB;                                                                     \
            xa_[j] = vo_;                                                                                               \
        }                                                                                                               \
    }
#define CW8_ISSUE_X(slot_, xa_, sl_)                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            xr[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, xa_[j] + (sl_) * 1024, 0, 0)); \
    }
#define CW8_ISSUE_RES(slot_, h_, A_)                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            rres[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CW8_KNOCK & 2) ? OOB : A_.yi[j], (h_) * 1024, PVR_NT_AUX(2))); \
    }
    int xa_c[2];                                            // the lane's byte offsets at the tap being loaded
    if (tile < t_hi) {
        CW8_TAP_ADDR(xa_c, 0, cur);
#pragma unroll
        for (int k = 0; k < XD; ++k) CW8_ISSUE_X(k, xa_c, k);
    }
    __syncthreads();                                        // biases and step 0's units visible

    const char *const frag = smem + lane * 16;              // + slot + part + fragment * 1024
    // ONE accumulator array: conv2's in a tile's first half-round, conv1''s in its second (t2 is made from it at the end of the first; the second starts from the
    // zero operand).  As two arrays hipcc keeps both live across the whole step loop - it cannot know that a wave alternates - and spills 568 bytes per lane.
    f32x4 acc[8][2];
    u32x4 t2[4][2], oe[2];
    static_assert(CMN == 0 || TN1 == 8, "conv1' accumulators share conv2's registers");
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int hr = 0; hr < n_half; ++hr) {
        const int rel = hr - grp;
        const bool act = rel >= 0 && tile < t_hi;           // a wave without a tile (quartet 1 in half-round 0, a quartet behind its last tile, a tile-less wave
        const bool phA = act && !(rel & 1), phB = act && (rel & 1);   // of the last round) only stages its share of the weights and meets the barriers
        // ring slot of step s of this half-round: consecutive steps alternate slots, and a half-round has an ODD number of steps, so the parity flips per half-round
        const int s_even = (hr & 1) * SLOT, s_odd = SLOT - s_even;
#define CW8_SLOT(s_) (((s_) & 1) ? s_odd : s_even)
        // The three bodies of a half-round (conv2 / conv3 + conv1' / no tile) are separate straight-line loops that meet the SAME nine barriers and stage the same
        // pieces (s_barrier counts arrivals, not program counters).  With one step loop and the phase test inside it hipcc keeps every phase's loop-carried
        // registers live across the other phase's code and spills 340-570 bytes per lane.
        // staging, first batch of step s: tap (s + 1) % 9 (in the registers since the middle of the previous step) -> the other slot; the next step's half-group
        // pair -> registers; second batch, in the middle of the step: that pair -> the other slot; tap (s + 2) % 9 -> registers
#define CW8_STEP_TOP()                                                                                                  \
            {                                                                                                           \
                CW8_T(s, 0);                                                                                            \
                CW8_W_STORE(CW8_SLOT(s + 1), NQ);                                                                       \
                if ((s + 1) % NS < 8) CW8_W_LOAD(NS + (s + 1) % NS, NQB);                                               \
            }
#define CW8_STEP_END()                                                                                                  \
            {                                                                                                           \
                CW8_T(s, 1);                                                                                            \
                CW8_BARRIER();          /* step s + 1's units visible; every wave is done with step s's slot */         \
                CW8_T(s, 2);                                                                                            \
            }
#define CW8_MID_STAGE()                                                                                                 \
            {                                                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                if ((s + 1) % NS < 8) CW8_W_STORE(CW8_SLOT(s + 1) + PART, NQB);                                         \
                CW8_W_LOAD((s + 2) % NS, NQ);                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
            }
        if (phA) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                CW8_STEP_TOP();
                // ---- conv2 3x3, tap s: 32 pixels x 128 couts x 128 channels.  The weight fragments of a K-step in two groups of four, each read while the
                // other group's eight MFMAs run (hipcc on its own reads a fragment pair right in front of its MFMAs: ~150 cycles of LDS latency per 64
                // cycles of matrix work, 54 % of the wave cycles parked in the first build).
                const char *const wu = frag + CW8_SLOT(s);
                V8 wa[4], wb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wa[i] = *reinterpret_cast<const V8 *>(wu + (i * 4) * 1024);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kt = 4 * s + ks;
                    const V8 x0 = __builtin_bit_cast(V8, xr[kt % XD][0]), x1 = __builtin_bit_cast(V8, xr[kt % XD][1]);
                    if (kt + XD < NK) {                     // refill the ring slot: K-step kt + XD = slice (kt + XD) & 3 of tap (kt + XD) / 4
                        if ((kt + XD) % 4 == 0) CW8_TAP_ADDR(xa_c, (kt + XD) / 4, cur);
                        CW8_ISSUE_X(kt % XD, xa_c, (kt + XD) & 3);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) wb[i] = *reinterpret_cast<const V8 *>(wu + ((4 + i) * 4 + ks) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][0] = mfma16<F16>(wa[i], x0, kt == 0 ? zero : acc[i][0]);
                        acc[i][1] = mfma16<F16>(wa[i], x1, kt == 0 ? zero : acc[i][1]);
                    }
                    if (ks < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) wa[i] = *reinterpret_cast<const V8 *>(wu + (i * 4 + ks + 1) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[4 + i][0] = mfma16<F16>(wb[i], x0, kt == 0 ? zero : acc[4 + i][0]);
                        acc[4 + i][1] = mfma16<F16>(wb[i], x1, kt == 0 ? zero : acc[4 + i][1]);
                    }
                    if (ks == 1) CW8_MID_STAGE()
                    else __builtin_amdgcn_sched_barrier(0);
                }
                if (s == NS - 1) {
                    // t2 = relu(acc2 + b2) -> 16 bit: tile pair q of pixel tile j IS conv3's B fragment of K step q; then the residual ring of the tile's second half-round
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bA = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq) * 4);
                        const float4 bB = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq + 4) * 4);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 lo = acc[2 * q][j], hi = acc[2 * q + 1][j];
                            const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                            u32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                            t2[q][j] = o;
                        }
                    }
#pragma unroll
                    for (int d = 0; d < RD; ++d) CW8_ISSUE_RES(d, d, cur);
                }
                CW8_STEP_END();
            }
        } else if (phB) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                CW8_STEP_TOP();
                if (s < 8) {
                // ---- conv3 (+ residual) and conv1', half-groups 2 s and 2 s + 1.  Fragment groups of a half-group: G1 = W3 K-steps 0, 1 (both cout tiles),
                // G2 = W3 K-steps 2, 3, G3 = W1' cout tiles 0-3, G4 = tiles 4-7; each group is read while the group in front of it feeds eight MFMAs
                const char *const wu = frag + CW8_SLOT(s) + PART;
                V8 ga[4], gb[4];
#define CW8_G12(dst_, hh_, k0_) { dst_[0] = *reinterpret_cast<const V8 *>(wu + (((hh_) * 2 + 0) * 4 + (k0_)) * 1024);      \
                                  dst_[1] = *reinterpret_cast<const V8 *>(wu + (((hh_) * 2 + 1) * 4 + (k0_)) * 1024);      \
                                  dst_[2] = *reinterpret_cast<const V8 *>(wu + (((hh_) * 2 + 0) * 4 + (k0_) + 1) * 1024);  \
                                  dst_[3] = *reinterpret_cast<const V8 *>(wu + (((hh_) * 2 + 1) * 4 + (k0_) + 1) * 1024); }
#define CW8_G34(dst_, hh_, i0_) { _Pragma("unroll") for (int i = 0; i < 4; ++i) dst_[i] = *reinterpret_cast<const V8 *>(wu + 16384 + (((i0_) + i) * 2 + (hh_)) * 1024); }
                CW8_G12(ga, 0, 0);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int h = 2 * s + hh;
                    u32x4 rp[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) rp[j] = rres[h % RD][j];
                    if (h + RD < NH) CW8_ISSUE_RES(h % RD, h + RD, cur);
                    f32x4 acc3[2][2];
                    CW8_G12(gb, hh, 2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {              // K-steps 0, 1 (ga = {t0 k0, t1 k0, t0 k1, t1 k1})
                        acc3[0][j] = mfma16<F16>(ga[0], __builtin_bit_cast(V8, t2[0][j]), zero);
                        acc3[1][j] = mfma16<F16>(ga[1], __builtin_bit_cast(V8, t2[0][j]), zero);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc3[0][j] = mfma16<F16>(ga[2], __builtin_bit_cast(V8, t2[1][j]), acc3[0][j]);
                        acc3[1][j] = mfma16<F16>(ga[3], __builtin_bit_cast(V8, t2[1][j]), acc3[1][j]);
                    }
                    if constexpr (CMN > 0) { CW8_G34(ga, hh, 0); }
                    else if (hh == 0) { CW8_G12(ga, 1, 0); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {              // K-steps 2, 3
                        acc3[0][j] = mfma16<F16>(gb[0], __builtin_bit_cast(V8, t2[2][j]), acc3[0][j]);
                        acc3[1][j] = mfma16<F16>(gb[1], __builtin_bit_cast(V8, t2[2][j]), acc3[1][j]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc3[0][j] = mfma16<F16>(gb[2], __builtin_bit_cast(V8, t2[3][j]), acc3[0][j]);
                        acc3[1][j] = mfma16<F16>(gb[3], __builtin_bit_cast(V8, t2[3][j]), acc3[1][j]);
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
                        CW8_G34(gb, hh, 4);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<F16>(ga[i], __builtin_bit_cast(V8, o[j]), h == 0 ? zero : acc[i][j]);
                        if (hh == 0) { CW8_G12(ga, 1, 0); }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16<F16>(gb[i], __builtin_bit_cast(V8, o[j]), h == 0 ? zero : acc[4 + i][j]);
                    }
                    if (hh == 0) CW8_MID_STAGE()
                    else __builtin_amdgcn_sched_barrier(0);
                }
#undef CW8_G12
#undef CW8_G34
                } else {
                CW8_MID_STAGE()
                {
                    // ---- step 8 of the second half-round: t1' = relu(acc1 + b1') (tile pair q = 8 consecutive couts per lane; always blocked: the next launch is
                    // this kernel), then this wave's next tile: its addresses and the first conv2 fragments
                    if constexpr (CMN > 0) {
#pragma unroll
                        for (int q = 0; q < TN1 / 2; ++q) {
                            const float4 bA = *reinterpret_cast<const float4 *>(smem + B1L + (32 * q + 8 * fq) * 4);
                            const float4 bB = *reinterpret_cast<const float4 *>(smem + B1L + (32 * q + 8 * fq + 4) * 4);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const f32x4 lo = acc[2 * q][j], hi = acc[2 * q + 1][j];
                                const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                                u32x4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = pack2_h<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                                cw8_store<0>(o, rs_t, cur.to[j], q * 1024);
                            }
                        }
                    }
                    tile += 8;
                    if (tile < t_hi) {
                        cw8_setup<OUTB>(cur, tile * 32, lane, p.M, p.H, p.W);
                        CW8_TAP_ADDR(xa_c, 0, cur);
#pragma unroll
                        for (int k = 0; k < XD; ++k) CW8_ISSUE_X(k, xa_c, k);
                    }
                }
                }
                CW8_STEP_END();
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                CW8_STEP_TOP();
                CW8_MID_STAGE()
                CW8_STEP_END();
            }
        }
#undef CW8_MID_STAGE
#undef CW8_STEP_TOP
#undef CW8_STEP_END
#undef CW8_SLOT
    }
#undef CW8_BARRIER
#undef CW8_W_LOAD
#undef CW8_W_STORE
#undef CW8_TAP_ADDR
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

#ifdef CW8_STAMP
}  // namespace pvr
extern "C" int pvr_debug_cw8_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pvr::cw8_stamps), sizeof(pvr::cw8_stamps)); }
namespace pvr {
#endif
static long long g_cw8_launches = 0;
long long chain_wave128_launches() { return g_cw8_launches; }

// OPT-IN: PVR_CHAIN_WAVE_L2=1 runs layer2's stride-1 tails on this form (bit-identical to the block form, measured no faster: profiles/experiments/
// r06_chain_wave128.txt); read when a plan is built
bool chain_wave128_supported(int cm, int cmn, int stride, int64_t M) {
    const char *e = getenv("PVR_CHAIN_WAVE_L2");              // (plan time only: plans built under different settings coexist in the tests)
    const int on = e ? atoi(e) : 0;
    return on && cm == 128 && (cmn == 128 || cmn == 0) && stride == 1 && M % 16 == 0;
}

static int cw8_num_cus() {
    static const int v = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    return v;
}

template <int CMN, bool F16, bool OUTB>
static pvr_status launch_cw8_one(ChainP &p, hipStream_t stream) {
    constexpr int XD = CW8_XD, RD = CW8_RD;
    const size_t lds = 4 * 32768 + 512 + 2048 + 512 + 8 * 2048;
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
