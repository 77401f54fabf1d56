// Per-frame fused bottleneck tail for layer3 (round 5): conv2 3x3 (256 -> 256) -> conv3 1x1 (256 -> 1024) + residual + ReLU of one 14 x 14
// image per workgroup, with the block's conv2 INPUT (t1: 196 pixels x 256 channels = 98 KB of 16-bit values) resident in LDS for the whole
// launch and conv2's output handed to conv3 through the same LDS image.  torchvision Bottleneck (conv2 / bn2 / relu / conv3 / bn3 / += identity /
// relu) reached from reference src/embeddings.py:118-120 and src/vision_models/moco.py:6-26; BatchNorm is folded into the weights (encoder.hip).
//
// Why per frame: a 14 x 14 image is 196 pixels, so a whole image is ONE pixel tile of a GEMM - the 3x3 neighbourhood never leaves the workgroup
// (no halo, no im2col gather from HBM: the nine taps are nine row offsets into the LDS image), 256 frames are 256 workgroups = one per CU (the
// deep-K launches of conv_pp256 fill 224 of 256 CUs at this shape), t2 never goes to HBM (-51 MB per block at batch 256), and two launches' fixed
// costs (ramp, first DMA round trip, output burst: ~11 us each, scripts/pp256_fixed_cost.py) become one.
//
// Structure (second form; the first one - conv_pp256's ping-pong phases with the weights in an LDS-DMA ring - measured 3140 cycles per 64-deep K tile
// against an MFMA issue floor of 1664: every half-phase paid ~390 cycles of LDS latency + barrier whatever its MFMA count,
// profiles/experiments/r05_bneck_frame.txt):
//   * NO barrier inside a convolution.  The pixel operand never changes while a convolution runs (the image is resident), so only the weights move -
//     and they go straight from L2 into registers as whole MFMA fragments: wave w owns 32 output channels (two 16-row A-operand tiles) x ALL 13
//     pixel tiles = 104 accumulator VGPRs, and per 64-deep K tile it needs four 1 KB weight fragments that no other wave of the workgroup needs.
//     The weights are pre-packed in the fragment-blocked layout [row tile][K / 8][16 rows][8] (rows permuted inside every 32-row block as in
//     conv_pp256: a lane's tile pair is 8 consecutive output channels), so a fragment is one contiguous 1 KB load, requested one K tile ahead.
//   * per K tile and pixel tile: two ds_read_b128 of the image (the two 32-deep k-steps; second address = first XOR 64) feed four MFMAs.  The two
//     waves of a SIMD interleave on their own: one wave's LDS latency is the other's MFMA time.
//   * conv2's B operand: pixel p = 16 j + (lane & 15) of tile j reads row p + 14 dy + dx of the image for tap (dy, dx); lanes whose neighbour is
//     outside the image (and the 12 padding pixels of tile 12) read the zero row instead.  The row address and its swizzle are per-lane values
//     computed once per tap and tile.
//   * LDS (109.5 KB): T = 4 channel slices x [209 rows][64 channels] (rows of 128 B, the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 as in
//     conv_pp256; row 208 of every slice stays zero) | b2, b3 as fp32.  Three workgroup barriers per launch: image landed, image free, t2 written.
//   * conv3 runs in four chunks of 256 couts over the t2 image; per chunk: 4 K tiles, then bias + residual + ReLU + 16-bit stores straight from the
//     accumulators (8 consecutive couts of one pixel per lane = one 16-byte load / store).
//   * NEXT1 (form 3: every block but the stage's last): the conv3 phase waits for HBM (identity in, y out: 206 MB per launch at batch 256) with the
//     matrix pipe 40 % busy, so the NEXT block's conv1 (1x1, 1024 -> 256) runs inside it instead of as its own launch that reads y again: conv3 goes
//     in eight rounds of 128 couts (waves 2 pixel halves x 4 cout quads), every round's y values go to HBM and, as a [208 rows][128 channels] image, to
//     a second LDS region; after a barrier each wave adds that round's 128-deep slice of W1' . y to ITS 32 channels x 13 pixel tiles of t1' (104 more
//     accumulator VGPRs, alive across the rounds).  Two raw barriers per round (LDS writes visible / image free again); identity loads one round ahead.
// Same GEMM view, operand roles (weights = MFMA A operand, pixels = B operand), K order (filter taps ascending, 64-channel slices ascending, two
// 32-deep MFMA steps per slice) and rounding points (t2 and y rounded to the 16-bit storage type after bias + ReLU) as the separate conv_pp256 /
// conv_expand launches: bit-identical to them (tests/test_gpu_encoder.py::test_frame_bottleneck_op_is_bit_identical).
#include "common.h"
#include "encoder_internal.h"
#include <cstddef>

namespace pvr {

struct BFP {
    const u16 *t1, *w2, *w3, *res;  // w2 / w3: fragment-blocked (pack_frag_weights)
    const float *b2, *b3;
    u16 *y, *t2_out;               // t2_out != nullptr (tests): conv2's output also goes to HBM, NHWC
    const u16 *w1f;                // FRONT1: this block's own conv1 weights (256, 1024), fragment-blocked, and bias: the launch then reads the block INPUT
    const float *b1f;              //         (= the identity tensor `res`) and computes t1 itself
    const u16 *w1n;                // NEXT1: the next block's conv1 weights (256, 1024), fragment-blocked; its bias; its output (n,14,14,256)
    const float *b1n;
    u16 *t1n;
    int n, phases;                 // phases 1: conv2 only, 3: conv2 + conv3, 7: + the next block's conv1; + 8: the block's own conv1 in front
    unsigned t1_bytes, w2_bytes, w3_bytes, res_bytes, y_bytes, t2_bytes, w1n_bytes, t1n_bytes, w1f_bytes;
    int stagger;                   // experiment (PVR_FRAME_STAGGER): odd workgroups start `stagger` x 8128 cycles late - de-phases the CUs' HBM and matrix phases
    unsigned long long *stamps;    // diagnostics (scripts/bneck_frame_time.py): s_memtime at the phase boundaries of block 8, waves 0 and 4; nullptr in the product
    int nblk;                      // RUN: consecutive bottlenecks of the stage this launch runs per frame (blk[0 .. nblk))
    BFBlk blk[6];
};

#define BF_LDS_PTR(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

// RUN (round 6): the launch takes every frame through `nblk` CONSECUTIVE bottlenecks (layer3.1 .. 3.5) - a frame's next bottleneck needs nothing but that
// frame's own output, so the workgroup that wrote y reads it back as the next x (and identity) without a launch boundary in between: the CUs stop moving in
// step (every launch boundary re-aligned all 256 of them: all in their HBM-bound front phase together, all in the traffic-free conv2 together), y is re-read
// while it is still in the Infinity Cache, and four launches' ramps go.  Between two bottlenecks: this wave's stores retired (vmcnt(0)) and a workgroup
// barrier.
// byte offset of channel c_ of pixel p_ of this frame in the block input / identity / output.  BF_LAYOUT_KO (timing experiments only: WRONG results against NHWC
// tensors): bit 0 / 1 / 2 = the front conv1's x reads / conv3's identity reads / the y stores address a frame as [channel >> 7][pixel][channel & 127] - every
// 128-channel half chunk of a frame one contiguous 50 KB block instead of 196 pieces of 256 B at a 2 KB stride
#ifndef BF_LAYOUT_KO
#define BF_LAYOUT_KO 0
#endif
#define BF_NHWC_(p_, c_) (((n * NPIX + (p_)) * CO + (c_)) * 2)
#define BF_BLK_(p_, c_) ((((n * 8 + ((c_) >> 7)) * NPIX + (p_)) * 128 + ((c_) & 127)) * 2)
#define BF_XADDR(p_, c_) ((BF_LAYOUT_KO & 1) ? BF_BLK_(p_, c_) : BF_NHWC_(p_, c_))
#define BF_RADDR(p_, c_) ((BF_LAYOUT_KO & 2) ? BF_BLK_(p_, c_) : BF_NHWC_(p_, c_))
#define BF_YADDR(p_, c_) ((BF_LAYOUT_KO & 4) ? BF_BLK_(p_, c_) : BF_NHWC_(p_, c_))
template <bool F16, bool NEXT1, bool FRONT1 = false, bool RUN = false>
__global__ __launch_bounds__(512, 1) void bneck_frame_kernel(BFP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int NPIX = 196, IW = 14, CM = 256, CO = 1024, NT = 13;
    constexpr int SROWS = 209, SLICE = SROWS * 128, ZROW = 208;
    constexpr int Y_OFF = 4 * SLICE, YS = 208 * 128;      // NEXT1: the round's y image, 2 slices x [208 rows][64 channels] behind T (107 008 + 53 248 = 160 256 B)
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int n = blockIdx.x;
    unsigned long long ts_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned long long fs_vm = 0, fs_bar = 0;     // (BF_FRONT_STAMPS builds: cycles the front phase spent in vmcnt waits / in its barriers)
#define BF_TS(k_) { if (p.stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_[k_]) :: "memory"); }
#define BF_TS_OUT() { if (p.stamps && blockIdx.x == 8 && lane0 == 0 && (wave0 & 3) == 0) { _Pragma("unroll") for (int k = 0; k < 10; ++k) p.stamps[(wave0 >> 2) * 10 + k] = ts_[k]; } }
    if (p.stagger) {
        // start delay in units of 8128 cycles: (stagger & 255) for the last of G groups, the others spread evenly below it; pattern = stagger >> 8:
        // 0: odd / even workgroups (= odd / even XCDs), 1: two groups inside every XCD, 2: four, 3: eight
        const int pat = p.stagger >> 8, units = p.stagger & 255;
        const int G = pat <= 1 ? 2 : pat == 2 ? 4 : 8;
        const int g = pat == 0 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & (G - 1));
        const int d = units * g / (G - 1);
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(127);
    }
    BF_TS(0);
    const int nblk = RUN ? p.nblk : 1;
    int blk = 0;
    do {                                                          // (non-RUN: `while (false)` - no loop at all; a one-trip `for` changed hipcc's hoisting decisions: 109 spilled VGPRs)
    int tid = tid0;
    if constexpr (RUN) asm volatile("" : "+v"(tid));              // (opaque per bottleneck: nothing derived from the lane index is hoisted out of the loop and kept alive across it)
    const int lane = tid & 63;
    const int wave = RUN ? __builtin_amdgcn_readfirstlane(tid >> 6) : wave0;
    const int fr = lane & 15, fq = lane >> 4;
    // (the table is read from the kernel-argument segment with a scalar load at a run-time offset: indexing the by-value struct itself makes hipcc copy it to scratch)
    typedef const __attribute__((address_space(4))) BFBlk *BlkPtr;
    typedef const __attribute__((address_space(4))) char *KaPtr;
    const BlkPtr bt = (BlkPtr)((KaPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BFP, blk)) + blk;
    const u16 *P_w1f = p.w1f, *P_w2 = p.w2, *P_w3 = p.w3, *P_res = p.res;
    const float *P_b1f = p.b1f, *P_b2 = p.b2, *P_b3 = p.b3;
    u16 *P_y = p.y;
    if constexpr (RUN) { P_w1f = bt->w1f; P_w2 = bt->w2; P_w3 = bt->w3; P_res = bt->res; P_b1f = bt->b1f; P_b2 = bt->b2; P_b3 = bt->b3; P_y = bt->y; }

    const auto rs_t1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.t1), 0, p.t1_bytes, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(P_w2), 0, p.w2_bytes, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(P_w3), 0, p.w3_bytes, 0x00020000);

    // ---- the frame's t1 image -> T: 4 slices x 26 groups of 8 rows, one 1 KB DMA each (rows >= 196: offset past num_records -> zeros)
    if constexpr (!FRONT1)
    for (int u = wave; u < 104; u += 8) {
        const int s = u / 26, g = u % 26;
        const int row = g * 8 + (lane >> 3), lch = (lane & 7) ^ ((row >> 1) & 7);
        const int vo = row < NPIX ? ((n * NPIX + row) * CM + s * 64 + lch * 8) * 2 : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t1, BF_LDS_PTR(s * SLICE + g * 1024), 16, vo, 0, 0, 0);
    }
    // the zero rows (ordinary stores: hipcc waits for the DMA above in front of them - the wait this prologue needs anyway)
    if (tid < 32) *reinterpret_cast<u32x4 *>(smem + (tid >> 3) * SLICE + ZROW * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- weights: fragment (row tile rt, 32-deep k-step kk) of a matrix with KC = K / 8 chunks per row = 1 KB at ((rt * KC + 4 kk) * 256) bytes;
    //      this wave's row tiles are 2 w and 2 w + 1 (conv3: + 16 per chunk of 256 couts)
    const int wlane = lane * 16;
    V8 wa[2][2], wb[2][2];                                // two K tiles of fragments [row tile][k-step]: the one in use and the one in flight
#define BF_LOAD_W(dst_, rs_, rt0_, KC_, kt_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            dst_[i][ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_, wlane, (((rt0_) + i) * (KC_) + 4 * (2 * (kt_) + ks)) * 256, 0));
    if constexpr (!FRONT1) { BF_LOAD_W(wa, rs_w2, 2 * wave, 9 * CM / 8, 0); }

    const int sw = (fr >> 1) & 7;
    const int zaddr = ZROW * 128 + (fq << 4);             // the zero row (its XOR-64 partner is in the row too)

    f32x4 acc[2][NT];
#define BF_ZERO_ACC()                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Four 64-deep K tiles (channel slices 0..3 of the image at the addresses xa[]) as ONE software pipeline of 52 (slice, pixel tile) steps: the
    // two fragment reads of step q + 2 are issued before the four MFMAs of step q, into a rotating set of three fragment pairs; inline-asm reads and
    // counted lgkmcnt waits (hipcc's own schedule kept every read next to its use: read, wait, 4 MFMAs - the LDS latency of every tile exposed).
    // W0_ / W1_ hold the weight fragments of slices 0, 2 / 1, 3; NEXT0_ .. NEXT3_ request the fragments of the following K tiles at the slice starts.
    V8 xs[3][2];
    int xbase = 0;                                        // byte offset of the first slice of a pipeline run (the front conv1 alternates between the image's two halves)
#define BF_XREAD(q_)                                                                                            \
    {                                                                                                          \
        const int a0_ = xa[(q_) % NT] + xbase + ((q_) / NT) * SLICE;                                            \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xs[(q_) % 3][0]) : "v"(a0_));                                 \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xs[(q_) % 3][1]) : "v"(a0_ ^ 64));                            \
    }
#define BF_HOOK(q_)                                     /* (a phase may hang one more request on a step: conv3's identity loads) */
#define BF_STEP(q_, W_, NQ_)                                                                                    \
    {                                                                                                          \
        BF_HOOK(q_)                                                                                             \
        if ((q_) + 2 < (NQ_)) BF_XREAD((q_) + 2);                                                               \
        if ((q_) + 2 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));      \
        else if ((q_) + 1 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1])); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        constexpr int j_ = (q_) % NT;                                                                           \
        acc[0][j_] = mfma16<F16>(W_[0][0], xs[(q_) % 3][0], acc[0][j_]);                                        \
        acc[1][j_] = mfma16<F16>(W_[1][0], xs[(q_) % 3][0], acc[1][j_]);                                        \
        acc[0][j_] = mfma16<F16>(W_[0][1], xs[(q_) % 3][1], acc[0][j_]);                                        \
        acc[1][j_] = mfma16<F16>(W_[1][1], xs[(q_) % 3][1], acc[1][j_]);                                        \
    }
#define BF_SLICE(s_, W_, NQ_)                                                                                   \
    BF_STEP((s_) * NT + 0, W_, NQ_) BF_STEP((s_) * NT + 1, W_, NQ_) BF_STEP((s_) * NT + 2, W_, NQ_) BF_STEP((s_) * NT + 3, W_, NQ_)  \
    BF_STEP((s_) * NT + 4, W_, NQ_) BF_STEP((s_) * NT + 5, W_, NQ_) BF_STEP((s_) * NT + 6, W_, NQ_) BF_STEP((s_) * NT + 7, W_, NQ_)  \
    BF_STEP((s_) * NT + 8, W_, NQ_) BF_STEP((s_) * NT + 9, W_, NQ_) BF_STEP((s_) * NT + 10, W_, NQ_) BF_STEP((s_) * NT + 11, W_, NQ_) \
    BF_STEP((s_) * NT + 12, W_, NQ_)
    // two K tiles (slices xbase / SLICE and the next one) with the fragments WA_ / WB_: the front conv1's half chunks
#define BF_TWO_KTILES(WA_, WB_)                                                                                 \
    {                                                                                                          \
        BF_XREAD(0); BF_XREAD(1);                                                                               \
        BF_SLICE(0, WA_, 2 * NT)                                                                                \
        BF_SLICE(1, WB_, 2 * NT)                                                                                \
    }
    // a slice whose first four steps each request ONE fragment of the K tile after the next (set DST_: row tile RT_ + (j >> 1), k-step j & 1 of K tile KT_):
    // four requests per wave at the slice start made the eight waves issue 32 KB together (see the front conv1's note)
#define BF_STEP_L(q_, W_, NQ_, DST_, RS_, RT_, KC_, KT_)                                                         \
    {                                                                                                          \
        if constexpr (((q_) % NT) < 4) {                                                                        \
            DST_[((q_) % NT) >> 1][((q_) % NT) & 1] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(RS_, wlane, (((RT_) + (((q_) % NT) >> 1)) * (KC_) + 4 * (2 * (KT_) + (((q_) % NT) & 1))) * 256, 0)); \
            asm volatile("" ::: "memory");                                                                      \
        }                                                                                                       \
        BF_STEP(q_, W_, NQ_)                                                                                    \
    }
#define BF_SLICE_L(s_, W_, NQ_, DST_, RS_, RT_, KC_, KT_)                                                        \
    BF_STEP_L((s_) * NT + 0, W_, NQ_, DST_, RS_, RT_, KC_, KT_) BF_STEP_L((s_) * NT + 1, W_, NQ_, DST_, RS_, RT_, KC_, KT_) BF_STEP_L((s_) * NT + 2, W_, NQ_, DST_, RS_, RT_, KC_, KT_) \
    BF_STEP_L((s_) * NT + 3, W_, NQ_, DST_, RS_, RT_, KC_, KT_) BF_STEP((s_) * NT + 4, W_, NQ_) BF_STEP((s_) * NT + 5, W_, NQ_) BF_STEP((s_) * NT + 6, W_, NQ_)        \
    BF_STEP((s_) * NT + 7, W_, NQ_) BF_STEP((s_) * NT + 8, W_, NQ_) BF_STEP((s_) * NT + 9, W_, NQ_) BF_STEP((s_) * NT + 10, W_, NQ_) BF_STEP((s_) * NT + 11, W_, NQ_)  \
    BF_STEP((s_) * NT + 12, W_, NQ_)
    // four K tiles (slices 0..3 with wa / wb in turns); (RSk_, RTk_, KCk_, KTk_): the K tile requested during slice k into the set slice k + 1 (k + 2) uses
#define BF_FOUR_KTILES(RS0_, RT0_, KC0_, KT0_, RS1_, RT1_, KC1_, KT1_, RS2_, RT2_, KC2_, KT2_, RS3_, RT3_, KC3_, KT3_)   \
    {                                                                                                          \
        BF_XREAD(0); BF_XREAD(1);                                                                               \
        BF_SLICE_L(0, wa, 4 * NT, wb, RS0_, RT0_, KC0_, KT0_)                                                   \
        BF_SLICE_L(1, wb, 4 * NT, wa, RS1_, RT1_, KC1_, KT1_)                                                   \
        BF_SLICE_L(2, wa, 4 * NT, wb, RS2_, RT2_, KC2_, KT2_)                                                   \
        BF_SLICE_L(3, wb, 4 * NT, wa, RS3_, RT3_, KC3_, KT3_)                                                   \
    }

    BF_ZERO_ACC();
#define BF_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int xa[NT];
    if constexpr (FRONT1) {
        // =============================================== conv1 (1x1, 1024 -> 256) of THIS block, in front =====================================
        // The block input x (= the identity tensor) goes through the image region in four chunks of 256 channels - the region is free until t1
        // exists - each chunk: DMA (the t1 load's code with x's row stride), wait, barrier, four barrier-free K tiles.  The chunk's DMA latency
        // is exposed (no second buffer: 107 KB of 160); what the launch saves is conv1's own launch (its ramp, prologue, output burst and the
        // 51 MB t1 round trip).  The weight fragments of a chunk's first K tile are requested BEFORE its DMA (loads retire in order: a
        // request behind the DMA could only be waited for together with it).
        const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(P_res), 0, p.res_bytes, 0x00020000);
        const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(P_w1f), 0, p.w1f_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NT; ++j) xa[j] = (16 * j + fr) * 128 + ((fq ^ sw) << 4);
        // Eight half chunks of 128 channels rotate through THREE 53 KB regions - the two halves of the image region (slices 0-1 / 2-3) and a third one
        // behind it (slices 4-5: 160 512 B of LDS in all) - and half chunk h + 2 is requested while h is computed: a half chunk's 1664 MFMA cycles
        // per wave do not cover an HBM round trip, and with one half chunk of prefetch the phase took 58 k cycles against an MFMA floor of 27 k
        // whatever the other workgroups were doing (PVR_FRAME_STAGGER experiment: not a lockstep effect, a per-CU latency chain).
        // Loads retire in order: the weight fragments of half chunk h + 1 (wc / wd or wa / wb in turns) are requested BEFORE the DMA of h + 2, so
        // "all but the last 7 requests" = vmcnt(7) is "pixels and weights of h + 1 are here".  Every wave issues 7 DMA instructions per half chunk
        // (52 row groups over 8 waves; the spare ones repeat the all-padding group 25 of slice 1: zeros over zeros).
        V8 wc[2][2], wd[2][2];
        auto stage_x1 = [&](int h, int i) {                        // DMA instruction i (0..6) of half chunk h: block-input channels [128 h, 128 h + 128) -> slices 2 (h % 3), + 1
            int lane_c = lane;
            asm volatile("" : "+v"(lane_c));
            const int b3 = h % 3;
            const int u0 = wave + 8 * i, u = u0 < 52 ? u0 : 51;
            const int s2 = u / 26, g = u % 26;
            const int row = g * 8 + (lane_c >> 3), lch = (lane_c & 7) ^ ((row >> 1) & 7);
            const int vo = row < NPIX ? BF_XADDR(row, h * 128 + s2 * 64 + lch * 8) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, BF_LDS_PTR((2 * b3 + s2) * SLICE + g * 1024), 16, vo, 0, 0, 0);
            asm volatile("" ::: "memory");
        };
        auto stage_x = [&](int h) {
#pragma unroll
            for (int i = 0; i < 7; ++i) stage_x1(h, i);
        };
        // The 15 requests of a half chunk - the NEXT half chunk's 8 weight fragments, then the 7 DMA instructions of the one after it - go out ONE PER
        // STEP between the MFMAs of the current one: issued together at its top they stalled every wave for ~1480 cycles (s_memtime stamps: a CU's
        // vector-memory path takes 64 B/clk and the eight waves of the workgroup issue 120 KB of requests at the same moment)
#define BF_ISSUE(q_, N0_, N1_, RS_, KC_, KT_, H_)                                                                \
        {                                                                                                      \
            if constexpr ((q_) < 4)                                                                             \
                N0_[(q_) >> 1][(q_) & 1] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(RS_, wlane, ((2 * wave + ((q_) >> 1)) * (KC_) + 4 * (2 * (KT_) + ((q_) & 1))) * 256, 0)); \
            else if constexpr ((q_) < 8)                                                                        \
                N1_[((q_) - 4) >> 1][(q_) & 1] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(RS_, wlane, ((2 * wave + (((q_) - 4) >> 1)) * (KC_) + 4 * (2 * ((KT_) + 1) + ((q_) & 1))) * 256, 0)); \
            else if constexpr ((q_) < 15) stage_x1(H_, (q_) - 8);                                               \
            if constexpr ((q_) < 8) asm volatile("" ::: "memory");                                              \
        }
#define BF_STEP_I(q_, W_, N0_, N1_, RS_, KC_, KT_, H_) { BF_ISSUE(q_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP(q_, W_, 2 * NT) }
#define BF_SLICE_I(s_, W_, N0_, N1_, RS_, KC_, KT_, H_)                                                          \
        BF_STEP_I((s_) * NT + 0, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 1, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 2, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        BF_STEP_I((s_) * NT + 3, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 4, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 5, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        BF_STEP_I((s_) * NT + 6, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 7, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 8, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        BF_STEP_I((s_) * NT + 9, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 10, W_, N0_, N1_, RS_, KC_, KT_, H_) BF_STEP_I((s_) * NT + 11, W_, N0_, N1_, RS_, KC_, KT_, H_) \
        BF_STEP_I((s_) * NT + 12, W_, N0_, N1_, RS_, KC_, KT_, H_)
#define BF_TWO_KTILES_I(WA_, WB_, N0_, N1_, RS_, KC_, KT_, H_)                                                   \
        {                                                                                                      \
            BF_XREAD(0); BF_XREAD(1);                                                                           \
            BF_SLICE_I(0, WA_, N0_, N1_, RS_, KC_, KT_, H_)                                                     \
            BF_SLICE_I(1, WB_, N0_, N1_, RS_, KC_, KT_, H_)                                                     \
        }
#ifdef BF_FRONT_STAMPS      // (experiment build: where does a half chunk of the front phase wait - for its pixels / weights, or for the other waves?)
#define BF_CHUNK_DONE(n_) { unsigned long long a_, b_, c_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_) :: "memory");                                                 \
        asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b_) :: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory");                                                 \
        fs_vm += b_ - a_; fs_bar += c_ - b_; __builtin_amdgcn_sched_barrier(0); }
#else
#define BF_CHUNK_DONE(n_) { asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#endif
        if (tid < 16) *reinterpret_cast<u32x4 *>(smem + (4 + (tid >> 3)) * SLICE + ZROW * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};   // (the third region's unused last rows: defined)
        BF_LOAD_W(wa, rs_w1, 2 * wave, CO / 8, 0);
        BF_LOAD_W(wb, rs_w1, 2 * wave, CO / 8, 1);
        stage_x(0);
        stage_x(1);
        BF_CHUNK_DONE(7);                                          // half chunk 0 and its weights
#pragma unroll 1
        for (int hh = 0; hh < 4; ++hh) {
            // (opaque per iteration: the fragment addresses do not change, and hipcc otherwise hoists every step's address and its XOR-64 partner out of
            //  this loop - 55 spilled VGPRs)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(xa[j]));
            const int h = 2 * hh;
            const int hn0 = h + 2 < 8 ? h + 2 : 7;                 // (past the last half chunk: a repeat into a region nobody reads any more)
            xbase = 2 * (h % 3) * SLICE;
            BF_TWO_KTILES_I(wa, wb, wc, wd, rs_w1, CO / 8, 2 * h + 2, hn0);
            BF_CHUNK_DONE(7);                                      // half chunk h + 1 and its weights have landed; every wave is done with h's region
            const bool lastc = hh == 3;
            // (the last requests: conv2's first K tiles - never a branch around loads)
            const auto rs_n = lastc ? rs_w2 : rs_w1;
            const int kc_n = lastc ? 9 * CM / 8 : CO / 8;
            const int hn1 = h + 3 < 8 ? h + 3 : 7;
            const int kt_n = lastc ? 0 : 2 * h + 4;                // (lastc: conv2's K tiles 0 and 1; conv2 requests tile 1 again itself - harmless)
            xbase = 2 * ((h + 1) % 3) * SLICE;
            BF_TWO_KTILES_I(wc, wd, wa, wb, rs_n, kc_n, kt_n, hn1);
            if (!lastc) BF_CHUNK_DONE(7);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the repeats past the last half chunk must not land in the image once t1 is written there)
        xbase = 0;
#undef BF_CHUNK_DONE
#undef BF_TWO_KTILES_I
#undef BF_SLICE_I
#undef BF_STEP_I
#undef BF_ISSUE
        BF_BARRIER();                                             // every wave's reads of the last chunk are done
        {
            const int c1 = 32 * wave + 8 * fq;
            const f32x4 bl = *reinterpret_cast<const f32x4 *>(P_b1f + c1), bh = *reinterpret_cast<const f32x4 *>(P_b1f + c1 + 4);
            char *tbase = smem + (wave >> 1) * SLICE + (((4 * (wave & 1) + fq) ^ sw) << 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int pp = 16 * j + fr;
                const f32x4 lo = acc[0][j], hi = acc[1][j];
                const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                if (pp >= NPIX) o = u32x4{0u, 0u, 0u, 0u};        // padding pixels of tile 12: zeros, as the t1 load leaves them
                *reinterpret_cast<u32x4 *>(tbase + pp * 128) = o;
            }
        }
        BF_ZERO_ACC();
        BF_BARRIER();                                             // t1 is in the image
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the image has landed (this wave's part; the first weight fragments too)
        __syncthreads();
    }
    BF_TS(1);

    // border masks of conv2: bit (3 (dy+1) + (dx+1)) of vmask[j] set <=> pixel 16 j + fr exists and its (dy, dx) neighbour is inside the image
    int vmask[NT];
    int frm = fr;
    asm volatile("" : "+v"(frm));                                  // (computed here, after the front phase: thirteen live registers less in it)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int pp = 16 * j + frm, py = pp / IW, px = pp % IW;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            m |= (int)(pp < NPIX && (unsigned)yy < (unsigned)IW && (unsigned)xx < (unsigned)IW) << t;
        }
        vmask[j] = m;
    }
    // =================================================== conv2: 9 taps x 4 slices =====================================================
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        // row offset and swizzle of this tap's neighbour pixel; per tile: the image row or the zero row
        const int off = (tap / 3 - 1) * IW + (tap % 3 - 1);
        const int rsw = ((fr + off + 32) >> 1) & 7;
        const int b0 = (fr + off) * 128 + ((fq ^ rsw) << 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) xa[j] = ((vmask[j] >> tap) & 1) ? b0 + j * 2048 : zaddr;
        const int kt = tap * 4;
        // (the last request of the last tap is conv3's first K tile - or, in the conv2-only mode, a harmless repeat: never a branch around loads,
        //  behind which hipcc can no longer count the loads in flight and waits for all of them)
        const bool last = tap == 8;
        const auto rs_n = (last && (p.phases & 15) > 1) ? rs_w3 : rs_w2;
        const int rt_n = (NEXT1 && last) ? 2 * (wave & 3) : 2 * wave, kc_n = (last && (p.phases & 15) > 1) ? CM / 8 : 9 * CM / 8, kt_n = last ? 0 : kt + 4;
        BF_FOUR_KTILES(rs_w2, 2 * wave, 9 * CM / 8, kt + 1, rs_w2, 2 * wave, 9 * CM / 8, kt + 2,
                       rs_w2, 2 * wave, 9 * CM / 8, kt + 3, rs_n, rt_n, kc_n, kt_n);
    }
    BF_TS(2);
    BF_BARRIER();                                               // every wave's reads of the t1 image are done
    // ---- t2 = relu(conv2 + b2), rounded to the storage type, into the image: this wave's 32 channels = half of slice w >> 1
    {
        const auto rs_t2 = __builtin_amdgcn_make_buffer_rsrc(p.t2_out, 0, p.t2_out ? p.t2_bytes : 0, 0x00020000);
        const int c = 32 * wave + 8 * fq;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(P_b2 + c), bh = *reinterpret_cast<const f32x4 *>(P_b2 + c + 4);
        char *tbase = smem + (wave >> 1) * SLICE + (((4 * (wave & 1) + fq) ^ sw) << 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = 16 * j + fr;
            if (pp < NPIX) {
                const f32x4 lo = acc[0][j], hi = acc[1][j];
                const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                *reinterpret_cast<u32x4 *>(tbase + pp * 128) = o;
                if (p.t2_out) __builtin_amdgcn_raw_buffer_store_b128(o, rs_t2, ((n * NPIX + pp) * CM + c) * 2, 0, 0);
            }
        }
    }
    BF_TS(3);
    if ((p.phases & 15) <= 1) { BF_TS_OUT(); return; }
    BF_BARRIER();                                               // every wave's part of t2 is in the image

    // =================================================== conv3: 4 chunks of 256 couts x 4 slices ========================================
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(P_res), 0, p.res_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(P_y, 0, p.y_bytes, 0x00020000);
    if constexpr (NEXT1) {
    // ---- form 3: conv3 in eight rounds of 128 couts + the next block's conv1 over each round's y image
    const int wr = wave >> 2, wc = wave & 3;
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1n), 0, p.w1n_bytes, 0x00020000);
    const auto rs_t1n = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);
    const int xc = fr * 128 + ((fq ^ sw) << 4);                 // centre tap of pixel tile 0; tile j: + 2048 j (no border masks in a 1x1 convolution)
    const int xc3 = xc + wr * (7 * 2048);                       // conv3: this wave's pixel tiles are 7 wr .. 7 wr + 6 (the second half has six)
    f32x4 acc1[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The round's 7 pixel tiles go in two halves (4 + 3) that share the round's weights: 56 + 28 live registers for the conv3 accumulators and the
    // identity values of all seven tiles, next to the 104 of t1', did not fit (41 spilled); a half needs 32 + 16.
    u32x4 rr[4];
#define BF_LOAD_RES(r_, h_)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
        const int jj = 4 * (h_) + j, pp = 16 * (7 * wr + jj) + fr;                                              \
        rr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (pp < NPIX && jj < (wr == 0 ? 7 : 6)) ? ((n * NPIX + pp) * CO + 128 * (r_) + 32 * wc + 8 * fq) * 2 : OOB, 0, PVR_NT_AUX(512))); \
    }
    // one 64-deep K tile of conv3 (image slice s_, pixel tiles 4 h_ .. 4 h_ + 3 of this wave's seven) / of the next conv1 (y image slice s_, all 13 pixel tiles)
#define BF_KT3(s_, h_, W_)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
        if (4 * (h_) + j < 6 || (4 * (h_) + j == 6 && wr == 0)) {                                               \
            const V8 x0 = *reinterpret_cast<const V8 *>(smem + xc3 + (4 * (h_) + j) * 2048 + (s_) * SLICE);     \
            const V8 x1 = *reinterpret_cast<const V8 *>(smem + ((xc3 + (4 * (h_) + j) * 2048 + (s_) * SLICE) ^ 64)); \
            acc3[0][j] = mfma16<F16>(W_[0][0], x0, acc3[0][j]); acc3[1][j] = mfma16<F16>(W_[1][0], x0, acc3[1][j]);   \
            acc3[0][j] = mfma16<F16>(W_[0][1], x1, acc3[0][j]); acc3[1][j] = mfma16<F16>(W_[1][1], x1, acc3[1][j]);   \
        }
#define BF_KT1(s_, W_)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                            \
        const V8 x0 = *reinterpret_cast<const V8 *>(smem + Y_OFF + xc + j * 2048 + (s_) * YS);                  \
        const V8 x1 = *reinterpret_cast<const V8 *>(smem + ((Y_OFF + xc + j * 2048 + (s_) * YS) ^ 64));         \
        acc1[0][j] = mfma16<F16>(W_[0][0], x0, acc1[0][j]); acc1[1][j] = mfma16<F16>(W_[1][0], x0, acc1[1][j]);       \
        acc1[0][j] = mfma16<F16>(W_[0][1], x1, acc1[0][j]); acc1[1][j] = mfma16<F16>(W_[1][1], x1, acc1[1][j]);       \
    }
    // y = relu(conv3 + b3 + identity), rounded, of the half's tiles: to HBM (NHWC) and into the y image (slice wc >> 1, row = pixel)
#define BF_EPI3(h_)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
        const int jj = 4 * (h_) + j, pp = 16 * (7 * wr + jj) + fr;                                              \
        const bool ok = pp < NPIX && jj < (wr == 0 ? 7 : 6);                                                    \
        const f32x4 lo = acc3[0][j], hi = acc3[1][j];                                                           \
        float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};   \
        u32x4 o;                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                         \
            v[2 * e] += from_h<F16>((u16)(rr[j][e] & 0xffffu));                                                 \
            v[2 * e + 1] += from_h<F16>((u16)(rr[j][e] >> 16));                                                 \
            o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);   \
        }                                                                                                       \
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, ok ? ((n * NPIX + pp) * CO + c) * 2 : OOB, 0, PVR_NT_AUX(256));   \
        if (ok) *reinterpret_cast<u32x4 *>(ybase + pp * 128) = o;                                               \
    }
#define BF_ZERO3()                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    BF_LOAD_RES(0, 0);
    BF_TS(4);
    char *const ybase = smem + Y_OFF + (wc >> 1) * YS + (((4 * (wc & 1) + fq) ^ sw) << 4);
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        f32x4 acc3[2][4];
        const int rt3 = 8 * r + 2 * wc;                         // W3 row tiles of this wave and round (shared with wave w ^ 4)
        const int c = 128 * r + 32 * wc + 8 * fq;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(P_b3 + c), bh = *reinterpret_cast<const f32x4 *>(P_b3 + c + 4);
        BF_ZERO3();
        BF_LOAD_W(wb, rs_w3, rt3, CM / 8, 1); BF_KT3(0, 0, wa);
        BF_LOAD_W(wa, rs_w3, rt3, CM / 8, 2); BF_KT3(1, 0, wb);
        BF_LOAD_W(wb, rs_w3, rt3, CM / 8, 3); BF_KT3(2, 0, wa);
        BF_LOAD_W(wa, rs_w3, rt3, CM / 8, 0); BF_KT3(3, 0, wb);
        if (r == 0) BF_TS(5);
        if (r > 0) BF_BARRIER();                                // every wave has finished reading the previous round's y image
        BF_EPI3(0);
        BF_LOAD_RES(r, 1);
        BF_ZERO3();
        BF_LOAD_W(wb, rs_w3, rt3, CM / 8, 1); BF_KT3(0, 1, wa);
        BF_LOAD_W(wa, rs_w3, rt3, CM / 8, 2); BF_KT3(1, 1, wb);
        BF_LOAD_W(wb, rs_w3, rt3, CM / 8, 3); BF_KT3(2, 1, wa);
        BF_LOAD_W(wa, rs_w1, 2 * wave, CO / 8, 2 * r); BF_KT3(3, 1, wb);       // next conv1: K tile 2 r of 16 (its K = the 1024 channels of y)
        BF_EPI3(1);
        BF_LOAD_RES(r < 7 ? r + 1 : 7, 0);                      // the next round's identity values arrive under this round's conv1 work
        if (r == 0) BF_TS(6);
        BF_BARRIER();                                           // every wave's part of the y image is written
        BF_LOAD_W(wb, rs_w1, 2 * wave, CO / 8, 2 * r + 1); BF_KT1(0, wa);
        BF_LOAD_W(wa, rs_w3, r < 7 ? rt3 + 8 : rt3, CM / 8, 0); BF_KT1(1, wb);
        if (r == 5) BF_TS(7);
    }
#undef BF_ZERO3
#undef BF_EPI3
    // ---- t1' = relu(conv1' + b1'), rounded, NHWC: this wave's 32 channels
    {
        const int c1 = 32 * wave + 8 * fq;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.b1n + c1), bh = *reinterpret_cast<const f32x4 *>(p.b1n + c1 + 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = 16 * j + fr;
            const f32x4 lo = acc1[0][j], hi = acc1[1][j];
            const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_t1n, pp < NPIX ? ((n * NPIX + pp) * CM + c1) * 2 : OOB, 0, 0);
        }
    }
#undef BF_KT1
#undef BF_KT3
#undef BF_LOAD_RES
    } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) xa[j] = (16 * j + fr) * 128 + ((fq ^ sw) << 4);     // centre tap (padding pixels read zero-filled rows; their columns are never stored)
    BF_TS(4);
    // The chunk's thirteen identity loads go out one per step inside its K loop (steps 17 .. 25 and 30 .. 33: behind the weight requests of slices 1 and 2),
    // so that they have landed when the epilogue starts: issued at the top of the epilogue their HBM latency was exposed once per chunk and wave, and the
    // eight waves' 104 KB of requests stalled each other in the vector-memory path.  (They are behind the previous chunk's stores in the one in-order
    // vmcnt queue: those have a whole K loop to drain.)
#undef BF_HOOK
#ifndef BF_LATE_RES
#define BF_LATE_RES 0           // (A/B builds: 1 = the identity loads at the top of the epilogue, as until round 5's last kernel commit)
#endif
#define BF_HOOK(q_)                                                                                             \
    if constexpr (!BF_LATE_RES && (((q_) >= 17 && (q_) < 26) || ((q_) >= 30 && (q_) < 34))) {                                     \
        constexpr int r_ = (q_) < 26 ? (q_) - 17 : (q_) - 30 + 9;                                               \
        const int pp_ = 16 * r_ + fr;                                                                           \
        rr[r_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (pp_ < NPIX && !(p.phases & 32)) ? BF_RADDR(pp_, cc) : OOB, 0, PVR_NT_AUX(512))); \
        asm volatile("" ::: "memory");                                                                          \
    }
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        BF_ZERO_ACC();
        const int rt0 = 16 * ch + 2 * wave;
        const int rt_n = ch < 3 ? rt0 + 16 : rt0;              // (after the last chunk: a harmless repeat, see conv2)
        const int cc = 256 * ch + 32 * wave + 8 * fq;
        u32x4 rr[NT];
        BF_FOUR_KTILES(rs_w3, rt0, CM / 8, 1, rs_w3, rt0, CM / 8, 2, rs_w3, rt0, CM / 8, 3, rs_w3, rt_n, CM / 8, 0);
        if (ch == 0) BF_TS(5);
        // ---- y = relu(conv3 + b3 + identity), rounded, NHWC; a lane's tile pair = 8 consecutive couts of one pixel
        const int c = cc;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(P_b3 + c), bh = *reinterpret_cast<const f32x4 *>(P_b3 + c + 4);
        if constexpr (BF_LATE_RES != 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int pp = 16 * j + fr;
                rr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (pp < NPIX && !(p.phases & 32)) ? ((n * NPIX + pp) * CO + c) * 2 : OOB, 0, PVR_NT_AUX(512)));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = 16 * j + fr;
            const f32x4 lo = acc[0][j], hi = acc[1][j];
            float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] += from_h<F16>((u16)(rr[j][e] & 0xffffu));
                v[2 * e + 1] += from_h<F16>((u16)(rr[j][e] >> 16));
                o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, (pp < NPIX && !(p.phases & 16)) ? BF_YADDR(pp, c) : OOB, 0, PVR_NT_AUX(256));
        }
        if (ch == 0) BF_TS(6);
        if (ch == 2) BF_TS(7);
    }
#undef BF_HOOK
#define BF_HOOK(q_)
    }
    if constexpr (RUN) {
        if (blk + 1 < nblk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's y stores have reached L2 ...
            __builtin_amdgcn_s_barrier();                         // ... every wave's have, and every wave is done with the t2 image
            // (no cache invalidate: a frame's region of the ping-pong buffers is written by THIS workgroup only, and a CU's vector L1 is coherent with
            //  that CU's own stores - workgroup scope needs no invalidate outside threadgroup-split mode.  `buffer_inv sc1` here cost 14 us per
            //  bottleneck: at agent scope it also drops the XCD's L2 lines, i.e. the weight fragments all 32 CUs of the XCD are reading.)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    } while (RUN && ++blk < nblk);                                // (the bottlenecks of a RUN launch)
    BF_TS(8);
    if (p.stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BF_TS(9);
    BF_TS_OUT();
#ifdef BF_FRONT_STAMPS
    if (p.stamps && blockIdx.x == 8 && lane0 == 0 && (wave0 & 3) == 0) { p.stamps[20 + (wave0 >> 2) * 2] = fs_vm; p.stamps[21 + (wave0 >> 2) * 2] = fs_bar; }
#endif
#undef BF_TS_OUT
#undef BF_TS
#undef BF_BARRIER
#undef BF_FOUR_KTILES
#undef BF_SLICE_L
#undef BF_STEP_L
#undef BF_TWO_KTILES
#undef BF_SLICE
#undef BF_STEP
#undef BF_HOOK
#undef BF_XREAD
#undef BF_ZERO_ACC
#undef BF_LOAD_W
}

// natural [rows][K] 16-bit weights -> the fragment-blocked layout the kernel reads: rows permuted inside every 32-row block (row 16 t + 4 a + c holds
// cout 8 a + 4 t + c: chain_row_source), then [row >> 4][k >> 3][row & 15][8]
__global__ __launch_bounds__(256) void pack_frag_weights_kernel(const u16 *w, u16 *out, int rows, int K) {
    const long long chunks = (long long)rows * (K / 8);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
        const int r = (int)(i % 16), kc = (int)((i / 16) % (K / 8)), rt = (int)(i / 16 / (K / 8));
        const int row = rt * 16 + r;
        const int src = (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);
        reinterpret_cast<u32x4 *>(out)[i] = *reinterpret_cast<const u32x4 *>(w + (size_t)src * K + kc * 8);
    }
}

pvr_status launch_pack_frag_weights(const void *w, void *out, int rows, int K, hipStream_t stream) {
    PVR_REQUIRE(w && out && rows % 32 == 0 && K % 32 == 0, "pack_frag_weights: rows and K must be multiples of 32");
    const long long chunks = (long long)rows * (K / 8);
    hipLaunchKernelGGL(pack_frag_weights_kernel, dim3((unsigned)((chunks + 255) / 256 < 1024 ? (chunks + 255) / 256 : 1024)), dim3(256), 0, stream, (const u16 *)w, (u16 *)out, rows, K);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

static long long g_bneck_frame_launches = 0;
long long bneck_frame_launches() { return g_bneck_frame_launches; }

// shapes the kernel is built for: layer3's stride-1 bottlenecks (14 x 14 x 256 -> 14 x 14 x 1024) in the 16-bit storage types
bool bneck_frame_supported(int n, int h, int w, int cm, int cout, int stride) {
    const char *e = getenv("PVR_FRAME_BNECK");                  // (read when a plan is built: A/B switch, default on)
    const int on = e ? atoi(e) : 1;
    return on && h == 14 && w == 14 && cm == 256 && cout == 1024 && stride == 1 && n >= 1 && (int64_t)n * 196 * 1024 * 2 < 0x7ffffff0ll;
}

// w2p / w3p / w1np / w1fp: fragment-blocked weights (launch_pack_frag_weights of the (256, 2304) / (1024, 256) / (256, 1024) / (256, 1024) matrices).
// phases: 1 conv2 only (t2_out), 3 conv2 + conv3, 7 + the NEXT block's conv1 (w1np, b1n -> t1n); + 8: the block's OWN conv1 in front (w1fp, b1f; the
// launch reads the block input `res` instead of t1); + 16 / 32: timing knock-outs.
pvr_status launch_bneck_frame(const void *t1, const void *w2p, const float *b2, const void *w3p, const float *b3, const void *res, void *y,
                              void *t2_out, int n, int phases, int dtype, hipStream_t stream, unsigned long long *stamps, const void *w1np,
                              const float *b1n, void *t1n, const void *w1fp, const float *b1f) {
    const int ph = phases & 7, front = (phases & 8) != 0;
    // the whole bottleneck (own conv1 in front) in the 64-channel tiling (bneck_frame64.hip, round 6) unless switched off or a diagnostic form is asked for
    if (front && ph == 3 && !(phases & ~15) && !t2_out && !stamps && !w1np && w1fp && b1f && w2p && b2 && w3p && b3 && res && y && frame64_on()) {
        ++g_bneck_frame_launches;                                 // (a per-frame bottleneck launch either way; bneck_frame64_launches() counts this tiling)
        return launch_bneck_frame64(w1fp, b1f, w2p, b2, w3p, b3, res, y, n, dtype, stream, nullptr);
    }
    PVR_REQUIRE(ph == 1 || ph == 3 || ph == 7, "bneck_frame: phases must be 1, 3 or 7 (+ 8: own conv1 in front; + 16 / 32: timing knock-outs of the y stores / identity loads)");
    PVR_REQUIRE((t1 || front) && w2p && b2 && (ph <= 1 || (w3p && b3 && res && y)) && (ph > 1 || t2_out) && (ph < 7 || (w1np && b1n && t1n)) &&
                (!front || (w1fp && b1f && res && ph == 3)), "bneck_frame: null argument (the front conv1 comes with phases 3 only)");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "bneck_frame: 16-bit storage types only");
    BFP p;
    p.t1 = (const u16 *)t1; p.w2 = (const u16 *)w2p; p.w3 = (const u16 *)w3p; p.res = (const u16 *)res; p.b2 = b2; p.b3 = b3;
    p.y = (u16 *)y; p.t2_out = (u16 *)t2_out; p.n = n; p.phases = phases & ~8; p.stamps = stamps;
    p.stagger = 0;
#ifdef PVR_EXPERIMENTS
    { static const int stg = [] { const char *e = getenv("PVR_FRAME_STAGGER"); return e ? atoi(e) : 0; }(); p.stagger = stg; }   // (negative result of round 5; read once)
#endif
    p.w1n = (const u16 *)w1np; p.b1n = b1n; p.t1n = (u16 *)t1n; p.w1f = (const u16 *)w1fp; p.b1f = b1f;
    p.t1_bytes = p.t2_bytes = p.t1n_bytes = (unsigned)((size_t)n * 196 * 256 * 2);
    p.w2_bytes = 256u * 9 * 256 * 2; p.w3_bytes = w3p ? 1024u * 256 * 2 : 0; p.w1n_bytes = w1np ? 256u * 1024 * 2 : 0; p.w1f_bytes = w1fp ? 256u * 1024 * 2 : 0;
    p.res_bytes = p.y_bytes = (unsigned)((size_t)n * 196 * 1024 * 2);
    constexpr int lds = 4 * 209 * 128, lds1 = lds + 2 * 208 * 128, ldsf = 6 * 209 * 128;   // (FRONT1: a third half-chunk region behind the image)
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
        attr_done.mark();
    }
    ++g_bneck_frame_launches;
    if (ph == 7) {
        if (dtype == PVR_F16) hipLaunchKernelGGL((bneck_frame_kernel<true, true>), dim3(n), dim3(512), lds1, stream, p);
        else hipLaunchKernelGGL((bneck_frame_kernel<false, true>), dim3(n), dim3(512), lds1, stream, p);
    } else if (front) {
        if (dtype == PVR_F16) hipLaunchKernelGGL((bneck_frame_kernel<true, false, true>), dim3(n), dim3(512), ldsf, stream, p);
        else hipLaunchKernelGGL((bneck_frame_kernel<false, false, true>), dim3(n), dim3(512), ldsf, stream, p);
    } else {
        if (dtype == PVR_F16) hipLaunchKernelGGL((bneck_frame_kernel<true, false>), dim3(n), dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((bneck_frame_kernel<false, false>), dim3(n), dim3(512), lds, stream, p);
    }
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// RUN launch: `nblk` consecutive whole bottlenecks (own conv1 in front) per frame; blocks[k].res = bottleneck k's input, blocks[k].y its output (blocks[k + 1].res
// == blocks[k].y).  stagger: odd workgroups start `stagger` x 8128 cycles late.
pvr_status launch_bneck_frame_run(const BFBlk *blocks, int nblk, int n, int dtype, hipStream_t stream, int stagger) {
    PVR_REQUIRE(blocks && nblk >= 1 && nblk <= 6 && n >= 1, "bneck_frame_run: 1 .. 6 bottlenecks");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "bneck_frame_run: 16-bit storage types only");
    BFP p = {};
    p.n = n; p.phases = 3; p.nblk = nblk; p.stagger = stagger;
    for (int k = 0; k < nblk; ++k) {
        const BFBlk &b = blocks[k];
        PVR_REQUIRE(b.w1f && b.w2 && b.w3 && b.b1f && b.b2 && b.b3 && b.res && b.y && b.res != b.y && (k == 0 || b.res == blocks[k - 1].y),
                    "bneck_frame_run: null argument or a bottleneck that does not read its predecessor's output");
        p.blk[k] = b;
    }
    p.w2_bytes = 256u * 9 * 256 * 2; p.w3_bytes = 1024u * 256 * 2; p.w1f_bytes = 256u * 1024 * 2;
    p.res_bytes = p.y_bytes = (unsigned)((size_t)n * 196 * 1024 * 2);
    constexpr int ldsf = 6 * 209 * 128;
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame_kernel<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        attr_done.mark();
    }
    ++g_bneck_frame_launches;
    if (dtype == PVR_F16) hipLaunchKernelGGL((bneck_frame_kernel<true, false, true, true>), dim3(n), dim3(512), ldsf, stream, p);
    else hipLaunchKernelGGL((bneck_frame_kernel<false, false, true, true>), dim3(n), dim3(512), ldsf, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
