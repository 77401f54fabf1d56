// Per-tile fused bottleneck for layer2's stride-1 blocks (round 5; the layer3 kernel's design, bneck_frame.hip, for a 28 x 28 image): the WHOLE block
// conv1 1x1 (512 -> 128) -> conv2 3x3 (128 -> 128) -> conv3 1x1 (128 -> 512) + identity + ReLU of a 7-row tile (196 output pixels = 13 MFMA pixel
// tiles, four tiles per image) per workgroup.  torchvision Bottleneck reached from reference src/embeddings.py:118-120, src/vision_models/moco.py:6-26.
//
//   * the tile's conv2 input t1 (7 + 2 halo rows x 28 pixels x 128 channels = 63 KB) is computed by the workgroup itself from the block input x and stays
//     in LDS: [2 channel slices][257 rows][64 channels] (rows of 128 B, chunk index XOR (row >> 1) & 7, row 256 all zeros).  Halo rows outside the image
//     are written as zeros (conv2 pads t1, not x); the two halo rows are recomputed by the neighbouring tiles (conv1 runs on 9 / 7 of the pixels: it is
//     13 % of the block's FLOPs).  x goes through the same region before t1 exists: eight 64-channel half chunks alternate between the two slices, chunk
//     h + 1 landing while h is computed.
//   * 4 waves, wave w owns 32 output channels x all pixel tiles of every convolution (conv3: four chunks of 128 couts); weights arrive as whole MFMA
//     fragments straight from L2 (fragment-blocked copies), two K tiles ahead; NO barrier inside a convolution; image reads hand-pipelined (inline-asm
//     ds_read_b128 two steps ahead, counted lgkmcnt).  65.8 KB of LDS and 256 threads per workgroup: TWO workgroups per CU, so one tile's HBM phases (x in,
//     identity in, y out) run under the other's matrix work - what the one-frame-per-CU layer3 kernel cannot do.
//   * conv2's taps: pixel p = 28 yo + xo of the tile reads image row p + 28 + 28 dy + dx; only the x borders need masks (lanes at xo + dx outside
//     [0, 28) read the zero row), the y borders are the zeroed halo rows.
// Same operand roles, K order and rounding points as the separate launches (and as bottleneck_chain.hip): bit-identical
// (tests/test_gpu_encoder.py::test_tile_bottleneck_*).
#include "common.h"

namespace pvr {

struct BTP {
    const u16 *x, *w1, *w2, *w3;   // x: block input = identity (n,28,28,512); weights fragment-blocked (launch_pack_frag_weights)
    const float *b1, *b2, *b3;
    u16 *y, *t1_out, *t2_out;      // t1_out / t2_out != nullptr (tests): conv1's / conv2's outputs of the tile's own 196 pixels also go to HBM, NHWC
    int n;
    unsigned x_bytes, w1_bytes, w2_bytes, w3_bytes, t_bytes;
};

#define BT_LDS_PTR(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

template <bool F16>
__global__ __launch_bounds__(256, 2) void bneck_tile_kernel(BTP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int IW = 28, TR = 7, NPIX = TR * IW, NT = 13, NT1 = 16, IR = (TR + 2) * IW, CM = 128, CO = 512;
    constexpr int SROWS = 257, SLICE = SROWS * 128, ZROW = 256;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, sw = (fr >> 1) & 7;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);     // an XCD owns a contiguous run of tiles: neighbouring tiles share their halo rows in its L2
    const int n = blk >> 2, q = blk & 3;
    // NHWC: the tile's 9 rows (and its 196 output pixels) are one contiguous run of pixels - image-local index lin0 + ir, lin0 = 28 (7 q - 1)
    const int lin0 = (TR * q - 1) * IW, pix0 = n * (IW * IW) + lin0;

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.x), 0, p.x_bytes, 0x00020000);
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1), 0, p.w1_bytes, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w2), 0, p.w2_bytes, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w3), 0, p.w3_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.x_bytes, 0x00020000);

    // block-input channels [64 h, 64 h + 64) of the tile's 252 pixels -> slice h & 1: 32 groups of 8 rows, one 1 KB DMA each (rows past the tile and
    // pixels outside the image: offset past num_records -> zeros)
    auto stage_x = [&](int h) {
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        for (int g = wave; g < 32; g += 4) {
            const int ir = g * 8 + (lane_c >> 3), lch = (lane_c & 7) ^ ((ir >> 1) & 7);
            const int vo = (ir < IR && (unsigned)(lin0 + ir) < (unsigned)(IW * IW)) ? ((pix0 + ir) * CO + h * 64 + lch * 8) * 2 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, BT_LDS_PTR((h & 1) * SLICE + g * 1024), 16, vo, 0, 0, 0);
        }
    };
    // weights: fragment (row tile rt, 32-deep k-step kk) of a matrix with KC = K / 8 chunks per row = 1 KB at ((rt * KC + 4 kk) * 256) bytes
    const int wlane = lane * 16;
    V8 wa[2][2], wb[2][2], wc[2][2], wd[2][2];
#define BT_LOAD_W(dst_, rs_, rt0_, KC_, kt_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            dst_[i][ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_, wlane, (((rt0_) + i) * (KC_) + 4 * (2 * (kt_) + ks)) * 256, 0));
#define BT_CHUNK_DONE() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define BT_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    BT_LOAD_W(wa, rs_w1, 2 * wave, CO / 8, 0);
    stage_x(0);
    // the zero rows (ordinary stores: hipcc waits for the DMA above in front of them - the wait this prologue needs anyway)
    if (tid < 16) *reinterpret_cast<u32x4 *>(smem + (tid >> 3) * SLICE + ZROW * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};

    // Image reads as a software pipeline of (slice, pixel tile) steps: the two fragment reads of step q + 2 are issued before the four MFMAs of step q
    // (bneck_frame.hip).  ADDR_(j): byte address of pixel tile j's fragment in slice 0 of the run; ACC_: accumulators [2][tiles]; NTL_: tiles per slice.
    V8 xs[3][2];
    // A fragment read = one base register + an instruction immediate: step q_ of a run reads pixel tile q_ % NTL_ of slice SB_ + q_ / NTL_ at
    // BASE0_(tile) [k-step 0] and BASE1_(tile) [k-step 1: the same address XOR 64, as a second base] + IMM_(tile) + slice * SLICE.  Centre-tap runs
    // (conv1, conv3) use two registers for everything (tile j = + 2048 j in the immediate); conv2's masked taps one address per tile.
#define BT_XREAD(q_, NTL_, SB_, BASE0_, BASE1_, IMM_)                                                           \
    {                                                                                                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][0]) : "v"(BASE0_((q_) % (NTL_))), "n"(IMM_((q_) % (NTL_)) + ((SB_) + (q_) / (NTL_)) * SLICE)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][1]) : "v"(BASE1_((q_) % (NTL_))), "n"(IMM_((q_) % (NTL_)) + ((SB_) + (q_) / (NTL_)) * SLICE)); \
    }
#define BT_STEP(q_, W_, NQ_, NTL_, SB_, BASE0_, BASE1_, IMM_, ACC_)                                             \
    {                                                                                                          \
        if ((q_) + 2 < (NQ_)) BT_XREAD((q_) + 2, NTL_, SB_, BASE0_, BASE1_, IMM_);                              \
        if ((q_) + 2 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));      \
        else if ((q_) + 1 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1])); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        constexpr int j_ = (q_) % (NTL_);                                                                       \
        ACC_[0][j_] = mfma16<F16>(W_[0][0], xs[(q_) % 3][0], ACC_[0][j_]);                                      \
        ACC_[1][j_] = mfma16<F16>(W_[1][0], xs[(q_) % 3][0], ACC_[1][j_]);                                      \
        ACC_[0][j_] = mfma16<F16>(W_[0][1], xs[(q_) % 3][1], ACC_[0][j_]);                                      \
        ACC_[1][j_] = mfma16<F16>(W_[1][1], xs[(q_) % 3][1], ACC_[1][j_]);                                      \
    }
#define BT_S(q_, W_, NQ_, NTL_, SB_, K_, ACC_) BT_STEP(q_, W_, NQ_, NTL_, SB_, K_##_B0, K_##_B1, K_##_IMM, ACC_)
#define BT_STEPS13(b_, W_, K_)                                                                                  \
    BT_S((b_) + 0, W_, 26, 13, 0, K_, acc) BT_S((b_) + 1, W_, 26, 13, 0, K_, acc) BT_S((b_) + 2, W_, 26, 13, 0, K_, acc) BT_S((b_) + 3, W_, 26, 13, 0, K_, acc)    \
    BT_S((b_) + 4, W_, 26, 13, 0, K_, acc) BT_S((b_) + 5, W_, 26, 13, 0, K_, acc) BT_S((b_) + 6, W_, 26, 13, 0, K_, acc) BT_S((b_) + 7, W_, 26, 13, 0, K_, acc)    \
    BT_S((b_) + 8, W_, 26, 13, 0, K_, acc) BT_S((b_) + 9, W_, 26, 13, 0, K_, acc) BT_S((b_) + 10, W_, 26, 13, 0, K_, acc) BT_S((b_) + 11, W_, 26, 13, 0, K_, acc)  \
    BT_S((b_) + 12, W_, 26, 13, 0, K_, acc)
#define BT_STEPS16(W_, SB_)                                                                                     \
    BT_S(0, W_, 16, 16, SB_, BT_C, acc1) BT_S(1, W_, 16, 16, SB_, BT_C, acc1) BT_S(2, W_, 16, 16, SB_, BT_C, acc1) BT_S(3, W_, 16, 16, SB_, BT_C, acc1)     \
    BT_S(4, W_, 16, 16, SB_, BT_C, acc1) BT_S(5, W_, 16, 16, SB_, BT_C, acc1) BT_S(6, W_, 16, 16, SB_, BT_C, acc1) BT_S(7, W_, 16, 16, SB_, BT_C, acc1)     \
    BT_S(8, W_, 16, 16, SB_, BT_C, acc1) BT_S(9, W_, 16, 16, SB_, BT_C, acc1) BT_S(10, W_, 16, 16, SB_, BT_C, acc1) BT_S(11, W_, 16, 16, SB_, BT_C, acc1)  \
    BT_S(12, W_, 16, 16, SB_, BT_C, acc1) BT_S(13, W_, 16, 16, SB_, BT_C, acc1) BT_S(14, W_, 16, 16, SB_, BT_C, acc1) BT_S(15, W_, 16, 16, SB_, BT_C, acc1)
    // one K tile over the 16 pixel tiles of slice SB_ (conv1); two K tiles over the 13 pixel tiles of slices 0 and 1 (K_ = BT_C: centre tap, conv3;
    // BT_M: conv2's masked tap addresses xa[] / xb[])
#define BT_KTILE16(W_, SB_) { BT_XREAD(0, 16, SB_, BT_C_B0, BT_C_B1, BT_C_IMM); BT_XREAD(1, 16, SB_, BT_C_B0, BT_C_B1, BT_C_IMM); BT_STEPS16(W_, SB_) }
#define BT_TWO_KTILES13(WA_, WB_, K_) { BT_XREAD(0, 13, 0, K_##_B0, K_##_B1, K_##_IMM); BT_XREAD(1, 13, 0, K_##_B0, K_##_B1, K_##_IMM); BT_STEPS13(0, WA_, K_) BT_STEPS13(13, WB_, K_) }
    const int xc = fr * 128 + ((fq ^ sw) << 4), xc64 = xc ^ 64;   // centre-tap address of pixel tile 0, k-steps 0 / 1
    int xa[NT];                                                    // conv2: per-tile address of a tap (image row or the zero row)
#define BT_C_B0(j_) xc
#define BT_C_B1(j_) xc64
#define BT_C_IMM(j_) ((j_) * 2048)
#define BT_M_B0(j_) xa[j_]
#define BT_M_B1(j_) (xa[j_] ^ 64)
#define BT_M_IMM(j_) 0

    // =================================================== conv1 (1x1, 512 -> 128) over the tile's 252 pixels ==================================
    {
        f32x4 acc1[2][NT1];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        BT_CHUNK_DONE();
#pragma unroll 1
        for (int hh = 0; hh < 4; ++hh) {
            const int h = 2 * hh;
            stage_x(h + 1);
            BT_LOAD_W(wb, rs_w1, 2 * wave, CO / 8, h + 1);
            BT_KTILE16(wa, 0);
            BT_CHUNK_DONE();                                       // half chunk h + 1 has landed; every wave is done with slice 0
            const bool lastc = hh == 3;
            if (!lastc) stage_x(h + 2);
            // (the last request: conv2's first K tile - never a branch around loads)
            BT_LOAD_W(wa, (lastc ? rs_w2 : rs_w1), 2 * wave, (lastc ? 9 * CM / 8 : CO / 8), (lastc ? 0 : h + 2));
            BT_KTILE16(wb, 1);
            if (!lastc) BT_CHUNK_DONE();
        }
        BT_LOAD_W(wb, rs_w2, 2 * wave, 9 * CM / 8, 1);
        BT_BARRIER();                                              // every wave's reads of the last half chunk are done
        // t1 = relu(conv1 + b1), rounded to the storage type, into the image; zeros in halo rows outside the image and in the rows past the tile
        const auto rs_t1o = __builtin_amdgcn_make_buffer_rsrc(p.t1_out, 0, p.t1_out ? p.t_bytes : 0, 0x00020000);
        const int c1 = 32 * wave + 8 * fq;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.b1 + c1), bh = *reinterpret_cast<const f32x4 *>(p.b1 + c1 + 4);
        char *tbase = smem + (wave >> 1) * SLICE + (((4 * (wave & 1) + fq) ^ sw) << 4);
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            const int ir = 16 * j + fr;
            const f32x4 lo = acc1[0][j], hi = acc1[1][j];
            const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
            if (ir >= IR || (unsigned)(lin0 + ir) >= (unsigned)(IW * IW)) o = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4 *>(tbase + ir * 128) = o;
            if (p.t1_out && ir >= IW && ir < IW + NPIX)            // (tests: the tile's own 196 pixels)
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_t1o, ((pix0 + ir) * CM + c1) * 2, 0, 0);
        }
    }
    BT_BARRIER();                                                  // t1 is in the image

    // =================================================== conv2: 9 taps x 2 slices ==========================================================
    f32x4 acc[2][NT];
#define BT_ZERO_ACC()                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    BT_ZERO_ACC();
    const int zaddr = ZROW * 128 + (fq << 4);
    // x-border masks, one bit per pixel tile: bit j of xmask[dx + 1] set <=> output pixel 16 j + fr exists and its column + dx is inside the image
    unsigned xmask[3] = {0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int po = 16 * j + fr, xo = po % IW;
        if (po < NPIX) {
            xmask[1] |= 1u << j;
            if (xo > 0) xmask[0] |= 1u << j;
            if (xo < IW - 1) xmask[2] |= 1u << j;
        }
    }
#define BT_SET_TAP(tap_)                                                                                        \
    {                                                                                                          \
        const int dy_ = (tap_) / 3 - 1, dx_ = (tap_) % 3 - 1, off_ = IW + dy_ * IW + dx_;                       \
        const int rsw_ = ((fr + off_) >> 1) & 7;                                                                \
        const int b0_ = (fr + off_) * 128 + ((fq ^ rsw_) << 4);                                                 \
        const unsigned m_ = dx_ < 0 ? xmask[0] : dx_ == 0 ? xmask[1] : xmask[2];                                \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) xa[j] = ((m_ >> j) & 1u) ? b0_ + j * 2048 : zaddr;       \
    }
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {
        const int t0 = 2 * tp;
        BT_SET_TAP(t0);
        BT_LOAD_W(wc, rs_w2, 2 * wave, 9 * CM / 8, 2 * t0 + 2);
        BT_LOAD_W(wd, rs_w2, 2 * wave, 9 * CM / 8, 2 * t0 + 3);
        BT_TWO_KTILES13(wa, wb, BT_M);
        BT_SET_TAP(t0 + 1);
        BT_LOAD_W(wa, rs_w2, 2 * wave, 9 * CM / 8, 2 * t0 + 4);
        BT_LOAD_W(wb, rs_w2, 2 * wave, 9 * CM / 8, 2 * t0 + 5);
        BT_TWO_KTILES13(wc, wd, BT_M);
    }
    BT_SET_TAP(8);
    BT_LOAD_W(wc, rs_w3, 2 * wave, CM / 8, 0);                    // conv3, chunk 0
    BT_LOAD_W(wd, rs_w3, 2 * wave, CM / 8, 1);
    BT_TWO_KTILES13(wa, wb, BT_M);
    BT_BARRIER();                                                  // every wave's reads of the t1 image are done
    // ---- t2 = relu(conv2 + b2), rounded, into the image (rows = the tile's 196 output pixels): this wave's 32 channels = half of slice w >> 1
    {
        const auto rs_t2o = __builtin_amdgcn_make_buffer_rsrc(p.t2_out, 0, p.t2_out ? p.t_bytes : 0, 0x00020000);
        const int c = 32 * wave + 8 * fq;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.b2 + c), bh = *reinterpret_cast<const f32x4 *>(p.b2 + c + 4);
        int fr2 = fr;
        asm volatile("" : "+v"(fr2));                              // (opaque: otherwise the t1 epilogue's store addresses are kept alive through conv2 for this one)
        char *tbase = smem + (wave >> 1) * SLICE + (((4 * (wave & 1) + fq) ^ ((fr2 >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int po = 16 * j + fr2;
            if (po < NPIX) {
                const f32x4 lo = acc[0][j], hi = acc[1][j];
                const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                *reinterpret_cast<u32x4 *>(tbase + po * 128) = o;
                if (p.t2_out) __builtin_amdgcn_raw_buffer_store_b128(o, rs_t2o, ((pix0 + IW + po) * CM + c) * 2, 0, 0);
            }
        }
    }
    BT_BARRIER();                                                  // every wave's part of t2 is in the image

    // =================================================== conv3: 4 chunks of 128 couts x 2 slices + identity ================================
    // y = relu(conv3 + b3 + identity), rounded, NHWC; a lane's tile pair = 8 consecutive couts of one pixel (all identity loads first, then the stores)
#define BT_EPI3(ch_)                                                                                            \
    {                                                                                                          \
        const int c = 128 * (ch_) + 32 * wave + 8 * fq;                                                         \
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.b3 + c), bh = *reinterpret_cast<const f32x4 *>(p.b3 + c + 4);   \
        u32x4 rr[NT];                                                                                           \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                        \
            const int po = 16 * j + fr;                                                                         \
            rr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, po < NPIX ? ((pix0 + IW + po) * CO + c) * 2 : OOB, 0, PVR_NT_AUX(512)));   \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                        \
            const int po = 16 * j + fr;                                                                         \
            const f32x4 lo = acc[0][j], hi = acc[1][j];                                                         \
            float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};   \
            u32x4 o;                                                                                            \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
                v[2 * e] += from_h<F16>((u16)(rr[j][e] & 0xffffu));                                             \
                v[2 * e + 1] += from_h<F16>((u16)(rr[j][e] >> 16));                                             \
                o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);   \
            }                                                                                                   \
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, po < NPIX ? ((pix0 + IW + po) * CO + c) * 2 : OOB, 0, PVR_NT_AUX(256));   \
        }                                                                                                       \
    }
#pragma unroll 1
    for (int cp = 0; cp < 2; ++cp) {
        const int c0 = 2 * cp;                                     // chunks c0 (fragments in wc / wd) and c0 + 1 (wa / wb)
        BT_ZERO_ACC();
        BT_LOAD_W(wa, rs_w3, 8 * (c0 + 1) + 2 * wave, CM / 8, 0);
        BT_LOAD_W(wb, rs_w3, 8 * (c0 + 1) + 2 * wave, CM / 8, 1);
        BT_TWO_KTILES13(wc, wd, BT_C);
        BT_EPI3(c0);
        BT_ZERO_ACC();
        const int cn = c0 + 2 < 4 ? c0 + 2 : 3;                    // (after the last chunk: a harmless repeat)
        BT_LOAD_W(wc, rs_w3, 8 * cn + 2 * wave, CM / 8, 0);
        BT_LOAD_W(wd, rs_w3, 8 * cn + 2 * wave, CM / 8, 1);
        BT_TWO_KTILES13(wa, wb, BT_C);
        BT_EPI3(c0 + 1);
    }
#undef BT_EPI3
#undef BT_SET_TAP
#undef BT_ZERO_ACC
#undef BT_M_IMM
#undef BT_M_B1
#undef BT_M_B0
#undef BT_C_IMM
#undef BT_C_B1
#undef BT_C_B0
#undef BT_S
#undef BT_TWO_KTILES13
#undef BT_KTILE16
#undef BT_STEPS16
#undef BT_STEPS13
#undef BT_STEP
#undef BT_XREAD
#undef BT_BARRIER
#undef BT_CHUNK_DONE
#undef BT_LOAD_W
}

static long long g_bneck_tile_launches = 0;
long long bneck_tile_launches() { return g_bneck_tile_launches; }

// shapes the kernel is built for: layer2's stride-1 bottlenecks (28 x 28 x 512 -> 128 -> 128 -> 512) in the 16-bit storage types
bool bneck_tile_supported(int n, int h, int w, int cm, int cout, int stride) {
    const char *e = getenv("PVR_TILE_BNECK");                   // (read when a plan is built: A/B switch, default on)
    const int on = e ? atoi(e) : 1;
    return on && h == 28 && w == 28 && cm == 128 && cout == 512 && stride == 1 && n >= 1 && (int64_t)n * 784 * 512 * 2 < 0x7ffffff0ll;
}

// w1p / w2p / w3p: fragment-blocked weights (launch_pack_frag_weights of the (128, 512) / (128, 1152) / (512, 128) matrices)
pvr_status launch_bneck_tile(const void *x, const void *w1p, const float *b1, const void *w2p, const float *b2, const void *w3p, const float *b3, void *y,
                             void *t1_out, void *t2_out, int n, int dtype, hipStream_t stream) {
    PVR_REQUIRE(x && w1p && b1 && w2p && b2 && w3p && b3 && y, "bneck_tile: null argument");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "bneck_tile: 16-bit storage types only");
    BTP p;
    p.x = (const u16 *)x; p.w1 = (const u16 *)w1p; p.w2 = (const u16 *)w2p; p.w3 = (const u16 *)w3p; p.b1 = b1; p.b2 = b2; p.b3 = b3;
    p.y = (u16 *)y; p.t1_out = (u16 *)t1_out; p.t2_out = (u16 *)t2_out; p.n = n;
    p.x_bytes = (unsigned)((size_t)n * 784 * 512 * 2); p.t_bytes = (unsigned)((size_t)n * 784 * 128 * 2);
    p.w1_bytes = 128u * 512 * 2; p.w2_bytes = 128u * 9 * 128 * 2; p.w3_bytes = 512u * 128 * 2;
    constexpr int lds = 2 * 257 * 128;
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    ++g_bneck_tile_launches;
    if (dtype == PVR_F16) hipLaunchKernelGGL(bneck_tile_kernel<true>, dim3(4 * n), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(bneck_tile_kernel<false>, dim3(4 * n), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
