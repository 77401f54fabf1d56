// Whole layer3 bottleneck per frame, second tiling (round 6): ONE wave per SIMD, each wave 64 output channels x all 13 pixel tiles.
// RESULT: bit-identical, SLOWER than the 32-channel tiling (142 vs 120 us isolated, layer3 0.90 vs 0.81 ms in the network) - opt-in (PVR_FRAME64=1), kept as the
// record of the experiment; profiles/experiments/r06_bneck_frame64.txt has the stamps and knock-outs.  The premise below holds for conv2 only, and there two
// waves per SIMD hide more than halved LDS traffic gains; the 1 x 1 phases are bound by how many cache lines a CU can have in flight, in either tiling.
//
// bneck_frame_kernel<.., FRONT1> (bneck_frame.hip: conv1 1x1 1024 -> 256, conv2 3x3, conv3 1x1 256 -> 1024 + identity + ReLU of one 14 x 14 image per
// workgroup; torchvision Bottleneck reached from reference src/embeddings.py:118-120, src/vision_models/moco.py:6-26, BatchNorm folded) gives every wave 32
// output channels x 13 pixel tiles: per step a wave reads ONE pixel fragment pair from LDS (2 KB) for FOUR MFMAs (64 matrix-pipe cycles).  With eight waves
// that is 128 B / clk / CU of LDS reads at full matrix rate - exactly the LDS peak - before the DMA's writes and the measured bank conflicts (0.33 of the
// LDS cycles): the kernel is LDS-bound at 45 % matrix-busy (profiles/r06_sq_counters_conv.txt), its front phase takes 48 k cycles against a matrix floor of 27 k
// with 1-5 k of them in memory waits (s_memtime stamps, profiles/experiments/r06_bneck_frame_run.txt).
// Here a wave owns 64 output channels (four 16-row A tiles): the same fragment pair feeds EIGHT MFMAs, LDS reads per MFMA halve.  208 accumulator registers
// per wave do not fit next to a second wave on the SIMD, so the workgroup is four waves, one per SIMD, with the accumulators in the AGPR half of the unified
// 512-register file; the software pipeline (fragment reads two steps ahead, weight fragments one K tile ahead, DMA two half chunks ahead) is the only latency
// hiding - a step is 128 matrix cycles, two steps cover an LDS round trip.
// Same GEMM view, operand roles, fragment layouts (pack_frag_weights), K order per accumulator and rounding points as bneck_frame_kernel and as the separate
// launches: bit-identical to both (tests/test_gpu_encoder.py::test_frame64_*).
#include "common.h"
#include "encoder_internal.h"

namespace pvr {

struct BF64P {
    const u16 *w1, *w2, *w3, *x;   // fragment-blocked weights; x = the block input = the identity tensor, NHWC (n, 14, 14, 1024)
    const float *b1, *b2, *b3;
    u16 *y;
    int n;
    unsigned x_bytes;
    unsigned long long *stamps;    // diagnostics: s_memtime at the phase boundaries of workgroup 8, wave 0; nullptr in the product
};

#define F64_LDS_PTR(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

template <bool F16>
__global__ __launch_bounds__(256, 1) void bneck_frame64_kernel(BF64P p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int NPIX = 196, IW = 14, CM = 256, CO = 1024, NT = 13;
    constexpr int SROWS = 209, SLICE = SROWS * 128, ZROW = 208;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = the wave's 64-channel quarter of every 256-channel operand
    const int fr = lane & 15, fq = lane >> 4;
    const int n = blockIdx.x;
    unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define F64_TS(k_) { if (p.stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_[k_]) :: "memory"); }
    F64_TS(0);

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.x), 0, p.x_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.x_bytes, 0x00020000);
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1), 0, 256u * 1024 * 2, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w2), 0, 256u * 9 * 256 * 2, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w3), 0, 1024u * 256 * 2, 0x00020000);

    // weights: fragment (row tile rt, 32-deep k-step kk) of a matrix with KC = K / 8 chunks per row = 1 KB at ((rt * KC + 4 kk) * 256) bytes; this wave's row
    // tiles are 4 w .. 4 w + 3 (conv3: + 16 per chunk of 256 couts); a K tile = 2 k-steps = 8 fragments = 32 registers
    const int wlane = lane * 16;
    V8 wa[4][2], wb[4][2], wc[4][2], wd[4][2];
#define F64_FRAG(rs_, rt_, KC_, kt_, ks_) __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_, wlane, ((rt_) * (KC_) + 4 * (2 * (kt_) + (ks_))) * 256, 0))
#define F64_LOAD_W(dst_, rs_, rt0_, KC_, kt_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) dst_[i][ks] = F64_FRAG(rs_, (rt0_) + i, KC_, kt_, ks);

    const int sw = (fr >> 1) & 7;
    const int zaddr = ZROW * 128 + (fq << 4);             // the zero row (its XOR-64 partner is in the row too)

    f32x4 acc[4][NT];
#define F64_ZERO_ACC()                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i][j])); }  \
    asm volatile("s_nop 7" ::: "memory");
    // A run of (slice, pixel tile) steps as ONE software pipeline: the two fragment reads of step q + 2 are issued before the eight MFMAs of step q, into a
    // rotating set of three fragment pairs; inline-asm reads and counted lgkmcnt waits (bneck_frame.hip's scheme).
    // Fragment reads.  With the step's address as C arithmetic (`xa[j] + slice * SLICE`, `^ 64` for the second k-step) hipcc computed all 104 addresses of a
    // 52-step run ahead of it, spilled them, and reloaded each through scratch behind an s_waitcnt vmcnt(0).  So the address arithmetic lives in the asm:
    //   * F64_XREAD_C (1 x 1 convolutions: every tile reads its own pixels): ONE base register pair (pixel fr of tile 0 and its XOR-64 partner; + 65 280 for
    //     offsets past the 16-bit field), tile and slice are the read's IMMEDIATE offset 2048 j + slice * SLICE (a multiple of 128: the partner moves with it);
    //   * F64_XREAD_A (conv2: a tap's neighbour row or the zero row per tile): xa[j] per tile, the slice as immediate offset, the partner one v_xor.
    V8 xs[3][2];
    int xa[NT];
    int xc = 0, xcx = 0, xc2 = 0, xc2x = 0;
#define F64_XOFF(q_) (2048 * ((q_) % NT) + ((q_) / NT) * SLICE)
#define F64_XREAD_C(q_)                                                                                         \
    {                                                                                                          \
        if constexpr (F64_XOFF(q_) < 65536) {                                                                   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][0]) : "v"(xc), "n"(F64_XOFF(q_)));                     \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][1]) : "v"(xcx), "n"(F64_XOFF(q_)));                    \
        } else {                                                                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][0]) : "v"(xc2), "n"(F64_XOFF(q_) - 65280));            \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][1]) : "v"(xc2x), "n"(F64_XOFF(q_) - 65280));           \
        }                                                                                                       \
    }
#define F64_XREAD_A(q_)                                                                                         \
    {                                                                                                          \
        int t1_;                                                                                                \
        if constexpr (((q_) / NT) * SLICE < 65536) {                                                            \
            asm volatile("v_xor_b32 %0, 64, %1" : "=v"(t1_) : "v"(xa[(q_) % NT]));                              \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][0]) : "v"(xa[(q_) % NT]), "n"(((q_) / NT) * SLICE));   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(q_) % 3][1]) : "v"(t1_), "n"(((q_) / NT) * SLICE));             \
        } else {                                                                                                \
            int t0_;                                                                                            \
            asm volatile("v_add_u32 %0, %2, %1" : "=v"(t0_) : "v"(xa[(q_) % NT]), "n"(((q_) / NT) * SLICE - 65280));                 \
            asm volatile("v_xor_b32 %0, 64, %1" : "=v"(t1_) : "v"(t0_));                                        \
            asm volatile("ds_read_b128 %0, %1 offset:65280" : "=v"(xs[(q_) % 3][0]) : "v"(t0_));                \
            asm volatile("ds_read_b128 %0, %1 offset:65280" : "=v"(xs[(q_) % 3][1]) : "v"(t1_));                \
        }                                                                                                       \
    }
#define F64_XREAD(q_) F64_XREAD_C(q_)
    // The MFMA as inline asm: accumulator IN PLACE in the accumulator file ("+a"), operands in architectural registers.  Through the builtin hipcc treated the
    // 52 accumulators as free-floating values (destination != source C, copies between the two files inside the loops), put weight fragments into the 48
    // spare accumulator-file registers, ran out and spilled - every reload behind an s_waitcnt vmcnt(0) that also waits for the DMA in flight.  What the
    // compiler cannot see inside asm is handled here: an accumulator's consecutive MFMAs are >= 4 MFMAs apart, and F64_MFMA_DRAIN() stands between the last
    // MFMA of a phase and the first v_accvgpr_read of its epilogue (two s_nop 15 > the 16-pass worst case).
#define F64_MFMA(c_, a_, b_)                                                                                    \
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c_) : "v"(a_), "v"(b_));    \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(a_), "v"(b_));
#define F64_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
#define F64_HOOK(q_)
#ifndef F64_KO
#define F64_KO 0                // (timing experiments, WRONG results: 1 = no fragment reads / LDS waits in the steps, 2 = no weight-fragment requests in the steps)
#endif
#define F64_STEP(q_, W_, NQ_)                                                                                   \
    {                                                                                                          \
        F64_HOOK(q_)                                                                                            \
        if constexpr (!(F64_KO & 1)) {                                                                          \
        if ((q_) + 2 < (NQ_)) F64_XREAD((q_) + 2);                                                              \
        if ((q_) + 2 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));      \
        else if ((q_) + 1 < (NQ_)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1])); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(q_) % 3][0]), "+v"(xs[(q_) % 3][1]));               \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        constexpr int j_ = (q_) % NT;                                                                           \
        F64_MFMA(acc[0][j_], W_[0][0], xs[(q_) % 3][0]) F64_MFMA(acc[1][j_], W_[1][0], xs[(q_) % 3][0])         \
        F64_MFMA(acc[2][j_], W_[2][0], xs[(q_) % 3][0]) F64_MFMA(acc[3][j_], W_[3][0], xs[(q_) % 3][0])         \
        F64_MFMA(acc[0][j_], W_[0][1], xs[(q_) % 3][1]) F64_MFMA(acc[1][j_], W_[1][1], xs[(q_) % 3][1])         \
        F64_MFMA(acc[2][j_], W_[2][1], xs[(q_) % 3][1]) F64_MFMA(acc[3][j_], W_[3][1], xs[(q_) % 3][1])         \
    }
    // a step whose slice position is < 4 also requests TWO fragments of the next K tile (set DST_: row tile RT_ + pos, both k-steps of K tile KT_): all
    // eight are in flight nine steps (1152 matrix cycles) before the next slice's first step needs them
#define F64_STEP_L(q_, W_, NQ_, DST_, RS_, RT_, KC_, KT_)                                                        \
    {                                                                                                          \
        if constexpr (((q_) % NT) < 4 && !(F64_KO & 2)) {                                                       \
            DST_[(q_) % NT][0] = F64_FRAG(RS_, (RT_) + ((q_) % NT), KC_, KT_, 0);                               \
            DST_[(q_) % NT][1] = F64_FRAG(RS_, (RT_) + ((q_) % NT), KC_, KT_, 1);                               \
            asm volatile("" ::: "memory");                                                                      \
        }                                                                                                       \
        F64_STEP(q_, W_, NQ_)                                                                                   \
    }
    // (the weight fragments of a slice pinned to ARCHITECTURAL registers where its first step needs all eight anyway: left to itself hipcc put some into
    //  the 48 accumulator-file registers the 208 accumulators leave, ran out there and spilled fragments - behind s_waitcnt vmcnt(0) - with 46 VGPRs free)
#define F64_PIN(W_)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(W_[i][ks]));
#define F64_SLICE_L(s_, W_, NQ_, DST_, RS_, RT_, KC_, KT_)                                                       \
    F64_PIN(W_)                                                                                                 \
    F64_STEP_L((s_) * NT + 0, W_, NQ_, DST_, RS_, RT_, KC_, KT_) F64_STEP_L((s_) * NT + 1, W_, NQ_, DST_, RS_, RT_, KC_, KT_) F64_STEP_L((s_) * NT + 2, W_, NQ_, DST_, RS_, RT_, KC_, KT_) \
    F64_STEP_L((s_) * NT + 3, W_, NQ_, DST_, RS_, RT_, KC_, KT_) F64_STEP((s_) * NT + 4, W_, NQ_) F64_STEP((s_) * NT + 5, W_, NQ_) \
    F64_STEP((s_) * NT + 6, W_, NQ_) F64_STEP((s_) * NT + 7, W_, NQ_) F64_STEP((s_) * NT + 8, W_, NQ_) F64_STEP((s_) * NT + 9, W_, NQ_) \
    F64_STEP((s_) * NT + 10, W_, NQ_) F64_STEP((s_) * NT + 11, W_, NQ_) F64_STEP((s_) * NT + 12, W_, NQ_)
    // four K tiles (slices 0..3 with wa / wb in turns); (RSk_, RTk_, KCk_, KTk_): the K tile requested during slice k into the set slice k + 1 (k + 2) uses
#define F64_FOUR_KTILES(RS0_, RT0_, KC0_, KT0_, RS1_, RT1_, KC1_, KT1_, RS2_, RT2_, KC2_, KT2_, RS3_, RT3_, KC3_, KT3_)  \
    {                                                                                                          \
        F64_XREAD(0); F64_XREAD(1);                                                                             \
        F64_SLICE_L(0, wa, 4 * NT, wb, RS0_, RT0_, KC0_, KT0_)                                                  \
        F64_SLICE_L(1, wb, 4 * NT, wa, RS1_, RT1_, KC1_, KT1_)                                                  \
        F64_SLICE_L(2, wa, 4 * NT, wb, RS2_, RT2_, KC2_, KT2_)                                                  \
        F64_SLICE_L(3, wb, 4 * NT, wa, RS3_, RT3_, KC3_, KT3_)                                                  \
    }
#define F64_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    F64_ZERO_ACC();
    // =============================================== conv1 (1x1, 1024 -> 256), in front ===========================================================
    // The block input goes through LDS in eight half chunks of 128 channels that rotate through three 53 KB regions (slices 0-1 / 2-3 / 4-5); half chunk
    // h + 2 is requested while h is computed.  Per half chunk a wave issues 29 requests - the NEXT half chunk's 16 weight fragments (steps 0 .. 12), then
    // its 13 of the 52 DMA instructions of the one after it (steps 13 .. 25) - between the MFMAs.  Loads retire in order: "all but the last 13 requests"
    // = vmcnt(13) is "pixels and weights of h + 1 are here".
    {
        // pixel fr of tile 0 (and its XOR-64 partner) in the region of half chunk 0; every later half chunk moves both by a region step
        xc = fr * 128 + ((fq ^ sw) << 4); xcx = xc ^ 64;
        auto stage_x1 = [&](int h, int i) {                        // DMA instruction i (0..12) of half chunk h: block-input channels [128 h, 128 h + 128) -> slices 2 (h % 3), + 1
            int lane_c = lane;
            asm volatile("" : "+v"(lane_c));
            const int b3 = h % 3;
            const int u = wave + 4 * i;
            const int s2 = u / 26, g = u % 26;
            const int row = g * 8 + (lane_c >> 3), lch = (lane_c & 7) ^ ((row >> 1) & 7);
            const int vo = (((n * NPIX + row) * CO + h * 128 + s2 * 64 + lch * 8) * 2) | (row < NPIX ? 0 : OOB);     // (rows >= 196: past num_records -> zeros; no branch)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, F64_LDS_PTR((2 * b3 + s2) * SLICE + g * 1024), 16, vo, 0, 0, 0);
            asm volatile("" ::: "memory");
        };
        auto stage_x = [&](int h) {
#pragma unroll
            for (int i = 0; i < 13; ++i) stage_x1(h, i);
        };
        // fragment f (0..15) of the two K tiles KT_, KT_ + 1: f < 8 -> N0_[f >> 1][f & 1], else N1_[(f - 8) >> 1][f & 1]
#define F64_REQ(f_, N0_, N1_, RS_, KC_, KT_)                                                                     \
        {                                                                                                      \
            if constexpr ((f_) < 8) N0_[(f_) >> 1][(f_) & 1] = F64_FRAG(RS_, 4 * wave + ((f_) >> 1), KC_, KT_, (f_) & 1);                         \
            else N1_[((f_) - 8) >> 1][(f_) & 1] = F64_FRAG(RS_, 4 * wave + (((f_) - 8) >> 1), KC_, (KT_) + 1, (f_) & 1);                          \
            asm volatile("" ::: "memory");                                                                      \
        }
#define F64_ISSUE(q_, N0_, N1_, RS_, KC_, KT_, H_)                                                               \
        {                                                                                                      \
            if constexpr ((q_) < 3) { F64_REQ(2 * (q_), N0_, N1_, RS_, KC_, KT_) F64_REQ(2 * (q_) + 1, N0_, N1_, RS_, KC_, KT_) }                \
            else if constexpr ((q_) < 13) { F64_REQ((q_) + 3, N0_, N1_, RS_, KC_, KT_) }                                                          \
            else stage_x1(H_, (q_) - 13);                                                                       \
        }
#define F64_STEP_I(q_, W_, N0_, N1_, RS_, KC_, KT_, H_) { F64_ISSUE(q_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP(q_, W_, 2 * NT) }
#define F64_SLICE_I(s_, W_, N0_, N1_, RS_, KC_, KT_, H_)                                                         \
        F64_PIN(W_)                                                                                             \
        F64_STEP_I((s_) * NT + 0, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 1, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 2, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        F64_STEP_I((s_) * NT + 3, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 4, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 5, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        F64_STEP_I((s_) * NT + 6, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 7, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 8, W_, N0_, N1_, RS_, KC_, KT_, H_)   \
        F64_STEP_I((s_) * NT + 9, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 10, W_, N0_, N1_, RS_, KC_, KT_, H_) F64_STEP_I((s_) * NT + 11, W_, N0_, N1_, RS_, KC_, KT_, H_) \
        F64_STEP_I((s_) * NT + 12, W_, N0_, N1_, RS_, KC_, KT_, H_)
#define F64_TWO_KTILES_I(WA_, WB_, N0_, N1_, RS_, KC_, KT_, H_)                                                  \
        {                                                                                                      \
            F64_XREAD(0); F64_XREAD(1);                                                                         \
            F64_SLICE_I(0, WA_, N0_, N1_, RS_, KC_, KT_, H_)                                                    \
            F64_SLICE_I(1, WB_, N0_, N1_, RS_, KC_, KT_, H_)                                                    \
        }
#define F64_CHUNK_DONE() { asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        // the zero rows of all six slices (the image's rows 196 .. 207 come as zeros from the DMA: offset past num_records)
        if (tid < 48) *reinterpret_cast<u32x4 *>(smem + (tid >> 3) * SLICE + ZROW * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
        F64_LOAD_W(wa, rs_w1, 4 * wave, CO / 8, 0);
        F64_LOAD_W(wb, rs_w1, 4 * wave, CO / 8, 1);
        stage_x(0);
        stage_x(1);
        F64_CHUNK_DONE();                                          // half chunk 0 and its weights
#pragma unroll 1
        for (int hh = 0; hh < 4; ++hh) {
            const int h = 2 * hh;
            const int hn0 = h + 2 < 8 ? h + 2 : 7;                 // (past the last half chunk: a repeat into a region nobody reads any more)
            F64_TWO_KTILES_I(wa, wb, wc, wd, rs_w1, CO / 8, 2 * h + 2, hn0);
            {
                const int d = (2 * ((h + 1) % 3) - 2 * (h % 3)) * SLICE;      // (regions are multiples of 128 B: the XOR-64 partner moves with its address)
                xc += d; xcx += d;
            }
            F64_CHUNK_DONE();                                      // half chunk h + 1 and its weights have landed; every wave is done with h's region
            const bool lastc = hh == 3;
            // (past the last K tile: a harmless repeat of tiles 14 / 15 - never a branch around loads; conv2's first K tile is requested behind the loop)
            const int hn1 = h + 3 < 8 ? h + 3 : 7;
            const int kt_n = lastc ? 14 : 2 * h + 4;
            F64_TWO_KTILES_I(wc, wd, wa, wb, rs_w1, CO / 8, kt_n, hn1);
            {
                const int d = (2 * ((h + 2) % 3) - 2 * ((h + 1) % 3)) * SLICE;
                xc += d; xcx += d;
            }
            if (!lastc) F64_CHUNK_DONE();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the repeats past the last half chunk must not land in the image once t1 is written there)
        F64_LOAD_W(wa, rs_w2, 4 * wave, 9 * CM / 8, 0);           // conv2's first K tile: arrives under the t1 epilogue
#undef F64_CHUNK_DONE
#undef F64_TWO_KTILES_I
#undef F64_SLICE_I
#undef F64_STEP_I
#undef F64_ISSUE
#undef F64_REQ
        F64_MFMA_DRAIN();
        F64_BARRIER();                                            // every wave's reads of the last half chunk are done
    }
    F64_TS(1);
    // relu(acc + bias), rounded to the storage type, into the image: this wave's 64 channels = slice `wave`; the tile pair (2 pr, 2 pr + 1) of a lane is 8
    // consecutive channels 64 wave + 32 pr + 8 fq of pixel 16 j + fr (chunk index 4 pr + fq, swizzled by the row)
    auto to_image = [&](const float *bias, bool zero_pad) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int c = 64 * wave + 32 * pr + 8 * fq;
            const f32x4 bl = *reinterpret_cast<const f32x4 *>(bias + c), bh = *reinterpret_cast<const f32x4 *>(bias + c + 4);
            char *tbase = smem + wave * SLICE + (((4 * pr + fq) ^ sw) << 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int pp = 16 * j + fr;
                const f32x4 lo = acc[2 * pr][j], hi = acc[2 * pr + 1][j];
                const float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                if (pp >= NPIX) { if (!zero_pad) continue; o = u32x4{0u, 0u, 0u, 0u}; }       // padding pixels of tile 12: zeros (t1), or left as they are (t2: never read as real pixels' neighbours)
                *reinterpret_cast<u32x4 *>(tbase + pp * 128) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    to_image(p.b1, true);
    F64_ZERO_ACC();
    F64_BARRIER();                                                // t1 is in the image

    // border masks of conv2: bit (3 (dy+1) + (dx+1)) of vmask[j] set <=> pixel 16 j + fr exists and its (dy, dx) neighbour is inside the image
    int vmask[NT];
    {
        int frm = fr;
        asm volatile("" : "+v"(frm));
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
    }
    // =================================================== conv2: 9 taps x 4 slices =====================================================
#undef F64_XREAD
#define F64_XREAD(q_) F64_XREAD_A(q_)
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3 - 1) * IW + (tap % 3 - 1);
        const int rsw = ((fr + off + 32) >> 1) & 7;
        const int b0 = (fr + off) * 128 + ((fq ^ rsw) << 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) xa[j] = ((vmask[j] >> tap) & 1) ? b0 + j * 2048 : zaddr;
        const int kt = tap * 4;
        // (the last request of the last tap is conv3's first K tile: never a branch around loads)
        const bool last = tap == 8;
        const auto rs_n = last ? rs_w3 : rs_w2;
        const int kc_n = last ? CM / 8 : 9 * CM / 8, kt_n = last ? 0 : kt + 4;
        F64_FOUR_KTILES(rs_w2, 4 * wave, 9 * CM / 8, kt + 1, rs_w2, 4 * wave, 9 * CM / 8, kt + 2,
                        rs_w2, 4 * wave, 9 * CM / 8, kt + 3, rs_n, 4 * wave, kc_n, kt_n);
    }
    F64_TS(2);
    F64_MFMA_DRAIN();
    F64_BARRIER();                                                // every wave's reads of the t1 image are done
    to_image(p.b2, false);
    F64_TS(3);
    F64_BARRIER();                                                // t2 is in the image

    // =================================================== conv3: 4 chunks of 256 couts x 4 slices ========================================
#undef F64_XREAD
#define F64_XREAD(q_) F64_XREAD_C(q_)
    // The identity values of the chunk's FIRST tile pair (13 loads) go out one per step inside its K loop (steps 26 .. 38: they have landed when the epilogue
    // starts); the second pair's are requested at the top of the epilogue and arrive under the first pair's arithmetic - all 26 in flight next to two K tiles
    // of weight fragments did not fit the 256 architectural registers.
#undef F64_HOOK
#define F64_HOOK(q_)                                                                                            \
    if constexpr ((q_) >= 26 && (q_) < 39) {                                                                    \
        constexpr int j_h = (q_) - 26;                                                                          \
        const int pp_ = 16 * j_h + fro;                                                                         \
        rr0[j_h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, pp_ < NPIX ? ((n * NPIX + pp_) * CO + cc) * 2 : OOB, 0, PVR_NT_AUX(512))); \
        asm volatile("" ::: "memory");                                                                          \
    }
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        F64_ZERO_ACC();
        // (the lane's coordinates opaque per chunk: every address below is recomputed here instead of living - in scratch - across the chunks)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int fro = lane_o & 15, fqo = lane_o >> 4;
        xc = fro * 128 + ((fqo ^ ((fro >> 1) & 7)) << 4); xcx = xc ^ 64; xc2 = xc + 65280; xc2x = xcx + 65280;     // centre tap (padding pixels' columns are never stored)
        const int rt0 = 16 * ch + 4 * wave;
        const int rt_n = ch < 3 ? rt0 + 16 : rt0;              // (after the last chunk: a harmless repeat)
        const int cc = 256 * ch + 64 * wave + 8 * fqo;
        u32x4 rr0[NT], rr1[NT];
        F64_FOUR_KTILES(rs_w3, rt0, CM / 8, 1, rs_w3, rt0, CM / 8, 2, rs_w3, rt0, CM / 8, 3, rs_w3, rt_n, CM / 8, 0);
        if (ch == 0) F64_TS(4);
        // ---- y = relu(conv3 + b3 + identity), rounded, NHWC
        __builtin_amdgcn_sched_barrier(0);
        F64_MFMA_DRAIN();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = 16 * j + fro;
            rr1[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, pp < NPIX ? ((n * NPIX + pp) * CO + cc + 32) * 2 : OOB, 0, PVR_NT_AUX(512)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int c = cc + 32 * pr;
            const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.b3 + c), bh = *reinterpret_cast<const f32x4 *>(p.b3 + c + 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int pp = 16 * j + fro;
                const f32x4 lo = acc[2 * pr][j], hi = acc[2 * pr + 1][j];
                float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned rv = pr ? rr1[j][e] : rr0[j][e];
                    v[2 * e] += from_h<F16>((u16)(rv & 0xffffu));
                    v[2 * e + 1] += from_h<F16>((u16)(rv >> 16));
                    o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, pp < NPIX ? ((n * NPIX + pp) * CO + c) * 2 : OOB, 0, PVR_NT_AUX(256));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ch == 0) F64_TS(5);
    }
#undef F64_HOOK
    F64_TS(6);
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        F64_TS(7);
        if (blockIdx.x == 8 && tid == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p.stamps[k] = ts_[k];
        }
    }
#undef F64_TS
#undef F64_BARRIER
#undef F64_FOUR_KTILES
#undef F64_SLICE_L
#undef F64_PIN
#undef F64_STEP_L
#undef F64_STEP
#undef F64_MFMA
#undef F64_MFMA_DRAIN
#undef F64_XREAD
#undef F64_XREAD_A
#undef F64_XREAD_C
#undef F64_XOFF
#undef F64_ZERO_ACC
#undef F64_LOAD_W
#undef F64_FRAG
}

static long long g_bneck_frame64_launches = 0;
long long bneck_frame64_launches() { return g_bneck_frame64_launches; }

// 1 the 64-channel tiling runs the whole-bottleneck frame launches, 0 bneck_frame_kernel<.., FRONT1>; -1 (default) the environment (PVR_FRAME64, default 0:
// measured slower, profiles/experiments/r06_bneck_frame64.txt)
static int g_frame64_mode = -1;
void set_frame64(int mode) { g_frame64_mode = mode; }
bool frame64_on() {
    if (g_frame64_mode >= 0) return g_frame64_mode != 0;
    static const bool on = [] { const char *e = getenv("PVR_FRAME64"); return e && atoi(e) != 0; }();
    return on;
}

// w1p / w2p / w3p: fragment-blocked weights (launch_pack_frag_weights of the (256, 1024) / (256, 2304) / (1024, 256) matrices); x: the block input (n,14,14,1024)
pvr_status launch_bneck_frame64(const void *w1p, const float *b1, const void *w2p, const float *b2, const void *w3p, const float *b3, const void *x, void *y, int n,
                                int dtype, hipStream_t stream, unsigned long long *stamps) {
    PVR_REQUIRE(w1p && b1 && w2p && b2 && w3p && b3 && x && y && n >= 1, "bneck_frame64: null argument");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "bneck_frame64: 16-bit storage types only");
    PVR_REQUIRE((int64_t)n * 196 * 1024 * 2 < 0x7ffffff0ll, "bneck_frame64: n = %d frames exceed a 2 GiB buffer descriptor", n);
    BF64P p;
    p.w1 = (const u16 *)w1p; p.w2 = (const u16 *)w2p; p.w3 = (const u16 *)w3p; p.x = (const u16 *)x; p.b1 = b1; p.b2 = b2; p.b3 = b3; p.y = (u16 *)y;
    p.n = n; p.x_bytes = (unsigned)((size_t)n * 196 * 1024 * 2); p.stamps = stamps;
    constexpr int lds = 6 * 209 * 128;
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)bneck_frame64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    ++g_bneck_frame64_launches;
    if (dtype == PVR_F16) hipLaunchKernelGGL((bneck_frame64_kernel<true>), dim3(n), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((bneck_frame64_kernel<false>), dim3(n), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
