// Implicit-GEMM convolution for the SMALL-M, deep-K launches of layer4 (round 5): 12 544 output pixels (batch 256 at 7 x 7) x 512 couts are 98 tiles of
// 256 x 256 or 196 of 128 x 256 on 256 CUs - conv_pp256 runs them at 36 % of a CU's peak on 77 % of the CUs.  This kernel cuts the same output into
// 112-pixel x 256-cout tiles (224 of them: 87.5 % of the CUs, one round) and takes the weight operand out of LDS:
//   * 8 waves, wave w owns 32 couts x all 7 pixel tiles (56 accumulator VGPRs); its weight fragments come straight from L2 as whole MFMA A fragments
//     (fragment-blocked copy of the matrix, launch_pack_frag_weights - every fragment is fetched ONCE per workgroup), three K chunks ahead, in registers;
//   * only the pixel operand goes through LDS: a 64-channel K chunk of the tile's 112 pixels = 14 KB, LDS-DMA'd three chunks ahead into a ring of four
//     buffers (rows of 128 B, chunk index XOR (row >> 1) & 7 - every fragment read is a conflict-free ds_read_b128); the 3 x 3 taps, the stride and the
//     padding live in the DMA's source addresses (out-of-image taps: offset past num_records -> zeros), so LDS only ever sees "centre" reads;
//   * one barrier per K chunk (28 MFMAs per wave); vmcnt is hand-counted: DMA and weight requests are issued unconditionally, in a fixed order.
// Torchvision Bottleneck convolutions reached from reference src/embeddings.py:118-120 (resnet50) and src/vision_models/moco.py:6-26.
// Same operand roles (A = weights, B = pixels), K order (tap-major, then channels, 32 per MFMA) and rounding points as conv_igemm / conv_pp256:
// bit-identical outputs (tests/test_gpu_encoder.py::test_conv_wfrag_is_bit_identical).
#include "common.h"

namespace pvr {

struct WFP {
    const u16 *in, *w, *res;
    const float *bias;
    void *out;
    int M, H, W, Cin, Cout, Ho, Wo, KH, KW, stride, pad, act;
    int MT, NCT;        // pixel tiles of 112 rows, cout tiles of 256
    int nch, cpt;       // K chunks of 64 channels in all / per tap (Cin / 64)
    int KC;             // K / 8: 16-byte k-chunks per weight row
    int ct_major;       // tile order inside an XCD's contiguous run: cout tile outermost (big weight matrices) or pixel tile outermost
    unsigned in_bytes, w_bytes, out_bytes, res_bytes;
    float *pool_out;     // POOL: the average over a frame's 49 pixels, fp32, row f at pool_out + f * pool_stride (nothing else is written)
    long long pool_stride;
    int frames;
};

#define WF_LDS_PTR(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

// KO (EXPERIMENTS builds, PVR_WFRAG_KO): timing knock-outs, results wrong by construction - 1 no pixel DMA, 2 no weight requests after the prologue's,
// 4 no barriers, 8 no MFMAs, 16 no fragment reads
// POOL: the launch is the trunk's last convolution on 7 x 7 maps and only AdaptiveAvgPool2d(1) reads its output (torchvision resnet.avgpool, reference
// src/embeddings.py:118-120): a tile is TWO whole frames (98 of its 112 rows), a wave owns every pixel of both for its 32 couts, and the epilogue
// reduces them in registers - the (n,7,7,2048) fp32 activation (103 MB written, 103 MB read back by avgpool_kernel at batch 256) never exists.
// The summation order is defined on the pixel index q inside a frame, not on its position in the tile: lane sums over q = 16 j' + l (j' ascending),
// then the row_shr 1 / 2 / 4 / 8 tree over l - avgpool_kernel follows the same order, so the fused and the stand-alone pool agree bit for bit
// and a frame's embedding does not depend on whether it sits first or second in its tile.
template <bool F16, int RES, bool OUT32, int KO = 0, bool POOL = false>
__global__ __launch_bounds__(512, 1) void conv_wfrag_kernel(WFP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int BMV = POOL ? 98 : 112;               // rows of the tile that are output pixels
    constexpr int BM = 112, NT = 7, XB = 16384;       // a ring buffer (four of them): 14 row groups of 1 KB + 2 KB nothing reads (waves 6 / 7's second, empty DMA)
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, sw = (fr >> 1) & 7;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int mt, ct;
    if (p.ct_major) { ct = t / p.MT; mt = t - ct * p.MT; }
    else { mt = t / p.NCT; ct = t - mt * p.NCT; }
    const int m0 = mt * BMV, co0 = ct * 256;

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- staging: row group g = wave (+ 8) of the tile, lane -> (row 8 g + lane / 8, 16-byte slot lane % 8 holding logical chunk slot ^ (row >> 1) & 7)
    int a_off[2], a_m[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = wave + 8 * i, r = g * 8 + (lane >> 3);
        const int lch = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = g < BM / 8 && r < BMV && m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.Wo, tq = mm / p.Wo, ho = tq % p.Ho, n = tq / p.Ho;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + lch * 8) * 2;
        int hb = 0, wb = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && t3 < p.KH && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
            wb |= (int)(t3 < p.KW && (unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
        }
        a_m[i] = hb | (wb << 3);
    }
    // the next K chunk to stage: (tap row, tap column, 64-channel chunk of the tap); past the last chunk the last one is staged again (a buffer nobody
    // reads any more) - requests are never conditional
    int s_c = 0, s_ky = 0, s_kx = 0, s_ch = 0;
    int nv[2];                                           // its two source offsets, computed a chunk early (WF_STAGE_NEXT): a DMA issue needs no temporaries -
                                                         // hipcc gave those the registers of a weight set between its last use and its reload, and then
                                                         // waited vmcnt(0) for "the load that may still write them"
#define WF_STAGE_NEXT(advance_)                                                                                 \
    {                                                                                                          \
        if ((advance_) && s_c + 1 < p.nch) {                                                                    \
            ++s_c;                                                                                              \
            if (++s_ch == p.cpt) { s_ch = 0; if (++s_kx == p.KW) { s_kx = 0; ++s_ky; } }                         \
        }                                                                                                       \
        const int toff_ = ((s_ky * p.W + s_kx) * p.Cin + s_ch * 64) * 2;                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
            nv[i] = ((a_m[i] >> s_ky) & (a_m[i] >> (3 + s_kx)) & 1) ? a_off[i] + toff_ : OOB;                   \
    }
#define WF_STAGE1(buf_, i_)                                                                                     \
    {                                                                                                          \
        if constexpr (!(KO & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, WF_LDS_PTR((buf_) * XB + (wave + 8 * (i_)) * 1024), 16, nv[i_], 0, 0, 0);   \
        asm volatile("" ::: "memory");                                                                          \
    }
#define WF_STAGE(buf_) { WF_STAGE1(buf_, 0) WF_STAGE1(buf_, 1) }
    // weights: fragment (row tile rt, 32-deep k-step kk) = 1 KB at ((rt * KC + 4 kk) * 256) bytes, lane * 16 inside; K chunk c = k-steps 2 c, 2 c + 1
    const int wlane = lane * 16;
    const int rt0 = (co0 >> 4) + 2 * wave;
    int w_c = 0;                                         // the next K chunk to request (clamped like the staging)
    V8 w0[2][2], w1[2][2], w2[2][2], w3[2][2];
#define WF_LOAD_W1(dst_, i_, ks_)                                                                               \
    {                                                                                                          \
        if constexpr (!(KO & 2)) dst_[i_][ks_] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, ((rt0 + (i_)) * p.KC + 4 * (2 * w_c + (ks_))) * 256, 0));   \
        asm volatile("" ::: "memory");            /* (pins the request here: hipcc otherwise sinks it towards its first use, past a loop exit) */ \
    }
#define WF_LOAD_W(dst_) WF_LOAD_W_(dst_, !(KO & 2))
#define WF_LOAD_W_(dst_, on_)                                                                                   \
    {                                                                                                          \
        if constexpr (on_) { _Pragma("unroll") for (int i = 0; i < 2; ++i)                                      \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
                dst_[i][ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, ((rt0 + i) * p.KC + 4 * (2 * w_c + ks)) * 256, 0)); }   \
        if (w_c + 1 < p.nch) ++w_c;                                                                             \
    }

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Pixel fragments: pixel tile j of ring buffer B at xc + B * XB + 2048 j (k-step 0) and the same XOR 64 (k-step 1): two address registers, the rest
    // in the instructions' immediates.  The reads run as ONE pipeline across K chunks, RD steps (pixel tiles) ahead of the MFMAs that use them: LDS
    // latency with eight waves reading is ~250 cycles, two steps of MFMAs are 128 (a knock-out run with the reads two steps ahead and a pipeline drained
    // at every chunk spent 29 of the launch's 64 us in the read stream alone).  xs[j] = the fragments of pixel tile j, of this chunk or the next.
    const int xc = fr * 128 + ((fq ^ sw) << 4), xc64 = xc ^ 64;
    V8 xs[NT][2];
    constexpr int RD = 3;
#define WF_XREAD(j_, B_)                                                                                        \
    {                                                                                                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[j_][0]) : "v"(xc), "n"((B_) * XB + (j_) * 2048));   \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[j_][1]) : "v"(xc64), "n"((B_) * XB + (j_) * 2048)); \
    }
    // step q of a chunk in buffer B_ (the next chunk's in BN_): request step q + RD, wait for step q's two fragments (2 RD newer reads may be in flight)
#define WF_STEP(q_, B_, BN_, W_)                                                                                \
    {                                                                                                          \
        if ((q_) + RD < NT) WF_XREAD((q_) + RD, B_) else WF_XREAD((q_) + RD - NT, BN_);                         \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xs[q_][0]), "+v"(xs[q_][1]));                                \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if constexpr (KO & 8) { asm volatile("" :: "v"(W_[0][0]), "v"(W_[1][0]), "v"(W_[0][1]), "v"(W_[1][1])); } else {   \
        acc[0][q_] = mfma16<F16>(W_[0][0], xs[q_][0], acc[0][q_]);                                              \
        acc[1][q_] = mfma16<F16>(W_[1][0], xs[q_][0], acc[1][q_]);                                              \
        acc[0][q_] = mfma16<F16>(W_[0][1], xs[q_][1], acc[0][q_]);                                              \
        acc[1][q_] = mfma16<F16>(W_[1][1], xs[q_][1], acc[1][q_]);                                              \
        }                                                                                                       \
    }
    // One K chunk c, in buffer B_ = c % 4 with weight set W_ = c % 4.  Its barrier publishes chunk c + 1 (every wave waits for its own part of that DMA
    // first) - chunk c itself was published an iteration ago, so the read pipeline runs on into the next chunk without waiting - and says every wave is
    // done with chunk c - 1, whose buffer and weight set the requests for chunk c + 3 then take.
    // The six requests of a chunk (4 weight fragments, then 2 DMA instructions) are issued ONE PER STEP, between the MFMAs: a CU's vector-memory path
    // takes 64 B/clk, eight waves x 6 KB issued together stall every wave at the top of the chunk for ~750 of its 896 MFMA cycles (s_memtime stamps of
    // the frame kernel's front phase: 1480 cycles to issue 15 requests per wave).  In front of the barrier a wave has issued, after its part of
    // chunk c + 1's DMA: chunk c + 2's weights (4) and DMA (2): vmcnt(6).
#define WF_ITER(B_, BN_, BF_, W_, WF_)                                                                          \
    {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                        \
        if constexpr (!(KO & 4)) __builtin_amdgcn_s_barrier();                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        WF_LOAD_W1(WF_, 0, 0) WF_STEP(0, B_, BN_, W_)                                                           \
        WF_LOAD_W1(WF_, 0, 1) WF_STEP(1, B_, BN_, W_)                                                           \
        WF_LOAD_W1(WF_, 1, 0) WF_STEP(2, B_, BN_, W_)                                                           \
        WF_LOAD_W1(WF_, 1, 1) WF_STEP(3, B_, BN_, W_)                                                           \
        if (w_c + 1 < p.nch) ++w_c;                                                                             \
        WF_STAGE1(BF_, 0) WF_STEP(4, B_, BN_, W_)                                                               \
        WF_STAGE1(BF_, 1) WF_STEP(5, B_, BN_, W_)                                                               \
        WF_STAGE_NEXT(1);                                                                                       \
        WF_STEP(6, B_, BN_, W_)                                                                                 \
    }

    // RES: the tile's identity values are requested FIRST - the oldest entries of the in-order vmcnt queue, so the K loop's counted waits never see
    // them - and have landed long before the epilogue wants them (requested at its top their HBM latency was exposed once per tile, and a tile of a
    // short-K launch - conv3: 8 chunks - is mostly epilogue)
    const int c = co0 + 32 * wave + 8 * fq;
    u32x4 rr[NT];
    if constexpr (RES == 1) {
        const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.res), 0, p.res_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int m = m0 + 16 * j + fr;
            rr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (m < p.M && 16 * j + fr < BMV) ? (m * p.Cout + c) * 2 : OOB, 0, 0));
        }
        asm volatile("" ::: "memory");
    }
    WF_STAGE_NEXT(0); WF_LOAD_W_(w0, true); WF_STAGE(0);
    WF_STAGE_NEXT(1); WF_LOAD_W_(w1, true); WF_STAGE(1);
    WF_STAGE_NEXT(1); WF_LOAD_W_(w2, true); WF_STAGE(2);
    if constexpr ((KO & 2) != 0) WF_LOAD_W_(w3, true);
    WF_STAGE_NEXT(1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // my part of chunk 0 (behind it: 4 + 2 + 4 + 2 requests)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WF_XREAD(0, 0); WF_XREAD(1, 0); WF_XREAD(2, 0);
#pragma unroll 1
    for (int c = 0;; c += 4) {
        WF_ITER(0, 1, 3, w0, w3);
        if (c + 1 >= p.nch) break;
        WF_ITER(1, 2, 0, w1, w0);
        if (c + 2 >= p.nch) break;
        WF_ITER(2, 3, 1, w2, w1);
        if (c + 3 >= p.nch) break;
        WF_ITER(3, 0, 2, w3, w2);
        if (c + 4 >= p.nch) break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the read-ahead into the chunk past the last one: nothing uses it, but its registers are reused below)
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: + bias (+ residual) (ReLU), a lane's two row tiles = 8 consecutive couts of one pixel (the packed weights' row permutation)
    const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.bias + c), bh = *reinterpret_cast<const f32x4 *>(p.bias + c + 4);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    float ps[2][8];                                       // POOL: this lane's sums over the pixel tiles, per frame of the tile
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) ps[f][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int m = m0 + 16 * j + fr;
        const f32x4 lo = acc[0][j], hi = acc[1][j];
        float v[8] = {lo[0] + bl[0], lo[1] + bl[1], lo[2] + bl[2], lo[3] + bl[3], hi[0] + bh[0], hi[1] + bh[1], hi[2] + bh[2], hi[3] + bh[3]};
        if constexpr (RES == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] += from_h<F16>((u16)(rr[j][e] & 0xffffu));
                v[2 * e + 1] += from_h<F16>((u16)(rr[j][e] >> 16));
            }
        }
        if (p.act) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (POOL) {
            const int pp = 16 * j + fr;                       // row of the tile: frame 0 = rows 0 .. 48, frame 1 = rows 49 .. 97
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ps[0][e] += pp < 49 ? v[e] : 0.f;
                ps[1][e] += (pp >= 49 && pp < 98) ? v[e] : 0.f;
            }
        } else if constexpr (OUT32) {
            const int o = m < p.M ? (m * p.Cout + c) * 4 : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rs_o, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rs_o, o, 16, 0);
        } else {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(v[2 * e]) | ((unsigned)to_h<F16>(v[2 * e + 1]) << 16);
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, m < p.M ? (m * p.Cout + c) * 2 : OOB, 0, 0);
        }
    }
    if constexpr (POOL) {
        // frame 1's pixel q = row - 49 sits one lane to the right of frame 0's pixel q (49 = 3 * 16 + 1): rotate its lane sums one lane to the left
        // (row_ror:15) and both frames go through the same tree
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float tot[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = ps[f][e];
                if (f == 1) t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x12f, 0xf, 0xf, false));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x111, 0xf, 0xf, true));   // row_shr:1, missing lanes add 0
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x112, 0xf, 0xf, true));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x114, 0xf, 0xf, true));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x118, 0xf, 0xf, true));
                tot[e] = t / 49.f;
            }
            const int frame = 2 * mt + f;
            if (fr == 15 && frame < p.frames) {
                float *o = p.pool_out + (long long)frame * p.pool_stride + c;
                *reinterpret_cast<f32x4 *>(o) = f32x4{tot[0], tot[1], tot[2], tot[3]};
                *reinterpret_cast<f32x4 *>(o + 4) = f32x4{tot[4], tot[5], tot[6], tot[7]};
            }
        }
    }
#undef WF_ITER
#undef WF_STEP
#undef WF_XREAD
#undef WF_LOAD_W1
#undef WF_LOAD_W_
#undef WF_LOAD_W
#undef WF_STAGE
#undef WF_STAGE1
#undef WF_STAGE_NEXT
}

static long long g_conv_wfrag_launches = 0;
long long conv_wfrag_launches() { return g_conv_wfrag_launches; }

// shapes the kernel accepts (whether it is the faster choice is the plan's decision: conv_wfrag_preferred)
bool conv_wfrag_supported(int64_t M, int64_t in_bytes, int cin, int cout, int kh, int kw, int pad, int act, int out_f32) {
    const int64_t K = (int64_t)kh * kw * cin;
    return cin % 64 == 0 && cout % 256 == 0 && kh <= 3 && kh == kw && (pad == kh / 2 || pad == 0) && (act == 0 || act == 1) && !(out_f32 & 2) && M >= 1 &&
           M * cout * ((out_f32 & 1) ? 4 : 2) < 0x7ffffff0ll && in_bytes < 0x7ffffff0ll && (int64_t)cout * K * 2 < 0x7ffffff0ll;
}

// the plan's rule: deep-K launches whose 256 x 256 (224 x 256) tiling leaves CUs idle - fewer than 160 such tiles - and whose 112 x 256 tiling fills
// at least 3/4 of them (layer4's conv1 / conv2 at batch 256: 98 -> 224 tiles)
bool conv_wfrag_preferred(int64_t M, int cin, int cout, int kh, int kw) {
    static const int on = [] { const char *e = getenv("PVR_CONV_WFRAG"); return e ? atoi(e) : 1; }();
    const int64_t K = (int64_t)kh * kw * cin, nct = cout / 256, t256 = ((M + 255) / 256) * nct, t112 = ((M + 111) / 112) * nct;
    if (on == 2) return K >= 512;                          // (A/B: every launch the kernel accepts)
    return on && K >= 1024 && t256 < 160 && t112 >= 192;
}

// wp: the fragment-blocked copy (launch_pack_frag_weights) of the (cout, kh * kw * cin) matrix
pvr_status launch_conv_wfrag(const void *in, const void *wp, const float *bias, const void *res, void *out, int n, int h, int w, int cin, int cout,
                             int kh, int kw, int stride, int pad, int act, int out_f32, int dtype, hipStream_t stream, float *pool_out, int64_t pool_stride) {
    // pool_out != nullptr: `out` is not written; pool_out[f * pool_stride + c] = mean over frame f's 7 x 7 outputs (fp32)
    PVR_REQUIRE(in && wp && bias && (out || pool_out), "conv_wfrag: null argument");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "conv_wfrag: 16-bit storage types only");
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    const int64_t M = (int64_t)n * ho * wo;
    PVR_REQUIRE(conv_wfrag_supported(M, (int64_t)n * h * w * cin * 2, cin, cout, kh, kw, pad, act, out_f32), "conv_wfrag: unsupported shape");
    WFP p;
    p.in = (const u16 *)in; p.w = (const u16 *)wp; p.res = (const u16 *)res; p.bias = bias; p.out = out;
    p.M = (int)M; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.Ho = ho; p.Wo = wo; p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad; p.act = act;
    p.MT = pool_out ? (n + 1) / 2 : (int)((M + 111) / 112); p.NCT = cout / 256;
    p.pool_out = pool_out; p.pool_stride = pool_stride; p.frames = n;
    PVR_REQUIRE(!pool_out || (ho == 7 && wo == 7 && res && act == 1 && pool_stride >= cout && ((size_t)pool_out & 15) == 0 && pool_stride % 4 == 0),
                "conv_wfrag: the pooled form is the 7 x 7 conv3 + identity + ReLU of the trunk's last block");
    p.cpt = cin / 64; p.nch = kh * kw * p.cpt; p.KC = kh * kw * cin / 8;
    const int64_t wbytes = (int64_t)cout * kh * kw * cin * 2;
    p.ct_major = wbytes > (5 << 19) ? 1 : 0;              // > 2.5 MB of weights: an XCD's L2 (4 MB) keeps ONE cout tile's slice of them
    p.in_bytes = (unsigned)((int64_t)n * h * w * cin * 2); p.w_bytes = (unsigned)wbytes;
    p.out_bytes = (unsigned)(M * cout * ((out_f32 & 1) ? 4 : 2)); p.res_bytes = res ? (unsigned)(M * cout * 2) : 0;
    constexpr int lds = 4 * 16384;
    const dim3 grid((unsigned)(p.MT * p.NCT)), block(512);
    ++g_conv_wfrag_launches;
    const bool f16 = dtype == PVR_F16, o32 = out_f32 & 1;
#define WF_GO(F_, R_, O_) hipLaunchKernelGGL((conv_wfrag_kernel<F_, R_, O_>), grid, block, lds, stream, p)
#ifdef PVR_EXPERIMENTS
    if (const char *e = getenv("PVR_WFRAG_KO"); e && atoi(e) && dtype == PVR_F16 && !res && !(out_f32 & 1)) {
#define WF_KO(k_) case k_: hipLaunchKernelGGL((conv_wfrag_kernel<true, 0, false, k_>), grid, block, lds, stream, p); break;
        switch (atoi(e)) { WF_KO(1) WF_KO(2) WF_KO(3) WF_KO(4) WF_KO(7) WF_KO(8) WF_KO(9) WF_KO(10) WF_KO(11) WF_KO(15) default: break; }
#undef WF_KO
        PVR_LAUNCH_CHECK();
        return PVR_OK;
    }
#endif
    if (pool_out) {
        if (f16) hipLaunchKernelGGL((conv_wfrag_kernel<true, 1, false, 0, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((conv_wfrag_kernel<false, 1, false, 0, true>), grid, block, lds, stream, p);
        PVR_LAUNCH_CHECK();
        return PVR_OK;
    }
    if (f16) {
        if (res) { if (o32) WF_GO(true, 1, true); else WF_GO(true, 1, false); }
        else { if (o32) WF_GO(true, 0, true); else WF_GO(true, 0, false); }
    } else {
        if (res) { if (o32) WF_GO(false, 1, true); else WF_GO(false, 1, false); }
        else { if (o32) WF_GO(false, 0, true); else WF_GO(false, 0, false); }
    }
#undef WF_GO
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
