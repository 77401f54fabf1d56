// Deep-K implicit-GEMM convolution / linear layer: 256 x 256 output tile, 8 waves in two groups that alternate
// between "feed" (LDS fragment reads + LDS-DMA prefetch) and "math" (16 MFMAs) phases, so the matrix pipe of every SIMD
// always has one wave issuing MFMAs while its partner wave loads.  Serves the 3x3 and K >= 512 1x1 convolutions of
// layer3 / layer4 (torchvision Bottleneck reached from reference src/embeddings.py:118-120) and the transformer
// linear layers of the CLIP / MAE encoders (reference src/embeddings.py:298-314, src/vision_models/mae.py:85-93).
// Same GEMM view, operand layouts, K order and epilogue arithmetic as conv_igemm.hip, so results agree with it to the
// last bit of the fp32 accumulation order per output (K slices ascending, 2 x 32-deep MFMA steps per slice).
//
//   out[m][co] = sum_k X[m][k] * W[co][k],  m = (n,ho,wo),  k = (kh,kw,c),  BK = 64 = one filter tap x 64 channels
//
// Structure (cdna_hip_programming.md "256^2 8-phase template", re-derived for NHWC implicit GEMM):
//   * waves 2 (pixels) x 4 (couts): a wave owns 128 pixels x 64 couts = 8 x 4 MFMA tiles = 128 accumulator VGPRs.
//   * LDS 128 KB = 2 buffers x {X-lo, X-hi, W-lo, W-hi} half tiles of [128 rows][64] 16-bit (16 KB each), rows of 128 B
//     with the 16-byte chunk index XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128 fragment reads).
//   * staging is LDS-DMA only (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write.  A wave instruction writes
//     1024 contiguous LDS bytes = 8 rows; the swizzle and the im2col gather live in the per-lane SOURCE offset.  Filter
//     taps outside the image, pixel-tile tails and cout tails use an offset past num_records: the DMA writes zeros.
//     The weight rows are permuted on the way in (LDS row 16t+4a+c of a 32-row block <- cout 8a+4t+c) so that a lane's
//     accumulators of an MFMA tile pair are 8 consecutive output channels: 16-byte epilogue stores / residual loads
//     straight from registers.
//   * one K tile = 4 phases of [feed | barrier | 16 MFMA | barrier]:
//         phase 0: read X0 (pixel tiles 0-3), W0 (cout tiles 0-1); DMA W-lo(t+1);   math W0 x X0
//         phase 1: read X1 (pixel tiles 4-7);                      DMA W-hi(t+1);   math W0 x X1
//         phase 2: read W1 (cout tiles 2-3);                       DMA X-lo(t+2);   math W1 x X1
//         phase 3:                                                 DMA X-hi(t+2);   math W1 x X0
//     The second wave group starts one barrier late, so its feed half-phase coincides with the first group's math
//     half-phase and vice versa.
//   * hazards (counted by hand; the compiler does not order LDS-DMA against ds_read):
//       WAR  every wave waits lgkmcnt(0) BEFORE the barrier that ends its feed half-phase, so a half tile last read in
//            phase p may be re-staged from phase p+1 on by either group.  X halves are last read in phase 1 -> re-staged in
//            phases 2/3 (same buffer, tile t+2); W halves are last read in phase 2 of tile t-1 -> re-staged in phases 0/1 of
//            tile t (other buffer, tile t+1).
//       RAW  one counted wait per K tile, in phase 3 after that phase's DMA issue: vmcnt(4) leaves X-lo/X-hi(t+2) in flight
//            and retires everything of tile t+1; every wave then passes at least one barrier before any wave reads tile t+1.
#include "common.h"
#include <type_traits>

namespace pvr {

struct ConvP;   // conv_igemm.hip
struct PPP {
    const u16 *in, *wgt, *res;
    const float *bias;
    void *out;
    int H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad, M, K;
    unsigned in_bytes, w_bytes, out_bytes, res_bytes;
    int act, out_f32;
    int n_tiles;
    int total_tiles;               // pixel tiles x cout tiles (the persistent form walks them with a stride of gridDim.x)
    int pointwise;                 // KH = KW = 1, stride 1, pad 0: pp_tile_setup needs no (n, ho, wo) decomposition
    int bias_lds;                  // the bias vector (<= 4096 floats, zero past Cout) is copied to LDS once per block: see the epilogue
    // DUAL (round 5): a second pixel operand appended along K - out = act(W[:, :K1] . in + W[:, K1:] . in2 + bias): a bottleneck's conv3 and its
    // 1 x 1 strided downsample (the identity branch) in ONE accumulation, the downsample's output never exists in HBM.  in2: (n, H2, W2, Cin2),
    // read at (ho * stride2, wo * stride2), no padding; nk2 = Cin2 / 64 K tiles (0: off); K = K1 + Cin2 is the weight rows' length.
    const u16 *in2;
    unsigned in2_bytes;
    int H2, W2, Cin2, stride2, nk2;
};

#define PP_LDS_PTR(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

// LDS reads of the epilogue as inline assembly: while LDS-DMA is in flight (the persistent form requests the next tile's first K tile in
// front of the epilogue) hipcc puts an s_waitcnt vmcnt(0) in front of every C++ LDS access.  The consumers are tied to the wait.
__device__ __forceinline__ void pp_lds_write16(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 pp_lds_read16(unsigned addr) { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void pp_lds_wait(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory"); }

// Diagnostic build only (scripts/pp256_stamps.hip defines PP_STAMP): s_memtime stamps of one steady-state K tile, one wave of
// each group of block 0, written to a buffer nothing else reads.  Compiles to nothing in the library.
#ifdef PP_STAMP
// PP_STAMP = K tile to sample, PP_PHASE = which phase's six stamps are compiled in (one build per phase keeps the SGPR cost
// at 12 + 6).  The stamps are inline asm (no compiler-inserted lgkmcnt wait: s_memtime returns through lgkmcnt like the LDS
// reads) and are only copied out after the tile's last barrier.
#define PP_STAMP_DECL(kt_) const bool stamp_on = (kt_) == PP_STAMP; int sti = 0; (void)sti; unsigned long long tt_[6] = {0, 0, 0, 0, 0, 0}
__device__ unsigned long long pp_stamps[2][8];
#define PP_T(k_) { const int kk_ = (k_); if (kk_ / 6 == PP_PHASE) asm volatile("s_memtime %0" : "=s"(tt_[kk_ % 6]) :: "memory"); }
#define PP_STAMP_LATCH() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (stamp_on) { _Pragma("unroll") for (int k = 0; k < 6; ++k) st_[k] = tt_[k]; } }
#else
#define PP_STAMP_DECL(kt_)
#define PP_T(k_)
#define PP_STAMP_LATCH()
#endif

// Second diagnostic (scripts/pp256_tile_stamps.hip defines PP_TSTAMP = the tile iteration to sample): s_memtime at the boundaries of one
// whole tile iteration of the persistent form (drain wait, barriers, K loop, next tile's setup + prologue issue, epilogue), block 8, one wave per group.
#ifdef PP_TSTAMP
__device__ unsigned long long pp_tstamps[2][8];
#define PP_TS(k_) { if (tile_it_ == PP_TSTAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tts_[k_]) :: "memory"); }
#else
#define PP_TS(k_)
#endif

// per-lane staging offsets of one output tile (pixel rows m0 .., couts co0 ..): see "staging" in the kernel
template <int BM, int XI>
__device__ __forceinline__ void pp_tile_setup(const PPP &p, int m0, int co0, int wave, int lane, int (&a_off)[2][XI], int (&a_mask)[2][XI],
                                              int (&b_off)[2][2], bool natural, int (*a_off2)[XI] = nullptr) {
    constexpr int XH = BM / 2, OOB = 0x7ffffff0;       // (BM = 224: 112 pixel rows per half tile in a 128-row LDS half; rows 112..127 stay zero)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int lch = (lane & 7) ^ ((r >> 1) & 7);                // logical chunk held by this physical slot
            if (i < XI) {
                const int m = m0 + h * XH + r;
                const bool ok = m < p.M && r < XH;
                const int mm = ok ? m : 0;
                if (p.pointwise) {
                    // 1 x 1, stride 1, no padding (every transformer GEMM, the conv1 launches): input pixel = output pixel, one tap, no border.
                    // The general path below costs four integer divisions per entry - ~3 k cycles of a 46 k-cycle tile iteration of the
                    // persistent form, once per tile (s_memtime stamps, scripts/pp256_tile_stamps.hip)
                    a_off[h][i < XI ? i : 0] = (mm * p.Cin + lch * 8) * 2;
                    a_mask[h][i < XI ? i : 0] = ok ? 1 : 0;
                }
                if (!p.pointwise) {
                const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                a_off[h][i < XI ? i : 0] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + lch * 8) * 2;
                int hb = 0, wb = 0;
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) {
                    hb |= (int)(ok && t3 < p.KH && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
                    wb |= (int)(t3 < p.KW && (unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
                }
                int mask = 0;
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * p.KW)) : 0;
                a_mask[h][i < XI ? i : 0] = mask;
                }
                if (a_off2) {                                           // DUAL: the second operand's pixel (1 x 1, stride2, no padding); bit 9 of the mask = row exists
                    const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
                    a_off2[h][i < XI ? i : 0] = (((n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + lch * 8) * 2;
                    a_mask[h][i < XI ? i : 0] |= ok ? 512 : 0;
                }
            }
            const int R = h * 128 + r;                                  // A-operand row inside the 256-cout tile
            // 16-bit outputs: rows permuted so that a lane's tile PAIR is 8 consecutive couts (one 16-byte store).  fp32 outputs with an fp32
            // (or no) residual - the transformer GEMMs: natural order, a lane's tile is 4 consecutive couts = 16 bytes of fp32 and the four
            // lanes of a pixel cover 64 contiguous bytes (the permuted order left 16-byte pieces 32 bytes apart: chain_wave.hip, memory layouts)
            const int co = natural ? co0 + R : co0 + (R & ~31) + 8 * ((R >> 2) & 3) + 4 * ((R >> 4) & 1) + (R & 3);
            b_off[h][i] = co < p.CoutPad ? (co * p.K + lch * 8) * 2 : OOB;
        }
}

// BM = 256: the structure above.  BM = 128 (launches whose 256-pixel tiles would not fill the chip, e.g. layer4 at batch 256):
// a wave owns 64 pixels x 64 couts, X half tiles are 64 rows (one DMA instruction each), a K tile is two phases:
//         phase 0: read X (pixel tiles 0-3), W0;  DMA W-lo(t+1), W-hi(t+1);           math W0 x X
//         phase 1: read W1;                       DMA X-lo(t+2), X-hi(t+2), vmcnt(2); math W1 x X
// PERSIST (launches of many more tiles than CUs: the transformer GEMMs): gridDim.x = 256 blocks walk the tiles v = blockIdx.x,
// blockIdx.x + 256, ... (v mod 8 = the block's XCD, so the XCD-aware tile order holds); the NEXT tile's prologue DMA (K tile 0 and the X
// half tiles of K tile 1) is issued between the main loop and the epilogue of the current tile, so its HBM / L2 latency runs under the
// epilogue's stores instead of in front of the first MFMA.  LDS is free at that point (every wave has passed the barrier that follows
// its last fragment read) and the epilogue does not touch LDS.
template <int BM, bool F16, int RES, bool PERSIST = false, bool DUAL = false>
__global__ __launch_bounds__(512, 1) void conv_pp256_kernel(PPP p) {
    typedef typename HT<F16>::V8 V8;
    // BM = 224 (round 2): the BM = 256 structure with 7 of the 8 pixel tiles per wave.  50 176 pixels (batch 256 at 14 x 14) are 196 tiles of
    // 256 = 77 % of the CUs for a full-length tile each, or 224 tiles of 224 = 88 % of the CUs for 7/8 of the time: the launch is 12 % shorter
    constexpr int TMW = BM / 32;                   // 16-pixel MFMA tiles per wave
    constexpr int XH = BM / 2;                     // pixel rows of an X half tile
    constexpr int XHB = (BM == 224 ? 128 : XH) * 128;   // its bytes in LDS
    constexpr int XI = (XH + 63) / 64;             // DMA instructions per X half tile (8 waves x 8 rows each)
    constexpr int T1 = TMW - 4;                    // pixel tiles of the wave's second fragment group (x1)
    constexpr int HALF = 16384;                    // W half tile: 128 couts
    constexpr int WOFF = 2 * XHB, BUF = WOFF + 2 * HALF;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int vb = blockIdx.x;                          // (virtual) block id of the tile being computed
    int m0, co0;

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wgt), 0, p.w_bytes, 0x00020000);
    const auto rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DUAL ? p.in2 : p.in), 0, DUAL ? p.in2_bytes : p.in_bytes, 0x00020000);
    // the bias vector -> LDS, behind the two staging buffers (read by the epilogues; the K loops' barriers order it before the first one)
    const unsigned bias_l = (unsigned)(size_t)PP_LDS_PTR(2 * BUF);
    if (p.bias_lds) {
        for (int c4 = tid; c4 * 4 < p.CoutPad; c4 += 512) {
            f32x4 b4;
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = c4 * 4 + e < p.Cout ? p.bias[c4 * 4 + e] : 0.f;
            *reinterpret_cast<f32x4 *>(smem + 2 * BUF + c4 * 16) = b4;
        }
    }

    // ---- staging: half tile h, DMA instruction i: this lane feeds LDS row r = (8i + wave)*8 + lane/8, physical chunk lane%8
    int a_off[2][XI], a_mask[2][XI], b_off[2][2];
    int a_off2[DUAL ? 2 : 1][XI];
    const bool natural = p.out_f32 && RES != 1;   // weight rows (= the accumulators' couts) in natural order: see pp_tile_setup
#define PP_TILE_SETUP(v_)                                                                                        \
    {                                                                                                            \
        const int swz_ = xcd_remap((v_), PERSIST ? p.total_tiles : (int)gridDim.x);                               \
        m0 = (swz_ / p.n_tiles) * BM; co0 = (swz_ % p.n_tiles) * 256;                                             \
        pp_tile_setup<BM, XI>(p, m0, co0, wave, lane, a_off, a_mask, b_off, natural, DUAL ? a_off2 : nullptr);     \
    }
    PP_TILE_SETUP(vb);
    const int cpt = p.Cin >> 6;                   // K tiles per filter tap
    const int nk = p.KH * p.KW * cpt + (DUAL ? p.nk2 : 0);
    int xs_tap = 0, xs_kh = 0, xs_kw = 0, xs_cs = 0;   // filter position of the next X tile to stage (wave-uniform)

#ifdef PP_KNOCK
#define PP_KNOCK_ PP_KNOCK
#else
#define PP_KNOCK_ 0            // timing experiments (scripts/pp256_stamps.hip -DPP_KNOCK=bits): 1 no RAW wait in phase 3, 2 no LDS-DMA in the K loop, 4 no fragment reads
#endif
#define PP_STAGE_X(h_, buf_)                                                                                     \
    if constexpr (!(PP_KNOCK_ & 2)) {                                                                            \
        if constexpr (!DUAL) {                                                                                   \
            const int tap_off = ((xs_kh * p.W + xs_kw) * p.Cin + xs_cs * 64) * 2;                                 \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                     \
                const int vo = ((a_mask[h_][i] >> xs_tap) & 1) ? a_off[h_][i] + tap_off : OOB;                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, PP_LDS_PTR((buf_) * BUF + (h_) * XHB + (i * 8 + wave) * 1024), 16, vo, 0, 0, 0); \
            }                                                                                                    \
        } else {                                                                                                 \
            /* past the first operand's last tap (xs_kh == KH) the K tiles come from the second one.  Pure ALU selects (masks), no ?: - hipcc turned */ \
            /* the nested conditionals into EXEC-masked in-place updates of the offset registers and put s_waitcnt vmcnt(0) in front of each        */ \
            const int s2m_ = -(int)(xs_kh >= p.KH);                                                               \
            const int tap_off = s2m_ ? xs_cs * 128 : ((xs_kh * p.W + xs_kw) * p.Cin + xs_cs * 64) * 2;            \
            const int bit_ = s2m_ ? 9 : xs_tap;                                                                   \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                     \
                const int base_ = a_off[h_][i] ^ ((a_off[h_][i] ^ a_off2[DUAL ? h_ : 0][i]) & s2m_);              \
                const int ok_ = -((a_mask[h_][i] >> bit_) & 1);                                                   \
                const int vo = ((base_ + tap_off) & ok_) | (OOB & ~ok_);                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s2m_ ? rs_in2 : rs_in, PP_LDS_PTR((buf_) * BUF + (h_) * XHB + (i * 8 + wave) * 1024), 16, vo, 0, 0, 0); \
            }                                                                                                    \
        }                                                                                                        \
    }
#define PP_ADVANCE_X()                                                                                           \
    { if (++xs_cs == ((DUAL && xs_kh >= p.KH) ? p.nk2 : cpt)) { xs_cs = 0; ++xs_tap; if (++xs_kw == p.KW) { xs_kw = 0; ++xs_kh; } } }
#define PP_STAGE_W(h_, kt_, buf_)                                                                                \
    if constexpr (!(PP_KNOCK_ & 2)) {                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, PP_LDS_PTR((buf_) * BUF + WOFF + (h_) * HALF + (i * 8 + wave) * 1024), 16, \
                                                     b_off[h_][i], (kt_) * 128, 0, 0);                           \
    }

    // ---- fragment read bases (per buffer and 32-deep k-step); tile offsets are instruction immediates --------------
    const int fr = lane & 15, fq = lane >> 4;
    int xrd[2][2], wrd[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ((ks * 4 + fq) ^ ((fr >> 1) & 7)) << 4;
            xrd[b][ks] = b * BUF + wr * XHB + fr * 128 + sw;
            wrd[b][ks] = b * BUF + WOFF + (wc >> 1) * HALF + (wc & 1) * 8192 + fr * 128 + sw;
        }

    f32x4 acc[4][TMW];

#ifdef PP_STAMP
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long clk0_, rt0_, clk1_, rt1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_) :: "memory");
#endif
    // ---- prologue: K tile 0 entirely, X of K tile 1 ------------------------------------------------------------------
#define PP_PROLOGUE_ISSUE()                                                                                      \
    {                                                                                                            \
        xs_tap = 0; xs_kh = 0; xs_kw = 0; xs_cs = 0;                                                             \
        PP_STAGE_X(0, 0); PP_STAGE_X(1, 0); PP_ADVANCE_X();                                                      \
        PP_STAGE_W(0, 0, 0); PP_STAGE_W(1, 0, 0);                                                                \
        if (nk > 1) { PP_STAGE_X(0, 1); PP_STAGE_X(1, 1); PP_ADVANCE_X(); }                                      \
    }
    PP_PROLOGUE_ISSUE();
    bool first_tile = true;
#ifdef PP_TSTAMP
    int tile_it_ = 0;
    unsigned long long tts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {                                      // tiles of this block (one pass unless PERSIST)
    PP_TS(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nk > 1 && (!PERSIST || first_tile)) {     // (later tiles: the previous tile's epilogue loads / stores were issued after the DMA:
        if constexpr (XI == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     //  everything drains; a counted wait that leaves the
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                       //  stores in flight measured no faster)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_TS(1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();   // second wave group runs one half-phase behind
    __builtin_amdgcn_sched_barrier(0);
    PP_TS(2);

#define PP_FEED_DONE()                                                                                           \
    PP_T(sti++);                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    PP_T(sti++);                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    PP_T(sti++);                                                                                                 \
    PP_T(sti++);                                                                                                 \
    __builtin_amdgcn_s_setprio(1);
#define PP_MATH_DONE()                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    PP_T(sti++);                                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    PP_T(sti++);                                                                                                 \
    PP_T(sti++);
#define PP_READ_X(dst_, j0_, B_)                                                                                 \
    if constexpr (!(PP_KNOCK_ & 4))                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
        _Pragma("unroll") for (int j = 0; j < ((j0_) == 0 ? 4 : T1); ++j)                                        \
            dst_[j][ks] = *reinterpret_cast<const V8 *>(smem + xrd[B_][ks] + ((j0_) + j) * 2048);
#define PP_READ_W(i0_, B_)                                                                                       \
    if constexpr (!(PP_KNOCK_ & 4))                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            wf[i][ks] = *reinterpret_cast<const V8 *>(smem + wrd[B_][ks] + ((i0_) + i) * 2048);
#define PP_MATH(i0_, x_, j0_)                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            _Pragma("unroll") for (int j = 0; j < ((j0_) == 0 ? 4 : T1); ++j)                                    \
                acc[(i0_) + i][(j0_) + j] = mfma16<F16>(wf[i][ks], x_[j][ks], acc[(i0_) + i][(j0_) + j]);

#define PP_KNOCK_WAIT() { if constexpr (!(PP_KNOCK_ & 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    // one K tile; B_ is a literal so that every LDS address is base register + immediate
#define PP_TILE(kt_, B_)                                                                                         \
    {                                                                                                            \
        const bool next1 = (kt_) + 1 < nk, next2 = (kt_) + 2 < nk;                                               \
        PP_STAMP_DECL(kt_);                                                                                      \
        V8 x0[4][2], x1[4][2], wf[2][2];                                                                         \
        PP_T(sti++);                                                                                             \
        PP_READ_X(x0, 0, B_); PP_READ_W(0, B_);                                                                  \
        if (next1) PP_STAGE_W(0, (kt_) + 1, 1 - (B_));                                                           \
        PP_FEED_DONE(); PP_MATH(0, x0, 0); PP_MATH_DONE();                                                       \
        PP_READ_X(x1, 4, B_);                                                                                    \
        if (next1) PP_STAGE_W(1, (kt_) + 1, 1 - (B_));                                                           \
        PP_FEED_DONE(); PP_MATH(0, x1, 4); PP_MATH_DONE();                                                       \
        PP_READ_W(2, B_);                                                                                        \
        if (next2) PP_STAGE_X(0, B_);                                                                            \
        PP_FEED_DONE(); PP_MATH(2, x1, 4); PP_MATH_DONE();                                                       \
        if (next2) {                                                                                             \
            PP_STAGE_X(1, B_); PP_ADVANCE_X();                                                                   \
            PP_KNOCK_WAIT();                                                                                     \
        } else {                                                                                                 \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        }                                                                                                        \
        PP_FEED_DONE(); PP_MATH(2, x0, 0); PP_MATH_DONE();                                                       \
        PP_STAMP_LATCH();                                                                                        \
    }
    // BM = 128: two phases per K tile
#define PP_TILE128(kt_, B_)                                                                                      \
    {                                                                                                            \
        const bool next1 = (kt_) + 1 < nk, next2 = (kt_) + 2 < nk;                                               \
        PP_STAMP_DECL(kt_);                                                                                      \
        V8 x0[4][2], wf[2][2];                                                                                   \
        PP_READ_X(x0, 0, B_); PP_READ_W(0, B_);                                                                  \
        if (next1) { PP_STAGE_W(0, (kt_) + 1, 1 - (B_)); PP_STAGE_W(1, (kt_) + 1, 1 - (B_)); }                   \
        PP_FEED_DONE(); PP_MATH(0, x0, 0); PP_MATH_DONE();                                                       \
        PP_READ_W(2, B_);                                                                                        \
        if (next2) {                                                                                             \
            PP_STAGE_X(0, B_); PP_STAGE_X(1, B_); PP_ADVANCE_X();                                                \
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                     \
        } else {                                                                                                 \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        }                                                                                                        \
        PP_FEED_DONE(); PP_MATH(2, x0, 0); PP_MATH_DONE();                                                       \
    }
    if constexpr (BM != 128) {
        for (int kt = 0; kt < nk; kt += 2) {
            PP_TILE(kt, 0);
            if (kt + 1 < nk) PP_TILE(kt + 1, 1);
        }
    } else {
        for (int kt = 0; kt < nk; kt += 2) {
            PP_TILE128(kt, 0);
            if (kt + 1 < nk) PP_TILE128(kt + 1, 1);
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with the late group's last barrier
    __builtin_amdgcn_sched_barrier(0);
    PP_TS(3);
    const int em0 = m0, eco0 = co0;               // the epilogue's tile
    bool more = false;
    if constexpr (PERSIST) {
        vb += gridDim.x;
        more = vb < p.total_tiles;
        if (more) {                               // next tile: offsets, then its prologue DMA - in flight during the epilogue below
            PP_TILE_SETUP(vb);
            PP_PROLOGUE_ISSUE();
        }
        first_tile = false;
        __builtin_amdgcn_sched_barrier(0);
    }
    PP_TS(4);
#ifdef PP_STAMP
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_), "=s"(rt1_) :: "memory");
    if (blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pp_stamps[wr][k] = st_[k];
        pp_stamps[wr][6] = clk1_ - clk0_; pp_stamps[wr][7] = rt1_ - rt0_;
    }
#endif

    // ---- epilogue: straight from the accumulators -------------------------------------------------------------------
    // D row 4*fq + reg of tile i is A-operand row 64*wc + 16*i + 4*fq + reg = cout co0 + 64*wc + 32*(i>>1) + 8*fq + 4*(i&1) + reg,
    // D column fr of tile j is pixel m0 + 128*wr + 16*j + fr: tiles (2q, 2q+1) give a lane 8 consecutive couts of one pixel.
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.res), 0, RES ? p.res_bytes : 0, 0x00020000);
    float bs[2][8];
    bool cok[2], cok2[2];                          // validity of the lane's first / second group of 4 couts (they differ only in natural order)
    if (p.bias_lds) {
        // bias from the block's LDS copy: a global load here returns in order BEHIND the next tile's prologue DMA (96 KB), i.e. it held the
        // epilogue until that DMA had landed - 2.2 k of a 46 k-cycle tile iteration (s_memtime stamps, scripts/pp256_tile_stamps.hip)
        f32x4 t[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = eco0 + wc * 64 + q * 32 + fq * (natural ? 4 : 8), c2 = c + (natural ? 16 : 4);
            cok[q] = c < p.Cout; cok2[q] = c2 < p.Cout;
            t[2 * q] = pp_lds_read16(bias_l + c * 4); t[2 * q + 1] = pp_lds_read16(bias_l + c2 * 4);
        }
        pp_lds_wait(t[0], t[1], t[2], t[3]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { bs[q][e] = t[2 * q][e]; bs[q][4 + e] = t[2 * q + 1][e]; }
    } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        // the lane's 8 values of tile pair q: couts c .. c + 7 (permuted rows), or c .. c + 3 and c + 16 .. c + 19 (natural rows)
        const int c = eco0 + wc * 64 + q * 32 + fq * (natural ? 4 : 8), c2 = c + (natural ? 16 : 4);
        cok[q] = c < p.Cout; cok2[q] = c2 < p.Cout;
        const float4 lo = cok[q] ? *reinterpret_cast<const float4 *>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 hi = cok2[q] ? *reinterpret_cast<const float4 *>(p.bias + c2) : make_float4(0.f, 0.f, 0.f, 0.f);
        bs[q][0] = lo.x; bs[q][1] = lo.y; bs[q][2] = lo.z; bs[q][3] = lo.w;
        bs[q][4] = hi.x; bs[q][5] = hi.y; bs[q][6] = hi.z; bs[q][7] = hi.w;
    }
    }
    // The activation and the output type are run-time parameters; the loop below is compiled once per (activation, output type) and the
    // choice made ONCE per tile: with the tests inside the loop every tile pair paid scalar branches and register shuffles between the
    // variants' register layouts (567 v_mov in the epilogue's ISA; ~800 cycles per pixel tile, s_memtime stamps).
    constexpr int esz_r = RES == 2 ? 4 : 2;
    const unsigned ep_slot = (unsigned)(size_t)PP_LDS_PTR(BUF + WOFF + wave * 4096);   // two 2 KB transposition slots per wave (fp32 outputs)
    auto epilogue = [&](auto act_c, auto of32_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool OF32 = decltype(of32_c)::value;
    constexpr int esz_o = OF32 ? 4 : 2;
#pragma unroll
    for (int j = 0; j < TMW; ++j) {
        if (j == 1) PP_TS(6);
        if (j == 4) PP_TS(7);
        if constexpr (OF32 && RES != 1) {
            // fp32 outputs in natural cout order (the transformer GEMMs; += into the fp32 residual stream when RES = 2): whole 128-byte lines.
            // A lane's 8 values of tile pair q are two 16-byte chunks (fq, 4 + fq) of the line (pixel fr, couts 32 q .. 32 q + 31); in that
            // layout a load / store instruction covers 16 rows x 64 B, which the texture-address path takes about twice as long over as
            // 8 rows x 128 B - and this epilogue is a bandwidth pass (458 KB per tile, 27 k of the out-projection's 62 k-cycle tile
            // iteration).  A wave-private 2 KB LDS slot per pair ([16 rows][128 B], chunk c of row r at c ^ (r & 7); inline-asm DS
            // accesses, see pp_lds_read16) turns lines into fragments and back: lane l moves rows l >> 3 and (l >> 3) + 8, chunk
            // (l & 7) ^ (l >> 3).  The slots are the W half tiles of buffer 1, which the next tile's prologue DMA does not touch.
            const int sw = fr & 7, lr = lane >> 3, ch = (lane & 7) ^ lr;
            const int mrow = em0 + wr * XH + j * 16 + lr;
            int lo_off[2], hi_off[2];
            f32x4 rl[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cq = eco0 + wc * 64 + q * 32 + ch * 4;
                lo_off[q] = (mrow < p.M && cq < p.Cout) ? (mrow * p.Cout + cq) * 4 : OOB;
                hi_off[q] = (mrow + 8 < p.M && cq < p.Cout) ? ((mrow + 8) * p.Cout + cq) * 4 : OOB;
                if constexpr (RES == 2) {
                    rl[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, lo_off[q], 0, PVR_NT_AUX(512)));
                    rl[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, hi_off[q], 0, PVR_NT_AUX(512)));
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned sq = ep_slot + q * 2048;
                const unsigned f0 = sq + fr * 128 + ((fq ^ sw) << 4), f1 = sq + fr * 128 + (((4 + fq) ^ sw) << 4);
                const f32x4 a_lo = acc[2 * q][j], a_hi = acc[2 * q + 1][j];
                f32x4 v0 = {a_lo[0] + bs[q][0], a_lo[1] + bs[q][1], a_lo[2] + bs[q][2], a_lo[3] + bs[q][3]};
                f32x4 v1 = {a_hi[0] + bs[q][4], a_hi[1] + bs[q][5], a_hi[2] + bs[q][6], a_hi[3] + bs[q][7]};
                if constexpr (RES == 2) {
                    pp_lds_write16(sq + lane * 16, rl[q][0]);
                    pp_lds_write16(sq + 1024 + lane * 16, rl[q][1]);
                    f32x4 r0 = pp_lds_read16(f0), r1 = pp_lds_read16(f1);
                    pp_lds_wait(r0, r1, v0, v1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] += r0[e]; v1[e] += r1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (ACT == 1) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                    else if constexpr (ACT == 2) { v0[e] = v0[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v0[e])); v1[e] = v1[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v1[e])); }
                    else if constexpr (ACT == 3) { v0[e] = gelu_erf(v0[e]); v1[e] = gelu_erf(v1[e]); }
                }
                pp_lds_write16(f0, v0);
                pp_lds_write16(f1, v1);
                f32x4 o_lo = pp_lds_read16(sq + lane * 16), o_hi = pp_lds_read16(sq + 1024 + lane * 16);
                pp_lds_wait(o_lo, o_hi, v0, v1);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_lo), rs_out, lo_off[q], 0, PVR_NT_AUX(256));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_hi), rs_out, hi_off[q], 0, PVR_NT_AUX(256));
            }
            continue;
        }
        const int m = em0 + wr * XH + j * 16 + fr;
        u32x4 rr[2][2];
        if constexpr (RES != 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = eco0 + wc * 64 + q * 32 + fq * (natural ? 4 : 8);
                const int ro = (m < p.M && cok[q]) ? (m * p.Cout + c) * esz_r : OOB;
                rr[q][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 0, PVR_NT_AUX(512)));
                if constexpr (RES == 2) {
                    const int ro2 = (m < p.M && cok2[q]) ? (m * p.Cout + c) * esz_r + (natural ? 64 : 16) : OOB;
                    rr[q][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro2, 0, PVR_NT_AUX(512)));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = eco0 + wc * 64 + q * 32 + fq * (natural ? 4 : 8);
            const f32x4 lo = acc[2 * q][j], hi = acc[2 * q + 1][j];
            float v[8] = {lo[0] + bs[q][0], lo[1] + bs[q][1], lo[2] + bs[q][2], lo[3] + bs[q][3],
                          hi[0] + bs[q][4], hi[1] + bs[q][5], hi[2] + bs[q][6], hi[3] + bs[q][7]};
            if constexpr (RES == 1) {
                const u32x4 r = rr[q][0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += from_h<F16>((u16)(r[e] & 0xffffu));
                    v[2 * e + 1] += from_h<F16>((u16)(r[e] >> 16));
                }
            } else if constexpr (RES == 2) {
                const f32x4 r0 = __builtin_bit_cast(f32x4, rr[q][0]), r1 = __builtin_bit_cast(f32x4, rr[q][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
            if constexpr (ACT == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if constexpr (ACT == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));   // v_rcp_f32 (1 ulp): an IEEE division here cost 15 % of the FC1 launch
            } else if constexpr (ACT == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
            }
            // byte offsets go into voffset (soffset stays 0): see store_b128_imm in bottleneck_chain.hip
            const int oo = (m < p.M && cok[q]) ? (m * p.Cout + c) * esz_o : OOB;
            if constexpr (OF32) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rs_out, oo, 0, PVR_NT_AUX(256));
                const int oo2 = (m < p.M && cok2[q]) ? (m * p.Cout + c) * esz_o + (natural ? 64 : 16) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rs_out, oo2, 0, PVR_NT_AUX(256));
            } else {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(v[2 * e]) | ((unsigned)to_h<F16>(v[2 * e + 1]) << 16);   // (pack2_h measured 2.6 % slower on ViT-B/16 here)
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, oo, 0, PVR_NT_AUX(256));
            }
        }
    }
    };
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using BT = std::integral_constant<bool, true>; using BF = std::integral_constant<bool, false>;
        if (p.out_f32) {
            if (p.act == 0) epilogue(I0{}, BT{}); else if (p.act == 1) epilogue(I1{}, BT{}); else if (p.act == 2) epilogue(I2{}, BT{}); else epilogue(I3{}, BT{});
        } else {
            if (p.act == 0) epilogue(I0{}, BF{}); else if (p.act == 1) epilogue(I1{}, BF{}); else if (p.act == 2) epilogue(I2{}, BF{}); else epilogue(I3{}, BF{});
        }
    }
    PP_TS(5);
#ifdef PP_TSTAMP
    if (tile_it_ == PP_TSTAMP && blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pp_tstamps[wr][k] = tts_[k];
    }
    ++tile_it_;
#endif
    if (!more) break;
  }
#undef PP_TILE128
#undef PP_TILE
#undef PP_MATH
#undef PP_READ_W
#undef PP_READ_X
#undef PP_MATH_DONE
#undef PP_FEED_DONE
#undef PP_PROLOGUE_ISSUE
#undef PP_TILE_SETUP
#undef PP_STAGE_W
#undef PP_ADVANCE_X
#undef PP_STAGE_X
}

static long long g_pp_persistent_launches = 0;
long long pp_persistent_launches() { return g_pp_persistent_launches; }

// PVR_PP_PERSIST: tiles per launch from which the persistent form is used (default 384 = 1.5 tiles per CU; 0 disables it)
static int pp_persist_min() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_PP_PERSIST"); v = e ? atoi(e) : 384; }
    return v;
}

// DUAL launches (conv3 & downsample of a stride-2 bottleneck): 224-pixel tiles, no residual operand
template <bool F16>
static pvr_status launch_pp_dual(PPP &p, hipStream_t stream) {
    constexpr int BM = 224, lds = 2 * (256 * 128 + 32768) + 16384;
    static DeviceOnce attr_done;
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_pp256_kernel<BM, F16, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_pp256_kernel<BM, F16, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    const int grid = ((p.M + BM - 1) / BM) * p.n_tiles;
    p.total_tiles = grid;
    p.pointwise = 0;
    p.bias_lds = p.CoutPad <= 4096 ? 1 : 0;
    if (pp_persist_min() > 0 && grid >= pp_persist_min()) {
        ++g_pp_persistent_launches;
        hipLaunchKernelGGL((conv_pp256_kernel<BM, F16, 0, true, true>), dim3(256), dim3(512), lds, stream, p);
    } else {
        hipLaunchKernelGGL((conv_pp256_kernel<BM, F16, 0, false, true>), dim3(grid), dim3(512), lds, stream, p);
    }
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

template <int BM, bool F16, int RES>
static pvr_status launch_pp_inst(PPP &p, hipStream_t stream) {
    constexpr int lds = 2 * ((BM == 224 ? 256 : BM) * 128 + 32768) + 16384;      // two staging buffers + the bias vector (<= 4096 floats)
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_pp256_kernel<BM, F16, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (BM != 128)
            PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_pp256_kernel<BM, F16, RES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    const int grid = ((p.M + BM - 1) / BM) * p.n_tiles;
    p.total_tiles = grid;
    p.pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
    p.bias_lds = p.CoutPad <= 4096 ? 1 : 0;
    if constexpr (BM != 128) {
        if (pp_persist_min() > 0 && grid >= pp_persist_min()) {       // one block per CU (128 KB of LDS each), several tiles per block
            ++g_pp_persistent_launches;
            hipLaunchKernelGGL((conv_pp256_kernel<BM, F16, RES, true>), dim3(256), dim3(512), lds, stream, p);
            PVR_LAUNCH_CHECK();
            return PVR_OK;
        }
    }
    hipLaunchKernelGGL((conv_pp256_kernel<BM, F16, RES>), dim3(grid), dim3(512), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

template <int BM>
static pvr_status launch_pp_bm(PPP &p, int rmode, int dtype, hipStream_t stream) {
    if (dtype == PVR_F16) {
        if (rmode == 0) return launch_pp_inst<BM, true, 0>(p, stream);
        return rmode == 1 ? launch_pp_inst<BM, true, 1>(p, stream) : launch_pp_inst<BM, true, 2>(p, stream);
    }
    if (rmode == 0) return launch_pp_inst<BM, false, 0>(p, stream);
    return rmode == 1 ? launch_pp_inst<BM, false, 1>(p, stream) : launch_pp_inst<BM, false, 2>(p, stream);
}

// shapes the kernel accepts (the caller decides whether it is the faster choice)
bool pp256_supported(int64_t M, int cin, int cout, int kh, int kw, int64_t in_bytes, int64_t w_bytes, int64_t out_bytes, int64_t res_bytes) {
    return cin % 64 == 0 && cout % 8 == 0 && kh <= 3 && kw <= 3 && M < (1ll << 31) && in_bytes < 0x7ffffff0ll && w_bytes < 0x7ffffff0ll &&
           out_bytes < 0x7ffffff0ll && res_bytes < 0x7ffffff0ll;
}

pvr_status launch_conv_pp256(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                             int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, int bm,
                             hipStream_t stream, const void *in2, int h2, int w2, int cin2, int stride2) {
    PPP p;
    p.in2 = (const u16 *)in2; p.in2_bytes = 0; p.H2 = h2; p.W2 = w2; p.Cin2 = cin2; p.stride2 = stride2; p.nk2 = 0;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.res = (const u16 *)res; p.bias = bias; p.out = out;
    p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64;
    p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - kh) / stride + 1;
    p.Wo = (w + 2 * pad - kw) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    p.K = kh * kw * cin + (in2 ? cin2 : 0);
    if (in2) {
        // wgt: (CoutPad, kh*kw*cin + cin2) - the second operand's columns behind the first's
        PVR_REQUIRE(!res && !out_f32 && cin2 % 64 == 0 && stride2 >= 1 && (h2 - 1) / stride2 + 1 == p.Ho && (w2 - 1) / stride2 + 1 == p.Wo,
                    "conv_pp256: the second operand must be a 1 x 1 (strided) view with the output's size, without a residual");
        PVR_REQUIRE((int64_t)n * h2 * w2 * cin2 * 2 < 0x7ffffff0ll, "conv_pp256: second operand larger than 2 GiB");
        p.in2_bytes = (unsigned)((int64_t)n * h2 * w2 * cin2 * 2); p.nk2 = cin2 / 64;
    }
    const int64_t inb = (int64_t)n * h * w * cin * 2, wb = (int64_t)p.CoutPad * p.K * 2, ob = M * cout * (out_f32 ? 4 : 2),
                  rb = res ? M * cout * (res_f32 ? 4 : 2) : 0;
    PVR_REQUIRE(pp256_supported(M, cin, cout, kh, kw, inb, wb, ob, rb), "conv_pp256: unsupported shape");
    PVR_REQUIRE(bm == 256 || bm == 224 || bm == 128, "conv_pp256: pixel tile must be 256, 224 or 128");
    p.M = (int)M; p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb; p.out_bytes = (unsigned)ob; p.res_bytes = (unsigned)rb;
    p.act = act; p.out_f32 = out_f32;
    p.n_tiles = (cout + 255) / 256;
    const int rmode = !res ? 0 : (res_f32 ? 2 : 1);
    if (in2) return dtype == PVR_F16 ? launch_pp_dual<true>(p, stream) : launch_pp_dual<false>(p, stream);
    if (bm == 224) return launch_pp_bm<224>(p, rmode, dtype, stream);
    return bm == 256 ? launch_pp_bm<256>(p, rmode, dtype, stream) : launch_pp_bm<128>(p, rmode, dtype, stream);
}

}  // namespace pvr
