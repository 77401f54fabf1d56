// Deep-K implicit-GEMM convolution / linear layer, round 3: 224 x 256 output tile on FOUR waves, each owning 112 pixels x 128 couts
// (7 x 8 MFMA tiles, 224 accumulator registers - 256 would need every AccVGPR of the wave), one wave per SIMD.
// STATUS: bit-identical to conv_igemm / conv_pp256 and 11-60 % SLOWER than conv_pp256; opt-in only (PVR_CONV_ALGO=4 /
// pvr_debug_set_conv_algo(4)), never chosen by the plan.  Kept with its measurements (profiles/experiments/r03_conv_w4.txt) because it
// answers the question DESIGN 8.1(a) asked after round 2.
//
// The idea: conv_pp256.hip's eight waves own 128 x 64 outputs each and read 0.375 ds_read_b128 per MFMA; per 64-deep K tile that is
// 1536 cycles of the CU's LDS port for fragment reads plus the LDS-DMA writes against 2048 cycles of MFMA issue, and its K tile
// measures 2810 cycles.  A 112 x 128 wave tile reads 0.27 fragments per MFMA.  With one wave per SIMD there is no partner wave to
// alternate with, so everything that is not an MFMA has to fit the 8 spare issue cycles of a 16-cycle MFMA gap.
// What the measurements say: fragment reads, address arithmetic and the global loads do fit (964 -> 969 cycles per 56-MFMA step);
// the staging does not - LDS-DMA holds the issuing wave 60-180 cycles per KB, and the register-staged form built here pays ~165
// cycles per step in the VGPR -> LDS store path and ~170 waiting for loads that three register sets can only request two steps ahead.
//
//   out[m][co] = sum_k X[m][k] * W[co][k],  m = (n,ho,wo),  k = (kh,kw,c); same K order (32-deep MFMA steps, ascending) and the
//   same epilogue arithmetic as conv_igemm.hip / conv_pp256.hip: bit-identical to both (tests/test_gpu_encoder.py).
//
// LDS: three stages each of X ([224 pixel rows][32 k], 14 KB) and W ([256 cout rows][32 k], 16 KB), 16-bit, rows of 64 B, + 1 spare KB.
// The 16-byte chunk c of row r sits at chunk position c ^ 2 * ((r >> 2) & 1): ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md, LDS table) over 64 banks of 4 B, and with 64-byte rows the unswizzled image puts
// rows r and r + 4k of one chunk column on the same banks - every fragment read two-way conflicted.
// im2col gather, tile tails and the weight-row permutation (LDS row 16t+4a+c of a 32-row block <- cout 8a+4t+c, so that a lane's
// accumulators of a tile pair are 8 consecutive output channels of one pixel) live in the per-lane source offset of the staging
// loads; out-of-range sources return zeros.  Waves 0 / 1 stage X (7 pieces of 16 rows each per step), waves 2 / 3 stage W (8).
//
// Pipeline (step s = one 32-deep slice; the X fragments of step s are in registers when the step starts).  For each of the 8 cout
// tiles i, one instruction per MFMA gap (W4_ROW):
//     request piece i of step s+4: 16-byte buffer load into register set (s+4) % 3
//     read the W fragment three tiles ahead (ring of 4) and X fragment i of step s+1 from LDS stage (s+1) % 3
//     s_waitcnt vmcnt(16): piece i of step s+2 has arrived (requested two steps ago; what was requested after it stays in flight)
//     ds_write_b128 it into LDS stage (s+2) % 3     (free: it held step s-1, whose fragments were read by the end of step s-1)
//   then s_waitcnt lgkmcnt(0); s_barrier.
//   Only plain loads, LDS writes and LDS reads; vmcnt retires in issue order and each wave counts only its own pieces.  Past the end of
//   the K range the requests go to an out-of-range offset (zeros, no memory access): the step body has no branch.
#ifdef PVR_EXPERIMENTS   // round-3 experiment (bit-identical, 11-60 % slower than conv_pp256: profiles/experiments/r03_conv_w4.txt); make EXPERIMENTS=1
#include <utility>
#include <vector>
#include "common.h"

namespace pvr {

struct PPP {                       // same launch parameters as conv_pp256.hip
    const u16 *in, *wgt, *res;
    const float *bias;
    void *out;
    int H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad, M, K;
    unsigned in_bytes, w_bytes, out_bytes, res_bytes;
    int act, out_f32;
    int n_tiles;
    int total_tiles;
    long long *stamps;      // diagnostics (PVR_W4_STAMPS): s_memtime at the top and at the end of the MFMAs of each step, one block; else null
};

#define W4_LDS(off_) ((__attribute__((address_space(3))) void *)(smem + (off_)))

// Accumulators: AccVGPRs a0 .. a223 by NAME.  Left to itself hipcc keeps loop-carried MFMA accumulators in VGPRs once there are
// more than ~128 of them and copies them into AccVGPRs in front of every MFMA (1100 v_accvgpr moves per step, 17-274 spilled
// registers in three formulations tried); an "+a" constraint per MFMA makes it copy the other way round.  So the kernel owns a fixed
// block of the accumulator file: accumulator (i, j) = a[(i * TM + j) * 4 .. + 3]; the MFMA, the zero-fill and the read-back are
// inline assembly with the register number as an immediate ("n"), and EVERY one of those statements declares the whole block a0 .. a223
// clobbered: no compiler value that lives across one of them can be allocated there (left alone, hipcc put the staging registers of
// the register-staged pipeline into a84 .. a147 - buffer_load and ds_write take AccVGPR data operands - and the MFMAs destroyed them).
// What is left to the compiler: 256 VGPRs + a224 .. a255.  Values that do not live across any of the statements could still land in
// the block, so the zero-fill comes after the prologue (whose loads and LDS writes have no MFMA between them).  Operands stay ordinary
// values ("v"), so the compiler's s_waitcnt for the fragment reads is in place; the same accumulator recurs 56 MFMAs later; s_nops
// separate the zero-fill from the first MFMA and the last MFMA from the read-back (the hazard recogniser cannot see inside the asm).
#define W4_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"
template <bool F16, int BASE>
__device__ __forceinline__ void mfma_fix(u32x4 a, u32x4 b) {       // (fragments as four dwords: a bf16x8 asm operand is re-packed with v_perm)
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(a), "v"(b), "n"(BASE), "n"(BASE + 3) : W4_ACC_CLOBBERS);
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(a), "v"(b), "n"(BASE), "n"(BASE + 3) : W4_ACC_CLOBBERS);
}
template <bool F16, int I, int TM_, int... J>
__device__ __forceinline__ void mfma_row(u32x4 w, const u32x4 (&x)[TM_], std::integer_sequence<int, J...>) {
    (mfma_fix<F16, (I * TM_ + J) * 4>(w, x[J]), ...);
}
template <int N>
__device__ __forceinline__ void acc_zero_one() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(N) : W4_ACC_CLOBBERS); }
template <int... N>
__device__ __forceinline__ void acc_zero(std::integer_sequence<int, N...>) { (acc_zero_one<N>(), ...); }
template <int BASE>
__device__ __forceinline__ f32x4 acc_read() {
    float r0, r1, r2, r3;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "n"(BASE), "n"(BASE + 1), "n"(BASE + 2), "n"(BASE + 3) : W4_ACC_CLOBBERS);
    return f32x4{r0, r1, r2, r3};
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <bool F16, int RES>
__global__ __launch_bounds__(256, 1) void conv_w4_kernel(PPP p) {
    typedef u32x4 V8;                              // an MFMA operand fragment: 8 x 16 bit, handled as four dwords
    constexpr int TM = 7, BMW = 2 * TM * 16;        // pixel tiles per wave, pixel rows per block (224)
    constexpr int NST = 3, XST = BMW * 64, WST = 256 * 64, WBASE = NST * XST, OOB = 0x7ffffff0;   // LDS stages per operand, bytes per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (swz / p.n_tiles) * BMW, co0 = (swz % p.n_tiles) * 256;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wgt), 0, p.w_bytes, 0x00020000);

    // ---- staging roles: waves 0 / 1 stage X (7 DMA instructions each: 16 pixel rows x 64 B per instruction), waves 2 / 3 stage W (8 each).
    // lane -> LDS row (lane / 4) of the instruction's 16 rows, 16-byte chunk lane % 4
    const bool xrole = wave < 2;
    int off[8], msk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (xrole) {
            const int r = (wave * 7 + i) * 16 + (lane >> 2);
            const int m = m0 + r;
            const bool ok = i < 7 && m < p.M;
            const int mm = ok ? m : 0;
            const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            off[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + (lane & 3) * 8) * 2;
            int hb = 0, wb = 0;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                hb |= (int)(ok && t3 < p.KH && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
                wb |= (int)(t3 < p.KW && (unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
            }
            int mask = 0;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * p.KW)) : 0;
            msk[i] = ~mask | 0x80000000;               // bit t set: tap t of this pixel row reads padding (or the row does not exist)
        } else {
            const int r = ((wave - 2) * 8 + i) * 16 + (lane >> 2);
            const int co = co0 + (r & ~31) + 8 * ((r >> 2) & 3) + 4 * ((r >> 4) & 1) + (r & 3);
            off[i] = co < p.CoutPad ? (co * p.K + (lane & 3) * 8) * 2 : OOB;
            msk[i] = 0x80000000;                        // bit 31: the position used for requests past the last step
        }
    }
    const int c32pt = p.Cin >> 5;                  // 32-deep steps per filter tap
    const int nk = p.KH * p.KW * c32pt;
    // Register staging: a step's 7 (X wave) / 8 (W wave) 16-row pieces are ordinary 16-byte buffer loads into one of three register
    // sets, requested FOUR steps ahead - one piece per cout-tile row of the MFMA stream, so the issue slots come out of the MFMA gaps -
    // and written to LDS (ds_write_b128) two steps before their MFMAs.  (The first builds used LDS-DMA: bit-identical, 35-45 % slower
    // than conv_pp256 whatever the prefetch depth - a buffer_load ... lds holds the ISSUING wave for 60-180 cycles, and with one wave
    // per SIMD that is MFMA time; conv_pp256 hides the same cost in its partner wave's math phase.  profiles/experiments/r03_conv_w4.txt)
    int xs_tap = 0, xs_kh = 0, xs_kw = 0, xs_c = 0, xs_s = 0;      // filter position / index of the step whose pieces are being requested
    int add_s = 0, soff_s = 0, bit_s = 0;                          // per step, scalar: byte offset added per lane, soffset, mask bit tested
    const auto rs_st = xrole ? rs_in : rs_w;
    auto piece_begin = [&](bool live) {
        const int tap_off = ((xs_kh * p.W + xs_kw) * p.Cin + xs_c * 32) * 2;
        add_s = xrole ? tap_off : 0;
        soff_s = xrole ? 0 : xs_s * 64;
        bit_s = live ? xs_tap : 31;
    };
    // A piece costs three VALU instructions and the load, each placed by hand in its own MFMA gap (W4_ROW):
    //   t = v_bfe_u32(msk, bit, 1);  a = off + add;  vo = (t << 31) | a  (out of range: the load returns zeros);  buffer_load_dwordx4
    auto piece_end = [&]() {
        ++xs_s;
        if (xrole) { if (++xs_c == c32pt) { xs_c = 0; ++xs_tap; if (++xs_kw == p.KW) { xs_kw = 0; ++xs_kh; } } }
    };
    // LDS address of piece i of stage st: lane part + role part + st * (XST | WST) + i * 1024; an X wave's eighth piece goes to a spare KB
    constexpr int SPARE = NST * (XST + WST);
    const int st_slot = (lane >> 2) * 64 + (((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);      // row lane / 4 of the piece, swizzled chunk
    const int st_lane = st_slot + (xrole ? wave * 7 * 1024 : WBASE + (wave - 2) * 8 * 1024);
    const int st_sz = xrole ? XST : WST;
    const int st_last = xrole ? 1 : 0;
    auto piece_store = [&](int i, int st_, const V8 &g) {
        const int a = (i == 7 && st_last) ? st_slot + SPARE : st_lane + st_ * st_sz + i * 1024;
        *reinterpret_cast<V8 *>(smem + a) = g;
    };
    auto piece_load_now = [&](int i, V8 (&g)[8]) {                 // prologue form
        const int t = __builtin_amdgcn_ubfe(msk[i], bit_s, 1);
        g[i] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_st, (t << 31) | (off[i] + add_s), soff_s, 0));
    };
    // fragment reads of one step: X tiles of this wave's pixel half, W tiles of its cout half
    const int fqs = (fq ^ (((fr >> 2) & 1) << 1)) * 16;
    const int xbase = (wr * TM * 16 + fr) * 64 + fqs, wbase = WBASE + (wc * 128 + fr) * 64 + fqs;
    auto read_x = [&](int st_, V8 (&xf)[TM]) {
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const V8 *>(smem + st_ * XST + xbase + j * 1024);
    };


    // ---- prologue: steps 0 / 1 in LDS, steps 2 / 3 requested, the fragments of step 0 in registers ---------------------------
    V8 g0[8], g1[8], g2[8];
    { piece_begin(0 < nk); _Pragma("unroll") for (int i = 0; i < 8; ++i) piece_load_now(i, g0); piece_end(); }
    { piece_begin(1 < nk); _Pragma("unroll") for (int i = 0; i < 8; ++i) piece_load_now(i, g1); piece_end(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) { piece_store(i, 0, g0[i]); piece_store(i, 1, g1[i]); }
    { piece_begin(2 < nk); _Pragma("unroll") for (int i = 0; i < 8; ++i) piece_load_now(i, g2); piece_end(); }
    { piece_begin(3 < nk); _Pragma("unroll") for (int i = 0; i < 8; ++i) piece_load_now(i, g0); piece_end(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    V8 xa[TM], xb[TM], wf[4];           // X fragments double-buffered by step; W fragments in a ring of four, read three cout tiles ahead
    read_x(0, xa);
#pragma unroll
    for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const V8 *>(smem + wbase + i * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    acc_zero(std::make_integer_sequence<int, 8 * TM * 4>{});        // from here to the read-back the kernel owns a0 .. a223
    asm volatile("s_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

// One wave per SIMD: a 16x16x32 MFMA occupies the matrix pipe for 16 cycles and the issue port for 8, so each gap hides about one
// other instruction (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost'); instructions bunched between the rows add their
// full issue time to the step - the first builds of this kernel did exactly that, and ran 40-45 % behind conv_pp256 whatever the
// staging.  So every cout tile I_ of step s_ is spelled out as 7 MFMAs with ONE instruction in each gap:
//   gaps 0-2  the address of piece I_ of step s_+4              gap 3  its buffer load (register set GL)
//   gap 4     the W fragment three tiles ahead (ring slot (I_+3) % 4; tiles 8..10 are tiles 0..2 of step s_+1)
//   gap 5     X fragment I_ of step s_+1 and piece I_ of step s_+2 (set GW, requested two steps ago) -> LDS
// Measured (profiles/experiments/r03_conv_w4.txt, s_memtime stamps and knock-out builds): 1300 cycles per step of 56 MFMAs where the
// MFMAs alone take 964; ~165 of the difference is the eight ds_write_b128 (two ds_write_b64 in separate gaps: no better), ~170 the
// wait for the loads requested two steps (~2600 cycles) earlier; the loads' and fragment reads' own issue slots cost < 10.
#ifndef W4_KNOCK
#define W4_KNOCK 0          // timing experiments only (scripts/w4_knock.sh): 1 no LDS writes, 2 no loads, 4 no fragment reads, 8 no barrier, 16 no vmcnt wait
#endif
#define W4_SB __builtin_amdgcn_sched_barrier(0);
#define W4_MF(I_, J_, XC) mfma_fix<F16, ((I_) * TM + (J_)) * 4>(wf[(I_) % 4], XC[J_]); W4_SB
#define W4_ROW(I_, XC, XN, GL, GW, s_)                                                                           \
        W4_MF(I_, 0, XC) const int t##I_ = __builtin_amdgcn_ubfe(msk[I_], bit_s, 1); W4_SB                       \
        W4_MF(I_, 1, XC) const int a##I_ = off[I_] + add_s; W4_SB                                                \
        W4_MF(I_, 2, XC) const int v##I_ = (t##I_ << 31) | a##I_; W4_SB                                          \
        W4_MF(I_, 3, XC) if constexpr (!(W4_KNOCK & 2)) GL[I_] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs_st, v##I_, soff_s, 0)); W4_SB \
        W4_MF(I_, 4, XC) if constexpr (!(W4_KNOCK & 4)) wf[((I_) + 3) % 4] = *reinterpret_cast<const V8 *>(((I_) + 3 < 8 ? wcur : wnext) + (((I_) + 3) % 8) * 1024); W4_SB \
        W4_MF(I_, 5, XC) if constexpr ((I_) < TM && !(W4_KNOCK & 4)) XN[(I_) < TM ? (I_) : 0] = *reinterpret_cast<const V8 *>(xnext + (I_) * 1024); W4_SB \
        if constexpr (!(W4_KNOCK & 16)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   /* in flight behind piece I_ of s_+2: 7 - I_ of its set, 8 of s_+3, I_ + 1 of s_+4 */ \
        if constexpr (!(W4_KNOCK & 1)) piece_store(I_, ((s_) + 2) % NST, GW[I_]); W4_SB                          \
        W4_MF(I_, 6, XC)
// Past the end of the K range the requests return zeros and the fragment reads fetch stale stages: harmless, and no branch.
#define W4_STEP(s_, XC, XN, GL, GW)                                                                              \
    {                                                                                                            \
        const char *wcur = smem + ((s_) % NST) * WST + wbase;                                                    \
        const char *wnext = smem + (((s_) + 1) % NST) * WST + wbase;                                             \
        const char *xnext = smem + (((s_) + 1) % NST) * XST + xbase;                                             \
        piece_begin((s_) + 4 < nk);                                                                              \
        if (stamp && (s_) < 128) p.stamps[(wave * 128 + (s_)) * 2] = __builtin_amdgcn_s_memtime();               \
        W4_SB                                                                                                    \
        W4_ROW(0, XC, XN, GL, GW, s_) W4_ROW(1, XC, XN, GL, GW, s_) W4_ROW(2, XC, XN, GL, GW, s_) W4_ROW(3, XC, XN, GL, GW, s_) \
        W4_ROW(4, XC, XN, GL, GW, s_) W4_ROW(5, XC, XN, GL, GW, s_) W4_ROW(6, XC, XN, GL, GW, s_) W4_ROW(7, XC, XN, GL, GW, s_) \
        piece_end();                                                                                             \
        if (stamp && (s_) < 128) p.stamps[(wave * 128 + (s_)) * 2 + 1] = __builtin_amdgcn_s_memtime();           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        W4_SB                                                                                                    \
        if constexpr (!(W4_KNOCK & 8)) __builtin_amdgcn_s_barrier();                                                   \
        W4_SB                                                                                                    \
    }
    const bool stamp = p.stamps && blockIdx.x == gridDim.x / 2 && lane == 0;
    // six steps per trip: the fragment sets alternate (a / b), the LDS stages rotate by three, the piece sets by two
    for (int s = 0;; s += 6) {
        W4_STEP(s, xa, xb, g1, g2);
        if (s + 1 >= nk) break;
        W4_STEP(s + 1, xb, xa, g2, g0);
        if (s + 2 >= nk) break;
        W4_STEP(s + 2, xa, xb, g0, g1);
        if (s + 3 >= nk) break;
        W4_STEP(s + 3, xb, xa, g1, g2);
        if (s + 4 >= nk) break;
        W4_STEP(s + 4, xa, xb, g2, g0);
        if (s + 5 >= nk) break;
        W4_STEP(s + 5, xb, xa, g0, g1);
        if (s + 6 >= nk) break;
    }
#undef W4_STEP
#undef W4_ROW
#undef W4_MF
#undef W4_SB
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // (MFMA results are read from the AccVGPRs below: well clear of the last issue)

    // ---- epilogue: straight from the accumulators -------------------------------------------------------------------
    // D row 4*fq + reg of cout tile i is cout co0 + 128*wc + 32*(i>>1) + 8*fq + 4*(i&1) + reg (weight rows permuted on the way in),
    // D column fr of pixel tile j is pixel m0 + 112*wr + 16*j + fr: tiles (2q, 2q+1) give a lane 8 consecutive couts of one pixel.
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.res), 0, RES ? p.res_bytes : 0, 0x00020000);
    const int esz_o = p.out_f32 ? 4 : 2;
    constexpr int esz_r = RES == 2 ? 4 : 2;
    static_for<4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const int c = co0 + wc * 128 + q * 32 + fq * 8;
        const bool cok = c < p.Cout;
        float bs[8];
        {
            const float4 lo = cok ? *reinterpret_cast<const float4 *>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 hi = cok ? *reinterpret_cast<const float4 *>(p.bias + c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            bs[0] = lo.x; bs[1] = lo.y; bs[2] = lo.z; bs[3] = lo.w; bs[4] = hi.x; bs[5] = hi.y; bs[6] = hi.z; bs[7] = hi.w;
        }
        static_for<TM>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int m = m0 + wr * TM * 16 + j * 16 + fr;
            const bool ok = m < p.M && cok;
            u32x4 rr0 = u32x4{0u, 0u, 0u, 0u}, rr1 = u32x4{0u, 0u, 0u, 0u};
            if constexpr (RES != 0) {
                const int ro = ok ? (m * p.Cout + c) * esz_r : OOB;
                rr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 0, 0));
                if constexpr (RES == 2) rr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 16, 0));
            }
            const f32x4 lo = acc_read<((2 * q) * TM + j) * 4>(), hi = acc_read<((2 * q + 1) * TM + j) * 4>();
            float v[8] = {lo[0] + bs[0], lo[1] + bs[1], lo[2] + bs[2], lo[3] + bs[3], hi[0] + bs[4], hi[1] + bs[5], hi[2] + bs[6], hi[3] + bs[7]};
            if constexpr (RES == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += from_h<F16>((u16)(rr0[e] & 0xffffu));
                    v[2 * e + 1] += from_h<F16>((u16)(rr0[e] >> 16));
                }
            } else if constexpr (RES == 2) {
                const f32x4 r0 = __builtin_bit_cast(f32x4, rr0), r1 = __builtin_bit_cast(f32x4, rr1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
            } else if (p.act == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
            }
            const int oo = ok ? (m * p.Cout + c) * esz_o : OOB;       // (byte offsets in voffset, soffset 0: see store_b128_imm in bottleneck_chain.hip)
            if (p.out_f32) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rs_out, oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rs_out, oo + 16, 0, 0);
            } else {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (unsigned)to_h<F16>(v[2 * e]) | ((unsigned)to_h<F16>(v[2 * e + 1]) << 16);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, oo, 0, 0);
            }
        });
    });
}

static long long g_w4_launches = 0;
long long conv_w4_launches() { return g_w4_launches; }

template <bool F16, int RES>
static pvr_status launch_w4_inst(PPP &p, hipStream_t stream) {
    constexpr int lds = 3 * (224 * 64) + 3 * (256 * 64) + 1024;      // three X stages + three W stages + a spare KB = 91 KB
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv_w4_kernel<F16, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.mark();
    }
    const int grid = ((p.M + 223) / 224) * p.n_tiles;
    p.total_tiles = grid;
    ++g_w4_launches;
    hipLaunchKernelGGL((conv_w4_kernel<F16, RES>), dim3(grid), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// same shapes as conv_pp256 (pp256_supported); Cin % 64 == 0 makes every 32-deep step a half of one tap's 64-channel slice
pvr_status launch_conv_w4(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                          int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, hipStream_t stream) {
    PPP p;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.res = (const u16 *)res; p.bias = bias; p.out = out;
    p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64;
    p.KH = kh; p.KW = kw; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - kh) / stride + 1;
    p.Wo = (w + 2 * pad - kw) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo;
    p.K = kh * kw * cin;
    const int64_t inb = (int64_t)n * h * w * cin * 2, wb = (int64_t)p.CoutPad * p.K * 2, ob = M * cout * (out_f32 ? 4 : 2),
                  rb = res ? M * cout * (res_f32 ? 4 : 2) : 0;
    PVR_REQUIRE(cin % 64 == 0 && cout % 8 == 0 && kh <= 3 && kw <= 3 && M < (1ll << 31) && inb < 0x7ffffff0ll && wb < 0x7ffffff0ll &&
                ob < 0x7ffffff0ll && rb < 0x7ffffff0ll, "conv_w4: unsupported shape");
    p.M = (int)M; p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb; p.out_bytes = (unsigned)ob; p.res_bytes = (unsigned)rb;
    p.act = act; p.out_f32 = out_f32;
    p.n_tiles = (cout + 255) / 256;
    p.stamps = nullptr;
    const char *sp = getenv("PVR_W4_STAMPS");
    if (sp && *sp) {
        PVR_HIP_TRY(hipMalloc((void **)&p.stamps, 4 * 128 * 2 * 8));
        PVR_HIP_TRY(hipMemset(p.stamps, 0, 4 * 128 * 2 * 8));
    }
    const int rmode = !res ? 0 : (res_f32 ? 2 : 1);
    pvr_status st;
    if (dtype == PVR_F16) st = rmode == 0 ? launch_w4_inst<true, 0>(p, stream) : rmode == 1 ? launch_w4_inst<true, 1>(p, stream) : launch_w4_inst<true, 2>(p, stream);
    else st = rmode == 0 ? launch_w4_inst<false, 0>(p, stream) : rmode == 1 ? launch_w4_inst<false, 1>(p, stream) : launch_w4_inst<false, 2>(p, stream);
    if (p.stamps) {             // diagnostics only: cycles per step of one block's four waves -> text file named by PVR_W4_STAMPS
        std::vector<long long> h(4 * 128 * 2);
        PVR_HIP_TRY(hipStreamSynchronize(stream));
        PVR_HIP_TRY(hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        PVR_HIP_TRY(hipFree(p.stamps));
        if (FILE *f = fopen(sp, "a")) {
            const int nk = kh * kw * (cin / 32);
            fprintf(f, "# conv_w4 M %d Cin %d Cout %d k %d: per step [top -> last MFMA issued | -> next top] cycles, waves 0..3\n", p.M, cin, cout, kh);
            for (int s2 = 0; s2 + 1 < nk && s2 + 1 < 128; ++s2) {
                fprintf(f, "step %3d:", s2);
                for (int w2 = 0; w2 < 4; ++w2)
                    fprintf(f, "  %5lld | %5lld", h[(w2 * 128 + s2) * 2 + 1] - h[(w2 * 128 + s2) * 2], h[(w2 * 128 + s2 + 1) * 2] - h[(w2 * 128 + s2) * 2 + 1]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return st;
}

}  // namespace pvr

#endif  // PVR_EXPERIMENTS
