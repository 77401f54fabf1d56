// Fused tail of a torchvision Bottleneck (reference src/embeddings.py:118-120 -> torchvision resnet50), barrier-free form for the
// stride-1 blocks of layer1 (Cm = 64):
//
//     t2  = relu(conv2_3x3(t1) + b2)                       (64 -> 64)
//     y   = relu(conv3_1x1(t2) + b3 + residual)            (64 -> 256)      [DS: + Wd . x instead of a residual tensor]
//     t1' = relu(conv1_1x1_of_the_NEXT_block(y) + b1')     (256 -> Cmn, optional)
//
// bottleneck_chain.hip carries a 128-pixel tile through the three GEMMs with four waves that exchange t2 and y through LDS: nine
// barrier-separated taps, four barrier-separated cout groups, and a block that asks HBM for nothing while it computes conv2.  Its
// launches move their (minimal) bytes at 3.2-4.2 TB/s because a block is a serial chain of phases and only 2-3 blocks share a CU
// (profiles/experiments/r03_chain_knockouts.txt).  This form removes every exchange between waves:
//
//   * A WAVE owns 32 pixels (two 16-pixel MFMA tiles) through all three GEMMs.  With the weights as the A operand and their rows
//     permuted inside every 32-row block (chain_row_source), a lane's accumulators of a tile pair are 8 CONSECUTIVE output channels
//     of one pixel - which is exactly the B-operand fragment of the next GEMM's 32-channel K step.  t2 and y therefore go from
//     accumulators to MFMA operands in registers (bias + ReLU + 16-bit rounding in between, as the unfused launches round them);
//     nothing is written to LDS after the prologue and no wave ever waits for another.
//   * All weights of the block stay in LDS for the whole launch (W2 72 KB, W3 32 KB, W1' 32 KB, biases): one 512-thread block per CU,
//     persistent, every wave walks its own list of 32-pixel tiles.  (Cmn = 128: W1' is 64 KB, so W3's pieces are read from L2
//     instead; DS: Wd's likewise.)
//   * Memory lane layouts.  The MFMA fragment layout (lane = 16 * chunk + pixel) makes a quarter wave - the unit the texture-address
//     path works in - touch 16 pixel rows x 16 B of an NHWC tensor: measured 4x slower through the TA than 4 rows x 64 B
//     (profiles/experiments/r04_chain_wave.txt).  Two answers, both used:
//       - NHWC tensors (the launch's boundary with other kernels): lane l moves pixel l >> 2 and one 16-byte chunk of a 64-byte piece;
//         a write + a read of a wave-private 1 KB LDS slot turn a loaded register into a fragment and a result into a store register
//         (cw_m2f / cw_f2m_pair; outputs leave as full 128-byte lines).
//       - BLOCKED tensors between two launches of this form ("P16C8": [pixel >> 4][channel >> 3][pixel & 15][8 channels], i.e. every
//         (16 pixels x 8 channels) fragment column is 256 contiguous bytes): the fragment layout IS the coalesced layout - loads,
//         stores and the residual need no permutation and a wave instruction moves 1 KB of contiguous memory.
//   * conv2's pixels.  NHWC input: 16-byte pieces per K-step, XD steps ahead of their MFMAs through a register ring.  Blocked input
//     (HALO): the tile's whole halo - the ten 16-pixel blocks around it, 20 KB - is loaded ONCE, one full phase ahead, and the nine
//     taps' fragments are made from those registers by DPP row shifts (pixel m + s of a 16-lane row = lane + (s & 15) of block
//     k or k + 1): no load is issued while conv2 runs, none ever waits behind a store, and t1 crosses the TA once instead of nine times.
//
// Numerics: the same rounding points and the same K order per accumulator as bottleneck_chain.hip and the unfused launches -
// bit-identical outputs (tests/test_gpu_encoder.py::test_fused_bottleneck_chain_is_bit_identical, scripts/chain_wave_bench.hip).
#include "chain_params.h"

namespace pvr {

__device__ __forceinline__ int cw_row_source(int row) { return (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

#ifndef CW_KNOCK
#define CW_KNOCK 0      // timing experiments (scripts/chain_wave_bench.hip -DCW_KNOCK=bits): 1 no y / t1' stores, 2 residual loads out of range (zeros, no
                        // traffic), 4 conv2 pixel loads out of range
#endif
#ifndef CW_AUX_ST
#define CW_AUX_ST PVR_NT_AUX(1)     // cache policy of the y / t1' stores and of the residual loads (raw-buffer aux bits: 2 = nt, streaming)
#endif
#ifndef CW_AUX_RES
#define CW_AUX_RES PVR_NT_AUX(2)
#endif
// 16-byte buffer store, byte offset in voffset + immediate (never soffset: bottleneck_chain.hip, store_b128_imm)
template <int AUX = CW_AUX_ST>
__device__ __forceinline__ void cw_store(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int imm, int never = 0) {
    if constexpr (CW_KNOCK & 1) { if (never) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, AUX); }
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, AUX);
}
#ifndef CW_AUX_T1
#define CW_AUX_T1 CW_AUX_ST          // t1' stores (A/B: -DCW_AUX_T1=0 keeps the next launch's conv2 input in L2 / MALL)
#endif

typedef float cw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cw_f16x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two 16-bit values (round to nearest even, as to_h): one v_cvt_pk instruction instead of two conversions + a pack
template <bool F16> __device__ __forceinline__ unsigned cw_pack2(float a, float b) {
    const cw_f32x2 v = {a, b};
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cw_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cw_bf16x2));
}

// NHWC <-> fragment layout of one 1 KB piece (16 pixels x 64 B) through a wave-private LDS slot: the memory layout's lane l = 4 r + q holds
// row r, 16-byte position q; the fragment layout's lane (fr, fq) wants row fr, chunk fq.  The slot is the memory image with the chunks of
// row r XOR-ed by (r >> 1) & 3 - lane l therefore carries chunk q ^ ((r >> 1) & 3), which its global address accounts for - so that both
// the contiguous side and the strided side are bank-conflict free.  A write + a read cost ~12 LDS cycles; four ds_bpermute_b32, the first
// form of this kernel, ~64 (they made every NHWC side of a launch ~45 us slower than its blocked side).  DS operations of one wave execute
// in order, so a slot needs no wait between its write and its read, nor between two uses.
__device__ __forceinline__ u32x4 cw_m2f(char *slot, int lane, int foff, u32x4 v) {
    *reinterpret_cast<u32x4 *>(slot + lane * 16) = v;
    return *reinterpret_cast<const u32x4 *>(slot + foff);
}
// Two consecutive 32-channel fragments (64 channels = 128 B per pixel) of a 16-pixel tile -> two registers of FULL 128-byte lines: lane l
// gets rows (l >> 3) and (l >> 3) + 8, chunk (l & 7) ^ (l >> 3).  Slot image: [16 rows][128 B], chunk c of row r at position c ^ (r & 7).
// (Stores of 16 x 64-byte pieces - half lines, the second half arriving a microsecond later - made an NHWC output side ~45 us slower than
// a blocked one at layer1's size; full lines take the same number of instructions.)
__device__ __forceinline__ void cw_f2m_pair(char *slot, int lane, u32x4 even, u32x4 odd, u32x4 &lo, u32x4 &hi) {
    const int fr = lane & 15, fq = lane >> 4, base = fr * 128, sw = fr & 7;
    *reinterpret_cast<u32x4 *>(slot + base + ((fq ^ sw) << 4)) = even;
    *reinterpret_cast<u32x4 *>(slot + base + (((4 + fq) ^ sw) << 4)) = odd;
    lo = *reinterpret_cast<const u32x4 *>(slot + lane * 16);
    hi = *reinterpret_cast<const u32x4 *>(slot + 1024 + lane * 16);
}

// lane fr of every 16-lane row <- lane fr + d of `lo` where that stays inside the row, else lane fr + d - 16 of `hi` (DPP row shifts)
__device__ __forceinline__ unsigned cw_row_shift(unsigned lo, unsigned hi, int d) {
#define CW_SH(D_) case D_: { const int a_ = __builtin_amdgcn_update_dpp(0, (int)lo, 0x100 + D_, 0xf, 0xf, true); \
                             return (unsigned)__builtin_amdgcn_update_dpp(a_, (int)hi, 0x110 + (16 - D_), 0xf, 0xf, false); }
    switch (d) {
        CW_SH(1) CW_SH(2) CW_SH(3) CW_SH(4) CW_SH(5) CW_SH(6) CW_SH(7) CW_SH(8) CW_SH(9) CW_SH(10) CW_SH(11) CW_SH(12) CW_SH(13) CW_SH(14) CW_SH(15)
        default: return lo;
    }
#undef CW_SH
}

struct CwTile {
    int xb[2];      // conv2 input: byte offset of the lane's piece at the centre tap (NHWC: (pixel lp, chunk lc); blocked ring: pixel index m0 + 16 j + fr)
    int mk[2];      // 9-bit "tap inside the image" mask of the pixel the lane loads (NHWC) / owns in the fragment (blocked)
    int yi[2];      // residual: byte offset of the lane's 16 bytes of half-group 0
    int yo[2];      // y: likewise for the store
    int to[2];      // t1': likewise (tile pair 0)
    int hb;         // HALO: byte offset of the lane's 16 bytes of halo block 0, K half 0
    int xs[2];      // DS: byte offset of the lane's piece of the block input x (always NHWC: the stem writes it)
};

template <int CMN, bool INB, bool OUTB>
__device__ __forceinline__ void cw_setup(CwTile &a, int m0, int lane, int M, int H, int W) {
    const int fr = lane & 15, fq = lane >> 4, lp = lane >> 2, lc = (lane & 3) ^ ((lane >> 3) & 3);   // NHWC lanes: row lp, chunk lc (cw_m2f)
    a.hb = ((m0 >> 4) - 4) * 2048 + fq * 256 + fr * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 16 * j + (INB ? fr : lp);       // the pixel whose taps this lane masks
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
        a.xb[j] = INB ? m : m * 128 + lc * 16;
        a.xs[j] = (m0 + 16 * j + lp) * 128 + lc * 16;
        const int blk = (m0 >> 4) + j;                     // (m0 is a multiple of 32: pixel tile j is block blk of a blocked tensor)
        a.yi[j] = INB ? blk * 8192 + fq * 256 + fr * 16 : (m0 + 16 * j + lp) * 512 + lc * 16;
        // NHWC outputs leave as FULL 128-byte lines (cw_f2m_pair): lane l stores rows (l >> 3) and (l >> 3) + 8, chunk (l & 7) ^ (l >> 3) of a 64-channel group
        a.yo[j] = OUTB ? blk * 8192 + fq * 256 + fr * 16 : (m0 + 16 * j + (lane >> 3)) * 512 + (((lane & 7) ^ (lane >> 3)) << 4);
        a.to[j] = OUTB ? blk * (CMN * 32) + fq * 256 + fr * 16 : (m0 + 16 * j + (lane >> 3)) * (CMN * 2) + (((lane & 7) ^ (lane >> 3)) << 4);
    }
}

// INB / OUTB: t1 + residual / y + t1' in the blocked layout; HALO (with INB, W = 56): conv2's pixels from halo registers
// XD: conv2 K-steps of pixel pieces in flight (ring forms; 8 VGPRs each); RD: residual half-groups in flight (8 VGPRs each);
// WD: half-groups of W3 / Wd pieces in flight when they come from L2 (16 VGPRs each)
template <int CMN, bool F16, bool DS, bool W3G, bool INB, bool OUTB, bool HALO, int XD, int RD, int WD>
__global__ __launch_bounds__(512, 2) void chain_wave_kernel(ChainP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int WW = 56, HB = 10;                        // HALO: image width (the launcher checks), 16-pixel blocks m0 / 16 - 4 .. + 5 cover m0 - 57 .. m0 + 88
    constexpr int NH = 8, NK = 18;                         // half-groups of 32 couts; conv2 K-steps of 32 channels (9 taps x 2)
    constexpr int TN1 = CMN / 16;
    constexpr int W2L = 0, W3L = 73728, W1L = W3G ? 73728 : 73728 + 32768;
    constexpr int B2L = W1L + CMN * 512, B3L = B2L + 256, B1L = B3L + 1024, SCR = B1L + 512;   // SCR: 2 KB of layout-conversion slots per wave
    constexpr int OOB = 0x7ffffff0;
    constexpr int YH = INB ? 1024 : 64;                    // byte step of a half-group (32 channels) in the residual
    static_assert(XD >= 1 && XD <= NK && RD >= 1 && RD <= NH && NH % RD == 0 && WD >= 1 && WD <= NH, "prefetch depths (the residual ring must close over a tile)");
    static_assert(!HALO || INB, "the halo form reads the blocked layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) << 4);   // fragment lane's 16 bytes inside a conversion slot (cw_m2f)
    char *const slot0 = smem + SCR + wave * 2048, *const slot1 = slot0 + 1024;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w2), 0, p.w2_bytes, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w3), 0, p.w3_bytes, 0x00020000);
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1n), 0, p.w1n_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.xds : p.res), 0, DS ? p.xds_bytes : p.y_bytes, 0x00020000);
    const auto rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.wdsb : p.w3b), 0, DS ? p.wds_bytes : p.w3_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);

    // ---- prologue: the block's weights and biases -> LDS ([rows][64] 16-bit tiles, 128-byte rows, chunk ^= (row >> 1) & 7) -------
#pragma unroll
    for (int q = 0; q < 9; ++q) {                          // W2: 9 taps x 64 rows x 8 chunks; LDS row r holds cout cw_row_source(r)
        const int r = tid >> 3, c = tid & 7;
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, (cw_row_source(r) * 576 + c * 8) * 2, q * 128, 0));
        *reinterpret_cast<u32x4 *>(smem + W2L + q * 8192 + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
    }
    if constexpr (!W3G) {                                  // W3 (rows already permuted by the host): [256][64]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 512 * q, r = idx >> 3, c = idx & 7;
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, idx * 16, 0, 0));
            *reinterpret_cast<u32x4 *>(smem + W3L + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
        }
    }
    if constexpr (CMN > 0) {                               // W1' (rows permuted): [CMN][256] -> four K groups of [CMN][64]
#pragma unroll
        for (int q = 0; q < CMN * 32 / 512; ++q) {
            const int idx = tid + 512 * q, r = idx >> 5, c32 = idx & 31;
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, idx * 16, 0, 0));
            *reinterpret_cast<u32x4 *>(smem + W1L + (c32 >> 3) * (CMN * 128) + r * 128 + (((c32 & 7) ^ ((r >> 1) & 7)) << 4)) = v;
        }
    }
    if (tid < 64) *reinterpret_cast<float *>(smem + B2L + tid * 4) = p.b2[tid];
    if (tid < 256) *reinterpret_cast<float *>(smem + B3L + tid * 4) = p.b3[tid];
    if constexpr (CMN > 0) { if (tid < CMN) *reinterpret_cast<float *>(smem + B1L + tid * 4) = p.b1n[tid]; }
    __syncthreads();                                       // the only barrier of the kernel

    // ---- this wave's tiles: chunk = 256 consecutive pixels (8 waves x 32); an XCD's blocks walk a contiguous run of chunks side by
    // side, so the rows two neighbouring tiles both read meet in that XCD's L2
    const int nch = (p.M + 255) >> 8, cx = (nch + 7) >> 3, L = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, c_end = min((xcd + 1) * cx, nch);
    int chunk = xcd * cx + (blockIdx.x >> 3);
    if (chunk >= c_end) return;

    // per-lane LDS fragment bases: row fr of a 16-row tile, 16-byte chunk fq (K step 0) / 4 + fq (K step 1)
    int lb0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), lb1 = lb0 ^ 64;
    const int Wb = p.W * 128;

    CwTile cur, nxt;
    cw_setup<CMN, INB, OUTB>(cur, chunk * 256 + wave * 32, lane, p.M, p.H, p.W);

    u32x4 xr[HALO ? 1 : XD][2];                            // ring forms: conv2 pixel pieces, K-steps kt .. kt + XD - 1
    u32x4 xh[HALO ? HB : 1][2];                            // HALO: fragments of the ten 16-pixel blocks around the tile, K halves 0 / 1
    u32x4 rres[DS ? 1 : RD][2];                            // residual pieces, half-groups h .. h + RD - 1
    V8 xd[2][2];                                           // DS: the block input's fragments (K steps 0 / 1) of the wave's two pixel tiles
    u32x4 wg[(W3G || DS) ? WD : 1][2][2];                  // W3 (W3G) or Wd (DS) pieces from L2: [half-group ring][cout tile][K step]
    // conv2 pixel pieces of K-step kt_ (tap kt_ / 2, channels 32 (kt_ & 1) ..) of tile A_ -> ring slot
#define CW_ISSUE_X(slot_, kt_, A_)                                                                                      \
    {                                                                                                                   \
        const int tp_ = (kt_) >> 1;                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
            int vo_;                                                                                                    \
            if constexpr (INB) {                           /* blocked: pixel pm's 16 bytes of chunk 4 ks + fq */        \
                const int pm_ = A_.xb[j] + (tp_ / 3 - 1) * p.W + (tp_ % 3 - 1);                                         \
                vo_ = (pm_ >> 4) * 2048 + (pm_ & 15) * 16 + (((kt_) & 1) * 4 + fq) * 256;                               \
            } else vo_ = A_.xb[j] + (tp_ / 3 - 1) * Wb + (tp_ % 3 - 1) * 128 + ((kt_) & 1) * 64;                        \
            vo_ = ((A_.mk[j] >> tp_) & 1) ? vo_ : OOB;                                                                  \
            if constexpr (CW_KNOCK & 4) vo_ = OOB;                                                                      \
            xr[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo_, 0, 0));          \
        }                                                                                                               \
    }
    // HALO: the whole halo of a tile in one burst (no load is issued during conv2); blocks before / past the tensor read zeros
#define CW_ISSUE_HALO(A_)                                                                                               \
    {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < HB; ++k)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
                xh[k][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (CW_KNOCK & 4) ? OOB : A_.hb + k * 2048 + ks * 1024, 0, 0)); \
    }
    // HALO: fragment of K-step kt_ for pixel tile j_: pixel 16 j + fr at tap (kh, kw) is pixel 64 + 16 j + (kh - 1) W + (kw - 1) + fr of
    // the halo, i.e. lane fr + d of block k0 (d = that offset & 15, k0 = offset >> 4) or, past the row's end, lane fr + d - 16 of block
    // k0 + 1; then zero where the tap falls outside the pixel's image
#define CW_FRAG(dst_, kt_, j_)                                                                                          \
    {                                                                                                                   \
        const int tp_ = (kt_) >> 1, ks_ = (kt_) & 1, o_ = 64 + 16 * (j_) + (tp_ / 3 - 1) * WW + tp_ % 3 - 1, k0_ = o_ >> 4, d_ = o_ & 15; \
        const bool in_ = (cur.mk[j_] >> tp_) & 1;                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                 \
            const unsigned v_ = cw_row_shift(xh[k0_][ks_][e], xh[k0_ + 1 < HB ? k0_ + 1 : k0_][ks_][e], d_);            \
            dst_[e] = in_ ? v_ : 0u;                                                                                    \
        }                                                                                                               \
    }
#define CW_ISSUE_RES(slot_, h_, A_)                                                                                     \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            rres[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CW_KNOCK & 2) ? OOB : A_.yi[j], (h_) * YH, CW_AUX_RES)); \
    }
    // W3 (W3G) / Wd (DS) from L2, row-permuted [256][64] in the blocked layout [row >> 4][chunk][row & 15][8]: the fragment of rows
    // 32 h + 16 t + fr, channels 32 ks + 8 fq .. is 1 KB of contiguous memory
#define CW_ISSUE_WG(slot_, h_)                                                                                          \
    {                                                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
                wg[slot_][t][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wd, wg_off + t * 2048 + ks * 1024, (h_) * 4096, 0)); \
    }
    const int wg_off = fq * 256 + fr * 16;

    if constexpr (HALO) CW_ISSUE_HALO(cur)
    else {
#pragma unroll
        for (int k = 0; k < XD; ++k) CW_ISSUE_X(k, k, cur);
    }
    if constexpr (!DS) {
#pragma unroll
        for (int d = 0; d < RD; ++d) CW_ISSUE_RES(d, d, cur);
    }

    for (;;) {
        const int chunk_n = chunk + L;
        const bool more = chunk_n < c_end;
        // (the weights in LDS are loop-invariant: without this hipcc hoists all 2 KB of a lane's fragment reads out of the tile loop - into scratch)
        asm volatile("" : "+v"(lb0), "+v"(lb1));

        // DS: the block input x at the wave's own pixels (NHWC, 64 channels); requested here, used from half-group 0 on
        u32x4 xdr[2][2];
        if constexpr (DS) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    xdr[ks][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, cur.xs[j], ks * 64, 0));
        }

        // ---- conv2 3x3: 32 pixels x 64 couts, K = 9 taps x 64 channels; weights from LDS -------------------------------------------
        f32x4 acc2[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xc[2], xn[2];                                // fragments of the current / next K-step
        if constexpr (HALO) { CW_FRAG(xc[0], 0, 0); CW_FRAG(xc[1], 0, 1); }
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j) xc[j] = INB ? xr[0][j] : cw_m2f(j ? slot1 : slot0, lane, foff, xr[0][j]);
            if (XD < NK) CW_ISSUE_X(0, XD, cur);
        }
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 1 < NK) {                             // the next step's fragments while this step's MFMAs run
                if constexpr (HALO) { CW_FRAG(xn[0], kt + 1, 0); CW_FRAG(xn[1], kt + 1, 1); }
                else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) xn[j] = INB ? xr[(kt + 1) % XD][j] : cw_m2f(j ? slot1 : slot0, lane, foff, xr[(kt + 1) % XD][j]);
                    if (kt + 1 + XD < NK) CW_ISSUE_X((kt + 1) % XD, kt + 1 + XD, cur);
                }
            }
            const char *wbase = smem + W2L + (kt >> 1) * 8192 + ((kt & 1) ? lb1 : lb0);
            V8 wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[i] = *reinterpret_cast<const V8 *>(wbase + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc2[i][0] = mfma16<F16>(wb[i], __builtin_bit_cast(V8, xc[0]), acc2[i][0]);
                acc2[i][1] = mfma16<F16>(wb[i], __builtin_bit_cast(V8, xc[1]), acc2[i][1]);
            }
            if (kt + 1 < NK) { xc[0] = xn[0]; xc[1] = xn[1]; }
            __builtin_amdgcn_sched_barrier(0);            // (bounds hipcc's hoisting of later steps' LDS reads: register pressure)
        }
        // t2 = relu(acc2 + b2) -> 16 bit: tile pair q of pixel tile j IS conv3's B fragment of K step q
        u32x4 t2[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 bA = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq) * 4);
            const float4 bB = *reinterpret_cast<const float4 *>(smem + B2L + (32 * q + 8 * fq + 4) * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 lo = acc2[2 * q][j], hi = acc2[2 * q + 1][j];
                const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = cw_pack2<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                t2[q][j] = o;
            }
        }

        // ---- the next tile's addresses and its conv2 pixels: requested before this tile's stores are issued -----------------------
        cw_setup<CMN, INB, OUTB>(nxt, more ? chunk_n * 256 + wave * 32 : ((p.M + 31) & ~31), lane, p.M, p.H, p.W);
        if (!more) nxt.hb = OOB;
        if constexpr (HALO) CW_ISSUE_HALO(nxt)
        else {
#pragma unroll
            for (int k = 0; k < XD; ++k) CW_ISSUE_X(k, k, nxt);
        }
        if constexpr (W3G || DS) {
#pragma unroll
            for (int d = 0; d < WD; ++d) CW_ISSUE_WG(d, d);
        }
        if constexpr (DS) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) xd[ks][j] = __builtin_bit_cast(V8, cw_m2f(j ? slot1 : slot0, lane, foff, xdr[ks][j]));
        }

        // ---- conv3 (+ residual / + Wd . x) and conv1', one 32-cout half-group at a time --------------------------------------------
        f32x4 acc1[CMN ? TN1 : 1][2];
        if constexpr (CMN > 0) {
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        u32x4 oe[2];                                       // NHWC out: y of the even half-group, held until its odd partner completes the 128-byte line
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // this half-group's pieces -> fragment layout, their ring slots refilled (this tile's later half-groups, then the next tile's first)
            u32x4 rp[2], wp[2][2];
            if constexpr (!DS) {
#pragma unroll
                for (int j = 0; j < 2; ++j) rp[j] = INB ? rres[h % RD][j] : cw_m2f(j ? slot1 : slot0, lane, foff, rres[h % RD][j]);
                if (h + RD < NH) CW_ISSUE_RES(h % RD, h + RD, cur)
                else CW_ISSUE_RES(h % RD, h + RD - NH, nxt)
            }
            if constexpr (W3G || DS) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) wp[t][ks] = wg[h % WD][t][ks];
                if (h + WD < NH) CW_ISSUE_WG(h % WD, h + WD);
            }
            f32x4 acc3[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc3[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 wb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if constexpr (W3G) wb[t] = __builtin_bit_cast(V8, wp[t][ks]);
                    else wb[t] = *reinterpret_cast<const V8 *>(smem + W3L + h * 4096 + t * 2048 + (ks ? lb1 : lb0));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc3[t][j] = mfma16<F16>(wb[t], __builtin_bit_cast(V8, t2[ks][j]), acc3[t][j]);
            }
            if constexpr (DS) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc3[t][j] = mfma16<F16>(__builtin_bit_cast(V8, wp[t][ks]), xd[ks][j], acc3[t][j]);
            }
            // y = relu(acc3 + b3 + residual): 8 consecutive couts per lane = conv1''s B fragment of K step h
            const float4 bA = *reinterpret_cast<const float4 *>(smem + B3L + (32 * h + 8 * fq) * 4);
            const float4 bB = *reinterpret_cast<const float4 *>(smem + B3L + (32 * h + 8 * fq + 4) * 4);
            u32x4 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 lo = acc3[0][j], hi = acc3[1][j];
                const float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
                if constexpr (DS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[j][e] = cw_pack2<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                } else {
                    const u32x4 r = rp[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = fmaxf(v[2 * e] + from_h<F16>((u16)(r[e] & 0xffffu)), 0.f);
                        const float v1 = fmaxf(v[2 * e + 1] + from_h<F16>((u16)(r[e] >> 16)), 0.f);
                        o[j][e] = cw_pack2<F16>(v0, v1);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (OUTB) cw_store(o[j], rs_y, cur.yo[j], h * 1024, p.stride == 77);
                else if (h & 1) {                          // NHWC: the pair (h - 1, h) = 128 bytes per pixel leaves as full lines
                    u32x4 lo, hi;
                    cw_f2m_pair(slot0, lane, oe[j], o[j], lo, hi);
                    cw_store(lo, rs_y, cur.yo[j], (h >> 1) * 128, p.stride == 77);
                    cw_store(hi, rs_y, cur.yo[j], (h >> 1) * 128 + 8 * 512, p.stride == 77);
                } else oe[j] = o[j];
            }
            if constexpr (CMN > 0) {
#pragma unroll
                for (int i = 0; i < TN1; ++i) {
                    const V8 wb = *reinterpret_cast<const V8 *>(smem + W1L + (h >> 1) * (CMN * 128) + i * 2048 + ((h & 1) ? lb1 : lb0));
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16<F16>(wb, __builtin_bit_cast(V8, o[j]), acc1[i][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- t1' = relu(acc1 + b1'): tile pair q = 8 consecutive couts per lane ----------------------------------------------------
        if constexpr (CMN > 0) {
            u32x4 te[2];
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
                    for (int e = 0; e < 4; ++e) o[e] = cw_pack2<F16>(fmaxf(v[2 * e], 0.f), fmaxf(v[2 * e + 1], 0.f));
                    if constexpr (OUTB) cw_store<CW_AUX_T1>(o, rs_t, cur.to[j], q * 1024, p.stride == 77);
                    else if (q & 1) {
                        u32x4 l0, l1;
                        cw_f2m_pair(slot0, lane, te[j], o, l0, l1);
                        cw_store<CW_AUX_T1>(l0, rs_t, cur.to[j], (q >> 1) * 128, p.stride == 77);
                        cw_store<CW_AUX_T1>(l1, rs_t, cur.to[j], (q >> 1) * 128 + 8 * (CMN * 2), p.stride == 77);
                    } else te[j] = o;
                }
            }
        }
        if (!more) break;
        cur = nxt;
        chunk = chunk_n;
    }
#undef CW_ISSUE_X
#undef CW_ISSUE_HALO
#undef CW_FRAG
#undef CW_ISSUE_RES
#undef CW_ISSUE_WG
}

static int cw_num_cus() {
    static int v = 0;
    if (!v) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
        else v = 256;
    }
    return v;
}

template <int CMN, bool F16, bool DS, bool W3G, bool INB, bool OUTB, bool HALO, int XD, int RD, int WD>
static pvr_status launch_cw_one(ChainP &p, hipStream_t stream) {
    const size_t lds = (size_t)(W3G ? 73728 : 73728 + 32768) + (size_t)CMN * 512 + 256 + 1024 + 512 + 8 * 2048;
    static DeviceOnce attr_done;                           // per device; NOT per launch: the call costs host time that shows up as a gap in front of the kernel
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)chain_wave_kernel<CMN, F16, DS, W3G, INB, OUTB, HALO, XD, RD, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    const int nch = (p.M + 255) / 256;
    int grid = cw_num_cus() & ~7;                          // one persistent block per CU; a multiple of 8 (blocks b and b + 8 share an XCD)
    if (grid < 8) grid = 8;
    const int need = ((nch + 7) / 8) * 8;                  // small launches: one chunk per block
    if (grid > need) grid = need;
    hipLaunchKernelGGL((chain_wave_kernel<CMN, F16, DS, W3G, INB, OUTB, HALO, XD, RD, WD>), dim3(grid), dim3(512), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// PVR_CHAIN_WAVE_HALO=0: blocked inputs through the per-K-step load ring instead of the halo registers (A/B runs)
static int cw_halo() {                                     // (read per call: plans built under different settings coexist in the tests)
    const char *e = getenv("PVR_CHAIN_WAVE_HALO");
    return e ? atoi(e) : 1;
}

template <bool F16>
static pvr_status launch_cw_dt(ChainP &p, int cmn, hipStream_t stream) {
    const bool ib = p.in_blk, ob = p.out_blk, halo = ib && p.W == 56 && cw_halo();
    if (p.xds) {                                           // layer1 block 0: x (from the stem) is NHWC; t1 is blocked when conv1's launch wrote it so
        if (cmn == 64 && halo && ob) return launch_cw_one<64, F16, true, false, true, true, true, 1, 1, 2>(p, stream);
        if (cmn == 64 && !ib) return ob ? launch_cw_one<64, F16, true, false, false, true, false, 4, 1, 2>(p, stream)
                                        : launch_cw_one<64, F16, true, false, false, false, false, 4, 1, 2>(p, stream);
    } else if (cmn == 64) {
        if (ib && ob) return halo ? launch_cw_one<64, F16, false, false, true, true, true, 1, 4, 1>(p, stream)
                                  : launch_cw_one<64, F16, false, false, true, true, false, 4, 4, 1>(p, stream);
        if (!ib && !ob) return launch_cw_one<64, F16, false, false, false, false, false, 4, 4, 1>(p, stream);
        if (!ib && ob) return launch_cw_one<64, F16, false, false, false, true, false, 4, 4, 1>(p, stream);
        return halo ? launch_cw_one<64, F16, false, false, true, false, true, 1, 4, 1>(p, stream)
                    : launch_cw_one<64, F16, false, false, true, false, false, 4, 4, 1>(p, stream);
    } else if (cmn == 128 && !ob) {                        // layer1's last block: t1' feeds layer2's block form (NHWC)
        if (halo) return launch_cw_one<128, F16, false, true, true, false, true, 1, 1, 1>(p, stream);   // 254 VGPRs: residual / W3 rings of depth 1 make room for the halo
        return ib ? launch_cw_one<128, F16, false, true, true, false, false, 3, 4, 2>(p, stream)
                  : launch_cw_one<128, F16, false, true, false, false, false, 3, 4, 2>(p, stream);
    } else if (cmn == 0 && !ob) {
        if (ib) return halo ? launch_cw_one<0, F16, false, false, true, false, true, 1, 4, 1>(p, stream)
                            : launch_cw_one<0, F16, false, false, true, false, false, 4, 4, 1>(p, stream);
        return launch_cw_one<0, F16, false, false, false, false, false, 4, 4, 1>(p, stream);
    }
    set_error("bottleneck chain (wave form): no instance for next Cm=%d%s, blocked in/out %d/%d", cmn, p.xds ? " with downsample" : "", (int)ib, (int)ob);
    return PVR_ERR_INVALID;
}

// Cmn = 128 (layer1's last block, t1' for layer2): as a launch of its own this instance is no faster than the block form (0.299 vs 0.279 ms
// inside the forward: W3 from L2, NHWC outputs, 228 VGPRs), but it takes its inputs in the blocked layout, which is what lets the tail in
// front of it run all-blocked (0.205 instead of 0.238 ms; with NHWC outputs that one takes 0.261).  PVR_CHAIN_WAVE_128=0 for the A/B.
bool chain_wave_supported(int cm, int cmn, int stride, bool ds) {
    if (cm != 64 || stride != 1) return false;
    if (ds) return cmn == 64;
    if (cmn == 128) { const char *e = getenv("PVR_CHAIN_WAVE_128"); return !e || atoi(e) != 0; }
    return cmn == 0 || cmn == 64;
}

// can the tensors between two consecutive wave-form launches (y = the next residual, t1' = the next conv2 input) use the blocked layout?
bool chain_wave_blocked_ok(int cmn_first, int h, int w) { return cmn_first == 64 && (h * w) % 32 == 0; }
bool chain_wave_halo_enabled() { return cw_halo() != 0; }

pvr_status launch_chain_wave(ChainP &p, int cmn, int dtype, hipStream_t stream) {
    PVR_REQUIRE((int64_t)(p.M + 64) * 512 < 0x7ffffff0ll, "bottleneck chain (wave form): operand larger than 2 GiB (use a smaller chunk)");
    PVR_REQUIRE(p.xds ? p.wdsb != nullptr : (cmn != 128 || p.w3b != nullptr), "bottleneck chain (wave form): the blocked copy of W3 / Wd is missing");
    PVR_REQUIRE(!(p.in_blk || p.out_blk) || p.M % 32 == 0, "bottleneck chain (wave form): the blocked layout needs a multiple of 32 pixels");
    return dtype == PVR_F16 ? launch_cw_dt<true>(p, cmn, stream) : launch_cw_dt<false>(p, cmn, stream);
}

}  // namespace pvr
