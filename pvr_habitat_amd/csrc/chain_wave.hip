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
//     persistent, every wave walks its own list of 32-pixel tiles.  (Cmn = 128: W1' is 64 KB, so W3's fragments are read from L2 in
//     MFMA layout instead; DS: Wd's fragments likewise.)
//   * conv2's pixel fragments are 16-byte global loads in operand layout (a lane reads channels 8q..8q+7 of pixel m + tap shift;
//     taps outside the image read past the buffer: the range check returns zeros).  The nine taps re-read each line from L1 / L2;
//     HBM sees t1 once.  Loads run XD K-steps ahead of their MFMAs through a register ring and the next tile's first steps and
//     residual half-groups are requested before the current tile's stores are issued (loads and stores share one in-order vmcnt queue).
//
// Eight independent waves per CU, each with ~25 KB of loads and stores in flight, keep HBM busy while other waves compute; MFMA work
// is ~25 % of the launch's HBM time, so the launch is a stream with arithmetic underneath.  Numerics: the same rounding points and the
// same K order per accumulator as bottleneck_chain.hip and the unfused launches - bit-identical outputs
// (tests/test_gpu_encoder.py::test_fused_bottleneck_chain_is_bit_identical, ::test_chain_wave_equals_block_form).
#include "chain_params.h"

namespace pvr {

int chain_row_source(int row);

__device__ __forceinline__ int cw_row_source(int row) { return (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

#ifndef CW_KNOCK
#define CW_KNOCK 0      // timing experiments (scripts/chain_wave_bench.hip -DCW_KNOCK=bits): 1 no y / t1' stores, 2 residual loads out of range (zeros, no
                        // traffic), 4 conv2 pixel loads out of range
#endif
// 16-byte buffer store, byte offset in voffset + immediate (never soffset: bottleneck_chain.hip, store_b128_imm)
__device__ __forceinline__ void cw_store(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int imm, int never = 0) {
    if constexpr (CW_KNOCK & 1) { if (never) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, 0); }
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + imm, 0, 0);
}

struct CwTile {
    int xb[2];      // byte offset of (pixel, 16-byte chunk lc) in t1 (128-byte rows); also x's offset in the DS form
    int mk[2];      // 9-bit "tap inside the image" mask of the pixel (0 for pixels past M)
    int yo[2];      // byte offset of (pixel, 16-byte chunk lc) in y / the residual tensor (512-byte rows)
};

// Addresses of a wave's 32-pixel tile in the MEMORY lane layout: lane l handles pixel lp = l >> 2 of a 16-pixel MFMA tile and the
// 16-byte chunk lc = l & 3 of a 64-byte piece of its row.  A quarter wave (the unit the texture-address path works in) then touches
// 4 rows x 64 B instead of the 16 rows x 16 B of the MFMA fragment layout (lane = 16 * chunk + pixel): 4x fewer row touches per
// wave instruction.  cw_perm_in / cw_perm_out (ds_bpermute: the LDS crossbar, no LDS memory) move a loaded register into fragment
// layout and a result register back.
__device__ __forceinline__ void cw_setup(CwTile &a, int m0, int lp, int lc, int M, int H, int W) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 16 * j + lp;
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
        a.xb[j] = m * 128 + lc * 16;
        a.mk[j] = mask;
        a.yo[j] = m * 512 + lc * 16;
    }
}

__device__ __forceinline__ u32x4 cw_perm(int addr, u32x4 v) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v[e]);
    return r;
}

// XD: conv2 K-steps of pixel pieces in flight (8 VGPRs each); RD: residual half-groups in flight (8 VGPRs each);
// WD: half-groups of W3 / Wd pieces in flight when they come from L2 (16 VGPRs each)
template <int CMN, bool F16, bool DS, bool W3G, int XD, int RD, int WD, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW / 4) void chain_wave_kernel(ChainP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int NH = 8, NK = 18;                         // half-groups of 32 couts; conv2 K-steps of 32 channels (9 taps x 2)
    constexpr int TN1 = CMN / 16;
    constexpr int W2L = 0, W3L = 73728, W1L = W3G ? 73728 : 73728 + 32768;
    constexpr int B2L = W1L + CMN * 512, B3L = B2L + 256, B1L = B3L + 1024;
    constexpr int OOB = 0x7ffffff0;
    static_assert(XD >= 1 && XD <= NK && RD >= 1 && RD <= NH && NH % RD == 0 && WD >= 1 && WD <= NH, "prefetch depths (the residual ring must close over a tile)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int lp = lane >> 2, lc = lane & 3;               // memory lane layout: pixel (row) of the 16-row tile, 16-byte chunk of the 64-byte piece
    const int pin = (fr * 4 + fq) * 4;                     // ds_bpermute source of fragment lane (fr, fq): memory lane 4 fr + fq
    const int pout = (lc * 16 + lp) * 4;                   // ... and of memory lane (lp, lc): fragment lane 16 lc + lp
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w2), 0, p.w2_bytes, 0x00020000);
    const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w3), 0, p.w3_bytes, 0x00020000);
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w1n), 0, p.w1n_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.xds : p.res), 0, DS ? p.xds_bytes : p.y_bytes, 0x00020000);
    const auto rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(DS ? p.wds : p.w3), 0, DS ? p.wds_bytes : p.w3_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);

    // ---- prologue: the block's weights and biases -> LDS ([rows][64] 16-bit tiles, 128-byte rows, chunk ^= (row >> 1) & 7) -------
    constexpr int NT = NW * 64;
#pragma unroll
    for (int q = 0; q < (4608 + NT - 1) / NT; ++q) {        // W2: 9 taps x 64 rows x 8 chunks; LDS row r holds cout cw_row_source(r)
        const int idx = tid + NT * q, tap = idx >> 9, r = (idx >> 3) & 63, c = idx & 7;
        if (idx < 4608) {
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, (cw_row_source(r) * 576 + tap * 64 + c * 8) * 2, 0, 0));
            *reinterpret_cast<u32x4 *>(smem + W2L + tap * 8192 + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
        }
    }
    if constexpr (!W3G) {                                  // W3 (rows already permuted by the host): [256][64]
#pragma unroll
        for (int q = 0; q < (2048 + NT - 1) / NT; ++q) {
            const int idx = tid + NT * q, r = idx >> 3, c = idx & 7;
            if (idx < 2048) {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, idx * 16, 0, 0));
                *reinterpret_cast<u32x4 *>(smem + W3L + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
            }
        }
    }
    if constexpr (CMN > 0) {                               // W1' (rows permuted): [CMN][256] -> four K groups of [CMN][64]
#pragma unroll
        for (int q = 0; q < (CMN * 32 + NT - 1) / NT; ++q) {
            const int idx = tid + NT * q, r = idx >> 5, c32 = idx & 31;
            if (idx < CMN * 32) {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, idx * 16, 0, 0));
                *reinterpret_cast<u32x4 *>(smem + W1L + (c32 >> 3) * (CMN * 128) + r * 128 + (((c32 & 7) ^ ((r >> 1) & 7)) << 4)) = v;
            }
        }
    }
    if (tid < 64) *reinterpret_cast<float *>(smem + B2L + tid * 4) = p.b2[tid];
    if (tid < 256) *reinterpret_cast<float *>(smem + B3L + tid * 4) = p.b3[tid];
    if constexpr (CMN > 0) { if (tid < CMN) *reinterpret_cast<float *>(smem + B1L + tid * 4) = p.b1n[tid]; }
    __syncthreads();                                       // the only barrier of the kernel

    // ---- this wave's tiles: chunk = NW x 32 consecutive pixels; an XCD's blocks walk a contiguous run of chunks side by
    // side, so the rows two neighbouring tiles both read meet in that XCD's L2
    constexpr int CH = NW * 32;                            // pixels per chunk
    const int nch = (p.M + CH - 1) / CH, cx = (nch + 7) >> 3, L = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, c_end = min((xcd + 1) * cx, nch);
    int chunk = xcd * cx + (blockIdx.x >> 3);
    if (chunk >= c_end) return;

    // per-lane LDS fragment bases: row fr of a 16-row tile, 16-byte chunk fq (K step 0) / 4 + fq (K step 1)
    int lb0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), lb1 = lb0 ^ 64;
    const int Wb = p.W * 128;

    CwTile cur, nxt;
    cw_setup(cur, chunk * CH + wave * 32, lp, lc, p.M, p.H, p.W);

    u32x4 xr[XD][2];                                       // conv2 pixel pieces (memory layout), K-steps kt .. kt + XD - 1
    u32x4 rres[DS ? 1 : RD][2];                            // residual pieces, half-groups h .. h + RD - 1
    V8 xd[2][2];                                           // DS: the block input's fragments (K steps 0 / 1) of the wave's two pixel tiles
    u32x4 wg[(W3G || DS) ? WD : 1][2][2];                  // W3 (W3G) or Wd (DS) pieces from L2: [half-group ring][cout tile][K step]
#define CW_ISSUE_X(slot_, kt_, A_)                                                                                      \
    {                                                                                                                   \
        const int tp_ = (kt_) >> 1, sh_ = (tp_ / 3 - 1) * Wb + (tp_ % 3 - 1) * 128 + ((kt_) & 1) * 64;                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
            int vo_ = ((A_.mk[j] >> tp_) & 1) ? A_.xb[j] + sh_ : OOB;                                                   \
            if constexpr (CW_KNOCK & 4) vo_ = OOB;                                                                      \
            xr[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo_, 0, 0));          \
        }                                                                                                               \
    }
#define CW_ISSUE_RES(slot_, h_, A_)                                                                                     \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            rres[slot_][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (CW_KNOCK & 2) ? OOB : A_.yo[j], (h_) * 64, 0)); \
    }
    // weight rows 32 h + 16 t + lp, channels 32 ks + 8 lc .. (128-byte rows: W3 [256][64] / Wd [256][64], both row-permuted)
#define CW_ISSUE_WG(slot_, h_)                                                                                          \
    {                                                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
                wg[slot_][t][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wd, wg_off + t * 2048 + ks * 64, (h_) * 4096, 0)); \
    }
    const int wg_off = lp * 128 + lc * 16;

#pragma unroll
    for (int k = 0; k < XD; ++k) CW_ISSUE_X(k, k, cur);
    if constexpr (!DS) {
#pragma unroll
        for (int d = 0; d < RD; ++d) CW_ISSUE_RES(d, d, cur);
    }

    for (;;) {
        const int chunk_n = chunk + L;
        const bool more = chunk_n < c_end;
        // (the weights in LDS are loop-invariant: without this hipcc hoists all 2 KB of a lane's fragment reads out of the tile loop - into scratch)
        asm volatile("" : "+v"(lb0), "+v"(lb1));

        // DS: the block input x at the wave's own pixels (the centre-tap pattern on the 64-channel tensor x); used from half-group 0 on
        u32x4 xdr[2][2];
        if constexpr (DS) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    xdr[ks][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, cur.xb[j], ks * 64, 0));
        }

        // ---- conv2 3x3: 32 pixels x 64 couts, K = 9 taps x 64 channels; weights from LDS, pixels from the register ring ----------
        f32x4 acc2[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xc[2], xn[2];                                // fragments of the current / next K-step
#pragma unroll
        for (int j = 0; j < 2; ++j) xc[j] = cw_perm(pin, xr[0][j]);
        if (XD < NK) CW_ISSUE_X(0, XD, cur);
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 1 < NK) {                             // the next step's pieces -> fragment layout while this step's MFMAs run
#pragma unroll
                for (int j = 0; j < 2; ++j) xn[j] = cw_perm(pin, xr[(kt + 1) % XD][j]);
                if (kt + 1 + XD < NK) CW_ISSUE_X((kt + 1) % XD, kt + 1 + XD, cur);
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
                    o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                t2[q][j] = o;
            }
        }

        // ---- the next tile's addresses and its first conv2 pieces: requested before this tile's stores are issued -----------------
        cw_setup(nxt, more ? chunk_n * CH + wave * 32 : p.M, lp, lc, p.M, p.H, p.W);
#pragma unroll
        for (int k = 0; k < XD; ++k) CW_ISSUE_X(k, k, nxt);
        if constexpr (W3G || DS) {
#pragma unroll
            for (int d = 0; d < WD; ++d) CW_ISSUE_WG(d, d);
        }
        if constexpr (DS) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) xd[ks][j] = __builtin_bit_cast(V8, cw_perm(pin, xdr[ks][j]));
        }

        // ---- conv3 (+ residual / + Wd . x) and conv1', one 32-cout half-group at a time --------------------------------------------
        f32x4 acc1[CMN ? TN1 : 1][2];
        if constexpr (CMN > 0) {
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // this half-group's pieces -> fragment layout, their ring slots refilled (this tile's later half-groups, then the next tile's first)
            u32x4 rp[2], wp[2][2];
            if constexpr (!DS) {
#pragma unroll
                for (int j = 0; j < 2; ++j) rp[j] = cw_perm(pin, rres[h % RD][j]);
                if (h + RD < NH) CW_ISSUE_RES(h % RD, h + RD, cur)
                else CW_ISSUE_RES(h % RD, h + RD - NH, nxt)
            }
            if constexpr (W3G || DS) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) wp[t][ks] = cw_perm(pin, wg[h % WD][t][ks]);
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
            // y = relu(acc3 + b3 + residual): 8 consecutive couts per lane = conv1''s B fragment of K step h; stored through the memory layout
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
                        o[j][e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                } else {
                    const u32x4 r = rp[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = fmaxf(v[2 * e] + from_h<F16>((u16)(r[e] & 0xffffu)), 0.f);
                        const float v1 = fmaxf(v[2 * e + 1] + from_h<F16>((u16)(r[e] >> 16)), 0.f);
                        o[j][e] = (unsigned)to_h<F16>(v0) | ((unsigned)to_h<F16>(v1) << 16);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) cw_store(cw_perm(pout, o[j]), rs_y, cur.yo[j], h * 64, p.stride == 77);
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
                    for (int e = 0; e < 4; ++e)
                        o[e] = (unsigned)to_h<F16>(fmaxf(v[2 * e], 0.f)) | ((unsigned)to_h<F16>(fmaxf(v[2 * e + 1], 0.f)) << 16);
                    // byte offset of (pixel, chunk lc) in t1': yo = m * 512 + lc * 16  ->  m * 2 CMN + lc * 16
                    const int to = ((cur.yo[j] - lc * 16) >> 9) * (CMN * 2) + lc * 16;
                    cw_store(cw_perm(pout, o), rs_t, to, q * 64, p.stride == 77);
                }
            }
        }
        if (!more) break;
        cur = nxt;
        chunk = chunk_n;
    }
#undef CW_ISSUE_X
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

template <int CMN, bool F16, bool DS, bool W3G, int XD, int RD, int WD, int NW = 8>
static pvr_status launch_cw_one(ChainP &p, hipStream_t stream) {
    const size_t lds = (size_t)(W3G ? 73728 : 73728 + 32768) + (size_t)CMN * 512 + 256 + 1024 + 512;
    PVR_HIP_TRY(hipFuncSetAttribute((const void *)chain_wave_kernel<CMN, F16, DS, W3G, XD, RD, WD, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nch = (p.M + NW * 32 - 1) / (NW * 32);
    int grid = cw_num_cus() & ~7;                          // one persistent block per CU; a multiple of 8 (blocks b and b + 8 share an XCD)
    if (grid < 8) grid = 8;
    const int need = ((nch + 7) / 8) * 8;                  // small launches: one chunk per block
    if (grid > need) grid = need;
    hipLaunchKernelGGL((chain_wave_kernel<CMN, F16, DS, W3G, XD, RD, WD, NW>), dim3(grid), dim3(NW * 64), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// PVR_CHAIN_WAVE_NW=12: twelve waves per CU (three per SIMD, <= 168 VGPRs) instead of eight (A/B runs)
static int cw_nw() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_CHAIN_WAVE_NW"); v = e ? atoi(e) : 8; }
    return v;
}

template <bool F16>
static pvr_status launch_cw_dt(ChainP &p, int cmn, hipStream_t stream) {
    if (p.xds) {
        if (cmn == 64) return launch_cw_one<64, F16, true, false, 4, 1, 1>(p, stream);
    } else {
        if (cmn == 64) return cw_nw() == 12 ? launch_cw_one<64, F16, false, false, 4, 4, 1, 12>(p, stream) : launch_cw_one<64, F16, false, false, 4, 4, 1>(p, stream);
        if (cmn == 128) return launch_cw_one<128, F16, false, true, 2, 2, 1>(p, stream);
        if (cmn == 0) return cw_nw() == 12 ? launch_cw_one<0, F16, false, false, 4, 4, 1, 12>(p, stream) : launch_cw_one<0, F16, false, false, 4, 4, 1>(p, stream);
    }
    set_error("bottleneck chain (wave form): no instance for next Cm=%d%s", cmn, p.xds ? " with downsample" : "");
    return PVR_ERR_INVALID;
}

bool chain_wave_supported(int cm, int cmn, int stride, bool ds) {
    return cm == 64 && stride == 1 && (ds ? cmn == 64 : (cmn == 0 || cmn == 64 || cmn == 128));
}

pvr_status launch_chain_wave(ChainP &p, int cmn, int dtype, hipStream_t stream) {
    PVR_REQUIRE((int64_t)(p.M + 64) * 512 < 0x7ffffff0ll, "bottleneck chain (wave form): operand larger than 2 GiB (use a smaller chunk)");
    return dtype == PVR_F16 ? launch_cw_dt<true>(p, cmn, stream) : launch_cw_dt<false>(p, cmn, stream);
}

}  // namespace pvr
