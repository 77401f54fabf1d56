// fp32 convolutions on the 16-bit matrix pipe: every fp32 operand as an exact (hi, lo) pair of f16 values.
//
// Where: the parity plan of the compressed PVRs (`*_l3` / `*_l4`, f16 storage: src/vision_models/moco.py:29-113, resnet.py:6-83 through
// src/embeddings.py:195-280) keeps its last trunk stage and the compression head in fp32 - these variants have no average pool, so the storage
// rounding of that stage reaches the output element by element (encoder.hip::build_resnet50).  Until round 5 that stage ran on the f32-input
// MFMA (conv_f32.hip), 1/16 of the 16-bit rate: the 5-crop uber PVR of BASELINE configs[4] ran at 0.115 of the 16-bit peak in its compliant plan.
//
// How: a = a_hi + 2^-11 a_lo with a_hi = f16(a) and a_lo = f16(2^11 (a - a_hi)) (the difference is exact in fp32; the 2^11 keeps the low
// part - and the low part of weights of magnitude 1e-2 - out of f16's subnormal range), likewise w.  Then
//      a w = a_hi w_hi + 2^-11 (a_lo w_hi + a_hi w_lo) + 2^-22 a_lo w_lo
// and the last term is below fp32's own rounding: three 16x16x32 MFMAs per fragment pair, the first into one fp32 accumulator, the two
// cross terms into a second one that is scaled once in the epilogue.  Products of 11-bit significands are exact in the fp32 accumulation,
// so the result differs from an fp32 dot product by the dropped 2^-22 term and the rounding of a_lo / w_lo (2^-22 relative): ~4e-7 per
// term, against 6e-8 for fp32 and 5e-4 for f16 storage.  3 MFMAs at 16x the f32-input rate.
//
// Kernel: implicit GEMM, BM pixels x BN couts per 512-thread workgroup, K steps of 32 channels of one filter tap.
//   * weights: split and laid out once (launch_split16_pack) as MFMA A fragments [cout >> 4][k >> 5][hi, lo][k chunk][cout & 15][8]:
//     a wave owns 16 couts and reads its two 1 KB fragments per step straight from L2 into registers, one step ahead - every fragment is
//     fetched once per workgroup and never touches LDS
//   * pixels: fp32 NHWC rows (128 B per pixel and step) are loaded one step ahead, split in registers and written to LDS as
//     [hi, lo][k chunk][pixel ^ (2 * chunk)][8] - 8-byte stores and 16-byte fragment reads are both conflict-free in that image - double
//     buffered, one barrier per step
//   * a wave: 16 couts x BM / NWP pixels; per pixel tile two ds_read_b128 and three MFMAs
#include "common.h"

namespace pvr {

struct ConvS {
    const float *in, *bias, *res;
    const u16 *w;
    float *out;
    // pair form (two convolutions of the SAME input in one launch: the compression head's conv1 and downsample): couts < n1 go to `out` (row stride n1, with
    // the activation), couts >= n1 to out2 (row stride Cout - n1, no activation); out16: 16-bit output (f16, row stride Cout) instead of fp32
    float *out2 = nullptr;
    int n1 = 0;
    u16 *out16 = nullptr;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, M, relu, n_ctiles, nsteps;
    unsigned in_bytes, w_bytes;
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// TERMS = 3: the exact split product above.  TERMS = 1: a_hi w_hi only - an ordinary 16-bit convolution that reads its fp32 input itself (the rounding of the
// operand to f16 happens in the staging pass: the fp32 -> 16-bit copy launch of the residual stream is gone) with the 16-bit plan's own weights (w_hi = f16(w)).
template <int BM, int BN, int TERMS = 3>
__global__ __launch_bounds__(512) void conv_split16_kernel(ConvS p) {
    constexpr int NWC = BN / 16, NWP = 8 / NWC, WP = BM / NWP, JT = WP / 16, RPT = BM / 64;
    static_assert(NWC * NWP == 8 && WP % 16 == 0 && RPT >= 1, "tile shape");
    __shared__ __attribute__((aligned(16))) u16 sm[2][2][BM * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int wc = wave % NWC, wp = wave / NWC;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (swz / p.n_ctiles) * BM, n0 = (swz % p.n_ctiles) * BN;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    // pixel rows this thread loads: row (tid >> 3) + 64 i, 16-byte piece tid & 7 of the step's 128 bytes
    const int piece = tid & 7, chunk = piece >> 1, half = piece & 1;
    int a_off[RPT], a_mask[RPT], s_off[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = (tid >> 3) + 64 * i, m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + piece * 4) * 4;
        int hb = 0, wb = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && t3 < p.KH && (unsigned)(hi0 + t3) < (unsigned)p.H) << t3;
            wb |= (int)(t3 < p.KW && (unsigned)(wi0 + t3) < (unsigned)p.W) << t3;
        }
        int mask = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb << (t3 * p.KW)) : 0;
        a_mask[i] = mask;
        s_off[i] = (chunk * BM + (row ^ (chunk << 1))) * 8 + half * 4;          // u16 elements
    }
    // weight fragments of this wave: [cout block][step][hi, lo][512 elements], lane-linear 16 bytes
    const int cb = (n0 >> 4) + wc;
    const int w_off = cb * p.nsteps * 2048 + lane * 16;                          // bytes; + step * 2048 (+ 1024 for lo)
    const int cpt = p.Cin / 32;
    int kh = 0, kw = 0, cs = 0, tap = 0;
    f32x4 ra[RPT];
    f16x8 a_hi, a_lo = {}, n_hi, n_lo = {};          // (the lo halves exist with TERMS == 3 only)
#define PVR_S_LOAD(ks_)                                                                                          \
    {                                                                                                            \
        const int tap_off = ((kh * p.W + kw) * p.Cin + cs * 32) * 4;                                             \
        _Pragma("unroll") for (int i = 0; i < RPT; ++i) {                                                        \
            const int vo = ((a_mask[i] >> tap) & 1) ? a_off[i] + tap_off : OOB;                                  \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0));           \
        }                                                                                                        \
        n_hi = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off, (ks_) * 2048, 0));   \
        if constexpr (TERMS == 3) n_lo = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off + 1024, (ks_) * 2048, 0)); \
        if (++cs == cpt) { cs = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }                                  \
    }
#define PVR_S_STORE(buf_)                                                                                        \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < RPT; ++i) {                                                        \
            const unsigned h01 = pack2_h<true>(ra[i][0], ra[i][1]), h23 = pack2_h<true>(ra[i][2], ra[i][3]);     \
            const pk_f16x2 q01 = __builtin_bit_cast(pk_f16x2, h01), q23 = __builtin_bit_cast(pk_f16x2, h23);     \
            *reinterpret_cast<uint2 *>(&sm[buf_][0][s_off[i]]) = make_uint2(h01, h23);                           \
            if constexpr (TERMS == 3) {                                                                          \
                const unsigned l01 = pack2_h<true>((ra[i][0] - (float)q01[0]) * 2048.f, (ra[i][1] - (float)q01[1]) * 2048.f); \
                const unsigned l23 = pack2_h<true>((ra[i][2] - (float)q23[0]) * 2048.f, (ra[i][3] - (float)q23[1]) * 2048.f); \
                *reinterpret_cast<uint2 *>(&sm[buf_][1][s_off[i]]) = make_uint2(l01, l23);                       \
            }                                                                                                    \
        }                                                                                                        \
    }
    f32x4 acc0[JT], acc1[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    PVR_S_LOAD(0);
    PVR_S_STORE(0);
    a_hi = n_hi; a_lo = n_lo;
    __syncthreads();
    // fragment read: pixel tile j of this wave, k chunk fq
    const int r_off = (fq * BM + ((wp * WP + fr) ^ (fq << 1))) * 8;                // + j * 16 * 8 (the XOR touches bits 1..2 only)
    int cur = 0;
    for (int ks = 0; ks < p.nsteps; ++ks) {
        const bool more = ks + 1 < p.nsteps;
        if (more) PVR_S_LOAD(ks + 1);
        const u16 *Bh = sm[cur][0] + r_off, *Bl = sm[cur][1] + r_off;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(Bh + j * 128);
            acc0[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bh, acc0[j], 0, 0, 0);
            if constexpr (TERMS == 3) {
                const f16x8 bl = *reinterpret_cast<const f16x8 *>(Bl + j * 128);
                acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bl, acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bh, acc1[j], 0, 0, 0);
            }
        }
        if (more) { PVR_S_STORE(cur ^ 1); a_hi = n_hi; a_lo = n_lo; }
        __syncthreads();
        cur ^= 1;
    }
#undef PVR_S_LOAD
#undef PVR_S_STORE
    // D: row = cout 4 fq + r, column = pixel fr: four consecutive couts of one pixel per lane
    const int co = n0 + wc * 16 + fq * 4;
    if (co >= p.Cout) return;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + co);
    // pair form: this wave's 16 couts belong to one of the two outputs (n1 is a multiple of 16)
    const bool second = p.out2 && co >= p.n1;
    float *const obase = second ? p.out2 : p.out;
    const int ostride = p.out2 ? (second ? p.Cout - p.n1 : p.n1) : p.Cout, oc = second ? co - p.n1 : co;
    const bool relu = p.relu && !second;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int m = m0 + wp * WP + j * 16 + fr;
        if (m >= p.M) continue;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (TERMS == 3 ? acc0[j][r] + acc1[j][r] * (1.f / 2048.f) : acc0[j][r]) + bv[r];
        if (p.res) {
            const f32x4 rv = *reinterpret_cast<const f32x4 *>(p.res + (size_t)m * p.Cout + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.out16) *reinterpret_cast<uint2 *>(p.out16 + (size_t)m * p.Cout + co) = make_uint2(pack2_h<true>(v[0], v[1]), pack2_h<true>(v[2], v[3]));
        else *reinterpret_cast<f32x4 *>(obase + (size_t)m * ostride + oc) = v;
    }
}

// fp32 weights [rows][K] (K = kh kw cin, cin % 32 == 0, rows % 16 == 0) -> the split fragment layout above
__global__ void split16_pack_kernel(const float *__restrict__ w, u16 *__restrict__ out, int rows, int K) {
    const long long total = (long long)rows * (K / 8);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / (K / 8)), k8 = (int)(idx % (K / 8));
        const float *src = w + (size_t)row * K + (size_t)k8 * 8;
        const size_t base = (((size_t)(row >> 4) * (K / 32) + (k8 >> 2)) * 2) * 512 + (size_t)((k8 & 3) * 16 + (row & 15)) * 8;
        u16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f16_t h = (f16_t)src[e];
            hi[e] = __builtin_bit_cast(u16, h);
            lo[e] = __builtin_bit_cast(u16, (f16_t)((src[e] - (float)h) * 2048.f));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { out[base + e] = hi[e]; out[base + 512 + e] = lo[e]; }
    }
}

pvr_status launch_split16_pack(const float *w, void *out, int rows, int K, hipStream_t stream) {
    PVR_REQUIRE(w && out && rows % 16 == 0 && K % 32 == 0, "split16_pack: rows %% 16 and K %% 32 must be 0 (rows %d, K %d)", rows, K);
    const long long total = (long long)rows * (K / 8);
    hipLaunchKernelGGL(split16_pack_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, stream, w, (u16 *)out, rows, K);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

bool conv_split16_supported(int cin, int cout, int k) { return cin % 32 == 0 && cout % 16 == 0 && k <= 3 && k >= 1; }

static long long g_split16_launches = 0;
long long conv_split16_launches() { return g_split16_launches; }

// in / res / out: fp32 NHWC (out and res with row stride cout); wsp: launch_split16_pack of the (cout rounded up to 64, k k cin) fp32 weights.
// out2 / n1: the pair form (see ConvS); out16: 16-bit output instead of `out`; terms: 3 = the exact split product, 1 = a 16-bit convolution of an fp32 input
pvr_status launch_conv_split16(const float *in, const void *wsp, const float *bias, const float *res, float *out, int n, int h, int w, int cin,
                               int cout, int k, int stride, int pad, int relu, hipStream_t stream, float *out2, int n1, void *out16, int terms) {
    PVR_REQUIRE(in && wsp && bias && (out || out16), "conv_split16: null argument");
    PVR_REQUIRE(conv_split16_supported(cin, cout, k), "conv_split16: cin %d must be a multiple of 32, cout %d of 16, k %d in 1..3", cin, cout, k);
    PVR_REQUIRE(terms == 3 || terms == 1, "conv_split16: terms must be 3 or 1");
    PVR_REQUIRE(!out2 || (n1 > 0 && n1 < cout && n1 % 16 == 0 && !res && !out16), "conv_split16 (pair form): n1 must split the couts at a multiple of 16, no residual");
    ConvS p;
    p.in = in; p.w = (const u16 *)wsp; p.bias = bias; p.res = res; p.out = out; p.out2 = out2; p.n1 = n1; p.out16 = (u16 *)out16;
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.KH = k; p.KW = k; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - k) / stride + 1; p.Wo = (w + 2 * pad - k) / stride + 1;
    const int cout_pad = (cout + 63) / 64 * 64;
    const int64_t M = (int64_t)n * p.Ho * p.Wo, inb = (int64_t)n * h * w * cin * 4, wb = (int64_t)cout_pad * k * k * cin * 4;
    PVR_REQUIRE(M < (1ll << 31) && inb < 0x7ffffff0ll && wb < 0x7ffffff0ll, "conv_split16: operand larger than 2 GiB (use a smaller chunk)");
    p.M = (int)M; p.relu = relu; p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb; p.nsteps = k * k * cin / 32;
    static const int cus = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
    ++g_split16_launches;
#define PVR_S16_LAUNCH(BM_, BN_, grid_)                                                                                              \
    {                                                                                                                                \
        if (terms == 3) hipLaunchKernelGGL((conv_split16_kernel<BM_, BN_, 3>), dim3((unsigned)(grid_)), dim3(512), 0, stream, p);    \
        else hipLaunchKernelGGL((conv_split16_kernel<BM_, BN_, 1>), dim3((unsigned)(grid_)), dim3(512), 0, stream, p);               \
    }
    if (cout_pad % 128 == 0) {
        p.n_ctiles = cout_pad / 128;
        const int64_t t128 = ((M + 127) / 128) * p.n_ctiles;
        if (t128 >= cus) PVR_S16_LAUNCH(128, 128, t128)
        else PVR_S16_LAUNCH(64, 128, ((M + 63) / 64) * p.n_ctiles)
    } else {
        p.n_ctiles = cout_pad / 64;
        const int64_t t128 = ((M + 127) / 128) * p.n_ctiles;
        if (t128 >= cus) PVR_S16_LAUNCH(128, 64, t128)
        else PVR_S16_LAUNCH(64, 64, ((M + 63) / 64) * p.n_ctiles)
    }
#undef PVR_S16_LAUNCH
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
