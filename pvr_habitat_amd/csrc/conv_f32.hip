// fp32 "reference-precision" encoder kernels (dtype PVR_F32): the ResNet50 family with fp32 storage and the
// f32-input MFMA (v_mfma_f32_16x16x4_f32 = an exact fp32 fma chain), i.e. the reference's own arithmetic type
// (the reference is fp32 end to end).  Runs at the f32 MFMA rate (1/16 of bf16) and twice the bytes, so it
// is the parity mode, not the throughput mode.  Same plan, topology and folded-BN weights as the 16-bit path.
#include "common.h"

namespace pvr {

__device__ __forceinline__ f32x4 mfma_f32x(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct ConvF {
    const float *in, *wgt, *bias, *res;
    float *out;
    int N, H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad, M, K, relu, n_tiles;
    unsigned in_bytes, w_bytes;
};

// NHWC implicit GEMM, 64 pixels x 64 couts x 32 k per step, 2x2 waves, 2-stage LDS pipeline.
// Cin % 32 == 0, so a 32-wide K slice is one filter tap and 128 contiguous bytes.
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF p) {
    constexpr int BM = 64, BN = 64, BK = 32, LD = 36, TILE = 64 * 36;
    __shared__ __attribute__((aligned(16))) float sm[2][2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (swz / p.n_tiles) * BM, n0 = (swz % p.n_tiles) * BN;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wgt), 0, p.w_bytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    int a_off[2], a_mask[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, n = t / p.Ho;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + c4 * 4) * 4;
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
        const int co = n0 + row;
        b_off[i] = co < p.CoutPad ? (co * p.K + c4 * 4) * 4 : OOB;
    }
    const int cpt = p.Cin / BK, nk = p.KH * p.KW * cpt;
    int kh = 0, kw = 0, cs = 0, tap = 0;
    f32x4 ra[2], rb[2];
#define PVR_F_LOAD(kt_)                                                                                  \
    {                                                                                                    \
        const int tap_off = ((kh * p.W + kw) * p.Cin + cs * BK) * 4;                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
            const int vo = ((a_mask[i] >> tap) & 1) ? a_off[i] + tap_off : OOB;                          \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0));   \
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], (kt_) * (BK * 4), 0)); \
        }                                                                                                \
        if (++cs == cpt) { cs = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }                          \
    }
#define PVR_F_STORE(buf_)                                                                                \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
            const int idx = tid + 256 * i;                                                               \
            *reinterpret_cast<f32x4 *>(&sm[buf_][0][(idx >> 3) * LD + (idx & 7) * 4]) = ra[i];           \
            *reinterpret_cast<f32x4 *>(&sm[buf_][1][(idx >> 3) * LD + (idx & 7) * 4]) = rb[i];           \
        }                                                                                                \
    }
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    PVR_F_LOAD(0);
    PVR_F_STORE(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) PVR_F_LOAD(kt + 1);
        const float *As = sm[cur][0], *Bs = sm[cur][1];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int k = ks * 4 + fq;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + fr) * LD + k];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(wn * 32 + j * 16 + fr) * LD + k];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32x(a[i], b[j], acc[i][j]);
        }
        if (more) PVR_F_STORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef PVR_F_LOAD
#undef PVR_F_STORE
    // D: row = pixel (4*fq + r), col = cout (fr)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 32 + j * 16 + fr;
        if (co >= p.Cout) continue;
        const float bv = p.bias[co];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.res) v += p.res[(size_t)m * p.Cout + co];
                if (p.relu) v = fmaxf(v, 0.f);
                p.out[(size_t)m * p.Cout + co] = v;
            }
    }
}

// conv1 7x7/2 pad 3 + folded BN + ReLU on the normalised fp32 NHWC4 image: wave = 16 output pixels x 64 channels,
// k-slot = input channel (slot 3 is the zero pad), one MFMA per (tap, 16 couts).  wgt [64][49][4].
__global__ __launch_bounds__(256) void stem_f32_kernel(const float *__restrict__ img, const float *__restrict__ wgt,
                                                       const float *__restrict__ bias, float *__restrict__ out, int n, int S) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int So = S / 2;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), npix = (long long)n * So * So;
    if (tile * 16 >= npix) return;
    const long long pix = tile * 16 + fr;
    const bool pok = pix < npix;
    const long long pp = pok ? pix : 0;
    const int ox = (int)(pp % So), oy = (int)((pp / So) % So), b = (int)(pp / ((long long)So * So));
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = 2 * oy + ky - 3;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = 2 * ox + kx - 3, tap = ky * 7 + kx;
            const bool ok = pok && (unsigned)iy < (unsigned)S && (unsigned)ix < (unsigned)S;
            const float a = ok ? img[(((size_t)b * S + iy) * S + ix) * 4 + fq] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma_f32x(a, wgt[((j * 16 + fr) * 49 + tap) * 4 + fq], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bv = bias[j * 16 + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long q = tile * 16 + fq * 4 + r;
            if (q < npix) out[(size_t)q * 64 + j * 16 + fr] = fmaxf(acc[j][r] + bv, 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float *__restrict__ in, float *__restrict__ out, int n, int h, int w,
                                                          int c, int ho, int wo) {
    const int cg = c / 4;
    const size_t total = (size_t)n * ho * wo * cg;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ch = (int)(idx % cg) * 4;
        size_t r = idx / cg;
        const int x = (int)(r % wo); r /= wo;
        const int y = (int)(r % ho), b = (int)(r / ho);
        f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * y - 1 + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * x - 1 + dx;
                if (xx < 0 || xx >= w) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)b * h + yy) * w + xx) * c + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        *reinterpret_cast<f32x4 *>(out + (((size_t)b * ho + y) * wo + x) * c + ch) = m;
    }
}

pvr_status launch_conv_f32(const float *in, const float *wgt, const float *bias, const float *res, float *out, int n, int h, int w,
                           int cin, int cout, int k, int stride, int pad, int relu, hipStream_t stream) {
    PVR_REQUIRE(cin % 32 == 0 && k <= 3, "conv_f32: cin %d must be a multiple of 32 and k <= 3", cin);
    ConvF p;
    p.in = in; p.wgt = wgt; p.bias = bias; p.res = res; p.out = out;
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.CoutPad = (cout + 63) / 64 * 64; p.KH = k; p.KW = k;
    p.stride = stride; p.pad = pad; p.Ho = (h + 2 * pad - k) / stride + 1; p.Wo = (w + 2 * pad - k) / stride + 1;
    const int64_t M = (int64_t)n * p.Ho * p.Wo, inb = (int64_t)n * h * w * cin * 4, wb = (int64_t)p.CoutPad * k * k * cin * 4;
    PVR_REQUIRE(M < (1ll << 31) && inb < 0x7ffffff0ll && wb < 0x7ffffff0ll, "conv_f32: operand larger than 2 GiB (use a smaller chunk)");
    p.M = (int)M; p.K = k * k * cin; p.relu = relu; p.in_bytes = (unsigned)inb; p.w_bytes = (unsigned)wb;
    p.n_tiles = (cout + 63) / 64;
    const int grid = ((p.M + 63) / 64) * p.n_tiles;
    hipLaunchKernelGGL(conv_f32_kernel, dim3(grid), dim3(256), 0, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_stem_f32(const float *img, const float *wgt, const float *bias, float *out, int n, int S, hipStream_t stream) {
    const long long tiles = ((long long)n * (S / 2) * (S / 2) + 15) / 16;
    hipLaunchKernelGGL(stem_f32_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, img, wgt, bias, out, n, S);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_maxpool_f32(const float *in, float *out, int n, int h, int w, int c, hipStream_t stream) {
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)n * ho * wo * (c / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3(blocks), dim3(256), 0, stream, in, out, n, h, w, c, ho, wo);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
