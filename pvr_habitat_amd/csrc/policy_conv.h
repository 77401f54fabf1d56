// PolicyNetWithConv feature extractor (reference src/models.py:107-118, 159-170), fp32, forward and backward.
//   x/255 -> per 3-channel frame: transpose(1,3) (H<->W swap) -> 5 x [Conv2d(k3, s2, p1, ->32) + ELU] -> cat on the
//   last axis -> view(T*B, -1)
// Activations are NHWC fp32 [frame][S][S][32] in the TRANSPOSED spatial domain the reference convolves in
// (row index a = original column, column index b = original row).  All three kernels are implicit GEMMs on the
// f32 MFMA (16x16x4); reductions are fixed-order (per-wave partials + colreduce), no atomics.
#pragma once
#include "policy_kernels.h"

namespace pvr {

struct ConvFP {
    const void *in;          // CIN==3: uint8 obs (N,64,64,3*nf); CIN==4: fp32 NHWC [F][Sin][Sin][4] (RGB + zero pad); else fp32 NHWC [F][Sin][Sin][32]
    const float *W, *bias;   // [32][9][CINP] (CINP = 4 or 32), [32]
    float *out;              // [F][So][So][32] post-ELU
    int F, Sin, So, nf;      // F frames (= N*nf)
};

constexpr int CONV_TPW = 8;      // 16-pixel tiles per wave of the few-channel forward kernel

// (float)x / 255.0f for x = 0 .. 255, bit for bit (the reference divides: models.py:163), without the ~10-instruction IEEE division
// sequence: one multiply by 1/255 and one Newton step through two FMAs give the correctly rounded quotient for all 256 inputs
// (checked exhaustively; the multiply alone is off by an ulp for 126 of them).
__device__ __forceinline__ float div255(float x) {
    const float inv = 1.0f / 255.0f;
    const float q = x * inv;
    return __fmaf_rn(__fmaf_rn(-q, 255.0f, x), inv, q);
}

// forward: wave = 16 output pixels x 32 channels
template <int CIN>
static __global__ __launch_bounds__(256) void conv_s2_fwd_kernel(ConvFP p) {
    constexpr int CP = CIN <= 4 ? 4 : 32;
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const long long npix = (long long)p.F * p.So * p.So;
    // few-channel first layer: the lane's 18 weight values stay in registers over CONV_TPW consecutive tiles (one tile per wave meant
    // 18 weight loads + 9 pixel loads for 18 MFMAs)
    constexpr int TPW = CIN <= 4 ? CONV_TPW : 1;
    float wv[CIN <= 4 ? 9 : 1][2];
    if constexpr (CIN <= 4) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 2; ++j) wv[tap][j] = p.W[((j * 16 + fr) * 9 + tap) * CP + fq];
    }
    // CIN == 3: the nine input bytes of the wave's NEXT tile are requested before the current tile's MFMAs (the kernel is a chain of
    // byte gathers: 9 loads -> 18 MFMAs -> stores per tile, 8 tiles per wave)
    uint8_t xnext[CIN == 3 ? 9 : 1];
    auto gather3 = [&](long long tile_, uint8_t (&dst)[CIN == 3 ? 9 : 1]) {
        if constexpr (CIN == 3) {
            const long long px_ = tile_ * 16 + fr;
            const bool pk = px_ < npix;
            const long long q_ = pk ? px_ : 0;
            const int ox_ = (int)(q_ % p.So), oy_ = (int)((q_ / p.So) % p.So), f_ = (int)(q_ / ((long long)p.So * p.So));
            const int n_ = f_ / p.nf, fi_ = f_ % p.nf;
            const uint8_t *x = (const uint8_t *)p.in;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = 2 * oy_ + tap / 3 - 1, ix = 2 * ox_ + tap % 3 - 1;
                const bool in = pk && (unsigned)iy < (unsigned)p.Sin && (unsigned)ix < (unsigned)p.Sin && fq < 3;
                dst[tap] = x[(((size_t)n_ * p.Sin + (in ? ix : 0)) * p.Sin + (in ? iy : 0)) * (3 * p.nf) + 3 * fi_ + (in ? fq : 0)];
            }
        }
    };
    const long long tile0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW;
    if constexpr (CIN == 3) gather3(tile0, xnext);
    for (int tt = 0; tt < TPW; ++tt) {
    const long long tile = tile0 + tt;
    if (tile * 16 >= npix) return;
    uint8_t xcur[CIN == 3 ? 9 : 1];
    if constexpr (CIN == 3) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) xcur[tap] = xnext[tap];
        if (tt + 1 < TPW) gather3(tile + 1, xnext);
    }
    const long long pix = tile * 16 + fr;                 // B-operand column of this lane
    const bool pok = pix < npix;
    const long long pp = pok ? pix : 0;
    const int ox = (int)(pp % p.So), oy = (int)((pp / p.So) % p.So);
    const int f = (int)(pp / ((long long)p.So * p.So));
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = 2 * oy + tap / 3 - 1, ix = 2 * ox + tap % 3 - 1;
        const bool ok = pok && (unsigned)iy < (unsigned)p.Sin && (unsigned)ix < (unsigned)p.Sin;
        if constexpr (CIN == 3) {
            // k-slot fq = input channel; conv input I'[iy][ix][c] = x[n][ix][iy][3*fr_ + c] / 255  (H<->W swap)
            // (clamped address + select instead of a branch around the load: hipcc waits for every load inside a branch on its own,
            //  nine serial memory latencies per tile; unconditional loads are issued back to back)
            const bool in = ok && fq < 3;
            const float a = in ? div255((float)xcur[tap]) : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mfma_f32(wv[tap][j], a, acc[j]);
        } else if constexpr (CIN == 4) {
            // 'random' PVR first layer: normalised fp32 image, k-slot fq = channel (slot 3 is the zero pad)
            const float av = ((const float *)p.in)[(((size_t)f * p.Sin + (ok ? iy : 0)) * p.Sin + (ok ? ix : 0)) * 4 + fq];
            const float a = ok ? av : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mfma_f32(wv[tap][j], a, acc[j]);
        } else {
            const float *src = (const float *)p.in + (((size_t)f * p.Sin + (ok ? iy : 0)) * p.Sin + (ok ? ix : 0)) * 32;
#pragma unroll
            for (int c0 = 0; c0 < 32; c0 += 16) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(src + c0 + fq * 4);
                if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(p.W + ((size_t)(j * 16 + fr) * 9 + tap) * CP + c0 + fq * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j] = mfma_f32(w[e], a[e], acc[j]);
                }
            }
        }
    }
    // The weights are the A operand (rows = channels), the pixels the B operand (columns): D row = 4*fq + r = channel (+16 j),
    // col = fr = pixel, so a lane owns 4 consecutive channels of its pixel -> one 16-byte store per j (the transposed product gave
    // four scattered 4-byte stores per j; same multiplies and the same accumulation order per element, bit-identical).
    if (pok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(p.bias + j * 16 + fq * 4);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[j][r] + b[r];
                o[r] = v > 0.f ? v : expf(v) - 1.f;                                 // ELU (alpha 1)
            }
            *reinterpret_cast<f32x4 *>(p.out + (size_t)pix * 32 + j * 16 + fq * 4) = o;
        }
    }
    }
}

// d(pre-activation) = d(out) * (out > 0 ? 1 : out + 1)   in place  (elu'(x) = exp(x) = out + 1 for x <= 0)
static __global__ __launch_bounds__(256) void elu_bwd_kernel(float *__restrict__ d, const float *__restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 g = reinterpret_cast<f32x4 *>(d)[i];
        const f32x4 o = reinterpret_cast<const f32x4 *>(out)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= o[e] > 0.f ? 1.f : o[e] + 1.f;
        reinterpret_cast<f32x4 *>(d)[i] = g;
    }
}

// weight gradient partials: wave w accumulates dW[co][tap][ci] over its pixel range; partial[w][co*9*CP + tap*CP + ci]
struct ConvWP {
    const void *in;          // layer input (uint8 obs for CIN==3, else fp32 NHWC)
    const float *dpre;       // [F][So][So][32]
    float *partial;          // [waves][32*9*CP]
    int F, Sin, So, nf, waves;
};

template <int CIN>
static __global__ __launch_bounds__(256) void conv_s2_wgrad_kernel(ConvWP p) {
    constexpr int CP = CIN == 3 ? 4 : 32;
    constexpr int NT = CIN == 3 ? 3 : 18;                 // 16-column tiles over the 9*CP columns (36 / 288)
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long npix = (long long)p.F * p.So * p.So;
    const long long per = ((npix + p.waves - 1) / p.waves + 3) / 4 * 4;
    const long long beg = (long long)wv * per, end = beg + per < npix ? beg + per : npix;
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ls = 31 - __builtin_clz(p.So);              // So is a power of two (64 >> layer): shifts, not 64-bit divisions, per step
    for (long long p0 = beg; p0 < end; p0 += 4) {
        const long long pix = p0 + fq;                    // k-slot fq = pixel
        const bool pok = pix < end;
        const unsigned pp = pok ? (unsigned)pix : 0u;     // (npix < 2^31, checked by the launcher)
        const int ox = (int)(pp & (unsigned)(p.So - 1)), oy = (int)((pp >> ls) & (unsigned)(p.So - 1));
        const int f = (int)(pp >> (2 * ls));
        float a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = pok ? p.dpre[(size_t)pp * 32 + i * 16 + fr] : 0.f;
        // all NT gathers of the step are issued back to back (clamped addresses, no branch around a load - inside branches hipcc
        // waited for every load separately: NT serial memory latencies per step), then the MFMAs consume them
        float b[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = j * 16 + fr, tap = col / CP, ci = col % CP;
            const int iy = 2 * oy + tap / 3 - 1, ix = 2 * ox + tap % 3 - 1;
            const bool in = pok && tap < 9 && (unsigned)iy < (unsigned)p.Sin && (unsigned)ix < (unsigned)p.Sin && (CIN != 3 || ci < 3);
            const int cy = in ? iy : 0, cx = in ? ix : 0, cf = in ? f : 0, cc = in ? ci : 0;
            float v;
            if constexpr (CIN == 3) {
                const int n = cf / p.nf, fi = cf % p.nf;
                v = div255((float)((const uint8_t *)p.in)[(((size_t)n * p.Sin + cx) * p.Sin + cy) * (3 * p.nf) + 3 * fi + cc]);
            } else {
                v = ((const float *)p.in)[(((size_t)cf * p.Sin + cy) * p.Sin + cx) * 32 + cc];
            }
            b[j] = in ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = mfma_f32(a[i], b[j], acc[i][j]);
    }
    // D: row = co (4*fq + r + 16 i), col = column fr + 16 j
    float *out = p.partial + (size_t)wv * 32 * 9 * CP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = j * 16 + fr;
            if (col >= 9 * CP) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(i * 16 + fq * 4 + r) * 9 * CP + col] = acc[i][j][r];
        }
}

// input gradient (layers 2..5): dIn[f][iy][ix][ci] = sum_{taps with matching parity} dPre[f][oy][ox][:] . Wt[tap][ci][:]
struct ConvDP {
    const float *dpre, *Wt;  // [F][So][So][32], [9][32 ci][32 co]
    float *din;              // [F][Sin][Sin][32]
    const float *act_in;     // the lower layer's ELU output, same shape as din: its ELU' is applied on the way out (elu_bwd_kernel fused)
    int F, Sin, So;
};

// A wave owns 16 input pixels of ONE parity class (iy & 1, ix & 1): with stride 2 / pad 1 / 3x3 an input pixel is reached by
// tap ky only when ky = iy + 1 (mod 2), so a class uses 1, 2, 2 or 4 of the 9 taps and the other taps are skipped for the whole
// wave (a mixed tile multiplied zeros for 75 % of its MFMAs).  Skipped taps contributed exact zeros: results are unchanged.
static __global__ __launch_bounds__(256) void conv_s2_dgrad_kernel(ConvDP p) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int H2 = p.Sin >> 1;
    const long long nq = (long long)p.F * H2 * H2;                  // pixels per parity class
    const long long tpc = (nq + 15) / 16;                            // tiles per class
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= 4 * tpc) return;
    const int cls = (int)(tile / tpc), py = cls >> 1, px = cls & 1;
    const long long t = tile % tpc;
    const long long q = t * 16 + fr;
    const bool pok = q < nq;
    const long long qq = pok ? q : 0;
    const int ix = 2 * (int)(qq % H2) + px, iy = 2 * (int)((qq / H2) % H2) + py;
    const int f = (int)(qq / ((long long)H2 * H2));
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (((tap / 3) & 1) == py || ((tap % 3) & 1) == px) continue;           // wave-uniform: this class never meets the tap
        const int ty = iy + 1 - tap / 3, tx = ix + 1 - tap % 3;     // = 2*oy, 2*ox
        const bool ok = pok && ty >= 0 && tx >= 0 && (ty >> 1) < p.So && (tx >> 1) < p.So;
        const float *src = p.dpre + (((size_t)f * p.So + (ok ? ty >> 1 : 0)) * p.So + (ok ? tx >> 1 : 0)) * 32;
#pragma unroll
        for (int c0 = 0; c0 < 32; c0 += 16) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(src + c0 + fq * 4);      // unconditional (clamped) load + select, see the forward kernel
            if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(p.Wt + ((size_t)tap * 32 + j * 16 + fr) * 32 + c0 + fq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j] = mfma_f32(a[e], w[e], acc[j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long qo = t * 16 + fq * 4 + r;
        if (qo >= nq) continue;
        const int ox = 2 * (int)(qo % H2) + px, oy = 2 * (int)((qo / H2) % H2) + py;
        const size_t o = (((size_t)(qo / ((long long)H2 * H2)) * p.Sin + oy) * p.Sin + ox) * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float ov = p.act_in[o + j * 16 + fr];
            p.din[o + j * 16 + fr] = acc[j][r] * (ov > 0.f ? 1.f : ov + 1.f);   // d(pre-activation) of the lower layer, as elu_bwd_kernel
        }
    }
}

// W [32 co][9][32 ci] -> Wt [9][32 ci][32 co]
static __global__ __launch_bounds__(256) void conv_wt_kernel(const float *__restrict__ W, float *__restrict__ Wt) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 9 * 32 * 32; i += gridDim.x * 256) {
        const int co = i % 32, ci = (i / 32) % 32, tap = i / 1024;
        Wt[i] = W[(co * 9 + tap) * 32 + ci];
    }
}

// OIHW (32,CIN,3,3) <-> [32][9][CP] packing of the reference parameter (and back for gradients)
template <int CIN>
static __global__ __launch_bounds__(256) void conv_pack_kernel(const float *__restrict__ oihw, float *__restrict__ packed, int to_packed) {
    constexpr int CP = CIN == 3 ? 4 : 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 32 * 9 * CP; i += gridDim.x * 256) {
        const int ci = i % CP, tap = (i / CP) % 9, co = i / (9 * CP);
        if (ci >= CIN) { if (to_packed) packed[i] = 0.f; continue; }
        const int o = (co * CIN + ci) * 9 + tap;
        if (to_packed) packed[i] = oihw[o]; else const_cast<float *>(oihw)[o] = packed[i];
    }
}

// features: feat[n][c*(4*nf) + a*(2*nf) + fi*2 + b] = act5[n*nf + fi][a][b][c]   (torch.cat(..., -1).view(T*B, -1))
static __global__ __launch_bounds__(256) void conv_feat_kernel(float *__restrict__ act5, float *__restrict__ feat, int N, int nf, int to_feat) {
    const int per = 32 * 4 * nf;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * per; i += gridDim.x * 256) {
        const int n = i / per, k = i % per;
        const int b = k % 2, fi = (k / 2) % nf, a = (k / (2 * nf)) % 2, c = k / (4 * nf);
        const size_t src = ((((size_t)n * nf + fi) * 2 + a) * 2 + b) * 32 + c;
        if (to_feat) feat[i] = act5[src]; else act5[src] = feat[i];
    }
}

// partial[g][c] = sum over rows [g*per, min((g+1)*per, R)) of X[r][c], C = 32 (bias gradients of the conv stack).
// A thread owns 4 channels of every 32nd row of its block's range (16-byte loads, 8 lanes per row); fixed summation order.
static __global__ __launch_bounds__(256) void rowblock_colsum_kernel(const float *__restrict__ X, float *__restrict__ partial, int R, int per) {
    __shared__ float s[32][33];
    const int q = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int beg = blockIdx.x * per, end = beg + per < R ? beg + per : R;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = beg + g; r < end; r += 32) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(X + (size_t)r * 32 + q * 4);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[g][q * 4 + e] = a[e];
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += s[i][threadIdx.x];
        partial[blockIdx.x * 32 + threadIdx.x] = t;
    }
}

// BatchNorm1d input gradient: dx = gamma*invstd*(dy - dbeta/N - xhat*dgamma/N)
static __global__ __launch_bounds__(256) void bn_dx_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ mean,
                                                    const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                    const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                    float *__restrict__ dx, int N, int C, int n_stat) {
    const size_t total = (size_t)N * C;
    const float inv_n = 1.0f / (float)n_stat;      // rows the statistics were taken over (N, or the global batch under SyncBN)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float xh = (x[i] - mean[c]) * invstd[c];
        dx[i] = gamma[c] * invstd[c] * (dy[i] - dbeta[c] * inv_n - xh * dgamma[c] * inv_n);
    }
}

}  // namespace pvr
