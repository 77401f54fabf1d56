// Byte kernels of the CLIP RN50 visual tower (openai/CLIP ModifiedResNet + AttentionPool2d, reached from reference
// src/embeddings.py:305-306 clip.load("RN50") and :375-376 encode_image).  The convolutions run on the shared implicit-GEMM
// kernels (every convolution of this tower has stride 1: the spatial reduction is AvgPool2d), the attention core on vit.hip's.
#include "encoder_internal.h"

namespace pvr {

// AvgPool2d(2) on NHWC 16-bit: fp32 sum of the 4 taps * 0.25, rounded once to the storage type
template <bool F16>
__global__ __launch_bounds__(256) void avgpool2_kernel(const u16 *__restrict__ in, u16 *__restrict__ out, int n, int h, int w, int c) {
    const int ho = h / 2, wo = w / 2, c8 = c / 8;
    const size_t total = (size_t)n * ho * wo * c8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i % c8) * 8;
        const size_t pix = i / c8;
        const int x = (int)(pix % wo), y = (int)((pix / wo) % ho), b = (int)(pix / ((size_t)wo * ho));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(in + (((size_t)b * h + 2 * y + dy) * w + 2 * x + dx) * c + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += from_h<F16>((u16)(v[e] & 0xffffu));
                    acc[2 * e + 1] += from_h<F16>((u16)(v[e] >> 16));
                }
            }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2_h<F16>(acc[2 * e] * 0.25f, acc[2 * e + 1] * 0.25f);
        *reinterpret_cast<u32x4 *>(out + pix * c + ch) = o;
    }
}

pvr_status launch_avgpool2(const void *in, void *out, int n, int h, int w, int c, int dtype, hipStream_t st) {
    PVR_REQUIRE(h % 2 == 0 && w % 2 == 0 && c % 8 == 0, "avgpool2: %dx%dx%d not poolable", h, w, c);
    const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 8);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (dtype == PVR_F16) hipLaunchKernelGGL(avgpool2_kernel<true>, dim3(blocks), dim3(256), 0, st, (const u16 *)in, (u16 *)out, n, h, w, c);
    else hipLaunchKernelGGL(avgpool2_kernel<false>, dim3(blocks), dim3(256), 0, st, (const u16 *)in, (u16 *)out, n, h, w, c);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// AttentionPool2d token assembly: tokens[b][0] = mean_p x[b][p] + pos[0], tokens[b][1+p] = x[b][p] + pos[1+p]   (x fp32 NHWC)
template <bool F16>
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const float *__restrict__ x, const float *__restrict__ pos,
                                                              u16 *__restrict__ tok, int hw, int c) {
    const int b = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    const float *xb = x + (size_t)b * hw * c + ch;
    u16 *tb = tok + (size_t)b * (hw + 1) * c + ch;
    float s = 0.f;
    for (int p = 0; p < hw; ++p) {
        const float v = xb[(size_t)p * c];
        s += v;
        tb[(size_t)(p + 1) * c] = to_h<F16>(v + pos[(size_t)(p + 1) * c + ch]);
    }
    tb[0] = to_h<F16>(s / (float)hw + pos[ch]);
}

pvr_status launch_attnpool_tokens(const float *x, const float *pos, void *tokens, int n, int hw, int c, int dtype, hipStream_t st) {
    dim3 grid((c + 255) / 256, n);
    if (dtype == PVR_F16) hipLaunchKernelGGL(attnpool_tokens_kernel<true>, grid, dim3(256), 0, st, x, pos, (u16 *)tokens, hw, c);
    else hipLaunchKernelGGL(attnpool_tokens_kernel<false>, grid, dim3(256), 0, st, x, pos, (u16 *)tokens, hw, c);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
