// 'random' PVR: the fixed 5-layer conv encoder of reference src/embeddings.py:90-106 behind EmbeddingNet
// (used by main_bc_1.py with seed-dependent orthogonal weights): default transforms (:80-85) ->
// 5 x [Conv2d(k3, s2, p1, ->32) + ELU] -> (N,32,7,7) -> C-major flatten (1568).  fp32 end to end on the f32 MFMA
// (the kernels are the PolicyNetWithConv forward kernels of policy_conv.h).  SURVEY 8f N4.
#include "encoder_internal.h"
#include "policy_conv.h"

namespace pvr {

// uint8 NHWC frame -> Resize(256, bilinear, round to uint8) -> CenterCrop(224) -> /255 -> Normalize, as fp32 NHWC4
__global__ __launch_bounds__(256) void normalize_nhwc4_kernel(const u16 *__restrict__ img_h, float *__restrict__ out, int n, int crop,
                                                              float m0, float m1, float m2, float s0, float s1, float s2, int f16) {
    // img_h is the stem input image written by preprocess_kernel: (n, crop+6, crop+8, 4) 16-bit holding x-128 exactly
    const size_t total = (size_t)n * crop * crop;
    const int PW = crop + 8, PH = crop + 6;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % crop), y = (int)((i / crop) % crop), b = (int)(i / ((size_t)crop * crop));
        const ushort4 v = *reinterpret_cast<const ushort4 *>(img_h + (((size_t)b * PH + y + 3) * PW + x + 3) * 4);
        const float r = (f16 ? from_h<true>(v.x) : from_h<false>(v.x)) + 128.f, g = (f16 ? from_h<true>(v.y) : from_h<false>(v.y)) + 128.f,
                    bl = (f16 ? from_h<true>(v.z) : from_h<false>(v.z)) + 128.f;
        f32x4 o;
        o[0] = (r / 255.0f - m0) / s0; o[1] = (g / 255.0f - m1) / s1; o[2] = (bl / 255.0f - m2) / s2; o[3] = 0.f;
        reinterpret_cast<f32x4 *>(out)[i] = o;
    }
}

}  // namespace pvr

namespace pvr {
pvr_status launch_normalize_nhwc4(const void *img_h, float *out, int n, int crop, const float *mean, const float *std_, int dtype, hipStream_t st) {
    const size_t tot = (size_t)n * crop * crop;
    hipLaunchKernelGGL(normalize_nhwc4_kernel, dim3((unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, st,
                       (const u16 *)img_h, out, n, crop, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], dtype == PVR_F16 ? 1 : 0);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}
}  // namespace pvr

using namespace pvr;

struct pvr_random5 {
    float *w[5] = {nullptr}, *b[5] = {nullptr}, *act[5] = {nullptr}, *img = nullptr;
    u16 *img_h = nullptr;
};

namespace pvr {

pvr_status random5_create(pvr_encoder *e) {
    e->rnd = new pvr_random5();
    e->out_size = 32 * 7 * 7;
    return PVR_OK;
}

pvr_status random5_finalize(pvr_encoder *e) {
    pvr_random5 *r = e->rnd;
    pvr_status s;
    for (int l = 0; l < 5; ++l) {
        const int cin = l == 0 ? 3 : 32, cp = l == 0 ? 4 : 32;
        const HostTensor *w, *b;
        const std::string nm = std::to_string(2 * l);
        if ((s = enc_need(e, nm + ".weight", &w, (size_t)32 * cin * 9))) return s;
        if ((s = enc_need(e, nm + ".bias", &b, 32))) return s;
        std::vector<float> hw((size_t)32 * 9 * cp, 0.f);
        for (int co = 0; co < 32; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) hw[((size_t)co * 9 + t) * cp + ci] = w->data[((size_t)co * cin + ci) * 9 + t];
        if ((s = enc_upload(&r->w[l], hw))) return s;
        if ((s = enc_upload(&r->b[l], b->data))) return s;
    }
    const size_t C = e->desc.chunk, crop = e->desc.crop;
    PVR_HIP_TRY(hipMalloc((void **)&r->img_h, C * (crop + 6) * (crop + 8) * 4 * 2));
    PVR_HIP_TRY(hipMemset(r->img_h, 0, C * (crop + 6) * (crop + 8) * 4 * 2));
    PVR_HIP_TRY(hipMalloc((void **)&r->img, C * crop * crop * 4 * sizeof(float)));
    size_t S = crop;
    for (int l = 0; l < 5; ++l) { S /= 2; PVR_HIP_TRY(hipMalloc((void **)&r->act[l], C * S * S * 32 * sizeof(float))); }
    return PVR_OK;
}

void random5_destroy(pvr_encoder *e) {
    pvr_random5 *r = e->rnd;
    if (!r) return;
    for (int l = 0; l < 5; ++l) { void *q[] = {r->w[l], r->b[l], r->act[l]}; for (void *x : q) if (x) (void)hipFree(x); }
    if (r->img) (void)hipFree(r->img);
    if (r->img_h) (void)hipFree(r->img_h);
    delete r;
    e->rnd = nullptr;
}

pvr_status random5_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride, hipStream_t st) {
    pvr_random5 *r = e->rnd;
    const int crop = e->desc.crop, dt = e->desc.dtype;
    pvr_status s;
    for (int f0 = 0; f0 < n; f0 += e->desc.chunk) {
        const int nb = (n - f0 < e->desc.chunk) ? n - f0 : e->desc.chunk;
        // the uint8 part of the transforms is shared with the ResNet path (bit-exact vs the oracle); x-128 is exact in 16 bits
        if ((s = launch_preprocess(frames + (size_t)f0 * h * w * 3, nb, h, w, e->desc.resize, crop, r->img_h, dt, st))) return s;
        const size_t tot = (size_t)nb * crop * crop;
        hipLaunchKernelGGL(normalize_nhwc4_kernel, dim3((unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, st,
                           r->img_h, r->img, nb, crop, e->desc.mean[0], e->desc.mean[1], e->desc.mean[2], e->desc.std_[0], e->desc.std_[1],
                           e->desc.std_[2], dt == PVR_F16 ? 1 : 0);
        const void *in = r->img;
        int S = crop;
        for (int l = 0; l < 5; ++l) {
            ConvFP c;
            c.in = in; c.W = r->w[l]; c.bias = r->b[l]; c.out = r->act[l]; c.F = nb; c.Sin = S; c.So = S / 2; c.nf = 1;
            const long long tiles = ((long long)nb * c.So * c.So + 15) / 16;
            if (l == 0) hipLaunchKernelGGL(conv_s2_fwd_kernel<4>, dim3((unsigned)((tiles + 4 * CONV_TPW - 1) / (4 * CONV_TPW))), dim3(256), 0, st, c);
            else hipLaunchKernelGGL(conv_s2_fwd_kernel<32>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, c);
            in = r->act[l]; S /= 2;
        }
        PVR_LAUNCH_CHECK();
        if ((s = launch_nhwc_to_chw(r->act[4], out + (size_t)f0 * out_stride, out_stride, nb, S * S, 32, 32, st))) return s;
        e->last_n = nb;
    }
    return PVR_OK;
}

}  // namespace pvr
