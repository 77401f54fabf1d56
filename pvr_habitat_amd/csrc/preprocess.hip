// K1+K2: uint8 NHWC frames -> Resize(short side) -> CenterCrop -> stem input image.
//
// Replaces reference src/embeddings.py:391-393 (.to(device), NHWC->NCHW transposes, the
// T.Resize/T.CenterCrop part of the transforms at :80-85).  ConvertImageDtype(/255) and
// Normalize are NOT applied here: they are folded into the stem weights (encoder.hip), so this
// kernel hands the stem the exact centred uint8 values x-128 (integers -128..127 are exact in bf16 and f16).
//
// torchvision 0.10 tensor Resize on uint8 = float32 bilinear (align_corners=False) then
// round-half-even back to uint8; restated in oracle/encoder_oracle.py:resize_u8.
//
// HBM-bound byte kernel: 3 B read (or 12 B for the 4 bilinear taps) + 8 B written per pixel.
// Output layout: (n, crop+6, crop+8, 4) 16-bit, pixel (y,x) at [y+3][x+3], channels
// (R-128,G-128,B-128,valid=1); the 3-pixel zero border (valid=0) is the conv1 padding and is never written.
#include "common.h"

namespace pvr {

struct PreP {
    const uint8_t *src;
    u16 *dst;
    int n, h, w;        // source frame size
    int rh, rw;         // size after Resize
    int top, left;      // crop offset inside the resized image
    int crop;
    int resize_needed;
    float scale_h, scale_w;
};

template <bool F16>
__global__ __launch_bounds__(256) void preprocess_kernel(PreP p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= p.crop || y >= p.crop) return;
    const int Y = y + p.top, X = x + p.left;
    const uint8_t *img = p.src + (size_t)n * p.h * p.w * 3;
    float v[3];
    if (!p.resize_needed) {
        const uint8_t *s = img + ((size_t)Y * p.w + X) * 3;
        v[0] = (float)s[0]; v[1] = (float)s[1]; v[2] = (float)s[2];
    } else {
#pragma clang fp contract(off)
        // ATen upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
        float sy = p.scale_h * ((float)Y + 0.5f) - 0.5f;
        float sx = p.scale_w * ((float)X + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const uint8_t *r0 = img + (size_t)y0 * p.w * 3, *r1 = img + (size_t)y1 * p.w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p00 = (float)r0[x0 * 3 + c], p01 = (float)r0[x1 * 3 + c];
            const float p10 = (float)r1[x0 * 3 + c], p11 = (float)r1[x1 * 3 + c];
            const float val = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
            v[c] = rintf(val);     // torch.round (half to even) then .to(uint8)
        }
    }
    const int PW = p.crop + 8, PH = p.crop + 6;
    ushort4 o;
    // centred values (x - 128, exact integers in bf16/f16): keeps the folded-Normalize products small, so the
    // 16-bit rounding of the stem weights is not amplified by the common-mode ~128 of every pixel
    o.x = to_h<F16>(v[0] - 128.f); o.y = to_h<F16>(v[1] - 128.f); o.z = to_h<F16>(v[2] - 128.f); o.w = to_h<F16>(1.0f);
    *reinterpret_cast<ushort4 *>(p.dst + (((size_t)n * PH + (y + 3)) * PW + (x + 3)) * 4) = o;
}

// host: resized size per torchvision resize(int size): smaller edge -> size, other edge int(size*long/short)
void resized_size(int h, int w, int size, int *rh, int *rw) {
    int sh = w <= h ? w : h, lg = w <= h ? h : w;
    if (sh == size) { *rh = h; *rw = w; return; }
    int ns = size, nl = (int)((double)size * (double)lg / (double)sh);
    if (w <= h) { *rw = ns; *rh = nl; } else { *rh = ns; *rw = nl; }
}

// the geometry launch_preprocess applies (for callers that fuse the no-resize case into their own kernel: stem_pool_lds_kernel<., true>)
void preprocess_geometry(int h, int w, int resize, int crop, int crop_pos, int *resize_needed, int *top, int *left) {
    int rh, rw;
    resized_size(h, w, resize, &rh, &rw);
    *resize_needed = (rh != h || rw != w) || rh < crop || rw < crop;
    *top = (int)nearbyint((rh - crop) / 2.0);
    *left = (int)nearbyint((rw - crop) / 2.0);
    if (crop_pos > 0) {
        *top = (crop_pos == 3 || crop_pos == 4) ? rh - crop : 0;
        *left = (crop_pos == 2 || crop_pos == 4) ? rw - crop : 0;
    }
}

// crop_pos: 0 centre (the reference's CenterCrop), 1 top-left, 2 top-right, 3 bottom-left, 4 bottom-right of the resized frame
// (the four corner crops of torchvision FiveCrop; build-defined 5-crop extension of BASELINE config 5)
pvr_status launch_preprocess(const uint8_t *frames, int n, int h, int w, int resize, int crop, void *out,
                             int dtype, hipStream_t stream, int crop_pos) {
    PVR_REQUIRE(n > 0 && h > 0 && w > 0, "preprocess: bad shape n=%d h=%d w=%d", n, h, w);
    PreP p;
    p.src = frames; p.dst = (u16 *)out; p.n = n; p.h = h; p.w = w; p.crop = crop;
    resized_size(h, w, resize, &p.rh, &p.rw);
    PVR_REQUIRE(p.rh >= crop && p.rw >= crop, "preprocess: resized frame %dx%d smaller than crop %d", p.rh, p.rw, crop);
    p.resize_needed = (p.rh != h || p.rw != w);
    // CenterCrop: int(round((H - crop) / 2.0)) with Python's round-half-even
    p.top = (int)nearbyint((p.rh - crop) / 2.0);
    p.left = (int)nearbyint((p.rw - crop) / 2.0);
    PVR_REQUIRE(crop_pos >= 0 && crop_pos <= 4, "preprocess: crop position %d outside 0..4", crop_pos);
    if (crop_pos > 0) {
        p.top = (crop_pos == 3 || crop_pos == 4) ? p.rh - crop : 0;
        p.left = (crop_pos == 2 || crop_pos == 4) ? p.rw - crop : 0;
    }
    p.scale_h = (float)h / (float)p.rh;
    p.scale_w = (float)w / (float)p.rw;
    dim3 grid((crop + 63) / 64, (crop + 3) / 4, n);
    if (dtype == PVR_F16) hipLaunchKernelGGL(preprocess_kernel<true>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(preprocess_kernel<false>, grid, dim3(256), 0, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

}  // namespace pvr
