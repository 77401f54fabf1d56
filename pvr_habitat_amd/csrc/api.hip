// C-ABI glue: error state, version, single-operator entry points (include/pvr_hip.h).
#include <stdarg.h>
#include "common.h"
#include "sample_rng.h"
#include <dlfcn.h>

namespace pvr {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const std::string &last_error() { return g_err; }

namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)();
struct Roctx {
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    Roctx() {
        const char *e = getenv("PVR_ROCTX");
        if (!e || atoi(e) == 0) return;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
                pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char *name) { if (roctx().push) roctx().push(name); }
void trace_pop() { if (roctx().pop) roctx().pop(); }

pvr_status launch_preprocess(const uint8_t *, int, int, int, int, int, void *, int, hipStream_t, int crop_pos = 0);
pvr_status launch_stem(const void *, const void *, const float *, void *, int, int, int, hipStream_t);
pvr_status launch_maxpool(const void *, void *, int, int, int, int, int, hipStream_t);
pvr_status launch_avgpool(const void *, float *, int64_t, int, int, int, int, int, hipStream_t);
bool stem_pool_u8_ok(const void *, int, int, int, int);
pvr_status launch_conv(const void *, const void *, const float *, const void *, void *, const void *, int, int, int, int,
                       int, int, int, int, int, int, int, int, hipStream_t);

void set_conv_algo(int a);
long long conv_expand_launches();
long long bneck_frame_launches();
pvr_status launch_pack_frag_weights(const void *w, void *out, int rows, int K, hipStream_t stream);
pvr_status launch_bneck_frame(const void *t1, const void *w2, const float *b2, const void *w3, const float *b3, const void *res, void *y,
                              void *t2_out, int n, int phases, int dtype, hipStream_t stream, unsigned long long *stamps = nullptr,
                              const void *w1np = nullptr, const float *b1n = nullptr, void *t1n = nullptr, const void *w1fp = nullptr, const float *b1f = nullptr);
pvr_status launch_bneck_frame64(const void *w1p, const float *b1, const void *w2p, const float *b2, const void *w3p, const float *b3, const void *x, void *y, int n,
                                int dtype, hipStream_t stream, unsigned long long *stamps);
void set_frame64(int mode);
long long bneck_frame64_launches();
long long pp_persistent_launches();
pvr_status launch_conv_pp256(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                             int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, int bm, hipStream_t stream,
                             const void *in2 = nullptr, int h2 = 0, int w2 = 0, int cin2 = 0, int stride2 = 1);
long long conv_wfrag_launches();
pvr_status launch_conv_wfrag(const void *in, const void *wp, const float *bias, const void *res, void *out, int n, int h, int w, int cin, int cout,
                             int kh, int kw, int stride, int pad, int act, int out_f32, int dtype, hipStream_t stream, float *pool_out = nullptr,
                             int64_t pool_stride = 0);

void set_stem_regpool(int v);
long long conv_split16_launches();
long long chain_wave128_launches();
pvr_status launch_split16_pack(const float *w, void *out, int rows, int K, hipStream_t stream);
pvr_status launch_conv_split16(const float *in, const void *wsp, const float *bias, const float *res, float *out, int n, int h, int w, int cin,
                               int cout, int k, int stride, int pad, int relu, hipStream_t stream, float *out2 = nullptr, int n1 = 0, void *out16 = nullptr,
                               int terms = 3);
pvr_status launch_conv_f32(const float *, const float *, const float *, const float *, float *, int, int, int, int, int, int, int, int, int, hipStream_t);

static void *g_zero = nullptr;
static pvr_status zero_page(void **out) {
    if (!g_zero) {
        PVR_HIP_TRY(hipMalloc(&g_zero, 256));
        PVR_HIP_TRY(hipMemset(g_zero, 0, 256));
        PVR_HIP_TRY(hipDeviceSynchronize());
    }
    *out = g_zero;
    return PVR_OK;
}

}  // namespace pvr

using namespace pvr;

extern "C" {

const char *pvr_version(void) { return "pvr_hip 0.1.0 (gfx950)"; }

int64_t pvr_debug_conv_expand_launches(void) { return (int64_t)conv_expand_launches(); }
int64_t pvr_debug_bneck_frame_launches(void) { return (int64_t)bneck_frame_launches(); }
// (rows, K) 16-bit weights in pvr_op_conv2d's layout -> the fragment-blocked layout pvr_op_bneck_frame reads (same size; rows % 32 == 0, K % 32 == 0)
pvr_status pvr_op_pack_frag_weights(const void *w, void *out, int32_t rows, int32_t k, void *stream) {
    return launch_pack_frag_weights(w, out, rows, k, (hipStream_t)stream);
}

// single-operator entry point of the per-frame fused layer3 bottleneck tail (bneck_frame.hip), for the op-level parity tests
pvr_status pvr_op_bneck_frame(const void *t1, const void *w2, const float *b2, const void *w3, const float *b3, const void *residual, void *y,
                              void *t2_out, const void *w1n, const float *b1n, void *t1n, const void *w1f, const float *b1f, int32_t n, int32_t phases,
                              int32_t dtype, void *stream) {
    PVR_REQUIRE(n > 0 && n <= 1300, "pvr_op_bneck_frame: n must be 1..1300");
    return launch_bneck_frame(t1, w2, b2, w3, b3, residual, y, t2_out, n, phases, dtype, (hipStream_t)stream, nullptr, w1n, b1n, t1n, w1f, b1f);
}
// the same launch with s_memtime stamps of block 8 (20 x uint64 on the device: waves 0 and 4, ten phase boundaries each) - diagnostics only
pvr_status pvr_debug_bneck_frame_stamps(const void *t1, const void *w2, const float *b2, const void *w3, const float *b3, const void *residual, void *y,
                                        const void *w1n, const float *b1n, void *t1n, const void *w1f, const float *b1f, int32_t n, int32_t dtype,
                                        uint64_t *stamps_dev, void *stream) {
    PVR_REQUIRE(stamps_dev && n > 8, "pvr_debug_bneck_frame_stamps: needs a stamp buffer and more than 8 frames");
    return launch_bneck_frame(t1, w2, b2, w3, b3, residual, y, nullptr, n, (w1n ? 7 : 3) | (w1f ? 8 : 0), dtype, (hipStream_t)stream, (unsigned long long *)stamps_dev,
                              w1n, b1n, t1n, w1f, b1f);
}
// round 6: which kernel runs the whole-bottleneck frame launches (pvr_op_bneck_frame with w1f, the plan's bneck_frame(front1) launches): 1 the 64-channel tiling
// (bneck_frame64.hip), 0 bneck_frame_kernel<.., FRONT1>, -1 back to the environment (PVR_FRAME64, default 0: bit-identical, measured slower).  Same bits either way.  Process-global.
pvr_status pvr_debug_set_frame64(int32_t mode) {
    PVR_REQUIRE(mode >= -1 && mode <= 1, "pvr_debug_set_frame64: -1 (environment: PVR_FRAME64, default off), 0 or 1");
    set_frame64(mode);
    return PVR_OK;
}
int64_t pvr_debug_bneck_frame64_launches(void) { return (int64_t)bneck_frame64_launches(); }
// the 64-channel tiling with s_memtime stamps of workgroup 8, wave 0 (8 x uint64: start, front conv1 done, conv2 loop done, t2 written, chunk 0 K loop, chunk 0
// epilogue, all issued, stores drained) - diagnostics only
pvr_status pvr_debug_bneck_frame64_stamps(const void *w1f, const float *b1f, const void *w2, const float *b2, const void *w3, const float *b3, const void *x, void *y,
                                          int32_t n, int32_t dtype, uint64_t *stamps_dev, void *stream) {
    PVR_REQUIRE(stamps_dev && n > 8, "pvr_debug_bneck_frame64_stamps: needs a stamp buffer and more than 8 frames");
    return launch_bneck_frame64(w1f, b1f, w2, b2, w3, b3, x, y, n, dtype, (hipStream_t)stream, (unsigned long long *)stamps_dev);
}
// single-operator entry point of the small-M implicit-GEMM kernel with weights as L2 fragments (conv_wfrag.hip), for the op-level parity tests
pvr_status pvr_op_conv_wfrag(const void *in, const void *wgt_packed, const float *bias, const void *residual, void *out, int32_t n, int32_t h, int32_t w,
                             int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t relu, int32_t out_f32, int32_t dtype,
                             void *stream) {
    PVR_REQUIRE(n > 0 && h > 0 && w > 0, "pvr_op_conv_wfrag: empty input");
    return launch_conv_wfrag(in, wgt_packed, bias, residual, out, n, h, w, cin, cout, kh, kw, stride, pad, relu, out_f32, dtype, (hipStream_t)stream);
}
// conv_pp256's two-operand form (conv3 & downsample in one accumulation), for the op-level parity tests
pvr_status pvr_op_conv2d_dual(const void *in, const void *in2, const void *wgt, const float *bias, void *out, int32_t n, int32_t h, int32_t w, int32_t cin,
                              int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t h2, int32_t w2, int32_t cin2, int32_t stride2,
                              int32_t relu, int32_t dtype, void *stream) {
    PVR_REQUIRE(in && in2 && wgt && bias && out && n > 0, "pvr_op_conv2d_dual: null argument");
    PVR_REQUIRE(dtype == PVR_BF16 || dtype == PVR_F16, "pvr_op_conv2d_dual: 16-bit storage types only");
    return launch_conv_pp256(in, wgt, bias, nullptr, out, n, h, w, cin, cout, kh, kw, stride, pad, relu, 0, 0, dtype, 224, (hipStream_t)stream, in2, h2, w2,
                             cin2, stride2);
}
// the pooled form of conv_wfrag: a 1 x 1 convolution on 7 x 7 maps + identity + ReLU whose only output is the average over each frame's 49 pixels
pvr_status pvr_op_conv_wfrag_pool(const void *in, const void *wgt_packed, const float *bias, const void *residual, float *pool_out, int64_t pool_stride,
                                  int32_t n, int32_t cin, int32_t cout, int32_t dtype, void *stream) {
    PVR_REQUIRE(n > 0 && pool_out, "pvr_op_conv_wfrag_pool: empty input");
    return launch_conv_wfrag(in, wgt_packed, bias, residual, nullptr, n, 7, 7, cin, cout, 1, 1, 1, 0, 1, 1, dtype, (hipStream_t)stream, pool_out, pool_stride);
}
int64_t pvr_debug_conv_wfrag_launches(void) { return (int64_t)conv_wfrag_launches(); }
// fp32 convolution on the 16-bit matrix pipe (conv_split16.hip): weights packed once, fp32 NHWC activations in and out
pvr_status pvr_op_split16_pack_weights(const float *w, void *out, int32_t rows, int32_t k, void *stream) {
    return launch_split16_pack(w, out, rows, k, (hipStream_t)stream);
}
pvr_status pvr_op_conv2d_split16(const float *in, const void *wgt_split, const float *bias, const float *residual, float *out, int32_t n, int32_t h,
                                 int32_t w, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t relu, void *stream) {
    PVR_REQUIRE(n > 0 && h > 0 && w > 0, "pvr_op_conv2d_split16: empty input");
    return launch_conv_split16(in, wgt_split, bias, residual, out, n, h, w, cin, cout, k, stride, pad, relu, (hipStream_t)stream);
}
// the same convolution on the f32-input MFMA (conv_f32.hip: the kernel of the PVR_F32 reference-precision mode), fp32 weights (cout_pad, k*k*cin)
pvr_status pvr_op_conv2d_f32(const float *in, const float *wgt, const float *bias, const float *residual, float *out, int32_t n, int32_t h, int32_t w,
                             int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t relu, void *stream) {
    PVR_REQUIRE(in && wgt && bias && out && n > 0, "pvr_op_conv2d_f32: null argument");
    return launch_conv_f32(in, wgt, bias, residual, out, n, h, w, cin, cout, k, stride, pad, relu, (hipStream_t)stream);
}
int64_t pvr_debug_conv_split16_launches(void) { return (int64_t)conv_split16_launches(); }
pvr_status pvr_debug_set_stem_regpool(int32_t mode) {
    PVR_REQUIRE(mode >= -1 && mode <= 1, "pvr_debug_set_stem_regpool: -1 (environment: PVR_STEM_REGPOOL, default on), 0 (LDS-tile pooling) or 1 (register pooling)");
    set_stem_regpool(mode);
    return PVR_OK;
}
int64_t pvr_debug_chain_wave128_launches(void) { return (int64_t)chain_wave128_launches(); }
int64_t pvr_debug_pp_persistent_launches(void) { return (int64_t)pp_persistent_launches(); }

size_t pvr_last_error(char *buf, size_t cap) {
    const std::string &e = last_error();
    if (buf && cap) {
        size_t n = e.size() < cap - 1 ? e.size() : cap - 1;
        memcpy(buf, e.data(), n);
        buf[n] = 0;
    }
    return e.size();
}

pvr_status pvr_op_preprocess(const uint8_t *frames, int32_t n, int32_t h, int32_t w, int32_t resize, int32_t crop,
                             void *out, int32_t dtype, void *stream) {
    PVR_REQUIRE(frames && out, "pvr_op_preprocess: null pointer");
    hipStream_t st = (hipStream_t)stream;
    PVR_HIP_TRY(hipMemsetAsync(out, 0, (size_t)n * (crop + 6) * (crop + 8) * 4 * 2, st));
    return launch_preprocess(frames, n, h, w, resize, crop, out, dtype, st);
}

pvr_status pvr_op_stem(const void *img, const void *wgt, const float *bias, void *out, int32_t n, int32_t dtype,
                       void *stream) {
    PVR_REQUIRE(img && wgt && bias && out, "pvr_op_stem: null pointer");
    return launch_stem(img, wgt, bias, out, n, 224, dtype, (hipStream_t)stream);
}

pvr_status pvr_op_maxpool(const void *in, void *out, int32_t n, int32_t h, int32_t w, int32_t c, int32_t dtype,
                          void *stream) {
    PVR_REQUIRE(in && out, "pvr_op_maxpool: null pointer");
    return launch_maxpool(in, out, n, h, w, c, dtype, (hipStream_t)stream);
}

pvr_status pvr_op_conv2d(const void *in, const void *wgt, const float *bias, const void *residual, void *out, int32_t n,
                         int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                         int32_t pad, int32_t relu, int32_t out_f32, int32_t dtype, void *stream) {
    PVR_REQUIRE(in && wgt && bias && out, "pvr_op_conv2d: null pointer");
    void *z;
    pvr_status s = zero_page(&z);
    if (s) return s;
    return launch_conv(in, wgt, bias, residual, out, z, n, h, w, cin, cout, kh, kw, stride, pad, relu, out_f32, dtype,
                       (hipStream_t)stream);
}

pvr_status pvr_debug_set_conv_algo(int32_t algo) {
    PVR_REQUIRE(algo >= -1 && algo <= 4, "pvr_debug_set_conv_algo: algo must be -1 (auto), 0 (conv_igemm), 1 / 2 / 3 (conv_pp256 with 256- / 128- / 224-pixel tiles), 4 (conv_w4)");
#ifndef PVR_EXPERIMENTS
    PVR_REQUIRE(algo != 4, "pvr_debug_set_conv_algo: conv_w4 is an experiment kernel; this library was built without it (make EXPERIMENTS=1)");
#endif
    set_conv_algo(algo);
    return PVR_OK;
}

pvr_status pvr_op_avgpool(const void *in, float *out, int64_t out_stride, int32_t n, int32_t hw, int32_t c, int32_t in_f32,
                          int32_t dtype, void *stream) {
    PVR_REQUIRE(in && out, "pvr_op_avgpool: null pointer");
    return launch_avgpool(in, out, out_stride, n, hw, c, in_f32, dtype, (hipStream_t)stream);
}

// Finite check of an embedding block ON THE DEVICE: rows x cols fp32 with row stride `stride`; *flag (a device int32 the caller zeroed) is
// set to 1 if any value is inf / NaN.  stream_embed checks every batch this way and reads the flag once at the end, instead of a host pass over
// the whole result (np.isfinite over 31 310 floats x N frames of the 5-crop uber PVR was 19 % of that leg's wall clock).
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float *x, long long rows, long long cols, long long stride, int *flag) {
    const long long n = rows * cols;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[(i / cols) * stride + (i % cols)];
        bad |= (__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0x7f800000u;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

pvr_status pvr_op_nonfinite_flag(const float *x, int64_t rows, int64_t cols, int64_t stride, int32_t *flag, void *stream) {
    PVR_REQUIRE(x && flag && rows >= 0 && cols > 0 && stride >= cols, "pvr_op_nonfinite_flag: bad argument");
    if (rows == 0) return PVR_OK;
    const long long n = rows * cols;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, (long long)cols, (long long)stride, flag);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

int32_t pvr_debug_stem_u8_geometry_ok(const void *frames, int32_t h, int32_t w, int32_t top, int32_t left) { return stem_pool_u8_ok(frames, h, w, top, left) ? 1 : 0; }

// test hook: the uniform in (0, 1) the action sampler makes of 32 random bits (sample_rng.h) - host arithmetic, no GPU work
float pvr_debug_sample_uniform(uint32_t bits) { return pvr::uniform_open01(bits); }

// 1 when the library carries the round-3 experiment kernels (conv_w4, split-bf16 GEMM, fused / persistent BPTT: make EXPERIMENTS=1), else 0
int32_t pvr_has_experiments(void) {
#ifdef PVR_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

// host-only helper used by the CPU tests: the weight conversion finalize() applies
pvr_status pvr_debug_convert(const float *src, uint16_t *dst, int64_t n, int32_t dtype) {
    PVR_REQUIRE(src && dst, "pvr_debug_convert: null pointer");
    for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_h(src[i], dtype);
    return PVR_OK;
}

}  // extern "C"
