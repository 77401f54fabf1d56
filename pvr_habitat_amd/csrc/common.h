// Shared host/device helpers for libpvr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <atomic>
#include "../../include/pvr_hip.h"

namespace pvr {
constexpr size_t PVR_ZERO_BYTES = 16384;   // size of an encoder's zero page (>= 4 * the widest Cout of a split-K launch)


// ---- error plumbing (thread-local message, integer status across the ABI) -------------------
void set_error(const char *fmt, ...);
// Optional roctx ranges around the C-ABI entry points (PVR_ROCTX=1; librocprofiler-sdk-roctx / libroctx64 is dlopen'ed on first use, so the
// library has no link-time dependency on a profiler).  `rocprofv3 --marker-trace --kernel-trace` then groups the launches per call.
void trace_push(const char *name);
void trace_pop();
struct TraceScope {
    explicit TraceScope(const char *name) { trace_push(name); }
    ~TraceScope() { trace_pop(); }
};
const std::string &last_error();

#define PVR_HIP_TRY(expr)                                                                      \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            pvr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                           __LINE__);                                                          \
            return PVR_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define PVR_LAUNCH_CHECK()                                                                     \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess) {                                                                \
            pvr::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),          \
                           __FILE__, __LINE__);                                                \
            return PVR_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define PVR_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            pvr::set_error(__VA_ARGS__);                                                       \
            return PVR_ERR_INVALID;                                                            \
        }                                                                                      \
    } while (0)

// "done once per device" flag for hipFuncSetAttribute(MaxDynamicSharedMemorySize): function attributes are per device, and a process that
// later uses a second GPU must raise the limit there too (a process-wide bool would leave that device's launches failing).  Bit d of the
// mask = device d; concurrent first calls at worst set the attribute twice.
// Cache policy of the pure streams (outputs written once, residuals read once): raw-buffer aux 2 = nt (streaming: the line is the
// first to leave L2), so that the rows a kernel re-reads (3x3 halos, the resident weights) stay.  PVR_NT is a build-time mask for
// A/B runs (profiles/experiments/r04_nt_streams.txt): 1 chain_wave stores, 2 chain_wave residual loads, 4 / 8 bottleneck_chain y /
// t1' stores, 16 its residual loads, 32 / 64 conv_expand stores / residual loads, 128 stem stores, 256 / 512 conv_pp256 stores /
// residual loads, 1024 the register-pooling stem's uint8 frame reads (round 6).
#ifndef PVR_NT
#define PVR_NT 119  // 1+2 wave stores / residual, 4+16 block-form y stores / residual, 32+64 conv_expand with a residual (layer3 / layer4 conv3)
#endif
#define PVR_NT_AUX(bit_) ((PVR_NT & (bit_)) ? 2 : 0)

struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    static int device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? (d & 63) : 0; }
    bool needed() const { return !((mask.load(std::memory_order_relaxed) >> device()) & 1ull); }
    void mark() { mask.fetch_or(1ull << device(), std::memory_order_relaxed); }
};

// ---- 16-bit storage types --------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// host-side round-to-nearest-even conversions (weights are converted once at finalize)
inline u16 f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
inline u16 f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (u16)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (u16)(sign | 0x7c00u);  // overflow -> inf (after rounding)
    if (x < 0x38800000u) {                               // subnormal / zero in f16
        if (x < 0x33000000u) return (u16)sign;
        int e = (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        int shift = 126 - e;                             // 14..24
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (u16)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13);
    uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (u16)(sign | r);
}
inline u16 f32_to_h(float f, int dtype) { return dtype == PVR_F16 ? f32_to_f16_bits(f) : f32_to_bf16_bits(f); }

// ---- device helpers --------------------------------------------------------------------------
template <bool F16> struct HT;
template <> struct HT<false> { typedef bf16_t T; typedef bf16x8 V8; };
template <> struct HT<true> { typedef f16_t T; typedef f16x8 V8; };

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(typename HT<F16>::V8 a, typename HT<F16>::V8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// GELU (erf form: timm's nn.GELU in the MAE blocks, reference src/vision_models/mae.py:85-93) for the GEMM epilogues.  erf through
// Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. <= 7.5e-8 |x| in the result: four orders of magnitude below the 16-bit
// rounding that follows): one v_rcp, one v_exp and five FMAs instead of the device library's erff, which cost 40 % of the FC1 launch.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * __expf(-x * x);
    return 0.5f * v * (1.f + (v < 0.f ? -erf_abs : erf_abs));
}

template <bool F16> __device__ __forceinline__ u16 to_h(float v) {
    typename HT<F16>::T t = (typename HT<F16>::T)v;
    return __builtin_bit_cast(u16, t);
}
// two floats -> one dword of two 16-bit values (round to nearest even, as to_h): ONE v_cvt_pk instruction.  `to_h(a) | to_h(b) << 16` makes
// hipcc convert with its own pairing and then re-pack with three more instructions per dword.
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pk_f16x2 __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ unsigned pack2_h(float a, float b) {
    const pk_f32x2 v = {a, b};
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2));
}
template <bool F16> __device__ __forceinline__ float from_h(u16 b) {
    return (float)__builtin_bit_cast(typename HT<F16>::T, b);
}

// bijective XCD-aware block remap (cdna_hip_programming.md, 256^2 template): blocks b and b+8 share
// an XCD; give every XCD a contiguous run of tiles so neighbouring tiles share operand panels in L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

}  // namespace pvr
