// BC batch assembly on the device (replaces the host loop of reference main_bc_2.py:186-204 / main_bc_finetune.py:173-188):
//     for i in starting_i:  rows = mod(arange(i, i + T), n_samples);  o.append(obs[rows]) ...;  np.stack(o, axis=1)
// The whole pre-embedded dataset lives in HBM (288 GB: 10^6 samples x 4096 fp32 = 16 GB), so one launch gathers the (T, B) batch
// - observations, actions and dones - from the B start indices; nothing crosses PCIe per iteration except 8 * B bytes of indices.
// HBM-bound: T*B rows of row_bytes are read once and written once (26 MB + 26 MB for T=100, B=16, 4096 floats).
#include "common.h"
#include "../../include/pvr_policy.h"

namespace pvr {

// one workgroup per (t, b) output row; 16-byte lanes; row_bytes % 16 == 0
__global__ __launch_bounds__(256) void bc_gather_kernel(const uint8_t *__restrict__ obs, const long long *__restrict__ action,
                                                       const uint8_t *__restrict__ done, long long n_samples, long long row_bytes,
                                                       const long long *__restrict__ starts, int T, int B, uint8_t *__restrict__ out_obs,
                                                       long long *__restrict__ out_action, uint8_t *__restrict__ out_done) {
    const int row = blockIdx.x, t = row / B, b = row % B;
    long long src = (starts[b] + t) % n_samples;
    if (src < 0) src += n_samples;
    const u32x4 *s = reinterpret_cast<const u32x4 *>(obs + src * row_bytes);
    u32x4 *d = reinterpret_cast<u32x4 *>(out_obs + (long long)row * row_bytes);
    const long long n16 = row_bytes >> 4;
    for (long long i = threadIdx.x; i < n16; i += 256) d[i] = s[i];
    if (threadIdx.x == 0) {
        if (out_action) out_action[row] = action[src];
        if (out_done) out_done[row] = done[src];
    }
}

}  // namespace pvr

extern "C" pvr_status pvr_bc_gather(const void *obs_dev, const int64_t *action_dev, const uint8_t *done_dev, int64_t n_samples,
                                    int64_t row_bytes, const int64_t *starts_dev, int32_t T, int32_t B, void *out_obs,
                                    int64_t *out_action, uint8_t *out_done, void *hip_stream) {
    PVR_REQUIRE(obs_dev && starts_dev && out_obs, "pvr_bc_gather: null argument");
    PVR_REQUIRE((out_action == nullptr) == (action_dev == nullptr) && (out_done == nullptr) == (done_dev == nullptr),
                "pvr_bc_gather: action / done inputs and outputs go together");
    PVR_REQUIRE(n_samples > 0 && T > 0 && B > 0, "pvr_bc_gather: n_samples=%lld T=%d B=%d", (long long)n_samples, T, B);
    PVR_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "pvr_bc_gather: row_bytes %lld must be a positive multiple of 16", (long long)row_bytes);
    hipLaunchKernelGGL(pvr::bc_gather_kernel, dim3((unsigned)(T * B)), dim3(256), 0, (hipStream_t)hip_stream, (const uint8_t *)obs_dev,
                       (const long long *)action_dev, done_dev, (long long)n_samples, (long long)row_bytes, (const long long *)starts_dev, T, B,
                       (uint8_t *)out_obs, (long long *)out_action, out_done);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}
