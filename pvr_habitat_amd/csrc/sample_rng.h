// Action sampling of PolicyNet.forward in training mode (reference src/models.py:78-80: torch.multinomial(F.softmax(policy_logits), 1)).
// One sample of softmax(l) per row through the Gumbel-max identity - argmax_a (l[a] - log(-log u_a)) is distributed as softmax(l) - with
// u from a counter-based Philox-4x32-10 stream keyed by the caller's seed and counted by (call, row, action / 4): the same (seed, call,
// row) gives the same action on any grid, nothing is stored between calls but the call counter.  Shared by the HIP plan (heads_kernel)
// and the host plan (host_policy.hip).  torch's own generator stream cannot be reproduced (its offsets depend on torch's launch
// geometry); the reference's contract here is the distribution, which tests/test_gpu_policy.py checks.
#pragma once
#include <cstdint>
#include <cmath>

namespace pvr {

__host__ __device__ inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// 32 random bits -> a uniform in the OPEN interval (0, 1): 23 bits, so that x + 0.5 is exact in fp32 (x < 2^23) and the result is never 0 or 1
// (with 24 bits, 2^24 - 1 + 0.5 rounds to 2^24: u = 1, -log(-log u) = +inf and that action would win whatever the logits say)
__host__ __device__ inline float uniform_open01(uint32_t bits) { return ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// one sample of softmax(l[0..A)), A <= 16
__host__ __device__ inline int sample_softmax_row(const float *l, int A, uint64_t seed, uint64_t call, uint64_t row) {
    int best = 0;
    float bv = 0.f;
    for (int a0 = 0; a0 < A; a0 += 4) {
        uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32) ^ ((uint32_t)(a0 >> 2) << 24), (uint32_t)call, (uint32_t)(call >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int j = 0; j < 4 && a0 + j < A; ++j) {
            const float u = uniform_open01(c[j]);
            const float v = l[a0 + j] - logf(-logf(u));
            if ((a0 + j) == 0 || v > bv) { bv = v; best = a0 + j; }
        }
    }
    return best;
}

}  // namespace pvr
