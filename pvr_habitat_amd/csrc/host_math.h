// Host-side arithmetic shared by the CPU backends (host_encoder.hip, host_policy.hip): a thread fan-out, a 4 x 4 register-blocked dot
// product kernel over 8-wide fp32 vectors (AVX2 + FMA when the CPU has them, chosen at run time), and the three GEMM forms the policy needs.
#pragma once
#include <atomic>
#include <thread>
#include <vector>
#include <string.h>

namespace pvr {

typedef float v8f __attribute__((vector_size(32)));

template <class F>
static void host_parallel_for(int n, int threads, F f) {
    if (threads > n) threads = n;
    if (threads <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next(0);
    auto work = [&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}

// out[p][co] = sum_k A[p][k] * W[co][k] for p < np (<= 4), co < nc (<= 4): a 4 x 4 block of 8-wide partial sums, K in steps of 8, tail scalar
#define PVR_HOST_DOT_BODY                                                                                          \
    v8f acc[4][4];                                                                                                 \
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v8f{0, 0, 0, 0, 0, 0, 0, 0};               \
    int k = 0;                                                                                                     \
    if (np == 4 && nc == 4) {                                                                                      \
        for (; k + 8 <= K; k += 8) {                                                                               \
            v8f a[4], w[4];                                                                                        \
            for (int i = 0; i < 4; ++i) memcpy(&a[i], A[i] + k, 32);                                               \
            for (int j = 0; j < 4; ++j) memcpy(&w[j], W[j] + k, 32);                                               \
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * w[j];                      \
        }                                                                                                          \
    } else {                                                                                                       \
        for (; k + 8 <= K; k += 8)                                                                                 \
            for (int i = 0; i < np; ++i) {                                                                         \
                v8f a; memcpy(&a, A[i] + k, 32);                                                                   \
                for (int j = 0; j < nc; ++j) { v8f w; memcpy(&w, W[j] + k, 32); acc[i][j] += a * w; }              \
            }                                                                                                      \
    }                                                                                                              \
    for (int i = 0; i < np; ++i)                                                                                   \
        for (int j = 0; j < nc; ++j) {                                                                             \
            float s = 0.f;                                                                                         \
            for (int e = 0; e < 8; ++e) s += acc[i][j][e];                                                         \
            for (int kk = k; kk < K; ++kk) s += A[i][kk] * W[j][kk];                                               \
            out[i][j] = s;                                                                                         \
        }
static void host_dot_block_generic(const float *const *A, const float *const *W, int K, int np, int nc, float out[4][4]) { PVR_HOST_DOT_BODY }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2,fma"))) static void host_dot_block_avx2(const float *const *A, const float *const *W, int K, int np, int nc, float out[4][4]) { PVR_HOST_DOT_BODY }
static void host_dot_block(const float *const *A, const float *const *W, int K, int np, int nc, float out[4][4]) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (avx2) host_dot_block_avx2(A, W, K, np, nc, out);
    else host_dot_block_generic(A, W, K, np, nc, out);
}
#else
static void host_dot_block(const float *const *A, const float *const *W, int K, int np, int nc, float out[4][4]) { host_dot_block_generic(A, W, K, np, nc, out); }
#endif
#undef PVR_HOST_DOT_BODY


inline int host_threads() {
    static const int n = [] {
        if (const char *t = getenv("PVR_HOST_THREADS")) return atoi(t) > 0 ? atoi(t) : 1;
        const unsigned hc = std::thread::hardware_concurrency();
        return hc ? (int)hc : 1;
    }();
    return n;
}

// C[M][N] = A[M][K] . B[N][K]^T (+ bias[N]) (relu): both operands K-contiguous
static void host_gemm_nt(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, bool relu, bool accumulate = false) {
    const int tiles = (M + 3) / 4;
    host_parallel_for(tiles, host_threads(), [&](int t) {
        const int m0 = t * 4, np = M - m0 < 4 ? M - m0 : 4;
        const float *ar[4];
        for (int i = 0; i < np; ++i) ar[i] = A + (size_t)(m0 + i) * K;
        for (int n0 = 0; n0 < N; n0 += 4) {
            const int nc = N - n0 < 4 ? N - n0 : 4;
            const float *br[4];
            for (int j = 0; j < nc; ++j) br[j] = B + (size_t)(n0 + j) * K;
            float o[4][4];
            host_dot_block(ar, br, K, np, nc, o);
            for (int i = 0; i < np; ++i)
                for (int j = 0; j < nc; ++j) {
                    float v = o[i][j] + (bias ? bias[n0 + j] : 0.f);
                    if (accumulate) v += C[(size_t)(m0 + i) * N + n0 + j];
                    C[(size_t)(m0 + i) * N + n0 + j] = relu ? (v > 0.f ? v : 0.f) : v;
                }
        }
    });
}

// C[M][K] (+)= A[M][N] . B[N][K]   (input gradients: dX = dY . W)
static void host_gemm_nn(const float *A, const float *B, float *C, int M, int N, int K, bool accumulate) {
    host_parallel_for(M, host_threads(), [&](int m) {
        float *c = C + (size_t)m * K;
        if (!accumulate) memset(c, 0, (size_t)K * 4);
        for (int n = 0; n < N; ++n) {
            const float a = A[(size_t)m * N + n];
            if (a == 0.f) continue;
            const float *b = B + (size_t)n * K;
            for (int k = 0; k < K; ++k) c[k] += a * b[k];
        }
    });
}

// C[N][K] = A[M][N]^T . B[M][K]   (weight gradients: dW = dY^T . X), rows of C split over the threads (no two threads share a row)
static void host_gemm_tn(const float *A, const float *B, float *C, int M, int N, int K) {
    host_parallel_for(N, host_threads(), [&](int n) {
        float *c = C + (size_t)n * K;
        memset(c, 0, (size_t)K * 4);
        for (int m = 0; m < M; ++m) {
            const float a = A[(size_t)m * N + n];
            if (a == 0.f) continue;
            const float *b = B + (size_t)m * K;
            for (int k = 0; k < K; ++k) c[k] += a * b[k];
        }
    });
}

}  // namespace pvr
