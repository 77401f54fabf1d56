// Micro-benchmark: what HBM rate do the residual-load / y-store access patterns of bottleneck_chain's phase B reach on their own?
//   A  current tiling: waves 2x2 (64 pixels x 32 couts of a 64-cout group): a wave instruction touches 16 pixel rows x 64 B
//   B  waves 4x1 (32 pixels x 64 couts): the two 64-B halves of a row's 128 B come from one wave, back to back
//   C  ceiling: consecutive lanes -> consecutive 16 B
// layer1 shape: M = 256*56*56 pixels, C4 = 256 channels (512 B per pixel), y = relu(res + 1) so the data is touched.
// build: hipcc --offload-arch=gfx950 -O3 scripts/store_pattern_bw.hip -o /tmp/spbw && /tmp/spbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int C4 = 256, BM = 128;

template <int PAT>
__global__ __launch_bounds__(256) void k(const unsigned short *__restrict__ res, unsigned short *__restrict__ y, int M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)blockIdx.x * BM;
    if (PAT == 2) {
        const u32x4 *s = reinterpret_cast<const u32x4 *>(res + m0 * C4);
        u32x4 *d = reinterpret_cast<u32x4 *>(y + m0 * C4);
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = s[i * 256 + tid];
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i][0] += 1; d[i * 256 + tid] = v[i]; }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long off;
            if (PAT == 0) { const int wm = wave >> 1, wn = wave & 1; off = (m0 + wm * 64 + j * 16 + fr) * C4 + g * 64 + wn * 32 + fq * 8; }
            else { off = (m0 + wave * 32 + (j >> 1) * 16 + fr) * C4 + g * 64 + (j & 1) * 32 + fq * 8; }
            v[j] = *reinterpret_cast<const u32x4 *>(res + off);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long off;
            if (PAT == 0) { const int wm = wave >> 1, wn = wave & 1; off = (m0 + wm * 64 + j * 16 + fr) * C4 + g * 64 + wn * 32 + fq * 8; }
            else { off = (m0 + wave * 32 + (j >> 1) * 16 + fr) * C4 + g * 64 + (j & 1) * 32 + fq * 8; }
            v[j][0] += 1;
            *reinterpret_cast<u32x4 *>(y + off) = v[j];
        }
    }
}

int main() {
    const int M = 256 * 56 * 56;
    const size_t bytes = (size_t)M * C4 * 2;
    unsigned short *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"A waves 2x2 (64 B per row per wave instruction)", "B waves 4x1 (128 B per row from one wave)", "C fully coalesced"};
    for (int rep = 0; rep < 2; ++rep)
    for (int pat = 0; pat < 3; ++pat) {
        for (int it = 0; it < 12; ++it) {
            if (it == 2) hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(M / BM), dim3(256), 0, 0, a, b, M);
            if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(M / BM), dim3(256), 0, 0, a, b, M);
            if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(M / BM), dim3(256), 0, 0, a, b, M);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %.3f ms per launch, %.2f TB/s (read + write %.0f MB)\n", names[pat], ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e12, 2.0 * bytes / 1e6);
    }
    // size sweep, coalesced pattern: how much of a short launch is ramp / tail?  (layer3 conv3 moves 103 MB in + 103 MB out + 26 MB)
    for (size_t mb : {26, 52, 103, 206, 411}) {
        const int Mx = (int)(mb * 1000000 / (C4 * 2) / BM * BM);
        for (int it = 0; it < 12; ++it) {
            if (it == 2) hipEventRecord(e0);
            hipLaunchKernelGGL(k<2>, dim3(Mx / BM), dim3(256), 0, 0, a, b, Mx);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("coalesced copy of %4zu MB (+ %4zu MB written): %.1f us per launch, %.2f TB/s\n", mb, mb, ms * 100, 2.0 * Mx * C4 * 2 / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
