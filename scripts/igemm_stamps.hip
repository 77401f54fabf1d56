// Diagnostic: where one block of conv_igemm spends its time for the layer3 conv3 shape (1x1, 256 -> 1024, + residual, 14x14, batch 256).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DIGEMM_STAMP scripts/igemm_stamps.hip -o /tmp/igemm_stamps
#include "../pvr_habitat_amd/csrc/conv_igemm.hip"
#include <stdarg.h>
#include <vector>
#include <random>
#include <algorithm>
namespace pvr {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
bool pp256_supported(int64_t, int, int, int, int, int64_t, int64_t, int64_t, int64_t) { return false; }
pvr_status launch_conv_pp256(const void *, const void *, const float *, const void *, void *, int, int, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t) { return PVR_ERR_INVALID; }
}
int main() {
    using namespace pvr;
    const int n = 256, h = 14, w = 14, cin = 256, cout = 1024;
    const size_t px = (size_t)n * h * w;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto rnd = [&](size_t cnt, float sc) { std::vector<u16> v(cnt); for (auto &x : v) x = f32_to_bf16_bits(nd(rng) * sc); return v; };
    auto up = [&](const std::vector<u16> &v) { u16 *d; hipMalloc(&d, v.size() * 2); hipMemcpy(d, v.data(), v.size() * 2, hipMemcpyHostToDevice); return d; };
    u16 *x = up(rnd(px * cin, 1.f)), *res = up(rnd(px * cout, 1.f)), *wt = up(rnd((size_t)cout * cin, 0.06f));
    u16 *y, *zero; hipMalloc(&y, px * cout * 2); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
    float *b; hipMalloc(&b, cout * 4); hipMemset(b, 0, cout * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int rep = 0; rep < reps + 3; ++rep) {
        if (rep == 3) hipEventRecord(e0, 0);
        if (launch_conv(x, wt, b, res, y, zero, n, h, w, cin, cout, 1, 1, 1, 0, 1, 0, PVR_BF16, 0)) { fprintf(stderr, "launch failed\n"); return 1; }
    }
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int grid = (int)((px + 127) / 128) * (cout / 128), nb = std::min(grid, 8192);
    std::vector<unsigned long long> st((size_t)8192 * 6);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(pvr::igemm_stamps), st.size() * 8);
    double seg[4] = {0, 0, 0, 0}, tot = 0;
    for (int b_ = 0; b_ < nb; ++b_) {
        const unsigned long long *q = &st[(size_t)b_ * 6];
        for (int k = 0; k < 4; ++k) seg[k] += (double)(q[k + 1] - q[k]);
        tot += (double)(q[4] - q[0]);
    }
    printf("conv3 1x1 256->1024 +res, 14x14, batch 256: %.3f ms per launch, grid %d\n", ms / reps, grid);
    const char *nm[4] = {"prologue: addresses, loads issued, slice 0 -> LDS", "K loop (4 slices)", "accumulators + bias -> LDS (fp32), barrier", "read back, + residual, ReLU, 16-B stores"};
    for (int k = 0; k < 4; ++k) printf("  %-52s %7.2f us  (%4.1f %%)\n", nm[k], seg[k] / nb / 100.0, 100.0 * seg[k] / tot);
    printf("  block total %.2f us\n", tot / nb / 100.0);
    return 0;
}
