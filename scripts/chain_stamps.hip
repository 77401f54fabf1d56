// Diagnostic: where one block of the fused bottleneck tail spends its time (s_memrealtime stamps, thread 0 of every block).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCHAIN_STAMP scripts/chain_stamps.hip -o /tmp/chain_stamps
#include "../pvr_habitat_amd/csrc/bottleneck_chain.hip"
#include <stdarg.h>
#include <vector>
#include <random>
#include <algorithm>
namespace pvr {
// (the wave form lives in chain_wave.hip; this harness times the block form only)
bool chain_wave_supported(int, int, int, bool) { return false; }
bool chain_wave_blocked_ok(int, int, int) { return false; }
bool chain_wave_halo_enabled() { return false; }
pvr_status launch_chain_wave(ChainP &, int, int, hipStream_t) { return PVR_ERR_INVALID; }
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
}
int main(int argc, char **argv) {
    using namespace pvr;
    const int layer = argc > 1 ? atoi(argv[1]) : 1;
    const bool ds = argc > 2 && atoi(argv[2]) != 0;             // layer1 block 0: downsample inside the chain (x and Wd instead of a residual tensor)
    const int n = 256, h = layer == 1 ? 56 : 28, w = h, cm = layer == 1 ? 64 : 128, cmn = cm, c4 = 4 * cm;
    const size_t px = (size_t)n * h * w;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto rnd = [&](size_t cnt, float sc) { std::vector<u16> v(cnt); for (auto &x : v) x = f32_to_bf16_bits(nd(rng) * sc); return v; };
    auto up = [&](const std::vector<u16> &v) { u16 *d; hipMalloc(&d, v.size() * 2); hipMemcpy(d, v.data(), v.size() * 2, hipMemcpyHostToDevice); return d; };
    u16 *t1 = up(rnd(px * cm, 1.f)), *res = up(rnd(px * c4, 1.f)), *w2 = up(rnd((size_t)cm * 9 * cm, 0.04f)), *w3 = up(rnd((size_t)c4 * cm, 0.1f)), *w1 = up(rnd((size_t)cmn * c4, 0.06f));
    u16 *y, *t1n; hipMalloc(&y, px * c4 * 2); hipMalloc(&t1n, px * cmn * 2);
    u16 *xds = ds ? up(rnd(px * 64, 1.f)) : nullptr, *wds = ds ? up(rnd((size_t)c4 * 64, 0.1f)) : nullptr;
    float *b; hipMalloc(&b, c4 * 4); hipMemset(b, 0, c4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int rep = 0; rep < reps + 3; ++rep) {
        if (rep == 3) hipEventRecord(e0, 0);
        if (launch_bottleneck_chain(t1, w2, b, w3, b, res, y, w1, b, t1n, n, h, w, cm, cmn, 1, PVR_BF16, 0, xds, wds)) { fprintf(stderr, "launch failed\n"); return 1; }
    }
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int grid = (int)((px + 127) / 128), nb = std::min(grid, 8192);
    std::vector<unsigned long long> st((size_t)8192 * 12);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(pvr::chain_stamps), st.size() * 8);
    double seg[5] = {0, 0, 0, 0, 0}, tot = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b_ = 0; b_ < nb; ++b_) {
        const unsigned long long *q = &st[(size_t)b_ * 12];
        for (int k = 0; k < 5; ++k) seg[k] += (double)(q[k + 1] - q[k]);
        tot += (double)(q[5] - q[0]); tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[5]);
    }
    if (ds) printf("DS form (downsample inside the chain)\n");
    printf("layer%d chain: %.3f ms per launch, grid %d; kernel span (first start .. last end of the stamped blocks) %.1f us\n", layer, ms / reps, grid, (tmax - tmin) / 100.0);
    const char *nm[5] = {"prologue (addresses, first slice -> LDS)", "phase A loop (conv2 3x3)", "t2 -> LDS, W3/W1' -> LDS, barrier", "phase B loop (conv3 + res + y, conv1')", "t1' epilogue"};
    for (int k = 0; k < 5; ++k) printf("  %-44s %7.2f us  (%4.1f %%)\n", nm[k], seg[k] / nb / 100.0, 100.0 * seg[k] / tot);
    printf("  block total %.2f us\n", tot / nb / 100.0);
    // one steady-state 64-channel group (g = 1) of phase B
    const char *gn[5] = {"conv3 MFMAs issued", "epilogue: bias + residual + y stores + y -> LDS, residual refill issued", "barrier 1", "W3[g+1] regs -> LDS (waits for the loads: vmcnt(0)), W3[g+2] loads issued", "conv1' MFMAs + barrier 2"};
    double gs[5] = {0, 0, 0, 0, 0};
    for (int b_ = 0; b_ < nb; ++b_) { const unsigned long long *q = &st[(size_t)b_ * 12]; for (int k = 0; k < 5; ++k) gs[k] += (double)(q[7 + k] - q[6 + k]); }
    for (int k = 0; k < 5; ++k) printf("    group 1: %-84s %6.2f us\n", gn[k], gs[k] / nb / 100.0);
    return 0;
}
