// Stand-alone A/B of the two forms of the fused layer1 bottleneck tail: bottleneck_chain.hip (block form) vs chain_wave.hip (wave form).
// Checks the outputs bit for bit (y and t1') and times both at batch 256.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/chain_wave_bench.hip -o scripts/build/chain_wave_bench
//   scripts/build/chain_wave_bench [frames=256] [reps=20]
#include "../pvr_habitat_amd/csrc/bottleneck_chain.hip"
#include "../pvr_habitat_amd/csrc/chain_wave.hip"
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include <random>
#include <algorithm>
namespace pvr {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
}
using namespace pvr;

// NHWC [M][C] <-> blocked [M / 16][C / 8][16][8] (chain_wave.hip)
static std::vector<u16> to_blocked(const std::vector<u16> &a, size_t M, int C) {
    std::vector<u16> b(a.size());
    for (size_t m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) b[((m >> 4) * (C / 8) + (c >> 3)) * 128 + (m & 15) * 8 + (c & 7)] = a[m * C + c];
    return b;
}
static std::vector<u16> from_blocked(const std::vector<u16> &b, size_t M, int C) {
    std::vector<u16> a(b.size());
    for (size_t m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) a[m * C + c] = b[((m >> 4) * (C / 8) + (c >> 3)) * 128 + (m & 15) * 8 + (c & 7)];
    return a;
}

static int run_case(int n, int cmn, bool ds, int dtype, int reps, bool ib = false, bool ob = false) {
    const int h = 56, w = 56, cm = 64, c4 = 256;
    const size_t px = (size_t)n * h * w;
    std::mt19937 rng(1 + cmn + (ds ? 7 : 0));
    std::normal_distribution<float> nd(0.f, 1.f);
    auto rnd = [&](size_t cnt, float sc, bool pos) { std::vector<u16> v(cnt); for (auto &x : v) { float f = nd(rng) * sc; if (pos && f < 0) f = 0; x = f32_to_h(f, dtype); } return v; };
    auto up = [&](const std::vector<u16> &v) { u16 *d; hipMalloc(&d, v.size() * 2 + 64); hipMemcpy(d, v.data(), v.size() * 2, hipMemcpyHostToDevice); return d; };
    auto upf = [&](size_t cnt) { std::vector<float> v(cnt); for (auto &x : v) x = nd(rng) * 0.2f; float *d; hipMalloc(&d, cnt * 4); hipMemcpy(d, v.data(), cnt * 4, hipMemcpyHostToDevice); return d; };
    const std::vector<u16> h_t1 = rnd(px * cm, 1.f, true), h_res = rnd(px * c4, 1.f, true);
    u16 *t1 = up(h_t1), *res = up(h_res);
    u16 *t1b = ib ? up(to_blocked(h_t1, px, cm)) : nullptr, *resb = ib ? up(to_blocked(h_res, px, c4)) : nullptr;
    const std::vector<u16> h_w3 = rnd((size_t)c4 * cm, 0.1f, false), h_wds = rnd((size_t)c4 * 64, 0.1f, false);
    u16 *w2 = up(rnd((size_t)cm * 9 * cm, 0.04f, false)), *w3 = up(h_w3), *w3b = up(to_blocked(h_w3, c4, cm));
    u16 *w1 = cmn ? up(rnd((size_t)cmn * c4, 0.06f, false)) : nullptr;
    u16 *xds = ds ? up(rnd(px * 64, 1.f, true)) : nullptr, *wds = ds ? up(h_wds) : nullptr, *wdsb = ds ? up(to_blocked(h_wds, c4, 64)) : nullptr;
    float *b2 = upf(cm), *b3 = upf(c4), *b1 = upf(cmn ? cmn : 1);
    u16 *y[2], *t1n[2];
    for (int k = 0; k < 2; ++k) {
        hipMalloc(&y[k], px * c4 * 2); hipMemset(y[k], 0x5a, px * c4 * 2);
        hipMalloc(&t1n[k], px * (cmn ? cmn : 1) * 2); hipMemset(t1n[k], 0x5a, px * (cmn ? cmn : 1) * 2);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < reps + 3; ++rep) {
            if (rep == 3) hipEventRecord(e0, 0);
            pvr_status s;
            if (form == 0) {
                s = launch_bottleneck_chain(t1, w2, b2, w3, b3, ds ? nullptr : res, y[0], w1, cmn ? b1 : nullptr, cmn ? t1n[0] : nullptr, n, h, w, cm, cmn, 1, dtype, 0, xds, wds);
            } else {
                ChainP p;
                p.in = ib ? t1b : t1; p.w2 = w2; p.w3 = w3; p.w1n = w1; p.res = ib ? resb : res; p.in_blk = ib; p.out_blk = ob; p.b2 = b2; p.b3 = b3; p.b1n = b1; p.y = y[1]; p.t1n = t1n[1];
                p.N = n; p.H = h; p.W = w; p.Ho = h; p.Wo = w; p.stride = 1; p.M = (int)px;
                p.in_bytes = (unsigned)(px * cm * 2); p.y_bytes = (unsigned)(px * c4 * 2); p.t1n_bytes = (unsigned)(px * (cmn ? cmn : 1) * 2);
                p.w2_bytes = cm * 9 * cm * 2; p.w3_bytes = c4 * cm * 2; p.w1n_bytes = cmn * c4 * 2;
                p.xds = xds; p.wds = wds; p.w3b = w3b; p.wdsb = wdsb; p.xds_bytes = ds ? (unsigned)(px * 128) : 0; p.wds_bytes = ds ? c4 * 128 : 0;
                s = launch_chain_wave(p, cmn, dtype, 0);
            }
            if (s) { fprintf(stderr, "launch failed (form %d)\n", form); return 1; }
        }
        hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "sync failed (form %d): %s\n", form, hipGetErrorString(hipGetLastError())); return 1; }
        hipEventElapsedTime(&ms[form], e0, e1);
        ms[form] /= reps;
    }
    std::vector<u16> a(px * c4), b(px * c4);
    hipMemcpy(a.data(), y[0], a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y[1], b.size() * 2, hipMemcpyDeviceToHost);
    if (ob) b = from_blocked(b, px, c4);
    size_t bad_y = 0, first_y = 0, nz = 0;
    for (size_t i = 0; i < a.size(); ++i) { if (a[i] != b[i]) { if (!bad_y) first_y = i; ++bad_y; } nz += a[i] != 0; }
    size_t bad_t = 0, first_t = 0;
    if (cmn) {
        std::vector<u16> c(px * cmn), d(px * cmn);
        hipMemcpy(c.data(), t1n[0], c.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(d.data(), t1n[1], d.size() * 2, hipMemcpyDeviceToHost);
        if (ob) d = from_blocked(d, px, cmn);
        for (size_t i = 0; i < c.size(); ++i) if (c[i] != d[i]) { if (!bad_t) first_t = i; ++bad_t; }
    }
    const double bytes = (double)px * ((ds ? 128 : 512) + 128 + 512 + cmn * 2);
    printf("n=%d cmn=%d ds=%d %s blk %d/%d: block form %.1f us (%.2f TB/s), wave form %.1f us (%.2f TB/s); mismatches y %zu / %zu (first at pixel %zu ch %zu), t1' %zu (first %zu); %.0f %% of y non-zero\n",
           n, cmn, (int)ds, dtype == PVR_F16 ? "f16" : "bf16", (int)ib, (int)ob, ms[0] * 1e3, bytes / ms[0] / 1e9, ms[1] * 1e3, bytes / ms[1] / 1e9, bad_y, a.size(), first_y / c4, first_y % c4,
           bad_t, first_t, 100.0 * nz / a.size());
    for (void *q : {(void *)w3b, (void *)wdsb, (void *)t1b, (void *)resb, (void *)t1, (void *)res, (void *)w2, (void *)w3, (void *)w1, (void *)xds, (void *)wds, (void *)b2, (void *)b3, (void *)b1, (void *)y[0], (void *)y[1], (void *)t1n[0], (void *)t1n[1]})
        if (q) hipFree(q);
    return (bad_y || bad_t) ? 2 : 0;
}

int main(int argc, char **argv) {
    setenv("PVR_CHAIN_WAVE", "0", 1);                       // launch_bottleneck_chain stays on the block form; the wave form is called directly
    const int n = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 20;
    int rc = 0;
    rc |= run_case(3, 64, false, PVR_BF16, 2);
    rc |= run_case(3, 64, true, PVR_F16, 2);
    rc |= run_case(5, 128, false, PVR_F16, 2);
    rc |= run_case(1, 0, false, PVR_BF16, 2);
    rc |= run_case(3, 64, false, PVR_F16, 2, true, true);
    rc |= run_case(3, 64, false, PVR_BF16, 2, false, true);
    rc |= run_case(3, 64, false, PVR_BF16, 2, true, false);
    rc |= run_case(3, 64, true, PVR_BF16, 2, false, true);
    rc |= run_case(3, 64, true, PVR_F16, 2, true, true);
    rc |= run_case(5, 128, false, PVR_BF16, 2, true, false);
    rc |= run_case(1, 0, false, PVR_F16, 2, true, false);
    rc |= run_case(n, 64, false, PVR_BF16, reps);
    rc |= run_case(n, 64, false, PVR_BF16, reps, true, true);
    rc |= run_case(n, 64, true, PVR_BF16, reps);
    rc |= run_case(n, 64, true, PVR_BF16, reps, false, true);
    rc |= run_case(n, 64, true, PVR_BF16, reps, true, true);
    rc |= run_case(n, 128, false, PVR_BF16, reps);
    rc |= run_case(n, 128, false, PVR_BF16, reps, true, false);
    rc |= run_case(n, 0, false, PVR_BF16, reps, true, false);
    printf(rc ? "FAILED\n" : "all bit-identical\n");
    return rc;
}
