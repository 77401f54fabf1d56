// Diagnostic: where one steady-state K tile of conv_pp256 spends its cycles (s_memtime stamps, one wave per group).
//   for p in 0 1 2 3: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPP_STAMP=8 -DPP_PHASE=$p scripts/pp256_stamps.hip -o /tmp/pp256_stamps_$p
#include "../pvr_habitat_amd/csrc/conv_pp256.hip"
#include <stdarg.h>
#include <vector>
#include <random>
namespace pvr {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
}
int main() {
    using namespace pvr;
    const int n = 256, h = 14, w = 14, cin = 256, cout = 256, k = 3;            // layer3 conv2
    const size_t xin = (size_t)n * h * w * cin, wn = (size_t)cout * k * k * cin, on = (size_t)n * h * w * cout;
    std::vector<u16> hx(xin), hw(wn);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : hx) v = f32_to_bf16_bits(nd(rng));
    for (auto &v : hw) v = f32_to_bf16_bits(nd(rng) * 0.02f);
    u16 *dx, *dw, *dout; float *db;
    hipMalloc(&dx, xin * 2); hipMalloc(&dw, wn * 2); hipMalloc(&dout, on * 2); hipMalloc(&db, cout * 4);
    hipMemcpy(dx, hx.data(), xin * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wn * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, cout * 4);
    for (int rep = 0; rep < 200; ++rep)
        if (launch_conv_pp256(dx, dw, db, nullptr, dout, n, h, w, cin, cout, k, k, 1, 1, 1, 0, 0, PVR_BF16, 256, 0)) return 1;
    hipDeviceSynchronize();
    unsigned long long st[2][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(pvr::pp_stamps), sizeof st);
    // stamps of phase PP_PHASE: 0 phase start, 1 reads + DMA issued, 2 after lgkmcnt(0), 3 after barrier A, 4 MFMAs issued, 5 after barrier B
    for (int g = 0; g < 2; ++g)
        printf("phase %d group %d: issue %5llu | lgkm wait %5llu | barrier A %5llu | math %5llu | barrier B %5llu | total %5llu\n", PP_PHASE, g,
               st[g][1] - st[g][0], st[g][2] - st[g][1], st[g][3] - st[g][2], st[g][4] - st[g][3], st[g][5] - st[g][4], st[g][5] - st[g][0]);
    // whole K loop of that block: s_memtime ticks vs s_memrealtime (100 MHz) -> in-kernel clock
    for (int g = 0; g < 2; ++g)
        printf("group %d: K loop %llu s_memtime ticks in %.2f us -> %.0f MHz; %.0f ticks per K tile (36 tiles; MFMA-issue floor 2048 cycles)\n", g,
               st[g][6], st[g][7] / 100.0, st[g][6] / (st[g][7] / 100.0), st[g][6] / 36.0);
    return 0;
}
