// Diagnostic: where one whole tile iteration of the PERSISTENT conv_pp256 goes (s_memtime at the iteration's boundaries), on the ViT-B/16
// QKV geometry (M = 256 x 197 rows, N = 2304, K = 768: 7 rounds of 12 K tiles).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPP_TSTAMP=3 scripts/pp256_tile_stamps.hip -o scripts/build/pp256_tile_stamps
#include "../pvr_habitat_amd/csrc/conv_pp256.hip"
#include <stdarg.h>
#include <vector>
#include <random>
namespace pvr {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
}
int main(int argc, char **argv) {
    using namespace pvr;
    const int n = 256, h = 197, w = 1, cin = argc > 1 ? atoi(argv[1]) : 768, cout = argc > 2 ? atoi(argv[2]) : 2304, k = 1;
    const int res32 = argc > 3 ? atoi(argv[3]) : 0;      // 1: fp32 output accumulated in place (out += ..., the transformer's residual stream)
    const size_t xin = (size_t)n * h * w * cin, wn = (size_t)cout * cin, on = (size_t)n * h * w * cout;
    std::vector<u16> hx(xin), hw(wn);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : hx) v = f32_to_bf16_bits(nd(rng));
    for (auto &v : hw) v = f32_to_bf16_bits(nd(rng) * 0.02f);
    u16 *dx, *dw, *dout; float *db;
    hipMalloc(&dx, xin * 2); hipMalloc(&dw, wn * 2); hipMalloc(&dout, on * 4); hipMemset(dout, 0, on * 4); hipMalloc(&db, cout * 4);
    hipMemcpy(dx, hx.data(), xin * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wn * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, cout * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) if (launch_conv_pp256(dx, dw, db, res32 ? (void *)dout : nullptr, dout, n, h, w, cin, cout, k, k, 1, 0, 0, res32, res32, PVR_BF16, cout == 768 ? 224 : 256, 0)) return 1;
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 50; ++rep) if (launch_conv_pp256(dx, dw, db, res32 ? (void *)dout : nullptr, dout, n, h, w, cin, cout, k, k, 1, 0, 0, res32, res32, PVR_BF16, cout == 768 ? 224 : 256, 0)) return 1;
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[2][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(pvr::pp_tstamps), sizeof st);
    printf("K %d N %d: %.1f us per launch\n", cin, cout, ms / 50 * 1e3);
    for (int g = 0; g < 2; ++g)
        printf("group %d, tile iteration %d: drain wait %6llu | barriers %6llu | K loop %6llu | next tile setup + prologue DMA issue %6llu | epilogue %6llu (bias + pixel tile 0: %llu, tiles 1-3: %llu, tiles 4-: %llu) | total %6llu cycles\n", g,
               PP_TSTAMP, st[g][1] - st[g][0], st[g][2] - st[g][1], st[g][3] - st[g][2], st[g][4] - st[g][3], st[g][5] - st[g][4], st[g][6] - st[g][4], st[g][7] - st[g][6], st[g][5] - st[g][7], st[g][5] - st[g][0]);
    return 0;
}
