// Diagnostic: where one image-tile of stem_pool_lds_kernel spends its cycles (s_memtime stamps of one block's eight waves).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSTEM_STAMP scripts/stem_stamps.hip -o /tmp/stem_stamps && /tmp/stem_stamps
#include "../pvr_habitat_amd/csrc/stem.hip"
#include <stdarg.h>
#include <vector>
namespace pvr {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const std::string &last_error() { static std::string s; return s; }
}
int main() {
    using namespace pvr;
    const int n = 256;
    const size_t img = (size_t)n * 230 * 232 * 4, wn = (size_t)64 * STEM_K, on = (size_t)n * 56 * 56 * 64;
    std::vector<u16> hi(img), hw(wn);
    unsigned s = 1;
    for (auto &v : hi) { s = s * 1664525u + 1013904223u; v = f32_to_bf16_bits(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = f32_to_bf16_bits((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.1f); }
    u16 *di, *dw, *dout; float *db;
    hipMalloc(&di, img * 2); hipMalloc(&dw, wn * 2); hipMalloc(&dout, on * 2); hipMalloc(&db, 64 * 4);
    hipMemcpy(di, hi.data(), img * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wn * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, 64 * 4);
    for (int rep = 0; rep < 100; ++rep)
        if (launch_stem_pool(di, dw, db, dout, n, 224, PVR_BF16, 0)) return 1;
    hipDeviceSynchronize();
    static long long st[8][64][5];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(pvr::stem_stamps), sizeof st);
    printf("per image of block (13, 1): cycles  [wait rows | 2 barriers | MFMA phase | barrier | pooling + loop]  waves 0, 3, 6, 7\n");
    for (int i = 1; i < 12; ++i) {
        printf("image %2d:", i);
        for (int w : {0, 3, 6, 7})
            printf("  %5lld %5lld %6lld %5lld %6lld |", st[w][i][1] - st[w][i][0], st[w][i][2] - st[w][i][1], st[w][i][3] - st[w][i][2],
                   st[w][i][4] - st[w][i][3], st[w][i + 1][0] - st[w][i][4]);
        printf("  total %lld\n", st[0][i + 1][0] - st[0][i][0]);
    }
    return 0;
}
