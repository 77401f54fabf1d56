// Host side of the "embeddings streamed from host frames" path: pageable (possibly strided) uint8 frames -> a pinned staging buffer, by
// native threads.  The reference hands torch a pageable NumPy array per batch (save_embedded_obs.py:148-154: .to(device) of a slice);
// here the frames cross PCIe by DMA from page-locked memory, and what feeds that DMA is a memory copy that one core cannot sustain
// (one memcpy thread moves ~3 GB/s on the bench host; 70 k frames/s of 256 x 256 x 3 frames need 13.8 GB/s).  The copy also gathers the
// one strided layout the drivers produce: a 3-channel plane of a (N, H, W, 3F) scene (runs of 3 bytes every 3F bytes).
#include <atomic>
#include <thread>
#include <vector>
#include "common.h"

namespace pvr {

// runs of 3 bytes every `stride` bytes -> contiguous; n runs.  Four runs per step: 4-byte loads (the 4th byte is discarded), 12 bytes out.
// The last run of a call is copied byte-wise, so nothing past src + (n - 1) * stride + 3 is read.
static void gather3(uint8_t *dst, const uint8_t *src, int64_t n, int64_t stride) {
    int64_t i = 0;
    for (; i + 5 <= n; i += 4) {                             // (i + 4 < n: the 4th run's 4-byte load stays inside the next run)
        uint32_t a, b, c, d;
        memcpy(&a, src + (i + 0) * stride, 4); memcpy(&b, src + (i + 1) * stride, 4);
        memcpy(&c, src + (i + 2) * stride, 4); memcpy(&d, src + (i + 3) * stride, 4);
        a &= 0xffffffu; b &= 0xffffffu; c &= 0xffffffu; d &= 0xffffffu;
        const uint64_t lo = (uint64_t)a | ((uint64_t)b << 24) | ((uint64_t)(c & 0xffffu) << 48);
        const uint32_t hi = (c >> 16) | (d << 8);
        memcpy(dst + 3 * i, &lo, 8); memcpy(dst + 3 * i + 8, &hi, 4);
    }
    for (; i < n; ++i) { dst[3 * i] = src[i * stride]; dst[3 * i + 1] = src[i * stride + 1]; dst[3 * i + 2] = src[i * stride + 2]; }
}

}  // namespace pvr

// rows x row_bytes contiguous bytes at dst  <-  row r = runs of run_bytes bytes every run_stride bytes starting at src + r * src_row_stride
// (run_bytes == row_bytes: plain rows).  Rows are split over `threads` native threads (>= 1); the calling thread works too.
extern "C" pvr_status pvr_stage_copy(void *dst, const void *src, int64_t rows, int64_t row_bytes, int64_t src_row_stride, int64_t run_bytes,
                                     int64_t run_stride, int32_t threads) {
    PVR_REQUIRE(dst && src && rows >= 0 && row_bytes > 0 && run_bytes > 0 && row_bytes % run_bytes == 0 && run_stride >= run_bytes && src_row_stride >= 0,
                "pvr_stage_copy: invalid geometry (rows %lld, row bytes %lld, run %lld every %lld)", (long long)rows, (long long)row_bytes,
                (long long)run_bytes, (long long)run_stride);
    uint8_t *d = (uint8_t *)dst;
    const uint8_t *s = (const uint8_t *)src;
    const int64_t runs = row_bytes / run_bytes;
    auto one = [&](int64_t r) {
        uint8_t *o = d + r * row_bytes;
        const uint8_t *i = s + r * src_row_stride;
        if (runs == 1) memcpy(o, i, (size_t)row_bytes);
        else if (run_bytes == 3) pvr::gather3(o, i, runs, run_stride);
        else for (int64_t k = 0; k < runs; ++k) memcpy(o + k * run_bytes, i + k * run_stride, (size_t)run_bytes);
    };
    // contiguous source rows: hand out pieces of ~1 MiB instead of whole rows (few large rows would leave threads idle)
    if (runs == 1 && src_row_stride == row_bytes) {
        const int64_t total = rows * row_bytes, piece = 1 << 20, np = (total + piece - 1) / piece;
        int nt = threads < 1 ? 1 : threads;
        if (nt > np) nt = np > 0 ? (int)np : 1;
        std::atomic<int64_t> next(0);
        auto work = [&] { for (int64_t p = next.fetch_add(1); p < np; p = next.fetch_add(1)) memcpy(d + p * piece, s + p * piece, (size_t)((p + 1) * piece <= total ? piece : total - p * piece)); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        return PVR_OK;
    }
    int nt = threads < 1 ? 1 : threads;
    if (nt > rows) nt = rows > 0 ? (int)rows : 1;
    std::atomic<int64_t> next(0);
    auto work = [&] { for (int64_t r = next.fetch_add(1); r < rows; r = next.fetch_add(1)) one(r); };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return PVR_OK;
}
