// Launch parameters shared by the two forms of the fused bottleneck tail (bottleneck_chain.hip, chain_wave.hip).
#pragma once
#include "common.h"

namespace pvr {

struct ChainP {
    const u16 *in, *w2, *w3, *w1n, *res;
    const float *b2, *b3, *b1n;
    u16 *y, *t1n;
    int N, H, W, Ho, Wo, stride, M;
    unsigned in_bytes, w2_bytes, w3_bytes, w1n_bytes, y_bytes, t1n_bytes;
    // DS form (block 0 of layer1): the identity branch is a 1x1 stride-1 convolution of the block input x (64 channels); it is
    // accumulated into conv3's fp32 accumulators (a K extension of 64) instead of being read back as a 16-bit residual tensor
    const u16 *xds = nullptr, *wds = nullptr;      // x [M][64]; Wd [4Cm][64] with W3's row permutation; b3 then holds b3 + bd
    unsigned xds_bytes = 0, wds_bytes = 0;
    // wave form: t1 + residual (in) / y + t1' (out) in the blocked layout [pixel >> 4][channel >> 3][pixel & 15][8] (chain_wave.hip)
    int in_blk = 0, out_blk = 0;
    int res_blk = 0, y_blk = 0;                    // block form: the residual / y in that layout (t1 and t1' stay NHWC)
    // ... and the row-permuted W3 / Wd in that layout ([row >> 4][channel >> 3][row & 15][8]) for the instances that read them from L2
    const u16 *w3b = nullptr, *wdsb = nullptr;
    // layer2 wave form (chain_wave128.hip): the launch's 17 weight units as LDS images (launch_chain_wave128_pack); t_blk: the block form writes t1' blocked for it
    const u16 *wpk = nullptr;
    unsigned wpk_bytes = 0;
    int t_blk = 0;
};

bool chain_supported(int cm, int cmn);
bool chain_ds_supported(int cm, int cmn, int cin, int stride);
bool chain_uses_wave_form(int cm, int cmn, int stride, bool ds);   // PVR_CHAIN_WAVE (default 1) and an instance exists; read when a plan is built
int chain_row_source(int row);
pvr_status launch_bottleneck_chain(const void *t1, const void *w2, const float *b2, const void *w3p, const float *b3, const void *res,
                                   void *y, const void *w1np, const float *b1n, void *t1n, int n, int h, int w, int cm, int cmn,
                                   int stride, int dtype, hipStream_t stream, const void *xds = nullptr, const void *wdsp = nullptr,
                                   const void *w3pb = nullptr, const void *wdspb = nullptr, int wave = 0, int in_blk = 0, int out_blk = 0,
                                   const void *wpk = nullptr);


// chain_wave.hip: the barrier-free form (stride-1 blocks with Cm = 64)
bool chain_wave_supported(int cm, int cmn, int stride, bool ds);
bool chain_wave_blocked_ok(int cmn_first, int h, int w);
bool chain_wave_halo_enabled();                    // PVR_CHAIN_WAVE_HALO != 0
pvr_status launch_chain_wave(ChainP &p, int cmn, int dtype, hipStream_t stream);

// chain_wave128.hip: the wave form of layer2's stride-1 tails (Cm = 128): wave-owned pixels, weights streamed through an LDS ring
bool chain_wave128_supported(int cm, int cmn, int stride, int64_t M);
bool chain_uses_wave128(int cm, int cmn, int stride, int64_t M);     // ... and PVR_CHAIN_WAVE != 0 (the all-block-form A/B baseline); read when a plan is built
size_t chain_wave128_pack_bytes();
pvr_status launch_chain_wave128_pack(const void *w2, const void *w3p, const void *w1np, void *out, hipStream_t stream);
pvr_status launch_chain_wave128(ChainP &p, int cmn, int dtype, hipStream_t stream);

}  // namespace pvr
