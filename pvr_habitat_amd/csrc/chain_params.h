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
};

// chain_wave.hip: the barrier-free form (stride-1 blocks with Cm = 64)
bool chain_wave_supported(int cm, int cmn, int stride, bool ds);
pvr_status launch_chain_wave(ChainP &p, int cmn, int dtype, hipStream_t stream);

}  // namespace pvr
