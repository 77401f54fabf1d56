// 3x3 / stride 1 / pad 1 convolution with Cin = Cout = 64 or 128 (+ folded-BN bias, optional 16-bit residual, optional ReLU):
// the BasicBlock convolutions of torchvision resnet18 / resnet34 (reference src/embeddings.py:112-117), the stride-1 3x3s of the
// CLIP ModifiedResNet and conv2 of a ResNet50 bottleneck when the fused tail is switched off.
//
// Same arithmetic as conv_igemm (taps outer, 64-channel slices inner, v_mfma_f32_16x16x32 in the same order: bit-identical), but the
// pixels are staged the way bottleneck_chain.hip's halo form does it: for stride 1 the input pixel of output pixel m (flattened
// n,h,w) at tap (kh,kw) is m + (kh-1)*W + (kw-1), so the 128 output pixels of a tile need ONE contiguous run of 128 + 2W + 2 input
// rows, whatever image borders it crosses.  That run is fetched once by LDS-DMA and the nine taps read their fragments from it at a
// row offset (taps outside their image are redirected to an all-zero row) - the activation goes through L2 1.9x instead of 9x,
// which is what bounded these launches (64 channels at 56x56: 925 MB of im2col reads per launch against 308 MB of HBM traffic).
// The weight slices go global -> registers (RL slices ahead) -> one of two LDS stages; plain loads, compiler-counted waits.
// LDS-DMA rule: the DMA is waited for once with vmcnt(0), then a barrier, then the address set-up, then the reads.
//
// Weight rows are permuted on their way into LDS (row 32b + 16t + 4a + c holds cout 32b + 8a + 4t + c) so that a lane's accumulators of
// an MFMA tile pair are 8 consecutive output channels of one pixel: 16-byte residual loads and stores straight from registers.
#include "common.h"

namespace pvr {

struct HaloP {
    const u16 *in, *wgt, *res;
    const float *bias;
    u16 *out;
    int H, W, M, relu;
    unsigned in_bytes, w_bytes;
};

__device__ __forceinline__ void halo_dma16(__amdgpu_buffer_rsrc_t rs, char *lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, 0);
}

template <int CM, bool F16>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloP p) {
    typedef typename HT<F16>::V8 V8;
    constexpr int BM = 128, BK = 64, B_CH = CM / 32, TM = 4, TN = CM / 32;
    constexpr int KS = CM / 64, HROWS = CM == 64 ? 256 : 192, RL = CM == 64 ? 3 : 2, nk = 9 * KS;
    constexpr int HSL = HROWS * 128, RING_OFF = KS * HSL, SLICE = CM * 128, ZERO_OFF = (HROWS - 1) * 128, HOPS = KS * HROWS / 32;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.in), 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(p.wgt), 0, p.w_bytes, 0x00020000);

    // the halo run, before anything else the block requests
    {
        const int HR = 128 + 2 * p.W + 2, hbase = m0 - p.W - 1;
#pragma unroll
        for (int i = 0; i < HOPS; ++i) {
            const int o = i * 4 + wave, sl = o / (HROWS / 8), rb = o % (HROWS / 8);
            const int r = rb * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            const int vo = (r < HR && hbase + r >= 0) ? ((hbase + r) * CM + sl * 64 + c * 8) * 2 : OOB;   // past the tensor: range miss -> zeros
            halo_dma16(rs_in, smem + sl * HSL + rb * 1024, vo);
        }
    }
    // weight staging: thread (srow, pch) moves chunk pch of LDS rows srow + 32 i; LDS row R holds cout perm(R)
    const int srow = tid >> 3, pch = tid & 7;
    int b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int row = srow + 32 * i;
        const int co = (row & ~31) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);
        b_off[i] = (co * (9 * CM) + (pch ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const int lds_st = srow * 128 + pch * 16;
    u32x4 w2r[RL][B_CH];
#pragma unroll
    for (int q = 0; q < RL; ++q)
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            w2r[q][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], q * (BK * 2), 0));

    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    // residual prefetch: the lane's 8 couts (pair bp of its wn half) of its pixel in tile j
    u32x4 rres[TN / 2][TM];
    int o_off[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        o_off[j] = m < p.M ? (m * CM + wn * (CM / 2) + fq * 8) * 2 : -1;
    }
    if (p.res) {
#pragma unroll
        for (int bp = 0; bp < TN / 2; ++bp)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                rres[bp][j] = o_off[j] >= 0 ? *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.res) + o_off[j] + bp * 64) : u32x4{0, 0, 0, 0};
    }
    int b_rd[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int row = wn * (CM / 2) + i * 16 + fr;
            b_rd[ks][i] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the halo (and everything else requested so far) has landed
#pragma unroll
    for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4 *>(smem + RING_OFF + lds_st + i * 32 * 128) = w2r[0][i];
    __syncthreads();

    int h_row[TM], h_mask[TM];                    // fragment rows inside the halo run, 9-bit "tap inside the image" masks
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pr = wm * 64 + j * 16 + fr, m = m0 + pr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int wo = mm % p.W, ho = (mm / p.W) % p.H;
        int hb = 0, wb_ = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            hb |= (int)(ok && (unsigned)(ho - 1 + t3) < (unsigned)p.H) << t3;
            wb_ |= (int)((unsigned)(wo - 1 + t3) < (unsigned)p.W) << t3;
        }
        int mask = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) mask |= ((hb >> t3) & 1) ? (wb_ << (t3 * 3)) : 0;
        h_row[j] = pr; h_mask[j] = mask;
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + RL < nk) {                       // register stage kt % RL held slice kt, which reached LDS during step kt - 1
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                w2r[kt % RL][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], (kt + RL) * (BK * 2), 0));
        }
        const int tp = kt / KS, csl = kt % KS, shift = (tp / 3) * p.W + tp % 3;
        int xo[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = h_row[j] + shift;
            const int a = csl * HSL + r * 128 + ((fq ^ ((r >> 1) & 7)) << 4);
            xo[j] = ((h_mask[j] >> tp) & 1) ? a : ZERO_OFF;
        }
        const char *ring = smem + RING_OFF + (kt & 1) * SLICE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V8 xa[TM], wb[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) xa[j] = *reinterpret_cast<const V8 *>(smem + (xo[j] ^ (ks * 64)));
#pragma unroll
            for (int i = 0; i < TN; ++i) wb[i] = *reinterpret_cast<const V8 *>(ring + b_rd[ks][i]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = mfma16<F16>(wb[i], xa[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                *reinterpret_cast<u32x4 *>(smem + RING_OFF + ((kt + 1) & 1) * SLICE + lds_st + i * 32 * 128) = w2r[(kt + 1) % RL][i];
            __syncthreads();                      // slice kt + 1 visible; every wave is done with stage kt & 1
        }
    }

    // epilogue straight from the accumulators: tile pair bp = 8 consecutive couts per lane
#pragma unroll
    for (int bp = 0; bp < TN / 2; ++bp) {
        const int co = wn * (CM / 2) + bp * 32 + fq * 8;
        const float4 bA = *reinterpret_cast<const float4 *>(p.bias + co), bB = *reinterpret_cast<const float4 *>(p.bias + co + 4);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const f32x4 lo = acc[2 * bp][j], hi = acc[2 * bp + 1][j];
            float v[8] = {lo[0] + bA.x, lo[1] + bA.y, lo[2] + bA.z, lo[3] + bA.w, hi[0] + bB.x, hi[1] + bB.y, hi[2] + bB.z, hi[3] + bB.w};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v0 = v[2 * e], v1 = v[2 * e + 1];
                if (p.res) { v0 += from_h<F16>((u16)(rres[bp][j][e] & 0xffffu)); v1 += from_h<F16>((u16)(rres[bp][j][e] >> 16)); }
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                o[e] = pack2_h<F16>(v0, v1);
            }
            if (o_off[j] >= 0) *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(p.out) + o_off[j] + bp * 64) = o;
        }
    }
}

// PVR_CONV_HALO=0: these shapes stay on conv_igemm (A/B runs; bit-identical)
static bool halo_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PVR_CONV_HALO"); v = e ? atoi(e) : 1; }
    return v != 0;
}

bool conv3x3_halo_supported(int64_t M, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32, int64_t in_bytes) {
    (void)h;
    return halo_enabled() && kh == 3 && kw == 3 && stride == 1 && pad == 1 && cin == cout && (cin == 64 || cin == 128) && relu <= 1 && out_f32 == 0 &&
           128 + 2 * w + 2 <= (cin == 64 ? 256 : 192) - 1 && M * cin * 2 < 0x7ffffff0ll && in_bytes < 0x7ffffff0ll;
}

template <int CM, bool F16>
static pvr_status launch_halo_inst(HaloP &p, hipStream_t stream) {
    const size_t lds = (size_t)(CM / 64) * (CM == 64 ? 256 : 192) * 128 + (size_t)2 * CM * 128;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_halo_kernel<CM, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<CM, F16>), dim3((p.M + 127) / 128), dim3(256), lds, stream, p);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status launch_conv3x3_halo(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int c, int relu,
                               int dtype, hipStream_t stream) {
    HaloP p;
    p.in = (const u16 *)in; p.wgt = (const u16 *)wgt; p.res = (const u16 *)res; p.bias = bias; p.out = (u16 *)out;
    p.H = h; p.W = w; p.M = n * h * w; p.relu = relu;
    p.in_bytes = (unsigned)((int64_t)p.M * c * 2); p.w_bytes = (unsigned)(c * 9 * c * 2);
    if (c == 64) return dtype == PVR_F16 ? launch_halo_inst<64, true>(p, stream) : launch_halo_inst<64, false>(p, stream);
    return dtype == PVR_F16 ? launch_halo_inst<128, true>(p, stream) : launch_halo_inst<128, false>(p, stream);
}

}  // namespace pvr
