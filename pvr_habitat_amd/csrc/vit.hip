// CLIP visual transformer (ViT-B/32, ViT-B/16) encode_image as a HIP plan.
//
// Replaces reference src/embeddings.py:298-314 (clip.load + transforms) and :375-376 (encode_image) with openai/CLIP's
// VisionTransformer restated in oracle/vit_oracle.py.  Plan per chunk of frames:
//   patchify (uint8 crop -> [N*g*g][P*P*3] 16-bit, Normalize and /255 folded into the patch-embed weights)
//   -> patch-embed GEMM (conv_igemm as a 1x1 conv over "pixels" = patches, fp32 out)
//   -> [CLS ; patches] + positional embedding -> ln_pre  (assemble_ln_kernel, fp32 residual stream)
//   -> 12 x { ln_1 -> QKV GEMM(+bias) -> attention_kernel -> out_proj GEMM(+bias, += fp32 residual)
//             ln_2 -> c_fc GEMM(+bias, QuickGELU) -> c_proj GEMM(+bias, += fp32 residual) }
//   -> ln_post(CLS) @ proj (cls_head_kernel, fp32)
// GEMMs: conv_igemm_kernel (MFMA 16x16x32, bf16 or f16 inputs, fp32 accumulate); the residual stream stays fp32.
// Attention: one workgroup per (image, head); scores are computed TRANSPOSED (S^T = K Q^T) so that the softmaxed
// accumulator tile is already the B operand of the second product (O^T = V^T P^T) — no LDS round trip for P
// (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"); softmax reductions are wave shuffles.
#include "encoder_internal.h"

namespace pvr {

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
// uint8 (n,h,w,3) centre crop -> A[(n*g*g + gy*g + gx)][(py*P + px)*3 + c] as 16-bit (exact centred values x-128)
template <bool F16>
__global__ __launch_bounds__(256) void patchify_kernel(const uint8_t *__restrict__ frames, u16 *__restrict__ A, int n, int h,
                                                       int w, int top, int left, int res, int P) {
    const int g = res / P, Kr = P * P * 3, K = (Kr + 63) / 64 * 64;   // rows padded to a multiple of 64 with zeros (patch 14: 588 -> 640)
    const size_t total = (size_t)n * g * g * (K / 8);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k8 = (int)(i % (K / 8)) * 8;
        const size_t row = i / (K / 8);
        const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
        u32x4 o;
        u16 *oe = reinterpret_cast<u16 *>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k8 + e, c = k % 3, px = (k / 3) % P, py = k / (3 * P);
            const int y = top + gy * P + py, x = left + gx * P + px;
            oe[e] = k < Kr ? to_h<F16>((float)frames[(((size_t)b * h + y) * w + x) * 3 + c] - 128.f) : (u16)0;     // centred, exact
        }
        *reinterpret_cast<u32x4 *>(A + row * K + k8) = o;
    }
}

// Antialiased bicubic Resize (embeddings.py:310: T.Resize(res, BICUBIC, antialias=True) on a uint8 tensor =
// float32 separable resampling, horizontal pass first, then vertical, clamp(0,255), round-half-even, uint8).
// Restates ATen's _upsample_bicubic2d_aa: per output index a window [xmin, xmin+xsize) and normalised weights of
// the a=-0.5 cubic evaluated at (j + xmin - center + 0.5) * invscale (tables built on the host, aa_tables()).
__global__ __launch_bounds__(256) void aa_resize_h_kernel(const uint8_t *__restrict__ src, float *__restrict__ tmp,
                                                          const int *__restrict__ xmin, const int *__restrict__ xsize,
                                                          const float *__restrict__ wts, int maxk, int n, int h, int w,
                                                          int left, int res) {
    const size_t total = (size_t)n * h * res * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % 3), xo = (int)((i / 3) % res);
        const size_t row = i / ((size_t)3 * res);                   // b*h + y
        const int X = xo + left, x0 = xmin[X], k = xsize[X];
        const uint8_t *s = src + (row * w + x0) * 3 + c;
        const float *wt = wts + (size_t)X * maxk;
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += wt[j] * (float)s[j * 3];
        tmp[i] = acc;
    }
}

__global__ __launch_bounds__(256) void aa_resize_v_kernel(const float *__restrict__ tmp, uint8_t *__restrict__ dst,
                                                          const int *__restrict__ ymin, const int *__restrict__ ysize,
                                                          const float *__restrict__ wts, int maxk, int n, int h, int top, int res) {
    const size_t total = (size_t)n * res * res * 3;
    const size_t rowlen = (size_t)res * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t xc = i % rowlen;
        const int yo = (int)((i / rowlen) % res), b = (int)(i / (rowlen * res));
        const int Y = yo + top, y0 = ymin[Y], k = ysize[Y];
        const float *t = tmp + ((size_t)b * h + y0) * rowlen + xc;
        const float *wt = wts + (size_t)Y * maxk;
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += wt[j] * t[(size_t)j * rowlen];
        acc = fminf(fmaxf(acc, 0.f), 255.f);                        // bicubic overshoots: clamp before the uint8 cast
        dst[i] = (uint8_t)rintf(acc);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per token row of width W (multiple of 256): LayerNorm (eps 1e-5, fp32) of x (+ optional assembly of the
// token sequence: row t = 0 -> cls, t > 0 -> patch_emb[n*g2 + t-1], plus pos[t]).  Writes fp32 (residual stream)
// and/or 16-bit (GEMM operand).
template <bool F16, int W>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ patch_emb,
                                                        const float *__restrict__ cls, const float *__restrict__ pos,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ out_f32, u16 *__restrict__ out_h, int rows, int T, float eps,
                                                        int normalize) {
    constexpr int PER = W / 64;                   // 12 for 768
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[PER];
    if (patch_emb) {
        const int t = row % T, b = row / T;
        const float *src = t == 0 ? cls : patch_emb + ((size_t)b * (T - 1) + t - 1) * W;
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) {
            const int k = (i * 64 + lane) * 4;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(src + k), p = *reinterpret_cast<const f32x4 *>(pos + (size_t)t * W + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i * 4 + e] = a[e] + p[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(x + (size_t)row * W + (i * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i * 4 + e] = a[e];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + eps);
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) {
        const int k = (i * 64 + lane) * 4;
        f32x4 o;
        if (normalize) {
            const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + k), bt = *reinterpret_cast<const f32x4 *>(beta + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i * 4 + e] - mean) * rstd * gm[e] + bt[e];
        } else {                                                  // MAE: tokens + pos-emb enter the blocks un-normalised
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i * 4 + e];
        }
        if (out_f32) *reinterpret_cast<f32x4 *>(out_f32 + (size_t)row * W + k) = o;
        if (out_h) {
            ushort4 r;
            r.x = to_h<F16>(o[0]); r.y = to_h<F16>(o[1]); r.z = to_h<F16>(o[2]); r.w = to_h<F16>(o[3]);
            *reinterpret_cast<ushort4 *>(out_h + (size_t)row * W + k) = r;
        }
    }
}

// Multi-head attention, head dim HD (64: CLIP / MAE-B / MAE-L; 80: MAE-H).  qkv: [N*T][3W] 16-bit (q | k | v, head h at columns
// h*HD..), out: [N*T][W].  grid (heads, N), 4 waves.  Keys are padded to TK (multiple of 32, <= 16*MAXNT): padded scores are -inf,
// padded V rows zero.  The contraction dim of QK^T is padded to KP = 32*ceil(HD/32) with zero chunks.
// EXACT: TK == 16 * MAXNT (every shape the plans use), so the key-tile count is a compile-time constant and only the last two tiles
// carry key-validity masks; the runtime-NT form cost ~40 % more VALU work per query tile (selects and moves on every score tile)
template <bool F16, int HD, int MAXNT, bool EXACT = true>
__global__ __launch_bounds__(256) void attention_kernel(const u16 *__restrict__ qkv, u16 *__restrict__ out, int T, int TK, int W) {
    typedef typename HT<F16>::V8 V8;
    constexpr int KP = (HD + 31) / 32 * 32, KCH = KP / 8, KSTEPS = KP / 32, MT = HD / 16;
    constexpr int KS = HD == 64 ? 128 : KP * 2 + 16;              // K row stride (bytes); 64: chunk-swizzled 128-B rows, else padded rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VS = TK + 4;                                        // V^T row stride (elements)
    char *Ks = smem;                                              // [TK][KP] 16-bit
    u16 *Vt = reinterpret_cast<u16 *>(smem + (size_t)TK * KS);    // [HD][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int hd = blockIdx.x, b = blockIdx.y;
    const size_t rs = (size_t)3 * W;
    const u16 *base = qkv + (size_t)b * T * rs + hd * HD;
    // branch-free loads: out-of-range rows / padded chunks take an offset past num_records and read zeros (a per-element
    // "load or zero" select makes hipcc branch around every load and serialise them)
    const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16 *>(base), 0, (unsigned)((size_t)T * rs * 2), 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    auto kaddr = [&](int row, int ch) { return HD == 64 ? row * 128 + ((ch ^ ((row >> 1) & 7)) << 4) : row * KS + (ch << 4); };
    // K / V fill.  A thread owns FOUR consecutive key rows x one 16-byte chunk: the four K chunks go to LDS as they are, the four V chunks are transposed in
    // registers (v_perm) and leave as eight 8-byte stores of four keys each - until round 5 every V element was its own 2-byte store (32 per item instead of
    // 8; round 6: no measurable change of the launch, 106 us per layer either way - the fill is not what the kernel waits for).  All global loads of a thread are issued before its first LDS write, so a wave pays the load latency once (the kernel is latency-bound:
    // SQ_WAIT_ANY 53 %, MFMA busy 6 %).
    constexpr int NG = MAXNT * 4, NITEM = NG * KCH, NIT = (NITEM + 255) / 256;      // row groups of 4 (TK <= 16 MAXNT), items, items per thread
    {
        u32x4 kv[NIT][4], vv[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it, rg = idx / KCH, ch = idx % KCH;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rg + r;
                const bool okl = idx < NITEM && row < T && row < TK && ch * 8 < HD;
                const int off = okl ? (int)(((size_t)row * rs + W + ch * 8) * 2) : OOB;
                kv[it][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, off, 0, 0));
                vv[it][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, okl ? off + W * 2 : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it, rg = idx / KCH, ch = idx % KCH;
            if (idx >= NITEM || 4 * rg >= TK) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<u32x4 *>(Ks + kaddr(4 * rg + r, ch)) = kv[it][r];
            if (ch * 8 < HD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;          // the odd / even halfword of both dwords
                    const uint2 o = {__builtin_amdgcn_perm(vv[it][1][e >> 1], vv[it][0][e >> 1], sel), __builtin_amdgcn_perm(vv[it][3][e >> 1], vv[it][2][e >> 1], sel)};
                    *reinterpret_cast<uint2 *>(Vt + (size_t)(ch * 8 + e) * VS + 4 * rg) = o;
                }
            }
        }
    }
    __syncthreads();
    const int NT = EXACT ? MAXNT : TK / 16;                       // key tiles (<= MAXNT)
    // scores are kept in the log2 domain (scale * log2(e) folded into one multiply) so the softmax exponent is a bare v_exp_f32
    const float scale = (HD == 64 ? 0.125f : 1.0f / sqrtf((float)HD)) * 1.44269504088896341f;
    // B operand of S^T = K Q^T : Q[query = fr][d = ks*32 + 8*fq + j]; the next tile's Q is fetched while this one is computed
    auto load_q = [&](int qt_, u32x4 (&t)[KSTEPS]) {
        const int q_ = qt_ * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int off = (q_ < T && ks * 32 + fq * 8 < HD) ? (int)(((size_t)q_ * rs + ks * 32 + fq * 8) * 2) : OOB;
            t[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, off, 0, 0));
        }
    };
    u32x4 qnext[KSTEPS];
    load_q(wave, qnext);
    for (int qt = wave; qt * 16 < T; qt += 4) {
        const int query = qt * 16 + fr;
        const bool qok = query < T;
        V8 qf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = __builtin_bit_cast(V8, qnext[ks]);
        load_q(qt + 4, qnext);
        f32x4 s[MAXNT];
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
            s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nt < NT) {
                const int krow = nt * 16 + fr;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const V8 kf = *reinterpret_cast<const V8 *>(Ks + kaddr(krow, ks * 4 + fq));
                    s[nt] = mfma16<F16>(kf, qf[ks], s[nt]);
                }
            }
        }
        // s[nt][r] = S^T[key = nt*16 + 4*fq + r][query = fr]; softmax over keys (scores / sqrt(HD))
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = nt * 16 + fq * 4 + r;
                float v = s[nt][r] * scale;
                if (!EXACT || nt >= MAXNT - 2) v = (nt < NT && key < T) ? v : -INFINITY;     // TK - T < 32: only the last two tiles can hold padding
                s[nt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[nt][r] - mx);    // 2^(-inf) = 0 for padded keys
                s[nt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // O^T = V^T P^T : k-slot (fq, j) <-> key 32*ks + 16*(j>>2) + 4*fq + (j&3), identical for both operands
        f32x4 o[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXNT / 2; ++ks) {
            if (ks * 32 < TK) {
                u32x4 pw;                                  // (one v_cvt_pk per dword: common.h::pack2_h)
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[e] = pack2_h<F16>(s[2 * ks + (e >> 1)][(2 * e) & 3], s[2 * ks + (e >> 1)][(2 * e + 1) & 3]);
                const V8 pf = __builtin_bit_cast(V8, pw);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u16 *vr = Vt + (size_t)(mt * 16 + fr) * VS + ks * 32 + fq * 4;
                    const uint2 lo = *reinterpret_cast<const uint2 *>(vr), hi = *reinterpret_cast<const uint2 *>(vr + 16);
                    const u32x4 vw = u32x4{lo.x, lo.y, hi.x, hi.y};
                    o[mt] = mfma16<F16>(__builtin_bit_cast(V8, vw), pf, o[mt]);
                }
            }
        }
        if (qok) {
            u16 *orow = out + ((size_t)b * T + query) * W + hd * HD + fq * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint2 r = {pack2_h<F16>(o[mt][0] * inv, o[mt][1] * inv), pack2_h<F16>(o[mt][2] * inv, o[mt][3] * inv)};
                *reinterpret_cast<uint2 *>(orow + mt * 16) = r;
            }
        }
    }
}

// ln_post(x[:,0,:]) @ proj  (W -> out_dim), one block per image, fp32
template <int W>
__global__ __launch_bounds__(256) void cls_head_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ proj,
                                                       float *__restrict__ out, int64_t out_stride, int T, int out_dim, float eps) {
    __shared__ float y[W];
    __shared__ float red[8];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *row = x + (size_t)b * T * W;
    float v[W / 256];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < W / 256; ++i) { v[i] = row[tid + i * 256]; s += v[i]; }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < W / 256; ++i) { const float d = v[i] - mean; q += d * d; }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)W + eps);
#pragma unroll
    for (int i = 0; i < W / 256; ++i) y[tid + i * 256] = (v[i] - mean) * rstd * gamma[tid + i * 256] + beta[tid + i * 256];
    __syncthreads();
    for (int j = tid; j < out_dim; j += 256) {
        float acc = 0.f;
        if (proj) for (int k = 0; k < W; ++k) acc += y[k] * proj[(size_t)k * out_dim + j];
        else acc = y[j];                                          // MAE: the normalised CLS token is the embedding
        out[(size_t)b * out_stride + j] = acc;
    }
}

}  // namespace pvr

// ------------------------------------------------------------------------------------------------------------------
// host plan
// ------------------------------------------------------------------------------------------------------------------
using namespace pvr;

struct VitBlock {
    u16 *w_qkv = nullptr, *w_out = nullptr, *w_fc = nullptr, *w_proj = nullptr;
    float *b_qkv = nullptr, *b_out = nullptr, *b_fc = nullptr, *b_proj = nullptr;
    float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
};

struct pvr_vit {
    int patch = 32, width = 768, layers = 12, heads = 12, out_dim = 512, res = 224, grid = 7, T = 50, TK = 64;
    bool mae = false;               // timm/MAE layout: patch bias, no ln_pre, LN eps 1e-6, erf GELU, CLS output without proj
    float eps = 1e-5f;
    int act = 2, resize_to = 224;
    std::vector<VitBlock> blocks;
    u16 *w_patch = nullptr;
    float *b_patch = nullptr, *cls = nullptr, *pos = nullptr, *lnpre_w = nullptr, *lnpre_b = nullptr, *lnpost_w = nullptr,
          *lnpost_b = nullptr, *proj = nullptr;
    // workspace (chunk frames)
    u16 *A = nullptr, *y = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *zero = nullptr;
    float *pe = nullptr, *x0 = nullptr, *x1 = nullptr;
    // antialiased-bicubic resize state, rebuilt when the frame size changes
    int rs_h = 0, rs_w = 0, rs_rh = 0, rs_rw = 0, rs_maxk_h = 0, rs_maxk_w = 0;
    int *rs_xmin = nullptr, *rs_xsize = nullptr, *rs_ymin = nullptr, *rs_ysize = nullptr;
    float *rs_wx = nullptr, *rs_wy = nullptr, *rs_tmp = nullptr;
    uint8_t *rs_u8 = nullptr;
    // second workspace lane (pvr_encoder_forward_lane): the members above are the CURRENT lane's pointers
    struct Ws { u16 *A = nullptr, *y = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr; float *pe = nullptr, *x0 = nullptr, *x1 = nullptr; bool valid = false; } ws[PVR_MAX_LANES];
    float *rs_tmp_l[PVR_MAX_LANES] = {nullptr};
    uint8_t *rs_u8_l[PVR_MAX_LANES] = {nullptr};
    int cur = 0;
    std::vector<void *> owned;
};

namespace pvr {

static pvr_status up_f32(pvr_encoder *e, const std::string &name, size_t numel, float **dptr) {
    const HostTensor *t;
    pvr_status s = enc_need(e, name, &t, numel);
    if (s) return s;
    if ((s = enc_upload(dptr, t->data))) return s;
    e->vit->owned.push_back(*dptr);
    return PVR_OK;
}

// nn.Linear weight [out][in] -> 16-bit rows (already K-major), rows padded to a multiple of 64
static pvr_status up_linear(pvr_encoder *e, const std::string &wname, const std::string &bname, int out_f, int in_f, u16 **dw, float **db) {
    const HostTensor *w, *b;
    pvr_status s;
    if ((s = enc_need(e, wname, &w, (size_t)out_f * in_f))) return s;
    if ((s = enc_need(e, bname, &b, (size_t)out_f))) return s;
    const int pad = (out_f + 63) / 64 * 64;
    std::vector<u16> hw((size_t)pad * in_f, 0);
    for (size_t i = 0; i < (size_t)out_f * in_f; ++i) hw[i] = f32_to_h(w->data[i], e->desc.dtype);
    std::vector<float> hb(pad, 0.f);
    for (int i = 0; i < out_f; ++i) hb[i] = b->data[i];
    if ((s = enc_upload(dw, hw))) return s;
    e->vit->owned.push_back(*dw);
    if ((s = enc_upload(db, hb))) return s;
    e->vit->owned.push_back(*db);
    return PVR_OK;
}

pvr_status vit_create(pvr_encoder *e) {
    pvr_vit *v = new pvr_vit();
    v->patch = e->desc.arch == PVR_ARCH_CLIP_VIT_B32 ? 32 : 16;
    v->mae = e->desc.arch == PVR_ARCH_MAE_VIT_B16 || e->desc.arch == PVR_ARCH_MAE_VIT_L16 || e->desc.arch == PVR_ARCH_MAE_VIT_H14;
    if (e->desc.arch == PVR_ARCH_MAE_VIT_L16) { v->width = 1024; v->layers = 24; v->heads = 16; }   // mae.py:283-288
    if (e->desc.arch == PVR_ARCH_MAE_VIT_H14) { v->width = 1280; v->layers = 32; v->heads = 16; v->patch = 14; }   // mae.py:291-296
    if (v->mae) { v->eps = 1e-6f; v->act = 3; v->out_dim = v->width; }
    v->res = e->desc.crop;
    v->resize_to = e->desc.resize;
    v->grid = v->res / v->patch;
    v->T = v->grid * v->grid + 1;
    v->TK = (v->T + 31) / 32 * 32;
    if (v->TK > 288 || v->width % v->heads || (v->width / v->heads != 64 && v->width / v->heads != 80)) {
        delete v; set_error("vit: more than 288 tokens or a head dim other than 64 / 80 is not built"); return PVR_ERR_INVALID;
    }
    e->vit = v;
    e->out_size = v->out_dim;
    return PVR_OK;
}

static pvr_status vit_alloc_ws(pvr_encoder *e) {
    pvr_vit *v = e->vit;
    const size_t C = e->desc.chunk, rows = C * v->T, prow = C * v->grid * v->grid, W = v->width, K = ((size_t)v->patch * v->patch * 3 + 63) / 64 * 64;
    auto alloc = [&](void **ptr, size_t bytes) -> pvr_status {
        PVR_HIP_TRY(hipMalloc(ptr, bytes));
        v->owned.push_back(*ptr);
        return PVR_OK;
    };
    pvr_status s;
    if ((s = alloc((void **)&v->A, prow * K * 2))) return s;
    if ((s = alloc((void **)&v->pe, prow * W * 4))) return s;
    if ((s = alloc((void **)&v->x0, rows * W * 4))) return s;
    if ((s = alloc((void **)&v->x1, rows * W * 4))) return s;
    if ((s = alloc((void **)&v->y, rows * W * 2))) return s;
    if ((s = alloc((void **)&v->qkv, rows * 3 * W * 2))) return s;
    if ((s = alloc((void **)&v->att, rows * W * 2))) return s;
    return alloc((void **)&v->hid, rows * 4 * W * 2);
}

pvr_status vit_finalize(pvr_encoder *e) {
    pvr_vit *v = e->vit;
    const int W = v->width, P = v->patch, Kr = P * P * 3, K = (Kr + 63) / 64 * 64, dt = e->desc.dtype;   // K: zero-padded row length
    pvr_status s;
    // patch embedding: conv1 [W][3][P][P] (no bias) -> [W][(py,px,c)], Normalize + /255 folded:
    //   sum w*((x/255-mean)/std) = sum (w/(255 std)) (x-128) + sum w (128 - 255 mean)/(255 std)
    const HostTensor *w;
    if ((s = enc_need(e, v->mae ? "patch_embed.proj.weight" : "visual.conv1.weight", &w, (size_t)W * Kr))) return s;
    const HostTensor *pb = nullptr;                              // timm PatchEmbed has a bias, CLIP conv1 does not
    if (v->mae && (s = enc_need(e, "patch_embed.proj.bias", &pb, (size_t)W))) return s;
    {
        std::vector<u16> hw((size_t)W * K, 0);
        std::vector<float> hb(W, 0.f);
        for (int co = 0; co < W; ++co) {
            double bsum = 0.0;
            for (int c = 0; c < 3; ++c)
                for (int py = 0; py < P; ++py)
                    for (int px = 0; px < P; ++px) {
                        const double wv = w->data[(((size_t)co * 3 + c) * P + py) * P + px];
                        hw[(size_t)co * K + (py * P + px) * 3 + c] = f32_to_h((float)(wv / (255.0 * e->desc.std_[c])), dt);
                        bsum += wv * (128.0 - 255.0 * e->desc.mean[c]) / (255.0 * e->desc.std_[c]);   // x = xc + 128
                    }
            hb[co] = (float)bsum + (pb ? pb->data[co] : 0.f);
        }
        if ((s = enc_upload(&v->w_patch, hw))) return s;
        v->owned.push_back(v->w_patch);
        if ((s = enc_upload(&v->b_patch, hb))) return s;
        v->owned.push_back(v->b_patch);
    }
    // names: openai/CLIP VisionTransformer (visual.*) or timm/MAE (mae.py:85-95)
    if ((s = up_f32(e, v->mae ? "cls_token" : "visual.class_embedding", W, &v->cls))) return s;
    if ((s = up_f32(e, v->mae ? "pos_embed" : "visual.positional_embedding", (size_t)v->T * W, &v->pos))) return s;
    if (!v->mae) {
        if ((s = up_f32(e, "visual.ln_pre.weight", W, &v->lnpre_w))) return s;
        if ((s = up_f32(e, "visual.ln_pre.bias", W, &v->lnpre_b))) return s;
        if ((s = up_f32(e, "visual.proj", (size_t)W * v->out_dim, &v->proj))) return s;
    }
    if ((s = up_f32(e, v->mae ? "norm.weight" : "visual.ln_post.weight", W, &v->lnpost_w))) return s;
    if ((s = up_f32(e, v->mae ? "norm.bias" : "visual.ln_post.bias", W, &v->lnpost_b))) return s;
    v->blocks.resize(v->layers);
    for (int i = 0; i < v->layers; ++i) {
        VitBlock &b = v->blocks[i];
        const std::string p = (v->mae ? "blocks." : "visual.transformer.resblocks.") + std::to_string(i) + ".";
        const char *n_qkv = v->mae ? "attn.qkv." : "attn.in_proj_", *n_out = v->mae ? "attn.proj." : "attn.out_proj.";
        const char *n_fc = v->mae ? "mlp.fc1." : "mlp.c_fc.", *n_pj = v->mae ? "mlp.fc2." : "mlp.c_proj.";
        const char *n_l1 = v->mae ? "norm1." : "ln_1.", *n_l2 = v->mae ? "norm2." : "ln_2.";
        if ((s = up_linear(e, p + n_qkv + "weight", p + n_qkv + "bias", 3 * W, W, &b.w_qkv, &b.b_qkv))) return s;
        if ((s = up_linear(e, p + n_out + "weight", p + n_out + "bias", W, W, &b.w_out, &b.b_out))) return s;
        if ((s = up_linear(e, p + n_fc + "weight", p + n_fc + "bias", 4 * W, W, &b.w_fc, &b.b_fc))) return s;
        if ((s = up_linear(e, p + n_pj + "weight", p + n_pj + "bias", W, 4 * W, &b.w_proj, &b.b_proj))) return s;
        if ((s = up_f32(e, p + n_l1 + "weight", W, &b.ln1_w))) return s;
        if ((s = up_f32(e, p + n_l1 + "bias", W, &b.ln1_b))) return s;
        if ((s = up_f32(e, p + n_l2 + "weight", W, &b.ln2_w))) return s;
        if ((s = up_f32(e, p + n_l2 + "bias", W, &b.ln2_b))) return s;
    }
    if ((s = vit_alloc_ws(e))) return s;
    v->ws[0] = {v->A, v->y, v->qkv, v->att, v->hid, v->pe, v->x0, v->x1, true};
    PVR_HIP_TRY(hipMalloc((void **)&v->zero, 256));
    v->owned.push_back(v->zero);
    PVR_HIP_TRY(hipMemset(v->zero, 0, 256));
    return PVR_OK;
}

pvr_status vit_use_lane(pvr_encoder *e, int lane) {
    pvr_vit *v = e->vit;
    if (lane == v->cur) return PVR_OK;
    if (!v->ws[lane].valid) {
        pvr_status s = vit_alloc_ws(e);
        if (s) return s;
        PVR_HIP_TRY(hipDeviceSynchronize());
        v->ws[lane] = {v->A, v->y, v->qkv, v->att, v->hid, v->pe, v->x0, v->x1, true};
    }
    if (v->rs_h != 0 && !v->rs_tmp_l[lane]) {                  // resize tables exist already: this lane's temporaries do not yet
        PVR_HIP_TRY(hipMalloc((void **)&v->rs_tmp_l[lane], (size_t)e->desc.chunk * v->rs_h * v->res * 3 * sizeof(float)));
        PVR_HIP_TRY(hipMalloc((void **)&v->rs_u8_l[lane], (size_t)e->desc.chunk * v->res * v->res * 3));
        PVR_HIP_TRY(hipDeviceSynchronize());
    }
    const auto &w = v->ws[lane];
    v->A = w.A; v->y = w.y; v->qkv = w.qkv; v->att = w.att; v->hid = w.hid; v->pe = w.pe; v->x0 = w.x0; v->x1 = w.x1;
    v->rs_tmp = v->rs_tmp_l[lane]; v->rs_u8 = v->rs_u8_l[lane];
    v->cur = lane;
    return PVR_OK;
}

void vit_destroy(pvr_encoder *e) {
    if (!e->vit) return;
    for (void *p : e->vit->owned) (void)hipFree(p);
    pvr_vit *v = e->vit;
    std::vector<void *> rs = {v->rs_xmin, v->rs_xsize, v->rs_ymin, v->rs_ysize, v->rs_wx, v->rs_wy};
    for (int l = 0; l < PVR_MAX_LANES; ++l) { rs.push_back(v->rs_tmp_l[l]); rs.push_back(v->rs_u8_l[l]); }
    for (void *q : rs) if (q) (void)hipFree(q);
    delete e->vit;
    e->vit = nullptr;
}

// ATen _compute_indices_weights_aa for one dimension (float arithmetic as in the fp32 kernel)
static void aa_tables(int in, int out, std::vector<int> &mn, std::vector<int> &sz, std::vector<float> &wt, int &maxk) {
    const float scale = (float)in / (float)out;
    const float support = scale >= 1.f ? 2.f * scale : 2.f;        // bicubic interp_size 4
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    maxk = (int)ceilf(support) * 2 + 1;
    mn.assign(out, 0); sz.assign(out, 0); wt.assign((size_t)out * maxk, 0.f);
    auto cubic = [](float x) {
        const float a = -0.5f;
        x = fabsf(x);
        if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
        if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
        return 0.f;
    };
    for (int i = 0; i < out; ++i) {
        const float center = scale * ((float)i + 0.5f);
        int lo = (int)(center - support + 0.5f); if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5f); if (hi > in) hi = in;
        const int k = hi - lo;
        float total = 0.f;
        for (int j = 0; j < k; ++j) { const float w = cubic(((float)(j + lo) - center + 0.5f) * invscale); wt[(size_t)i * maxk + j] = w; total += w; }
        for (int j = 0; j < k; ++j) if (total != 0.f) wt[(size_t)i * maxk + j] /= total;
        mn[i] = lo; sz[i] = k;
    }
}

// ATen upsample_bicubic2d (no antialias, align_corners=False, A = -0.75) for one dimension, in window form: the 4 taps
// ix-1..ix+2 are clamped to the image, so border duplicates merge into one weight (torchvision 0.10 Resize(256,
// interpolation=3) on tensors, reference src/embeddings.py:81 for the 'mae' names)
static void cubic_tables(int in, int out, std::vector<int> &mn, std::vector<int> &sz, std::vector<float> &wt, int &maxk) {
    const float scale = (float)in / (float)out, A = -0.75f;
    maxk = 4;
    mn.assign(out, 0); sz.assign(out, 0); wt.assign((size_t)out * maxk, 0.f);
    for (int i = 0; i < out; ++i) {
        const float src = scale * ((float)i + 0.5f) - 0.5f;
        const float fl = floorf(src);
        const int ix = (int)fl;
        const float t = src - fl;
        float c[4];
        float x = t + 1.f;  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
        x = t;              c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        x = 1.f - t;        c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        x = 2.f - t;        c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
        int lo = ix - 1 < 0 ? 0 : (ix - 1 > in - 1 ? in - 1 : ix - 1);
        mn[i] = lo;
        int last = 0;
        for (int j = 0; j < 4; ++j) {
            int idx = ix - 1 + j; idx = idx < 0 ? 0 : (idx > in - 1 ? in - 1 : idx);
            wt[(size_t)i * maxk + (idx - lo)] += c[j];
            last = idx - lo;
        }
        sz[i] = last + 1;
    }
}

static pvr_status aa_prepare(pvr_encoder *e, int h, int w) {
    pvr_vit *v = e->vit;
    if (v->rs_h == h && v->rs_w == w) return PVR_OK;
    const int sh = w <= h ? w : h, lg = w <= h ? h : w;
    const int ns = v->resize_to, nl = (int)((double)v->resize_to * (double)lg / (double)sh);     // torchvision resize(int size)
    v->rs_rw = w <= h ? ns : nl; v->rs_rh = w <= h ? nl : ns;
    std::vector<int> xm, xs, ym, ys;
    std::vector<float> wx, wy;
    if (v->mae) {
        cubic_tables(w, v->rs_rw, xm, xs, wx, v->rs_maxk_w);
        cubic_tables(h, v->rs_rh, ym, ys, wy, v->rs_maxk_h);
    } else {
        aa_tables(w, v->rs_rw, xm, xs, wx, v->rs_maxk_w);
        aa_tables(h, v->rs_rh, ym, ys, wy, v->rs_maxk_h);
    }
    std::vector<void *> old = {v->rs_xmin, v->rs_xsize, v->rs_ymin, v->rs_ysize, v->rs_wx, v->rs_wy};
    for (int l = 0; l < PVR_MAX_LANES; ++l) { old.push_back(v->rs_tmp_l[l]); old.push_back(v->rs_u8_l[l]); }
    PVR_HIP_TRY(hipDeviceSynchronize());
    for (void *q : old) if (q) (void)hipFree(q);
    pvr_status s;
    if ((s = enc_upload(&v->rs_xmin, xm)) || (s = enc_upload(&v->rs_xsize, xs)) || (s = enc_upload(&v->rs_ymin, ym)) ||
        (s = enc_upload(&v->rs_ysize, ys)) || (s = enc_upload(&v->rs_wx, wx)) || (s = enc_upload(&v->rs_wy, wy))) return s;
    for (int l = 0; l < PVR_MAX_LANES; ++l) {
        if (l > 0 && !v->ws[l].valid) { v->rs_tmp_l[l] = nullptr; v->rs_u8_l[l] = nullptr; continue; }   // lanes never used need no buffers
        PVR_HIP_TRY(hipMalloc((void **)&v->rs_tmp_l[l], (size_t)e->desc.chunk * h * v->res * 3 * sizeof(float)));
        PVR_HIP_TRY(hipMalloc((void **)&v->rs_u8_l[l], (size_t)e->desc.chunk * v->res * v->res * 3));
    }
    v->rs_tmp = v->rs_tmp_l[v->cur]; v->rs_u8 = v->rs_u8_l[v->cur];
    PVR_HIP_TRY(hipDeviceSynchronize());
    v->rs_h = h; v->rs_w = w;
    return PVR_OK;
}

// one launch of the attention core; EXACT instantiations for the token counts the plans use, a runtime-NT fallback otherwise
template <bool F16, int HD, int MAXNT, bool EXACT>
static pvr_status launch_attention_inst(const u16 *qkv, u16 *out, int T, int TK, int W, int heads, int nb, hipStream_t st) {
    const size_t lds = (size_t)TK * (HD == 64 ? 128 : ((HD + 31) / 32 * 64 + 16)) + (size_t)HD * (TK + 4) * 2;
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)attention_kernel<F16, HD, MAXNT, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
        attr_done.mark();
    }
    hipLaunchKernelGGL((attention_kernel<F16, HD, MAXNT, EXACT>), dim3(heads, nb), dim3(256), lds, st, qkv, out, T, TK, W);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

template <bool F16>
static pvr_status launch_attention_any(const u16 *qkv, u16 *out, int T, int TK, int W, int heads, int nb, hipStream_t st) {
    const int hd = W / heads, nt = TK / 16;
    PVR_REQUIRE(W == heads * hd && (hd == 64 || hd == 80) && TK % 32 == 0 && TK >= T && TK <= 288, "attention: head dim %d / %d keys not built", hd, TK);
    if (hd == 64 && nt == 4) return launch_attention_inst<F16, 64, 4, true>(qkv, out, T, TK, W, heads, nb, st);
    if (hd == 64 && nt == 14) return launch_attention_inst<F16, 64, 14, true>(qkv, out, T, TK, W, heads, nb, st);
    if (hd == 64) return launch_attention_inst<F16, 64, 18, false>(qkv, out, T, TK, W, heads, nb, st);
    if (nt == 18) return launch_attention_inst<F16, 80, 18, true>(qkv, out, T, TK, W, heads, nb, st);
    return launch_attention_inst<F16, 80, 18, false>(qkv, out, T, TK, W, heads, nb, st);
}

template <bool F16, int WD>
static pvr_status vit_forward_t(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride, hipStream_t st) {
    pvr_vit *v = e->vit;
    const int W = v->width, P = v->patch, K = (P * P * 3 + 63) / 64 * 64, T = v->T, g2 = v->grid * v->grid, dt = e->desc.dtype;
    PVR_REQUIRE(W == WD, "vit: width %d does not match the instantiated plan", W);
    // transforms (embeddings.py:309-314): Resize(res, BICUBIC, antialias) is the identity when the short side is res
    const int sh = w <= h ? w : h;
    const bool resize = sh != v->resize_to;
    pvr_status s;
    if (resize && (s = aa_prepare(e, h, w))) return s;
    const int eh = resize ? v->rs_rh : h, ew = resize ? v->rs_rw : w;              // size after Resize
    const int top = (int)nearbyint((eh - v->res) / 2.0), left = (int)nearbyint((ew - v->res) / 2.0);
    for (int f0 = 0; f0 < n; f0 += e->desc.chunk) {
        const int nb = (n - f0 < e->desc.chunk) ? n - f0 : e->desc.chunk;
        const int rows = nb * T, prow = nb * g2;
        const uint8_t *fr = frames + (size_t)f0 * h * w * 3;
        const size_t tot = (size_t)prow * (K / 8);
        if (resize) {
            const size_t t1 = (size_t)nb * h * v->res * 3, t2 = (size_t)nb * v->res * v->res * 3;
            hipLaunchKernelGGL(aa_resize_h_kernel, dim3((int)((t1 + 255) / 256 > 8192 ? 8192 : (t1 + 255) / 256)), dim3(256), 0, st, fr,
                               v->rs_tmp, v->rs_xmin, v->rs_xsize, v->rs_wx, v->rs_maxk_w, nb, h, w, left, v->res);
            hipLaunchKernelGGL(aa_resize_v_kernel, dim3((int)((t2 + 255) / 256 > 8192 ? 8192 : (t2 + 255) / 256)), dim3(256), 0, st,
                               v->rs_tmp, v->rs_u8, v->rs_ymin, v->rs_ysize, v->rs_wy, v->rs_maxk_h, nb, h, top, v->res);
            hipLaunchKernelGGL(patchify_kernel<F16>, dim3((int)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, st,
                               v->rs_u8, v->A, nb, v->res, v->res, 0, 0, v->res, P);
        } else {
            hipLaunchKernelGGL(patchify_kernel<F16>, dim3((int)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256)), dim3(256), 0, st,
                               fr, v->A, nb, h, w, top, left, v->res, P);
        }
        PVR_LAUNCH_CHECK();
        // patch embedding GEMM -> fp32 [prow][W]
        if ((s = launch_conv(v->A, v->w_patch, v->b_patch, nullptr, v->pe, v->zero, prow, 1, 1, K, W, 1, 1, 1, 0, 0, 1, dt, st))) return s;
        // tokens + positional embedding + ln_pre -> residual stream x0 (fp32)
        hipLaunchKernelGGL((layernorm_kernel<F16, WD>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float *)nullptr, v->pe, v->cls,
                           v->pos, v->lnpre_w, v->lnpre_b, v->x0, (u16 *)nullptr, rows, T, v->eps, v->mae ? 0 : 1);
        PVR_LAUNCH_CHECK();
        e->last_n = nb;
        const std::string &stop = e->stop_after;
        if (stop == "pe" || stop == "ln_pre") return PVR_OK;
        float *x = v->x0, *xn = v->x1;
        int bi = 0;
        for (auto &b : v->blocks) {
            hipLaunchKernelGGL((layernorm_kernel<F16, WD>), dim3((rows + 3) / 4), dim3(256), 0, st, x, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, b.ln1_w, b.ln1_b, (float *)nullptr, v->y, rows, T, v->eps, 1);
            if ((s = launch_conv(v->y, b.w_qkv, b.b_qkv, nullptr, v->qkv, v->zero, rows, 1, 1, W, 3 * W, 1, 1, 1, 0, 0, 0, dt, st))) return s;
            if (bi == 0 && stop == "qkv0") return PVR_OK;
            if ((s = launch_attention_any<F16>(v->qkv, v->att, T, v->TK, W, v->heads, nb, st))) return s;
            PVR_LAUNCH_CHECK();
            if (bi == 0 && stop == "att0") return PVR_OK;
            // x' = x + out_proj(att): fp32 residual in (bit1), fp32 out (bit0)
            if ((s = launch_conv(v->att, b.w_out, b.b_out, x, xn, v->zero, rows, 1, 1, W, W, 1, 1, 1, 0, 0, 3, dt, st))) return s;
            if (bi == 0 && stop == "res0") return PVR_OK;
            hipLaunchKernelGGL((layernorm_kernel<F16, WD>), dim3((rows + 3) / 4), dim3(256), 0, st, xn, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, b.ln2_w, b.ln2_b, (float *)nullptr, v->y, rows, T, v->eps, 1);
            if ((s = launch_conv(v->y, b.w_fc, b.b_fc, nullptr, v->hid, v->zero, rows, 1, 1, W, 4 * W, 1, 1, 1, 0, v->act, 0, dt, st))) return s;   // QuickGELU / GELU
            if (bi == 0 && stop == "fc0") return PVR_OK;
            if ((s = launch_conv(v->hid, b.w_proj, b.b_proj, xn, x, v->zero, rows, 1, 1, 4 * W, W, 1, 1, 1, 0, 0, 3, dt, st))) return s;
            if (stop == "block" + std::to_string(bi)) return PVR_OK;
            ++bi;
        }
        hipLaunchKernelGGL((cls_head_kernel<WD>), dim3(nb), dim3(256), 0, st, x, v->lnpost_w, v->lnpost_b, v->proj,
                           out + (size_t)f0 * out_stride, out_stride, T, v->out_dim, v->eps);
        PVR_LAUNCH_CHECK();
        e->last_n = nb;
    }
    return PVR_OK;
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)in[i];
}
static pvr_status launch_u8_to_f32(const uint8_t *in, float *out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, in, out, n);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// parity taps (after a forward stopped with pvr_encoder_debug_stop_after): fp32 copies of plan buffers
pvr_status vit_tap(pvr_encoder *e, const char *name, float *out, int64_t cap, int64_t *count, hipStream_t st) {
    pvr_vit *v = e->vit;
    const std::string nm = name;
    const size_t rows = (size_t)e->last_n * v->T, W = v->width;
    const void *src = nullptr;
    size_t elems = 0;
    bool f32 = true;
    if (nm == "pe") { src = v->pe; elems = (size_t)e->last_n * v->grid * v->grid * W; }
    else if (nm == "ln_pre" || nm.rfind("block", 0) == 0) { src = v->x0; elems = rows * W; }
    else if (nm == "qkv0") { src = v->qkv; elems = rows * 3 * W; f32 = false; }
    else if (nm == "att0") { src = v->att; elems = rows * W; f32 = false; }
    else if (nm == "res0") { src = v->x1; elems = rows * W; }
    else if (nm == "fc0") { src = v->hid; elems = rows * 4 * W; f32 = false; }
    else if (nm == "resized") {                      // uint8 crop after the antialiased Resize, as fp32
        PVR_REQUIRE(v->rs_u8 != nullptr, "no resize has run");
        elems = (size_t)e->last_n * v->res * v->res * 3;
        PVR_REQUIRE((int64_t)elems <= cap, "tap resized needs %zu elements", elems);
        *count = (int64_t)elems;
        return launch_u8_to_f32(v->rs_u8, out, elems, st);
    }
    else { set_error("unknown vit tap %s", name); return PVR_ERR_INVALID; }
    PVR_REQUIRE((int64_t)elems <= cap, "tap %s needs %zu elements", name, elems);
    *count = (int64_t)elems;
    if (f32) { PVR_HIP_TRY(hipMemcpyAsync(out, src, elems * 4, hipMemcpyDeviceToDevice, st)); return PVR_OK; }
    return launch_h_to_f32(src, out, elems, e->desc.dtype, st);
}

// attention core for callers outside the ViT plan (CLIP RN50 attention pool)
pvr_status launch_attention(const void *qkv, void *out, int T, int W, int heads, int nb, int dtype, hipStream_t st) {
    const int TK = (T + 31) / 32 * 32;
    return dtype == PVR_F16 ? launch_attention_any<true>((const u16 *)qkv, (u16 *)out, T, TK, W, heads, nb, st)
                            : launch_attention_any<false>((const u16 *)qkv, (u16 *)out, T, TK, W, heads, nb, st);
}

// Resize(res, BICUBIC, antialias=True) + CenterCrop(res) as a stand-alone service (CLIP RN50 plan): a weight-less pvr_vit that
// only owns the resampling tables and per-lane temporaries
pvr_status resizer_create(pvr_encoder *e) {
    pvr_vit *v = new pvr_vit();
    v->res = e->desc.crop; v->resize_to = e->desc.resize; v->mae = false;
    for (auto &w : v->ws) w.valid = true;                       // every lane gets its resize temporaries
    e->resizer = v;
    return PVR_OK;
}

pvr_status resizer_run(pvr_encoder *e, int lane, const uint8_t *frames, int nb, int h, int w, hipStream_t st, const uint8_t **u8, int *oh, int *ow) {
    pvr_vit *v = e->resizer;
    const int sh = w <= h ? w : h;
    if (sh == v->resize_to) { *u8 = frames; *oh = h; *ow = w; return PVR_OK; }      // Resize is the identity: the caller centre-crops
    pvr_vit *saved = e->vit;
    e->vit = v;
    pvr_status s = aa_prepare(e, h, w);
    e->vit = saved;
    if (s) return s;
    const int top = (int)nearbyint((v->rs_rh - v->res) / 2.0), left = (int)nearbyint((v->rs_rw - v->res) / 2.0);
    const size_t t1 = (size_t)nb * h * v->res * 3, t2 = (size_t)nb * v->res * v->res * 3;
    hipLaunchKernelGGL(aa_resize_h_kernel, dim3((int)((t1 + 255) / 256 > 8192 ? 8192 : (t1 + 255) / 256)), dim3(256), 0, st, frames,
                       v->rs_tmp_l[lane], v->rs_xmin, v->rs_xsize, v->rs_wx, v->rs_maxk_w, nb, h, w, left, v->res);
    hipLaunchKernelGGL(aa_resize_v_kernel, dim3((int)((t2 + 255) / 256 > 8192 ? 8192 : (t2 + 255) / 256)), dim3(256), 0, st,
                       v->rs_tmp_l[lane], v->rs_u8_l[lane], v->rs_ymin, v->rs_ysize, v->rs_wy, v->rs_maxk_h, nb, h, top, v->res);
    PVR_LAUNCH_CHECK();
    *u8 = v->rs_u8_l[lane]; *oh = v->res; *ow = v->res;
    return PVR_OK;
}

void resizer_destroy(pvr_encoder *e) {
    if (!e->resizer) return;
    pvr_vit *saved = e->vit;
    e->vit = e->resizer;
    vit_destroy(e);
    e->vit = saved;
    e->resizer = nullptr;
}

pvr_status vit_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride, hipStream_t st) {
    if (e->vit->width == 1024)
        return e->desc.dtype == PVR_F16 ? vit_forward_t<true, 1024>(e, frames, n, h, w, out, out_stride, st)
                                        : vit_forward_t<false, 1024>(e, frames, n, h, w, out, out_stride, st);
    if (e->vit->width == 1280)
        return e->desc.dtype == PVR_F16 ? vit_forward_t<true, 1280>(e, frames, n, h, w, out, out_stride, st)
                                        : vit_forward_t<false, 1280>(e, frames, n, h, w, out, out_stride, st);
    return e->desc.dtype == PVR_F16 ? vit_forward_t<true, 768>(e, frames, n, h, w, out, out_stride, st)
                                    : vit_forward_t<false, 768>(e, frames, n, h, w, out, out_stride, st);
}

}  // namespace pvr
