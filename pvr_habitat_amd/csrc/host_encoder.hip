// Host (CPU) backend of the frozen ResNet encoders behind the SAME C-ABI (pvr_encoder_*): BASELINE configs[0] is "embed 1k saved
// 128x128 frames ON CPU via save_embedded_obs.py ... (plumbing, no GPU)", and the reference picks the CPU when disable_cuda is set or no
// GPU is present (src/embeddings.py:367-370).  pvr_encoder_set_host_backend(enc, 1) between create and finalize selects it: finalize folds
// BatchNorm into fp32 weights that stay on the host, and pvr_encoder_forward then takes HOST pointers (frames and output) and runs the
// plan below - plain C++ loops over the encoder's own op list (the one the HIP plan launches), fp32 NHWC, threads of this process.
// It is the product's slow path for boxes without a GPU, not the test oracle (oracle/ is torch) and not a fallback: a GPU encoder never
// routes through it, and a host encoder never touches HIP.
//
//   transforms   Resize(short side, bilinear, align_corners=False, round half-even back to uint8) -> CenterCrop(224) -> /255 ->
//                Normalize (embeddings.py:80-85), the arithmetic of preprocess.hip restated for the host
//   stem         conv1 7x7/2 pad 3 + bn1 + relu, maxpool 3x3/2 pad 1 (torchvision resnet, embeddings.py:118-120)
//   plan         every convolution of the op list as a direct GEMM over (pixel tile) x (cout) x K with a 4 x 4 register block of 8-wide
//                fp32 vectors; BN folded (eps 1e-5), bias, residual, ReLU in the epilogue
//   head         global average pool (2048 / 512) or the C-major flatten of the compression heads (moco.py:57-60)
// ResNet50 / _l3 / _l4 / ResNet18 / ResNet34 (the torchvision family); PVR_F32 plans only.
#include "encoder_internal.h"
#include "host_math.h"

namespace pvr {

void resized_size(int h, int w, int size, int *rh, int *rw);

struct HostConv {
    std::vector<float> w, b;       // [cout][k][k][cin] with BN's scale folded in; bias = BN shift (+ scale * conv bias)
    int in_buf, out_buf, res_buf, h, w_, cin, cout, k, stride, pad, relu;
};

struct HostPlan {
    std::vector<float> stem_w, stem_b;                   // [64][7][7][3], [64]
    std::vector<HostConv> ops;
    std::vector<float> buf[B_COUNT], img, stem;          // activations (fp32 NHWC, real channel counts), normalised crop, un-pooled stem
    int threads = 1;
};

// one convolution over n frames: direct for 1x1 / stride 1 (the input row IS the K vector), im2col rows per pixel tile otherwise
static void host_conv(const HostConv &c, const float *in, const float *res, float *out, int n, int threads) {
    const int ho = (c.h + 2 * c.pad - c.k) / c.stride + 1, wo = (c.w_ + 2 * c.pad - c.k) / c.stride + 1;
    const int K = c.k * c.k * c.cin, TP = 32;
    const long M = (long)n * ho * wo;
    const int tiles = (int)((M + TP - 1) / TP);
    const bool direct = c.k == 1 && c.stride == 1 && c.pad == 0;
    host_parallel_for(tiles, threads, [&](int t) {
        const long m0 = (long)t * TP;
        const int np_all = (int)(M - m0 < TP ? M - m0 : TP);
        std::vector<float> col;
        const float *rows[TP];
        if (direct) {
            for (int p = 0; p < np_all; ++p) rows[p] = in + (m0 + p) * c.cin;
        } else {
            col.assign((size_t)np_all * K, 0.f);
            for (int p = 0; p < np_all; ++p) {
                const long m = m0 + p;
                const int x = (int)(m % wo), y = (int)((m / wo) % ho), f = (int)(m / ((long)wo * ho));
                float *dst = col.data() + (size_t)p * K;
                for (int a = 0; a < c.k; ++a) {
                    const int yy = y * c.stride - c.pad + a;
                    if (yy < 0 || yy >= c.h) continue;
                    for (int b = 0; b < c.k; ++b) {
                        const int xx = x * c.stride - c.pad + b;
                        if (xx < 0 || xx >= c.w_) continue;
                        memcpy(dst + ((size_t)a * c.k + b) * c.cin, in + (((size_t)f * c.h + yy) * c.w_ + xx) * c.cin, (size_t)c.cin * 4);
                    }
                }
                rows[p] = dst;
            }
        }
        for (int p0 = 0; p0 < np_all; p0 += 4) {
            const int np = np_all - p0 < 4 ? np_all - p0 : 4;
            for (int c0 = 0; c0 < c.cout; c0 += 4) {
                const int nc = c.cout - c0 < 4 ? c.cout - c0 : 4;
                const float *wr[4];
                for (int j = 0; j < nc; ++j) wr[j] = c.w.data() + (size_t)(c0 + j) * K;
                float o[4][4];
                host_dot_block(rows + p0, wr, K, np, nc, o);
                for (int i = 0; i < np; ++i)
                    for (int j = 0; j < nc; ++j) {
                        const size_t oi = (size_t)(m0 + p0 + i) * c.cout + c0 + j;
                        float v = o[i][j] + c.b[c0 + j];
                        if (res) v += res[oi];
                        out[oi] = c.relu ? (v > 0.f ? v : 0.f) : v;
                    }
            }
        }
    });
}

// Resize + CenterCrop (+ crop position) + ConvertImageDtype + Normalize: uint8 (n,h,w,3) -> fp32 (n,crop,crop,3)
static void host_preprocess(const pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *img, int threads) {
    const int crop = e->desc.crop;
    int rh, rw, need, top, left;
    resized_size(h, w, e->desc.resize, &rh, &rw);
    preprocess_geometry(h, w, e->desc.resize, crop, e->crop_pos, &need, &top, &left);
    const bool resize = rh != h || rw != w;
    const float sh = (float)h / (float)rh, sw = (float)w / (float)rw;
    host_parallel_for(n * crop, threads, [&](int job) {
        const int f = job / crop, y = job % crop, Y = y + top;
        const uint8_t *src = frames + (size_t)f * h * w * 3;
        float *dst = img + ((size_t)f * crop + y) * crop * 3;
        for (int x = 0; x < crop; ++x) {
            const int X = x + left;
            float v[3];
            if (!resize) {
                const uint8_t *s = src + ((size_t)Y * w + X) * 3;
                v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
            } else {
#pragma clang fp contract(off)
                float sy = sh * ((float)Y + 0.5f) - 0.5f, sx = sw * ((float)X + 0.5f) - 0.5f;      // ATen upsample_bilinear2d, align_corners=False
                sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
                const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
                const uint8_t *r0 = src + (size_t)y0 * w * 3, *r1 = src + (size_t)y1 * w * 3;
                for (int c = 0; c < 3; ++c) {
                    const float val = hy * (hx * (float)r0[x0 * 3 + c] + lx * (float)r0[x1 * 3 + c]) + ly * (hx * (float)r1[x0 * 3 + c] + lx * (float)r1[x1 * 3 + c]);
                    v[c] = rintf(val);                      // torch.round (half to even) back to uint8
                }
            }
            for (int c = 0; c < 3; ++c) dst[x * 3 + c] = (v[c] / 255.0f - e->desc.mean[c]) / e->desc.std_[c];
        }
    });
}

pvr_status host_finalize(pvr_encoder *e) {
    const int a = e->desc.arch;
    PVR_REQUIRE(a == PVR_ARCH_RESNET50 || a == PVR_ARCH_RESNET50_L3 || a == PVR_ARCH_RESNET50_L4 || a == PVR_ARCH_RESNET18 || a == PVR_ARCH_RESNET34,
                "host backend: only the torchvision ResNet family has a CPU plan (arch %d)", a);
    PVR_REQUIRE(e->desc.dtype == PVR_F32, "host backend: create the encoder with dtype PVR_F32 (the CPU plan is fp32)");
    HostPlan *hp = new HostPlan();
    e->hplan = hp;
    unsigned hc = std::thread::hardware_concurrency();
    hp->threads = hc ? (int)hc : 1;
    if (const char *t = getenv("PVR_HOST_THREADS")) hp->threads = atoi(t) > 0 ? atoi(t) : 1;
    pvr_status s;
    // stem: [64][3][7][7] -> [64][7][7][3], BN folded
    const HostTensor *w;
    if ((s = enc_need(e, "conv1.weight", &w, (size_t)64 * 3 * 49))) return s;
    std::vector<float> scale, shift;
    {
        const HostTensor *g, *b, *m, *v;
        if ((s = enc_need(e, "bn1.weight", &g, 64)) || (s = enc_need(e, "bn1.bias", &b, 64)) || (s = enc_need(e, "bn1.running_mean", &m, 64)) ||
            (s = enc_need(e, "bn1.running_var", &v, 64))) return s;
        scale.resize(64); shift.resize(64);
        for (int i = 0; i < 64; ++i) { scale[i] = g->data[i] / sqrtf(v->data[i] + 1e-5f); shift[i] = b->data[i] - m->data[i] * scale[i]; }
    }
    hp->stem_w.resize((size_t)64 * 147); hp->stem_b = shift;
    for (int co = 0; co < 64; ++co)
        for (int c = 0; c < 3; ++c)
            for (int t = 0; t < 49; ++t) hp->stem_w[((size_t)co * 49 + t) * 3 + c] = w->data[((size_t)co * 3 + c) * 49 + t] * scale[co];
    for (const ConvOp &op : e->ops) {
        PVR_REQUIRE(op.kind == 0, "host backend: op kind %d has no CPU form", op.kind);
        HostConv c;
        c.in_buf = op.in_buf; c.out_buf = op.out_buf; c.res_buf = op.res_buf; c.h = op.h; c.w_ = op.w; c.cin = op.cin_real; c.cout = op.cout_real;
        c.k = op.k; c.stride = op.stride; c.pad = op.pad; c.relu = op.relu;
        const HostTensor *cw;
        if ((s = enc_need(e, op.conv + ".weight", &cw, (size_t)c.cout * c.cin * c.k * c.k))) return s;
        const HostTensor *g, *b, *m, *v;
        if ((s = enc_need(e, op.bn + ".weight", &g, c.cout)) || (s = enc_need(e, op.bn + ".bias", &b, c.cout)) ||
            (s = enc_need(e, op.bn + ".running_mean", &m, c.cout)) || (s = enc_need(e, op.bn + ".running_var", &v, c.cout))) return s;
        c.w.resize((size_t)c.cout * c.k * c.k * c.cin); c.b.resize(c.cout);
        const HostTensor *cb = enc_find(e, op.conv + ".bias");
        for (int co = 0; co < c.cout; ++co) {
            const float sc = g->data[co] / sqrtf(v->data[co] + 1e-5f);
            c.b[co] = b->data[co] - m->data[co] * sc + (cb ? sc * cb->data[co] : 0.f);
            for (int ci = 0; ci < c.cin; ++ci)
                for (int t = 0; t < c.k * c.k; ++t)
                    c.w[((size_t)co * c.k * c.k + t) * c.cin + ci] = cw->data[((size_t)co * c.cin + ci) * c.k * c.k + t] * sc;
        }
        hp->ops.push_back(std::move(c));
    }
    e->weights.clear();
    e->finalized = true;
    return PVR_OK;
}

pvr_status host_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride) {
    HostPlan *hp = e->hplan;
    PVR_REQUIRE(hp && frames && out && n > 0, "host backend: bad forward arguments");
    const int crop = e->desc.crop, T = hp->threads;
    int rh, rw;
    resized_size(h, w, e->desc.resize, &rh, &rw);
    PVR_REQUIRE(rh >= crop && rw >= crop, "preprocess: resized frame %dx%d smaller than crop %d", rh, rw, crop);
    // buffers: sized for this call (a host encoder serves plumbing-sized batches; nothing is kept between calls but the vectors' capacity)
    size_t need[B_COUNT] = {0};
    need[B_X0] = (size_t)n * 56 * 56 * 64;
    for (const HostConv &c : hp->ops) {
        const int ho = (c.h + 2 * c.pad - c.k) / c.stride + 1, wo = (c.w_ + 2 * c.pad - c.k) / c.stride + 1;
        const size_t o = (size_t)n * ho * wo * c.cout, i = (size_t)n * c.h * c.w_ * c.cin;
        if (o > need[c.out_buf]) need[c.out_buf] = o;
        if (i > need[c.in_buf]) need[c.in_buf] = i;
    }
    for (int b = 0; b < B_COUNT; ++b) if (hp->buf[b].size() < need[b]) hp->buf[b].resize(need[b]);
    hp->img.resize((size_t)n * crop * crop * 3);
    hp->stem.resize((size_t)n * 112 * 112 * 64);
    host_preprocess(e, frames, n, h, w, hp->img.data(), T);
    HostConv st;
    st.h = crop; st.w_ = crop; st.cin = 3; st.cout = 64; st.k = 7; st.stride = 2; st.pad = 3; st.relu = 1;
    st.w = hp->stem_w; st.b = hp->stem_b;
    host_conv(st, hp->img.data(), nullptr, hp->stem.data(), n, T);
    float *x0 = hp->buf[B_X0].data();
    const float *sp = hp->stem.data();
    host_parallel_for(n * 56, T, [&](int job) {                          // maxpool 3x3 / 2, pad 1 (-inf padding)
        const int f = job / 56, y = job % 56;
        for (int x = 0; x < 56; ++x)
            for (int c = 0; c < 64; ++c) {
                float m = -INFINITY;
                for (int a = 0; a < 3; ++a) {
                    const int yy = 2 * y - 1 + a;
                    if (yy < 0 || yy >= 112) continue;
                    for (int b = 0; b < 3; ++b) {
                        const int xx = 2 * x - 1 + b;
                        if (xx < 0 || xx >= 112) continue;
                        const float v = sp[(((size_t)f * 112 + yy) * 112 + xx) * 64 + c];
                        m = v > m ? v : m;
                    }
                }
                x0[(((size_t)f * 56 + y) * 56 + x) * 64 + c] = m;
            }
    });
    const HostConv *last = nullptr;
    for (const HostConv &c : hp->ops) {
        host_conv(c, hp->buf[c.in_buf].data(), c.res_buf == B_NONE ? nullptr : hp->buf[c.res_buf].data(), hp->buf[c.out_buf].data(), n, T);
        last = &c;
    }
    PVR_REQUIRE(last, "host backend: empty plan");
    const float *y = hp->buf[last->out_buf].data();
    const int hw = e->final_hw, cr = e->final_creal;
    const bool pooled = e->desc.arch == PVR_ARCH_RESNET50 || e->desc.arch == PVR_ARCH_RESNET18 || e->desc.arch == PVR_ARCH_RESNET34;
    for (int f = 0; f < n; ++f) {
        float *o = out + (size_t)f * out_stride;
        if (pooled) {
            for (int c = 0; c < cr; ++c) {
                float s = 0.f;
                for (int p = 0; p < hw; ++p) s += y[((size_t)f * hw + p) * cr + c];
                o[c] = s / (float)hw;
            }
        } else {
            for (int c = 0; c < cr; ++c)
                for (int p = 0; p < hw; ++p) o[(size_t)c * hw + p] = y[((size_t)f * hw + p) * cr + c];     // C-major flatten (moco.py:57-60)
        }
    }
    e->last_n = n;
    return PVR_OK;
}

void host_destroy(pvr_encoder *e) {
    delete e->hplan;
    e->hplan = nullptr;
}

}  // namespace pvr
