// Frozen-encoder plan: ResNet50 family (conv5 / l4-compressed / l3-compressed) as a static list of
// fused launches over a pre-allocated HBM workspace.
//
// Replaces: reference src/embeddings.py:386-402 (EmbeddingNet.forward), the model construction at
// src/vision_models/moco.py:6-113 and resnet.py:6-104 (topology edits), and torchvision's
// ResNet._forward_impl.  Weight names are torchvision's (SURVEY 8a "State-dict keys").
//
// Host-side work done once in finalize():
//   * BN eval fold:  scale = gamma / sqrt(var + 1e-5),  W' = W*scale,  b' = beta - mean*scale (+ scale*conv_bias)
//   * stem: Normalize + /255 folded into conv1 (see stem.hip), K laid out (kh, kw[8], c[4])
//   * OIHW fp32 -> [cout_pad][kh][kw][cin_pad] 16-bit (bf16 or f16), zero padded
#include "encoder_internal.h"

namespace pvr {

pvr_status launch_split16_pack(const float *w, void *out, int rows, int K, hipStream_t stream);
bool conv_split16_supported(int cin, int cout, int k);
pvr_status launch_conv_split16(const float *in, const void *wsp, const float *bias, const float *res, float *out, int n, int h, int w, int cin,
                               int cout, int k, int stride, int pad, int relu, hipStream_t stream, float *out2 = nullptr, int n1 = 0, void *out16 = nullptr,
                               int terms = 3);

// every environment switch of the encoder plans, read ONCE per encoder in pvr_encoder_create (never on the forward path)
static void read_switches(PlanSwitches &sw) {
    auto get = [](const char *name, int def) { const char *v = getenv(name); return v ? atoi(v) : def; };
    sw.pool_fuse = get("PVR_POOL_FUSE", 1);
    sw.stem_u8 = get("PVR_STEM_U8", 1);
    sw.stem_lds = get("PVR_STEM_LDS", 1);
    sw.frame_front1 = get("PVR_FRAME_FRONT1", 1);
    sw.frame_next1 = get("PVR_FRAME_NEXT1", 0);
    sw.dual_ds = get("PVR_DUAL_DS", 1);
    sw.chain_ds = get("PVR_CHAIN_DS", 1);
    sw.chain_blocked = get("PVR_CHAIN_BLOCKED", 1);
    sw.splitk = get("PVR_SPLITK", 1);
    sw.smallk_div = get("PVR_SMALLK_DIV", 4);
    if (sw.smallk_div < 1) sw.smallk_div = 1;
    sw.frame_min_n = get("PVR_FRAME_MIN_N", 128);
    sw.frame_run = get("PVR_FRAME_RUN", 0);       // (measured equal to one launch per bottleneck: opt-in, profiles/experiments/r06_bneck_frame_run.txt)
    sw.frame_stagger = get("PVR_FRAME_RUN_STAGGER", 0);
    sw.split16 = get("PVR_SPLIT16", 1);
    sw.stem_conv1 = get("PVR_STEM_CONV1", 1);
    sw.resid32 = get("PVR_RESID32", 1);
    sw.tail_f32 = get("PVR_TAIL_F32", 1);
    sw.fuse = get("PVR_FUSE", 1);
}

// f16 range validation (pvr_encoder_check_range): any inf / NaN among the first `n` 16-bit (or fp32) values of a launch's output -> flags[idx] = 1
template <bool F32>
__global__ __launch_bounds__(256) void range_flag_kernel(const void *x, size_t n8, int dtype, int *flags, int idx) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        if constexpr (F32) {
            const u32x4 a = reinterpret_cast<const u32x4 *>(x)[2 * i], b = reinterpret_cast<const u32x4 *>(x)[2 * i + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= (a[e] & 0x7f800000u) == 0x7f800000u || (b[e] & 0x7f800000u) == 0x7f800000u;
        } else {
            const u32x4 a = reinterpret_cast<const u32x4 *>(x)[i];
            const unsigned em = dtype == PVR_F16 ? 0x7c00u : 0x7f80u;          // exponent field all ones: inf or NaN
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= ((a[e] & em) == em) || (((a[e] >> 16) & em) == em);
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flags + idx, 1);
}

const char *launch_kind_name(int k) {
    static const char *nm[] = {"conv", "bneck_frame(front1)", "bneck_frame", "bneck_frame(run)", "(in the run)", "frame_members", "conv_pp256(dual)", "dual_members", "chain", "cast",
                               "conv_f32", "conv_split16", "conv_split16(pair)", "conv_split16(in32)", "splitk(small)", "splitk", "conv_expand(blocked)",
                               "conv_wfrag(pool)", "conv_wfrag"};
    return k >= 0 && k < (int)(sizeof nm / sizeof nm[0]) ? nm[k] : "?";
}

static void add_conv(pvr_encoder *e, const std::string &conv, const std::string &bn, int in_buf, int out_buf,
                     int res_buf, int h, int w, int cin, int cin_real, int cout, int cout_real, int k, int stride,
                     int relu, int out_f32 = 0) {
    ConvOp op;
    op.conv = conv; op.bn = bn; op.in_buf = in_buf; op.out_buf = out_buf; op.res_buf = res_buf;
    op.h = h; op.w = w; op.cin = cin; op.cin_real = cin_real; op.cout = cout; op.cout_real = cout_real;
    op.k = k; op.stride = stride; op.pad = k / 2; op.relu = relu; op.out_f32 = out_f32;
    e->ops.push_back(op);
}

static void add_cast(pvr_encoder *e, int in_buf, int out_buf, int hw, int c) {
    ConvOp op;
    op.kind = 2; op.conv = "cast"; op.in_buf = in_buf; op.out_buf = out_buf; op.res_buf = B_NONE;
    op.h = hw; op.w = hw; op.cin = op.cout = op.cout_real = c; op.cin_real = 0; op.k = 1; op.stride = 1; op.pad = 0; op.relu = 0; op.out_f32 = 0;
    e->ops.push_back(op);
}

// torchvision resnet50 v1.5: layers [3,4,6,3], stride on the 3x3 (conv2), downsample on block 0
//
// Parity plan of the compressed PVRs (resid32: *_l3 / *_l4 in f16 storage).  These variants have no final average pool, so the
// trunk's accumulated storage rounding reaches the output element by element (measured 1.09e-3 / 9.6e-4 rel-L2 with every
// activation in f16).  From layer3 on the residual stream y is therefore kept in fp32 (conv3 adds an fp32 identity / fp32
// downsample output and writes fp32; a 16-bit copy feeds the next block's convolutions), and the compression head - three
// small 3x3 convolutions over K = 9216 / 18432 - runs from that fp32 stream with fp32 weights on the f32-input MFMA
// (conv_f32.hip).  layer3 / layer4 are MFMA-bound at 14x14 / 7x7, so the extra fp32 bytes cost little.  PVR_RESID32=0 restores
// the all-16-bit plan (A/B).
static void build_resnet50(pvr_encoder *e) {
    const int arch = e->desc.arch;
    const int stages = arch == PVR_ARCH_RESNET50_L3 ? 3 : 4;
    const int nblk[4] = {3, 4, 6, 3};
    e->resid32 = e->desc.dtype == PVR_F16 && arch != PVR_ARCH_RESNET50 && e->sw.resid32 != 0;
    // Round 3: the fp32 residual stream alone left *_l3 at 9.75e-4 of a 1e-3 bound, and CPU emulation over three weight seeds
    // (scripts/emulate_l3_rounding.py) puts that plan at 8.6e-4 ... 1.01e-3: one seed from red.  What gives real margin is the LAST
    // trunk stage entirely in fp32 (fp32 weights, fp32 operands: conv_f32.hip, the kernels of the PVR_F32 mode) with the fp32
    // residual stream starting at layer2: 5.6e-4 ... 6.4e-4 on the same seeds for *_l3, 6.3e-4 ... 6.7e-4 for *_l4 (16-bit weights alone cost ~6e-4
    // at layer3, whatever the activations do).  That stage is 36 % (layer3) / 20 % (layer4) of the trunk's FLOPs at the f32-MFMA
    // rate: the parity mode of the compressed PVRs pays for its margin in throughput (DESIGN.md section 2 has the numbers);
    // PVR_TAIL_F32=0 restores round 2's plan, bf16 (the throughput mode) never uses either.
    e->tail32 = e->resid32 && e->sw.tail_f32 != 0;
    const int r32_from = e->tail32 ? 1 : 2;             // fp32 residual stream from layer2 on (emulated: *_l3 5.6e-4 ... 6.4e-4, *_l4 6.3e-4 ... 6.7e-4)
    // Round 6: with conv_split16 the convolutions that consume the fp32 stream as a 16-bit operand (the next block's conv1, a stage's downsample) read it
    // themselves and round it in their staging pass (ConvOp::from32): no fp32 -> 16-bit copy launches (3 x 0.11 ms in *_l3, 4 x 0.11 + 5 x 0.05 ms in *_l4)
    const bool in32 = e->resid32 && e->sw.split16 != 0;
    bool x_is_32 = false;                               // the block input exists as fp32 only (the previous block wrote y32 and no 16-bit copy)
    int hw = 56, inpl = 64, x = B_X0, x32 = B_NONE;
    for (int li = 0; li < stages; ++li) {
        const int planes = 64 << li;
        const bool nested = (arch == PVR_ARCH_RESNET50_L4 && li == 3) || (arch == PVR_ARCH_RESNET50_L3 && li == 2);
        const bool r32 = e->resid32 && li >= r32_from;
        const bool f32stage = e->tail32 && li == stages - 1;
        for (int bi = 0; bi < nblk[li]; ++bi) {
            char pfx[64];
            if (nested) snprintf(pfx, sizeof pfx, "layer%d.0.%d", li + 1, bi);
            else snprintf(pfx, sizeof pfx, "layer%d.%d", li + 1, bi);
            const std::string p = pfx;
            const int stride = (bi == 0 && li > 0) ? 2 : 1;
            const int ohw = hw / stride;
            const int y = x == B_X0 ? B_X1 : B_X0;
            const bool last = (li == stages - 1 && bi == nblk[li] - 1);
            char tn[16]; snprintf(tn, sizeof tn, "layer%d", li + 1);
            if (f32stage) {
                // every tensor of the block is fp32 (the 16-bit ping-pong buffers are large enough: the stage's activations are
                // 1/4 ... 1/16 of layer1's elements); the block reads the fp32 stream directly
                const size_t first = e->ops.size();
                const int y32 = x32 == B_Y0 ? B_Y1 : B_Y0;
                add_conv(e, p + ".conv1", p + ".bn1", x32, B_T1, B_NONE, hw, hw, inpl, inpl, planes, planes, 1, 1, 1);
                add_conv(e, p + ".conv2", p + ".bn2", B_T1, B_T2, B_NONE, hw, hw, planes, planes, planes, planes, 3, stride, 1);
                int res32 = x32;
                if (bi == 0) {
                    add_conv(e, p + ".downsample.0", p + ".downsample.1", x32, B_DS, B_NONE, hw, hw, inpl, inpl, planes * 4, planes * 4, 1, stride, 0, 1);
                    res32 = B_DS;
                }
                add_conv(e, p + ".conv3", p + ".bn3", B_T2, y32, res32, ohw, ohw, planes, planes, planes * 4, planes * 4, 1, 1, 1, 1 | 2);
                for (size_t i = first; i < e->ops.size(); ++i) e->ops[i].f32op = true;
                if (bi == nblk[li] - 1) {
                    e->ops.back().tap = tn;
                    e->taps[tn] = {y32, {ohw, ohw, planes * 4, 1}};
                }
                x32 = y32; hw = ohw; inpl = planes * 4;
                continue;
            }
            add_conv(e, p + ".conv1", p + ".bn1", x_is_32 ? x32 : x, B_T1, B_NONE, hw, hw, inpl, inpl, planes, planes, 1, 1, 1);
            e->ops.back().from32 = x_is_32;
            add_conv(e, p + ".conv2", p + ".bn2", B_T1, B_T2, B_NONE, hw, hw, planes, planes, planes, planes, 3, stride, 1);
            int res = r32 ? x32 : x;
            if (bi == 0) {
                add_conv(e, p + ".downsample.0", p + ".downsample.1", x_is_32 ? x32 : x, B_DS, B_NONE, hw, hw, inpl, inpl, planes * 4,
                         planes * 4, 1, stride, 0, r32 ? 1 : 0);
                e->ops.back().from32 = x_is_32;
                res = B_DS;
            }
            if (r32) {
                const int y32 = x32 == B_Y0 ? B_Y1 : B_Y0;
                add_conv(e, p + ".conv3", p + ".bn3", B_T2, y32, res, ohw, ohw, planes, planes, planes * 4, planes * 4, 1, 1, 1, 1 | 2);
                // the 16-bit copy feeds the next block's convolutions - unless that block is fp32 (it reads the stream itself), as the head does
                const bool next_f32 = e->tail32 && li == stages - 2 && bi == nblk[li] - 1;
                if (!last && !next_f32 && !in32) add_cast(e, y32, y, ohw, planes * 4);
                if (bi == nblk[li] - 1) {
                    e->ops.back().tap = tn;
                    e->taps[tn] = {y32, {ohw, ohw, planes * 4, 1}};
                }
                x32 = y32; x = y; hw = ohw; inpl = planes * 4;
                x_is_32 = in32;
                continue;
            }
            // the trunk's last block feeds avgpool: keep fp32 (conv5 variant only)
            const int f32 = (last && arch == PVR_ARCH_RESNET50) ? 1 : 0;
            add_conv(e, p + ".conv3", p + ".bn3", B_T2, f32 ? B_F32 : y, res, ohw, ohw, planes, planes, planes * 4,
                     planes * 4, 1, 1, 1, f32);
            if (bi == nblk[li] - 1) {
                e->ops.back().tap = tn;
                e->taps[tn] = {f32 ? B_F32 : y, {ohw, ohw, planes * 4, f32}};
            }
            x = y; hw = ohw; inpl = planes * 4;
        }
    }
    if (arch == PVR_ARCH_RESNET50) {
        e->out_size = 2048; e->final_hw = 49; e->final_c = 2048; e->final_creal = 2048;
        return;
    }
    // compression head: BasicBlock(C -> c) with a conv3x3(+bias)+BN downsample (moco.py:35-50 / 79-94)
    const int cin = arch == PVR_ARCH_RESNET50_L3 ? 1024 : 2048;
    const int c = arch == PVR_ARCH_RESNET50_L3 ? 11 : 42;
    const std::string p = arch == PVR_ARCH_RESNET50_L3 ? "layer3.1" : "layer4.1";
    const int hx = e->resid32 ? x32 : x;
    add_conv(e, p + ".conv1", p + ".bn1", hx, B_T1, B_NONE, hw, hw, cin, cin, 64, c, 3, 1, 1);
    add_conv(e, p + ".downsample.0", p + ".downsample.1", hx, B_DS, B_NONE, hw, hw, cin, cin, 64, c, 3, 1, 0);
    add_conv(e, p + ".conv2", p + ".bn2", B_T1, B_F32, B_DS, hw, hw, 64, c, 64, c, 3, 1, 1, 1);
    if (e->resid32) for (size_t i = e->ops.size() - 3; i < e->ops.size(); ++i) e->ops[i].f32op = true;
    e->out_size = c * hw * hw; e->final_hw = hw * hw; e->final_c = 64; e->final_creal = c;
}

// torchvision resnet18 / resnet34 (reference embeddings.py:112-117): BasicBlock = conv3x3(stride) bn relu, conv3x3 bn, (+ identity
// or 1x1(stride)+bn downsample), relu; layers [2,2,2,2] / [3,4,6,3], widths 64..512, global average pool -> 512
static void build_basic_resnet(pvr_encoder *e) {
    const int nb18[4] = {2, 2, 2, 2}, nb34[4] = {3, 4, 6, 3};
    const int *nblk = e->desc.arch == PVR_ARCH_RESNET18 ? nb18 : nb34;
    int hw = 56, inpl = 64, x = B_X0;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        for (int bi = 0; bi < nblk[li]; ++bi) {
            char pfx[64];
            snprintf(pfx, sizeof pfx, "layer%d.%d", li + 1, bi);
            const std::string p = pfx;
            const int stride = (bi == 0 && li > 0) ? 2 : 1;
            const int ohw = hw / stride;
            const int y = x == B_X0 ? B_X1 : B_X0;
            const bool last = (li == 3 && bi == nblk[li] - 1);
            add_conv(e, p + ".conv1", p + ".bn1", x, B_T1, B_NONE, hw, hw, inpl, inpl, planes, planes, 3, stride, 1);
            int res = x;
            if (stride > 1 || inpl != planes) {
                add_conv(e, p + ".downsample.0", p + ".downsample.1", x, B_DS, B_NONE, hw, hw, inpl, inpl, planes, planes, 1, stride, 0);
                res = B_DS;
            }
            add_conv(e, p + ".conv2", p + ".bn2", B_T1, last ? B_F32 : y, res, ohw, ohw, planes, planes, planes, planes, 3, 1, 1, last ? 1 : 0);
            if (bi == nblk[li] - 1) {
                char tn[16]; snprintf(tn, sizeof tn, "layer%d", li + 1);
                e->ops.back().tap = tn;
                e->taps[tn] = {last ? B_F32 : y, {ohw, ohw, planes, last ? 1 : 0}};
            }
            x = y; hw = ohw; inpl = planes;
        }
    }
    e->out_size = 512; e->final_hw = 49; e->final_c = 512; e->final_creal = 512;
}

// openai/CLIP ModifiedResNet-50 (reference embeddings.py:305-306): stem conv1 (3x3/2, run by the stem kernel as a 7x7 with only
// its centre taps set) is not in the list; conv2 / conv3 of the stem, AvgPool2d(2), then Bottlenecks whose convolutions all have
// stride 1 - the stride is an AvgPool2d after conv2 and in front of the downsample convolution.  Channels 32 are padded to 64.
static void add_pool(pvr_encoder *e, int in_buf, int out_buf, int h, int c) {
    ConvOp op;
    op.kind = 1; op.conv = "avgpool2"; op.in_buf = in_buf; op.out_buf = out_buf; op.res_buf = B_NONE;
    op.h = h; op.w = h; op.cin = op.cin_real = op.cout = op.cout_real = c; op.k = 2; op.stride = 2; op.pad = 0; op.relu = 0; op.out_f32 = 0;
    e->ops.push_back(op);
}

static void build_clip_rn50(pvr_encoder *e) {
    const std::string v = "visual.";
    add_conv(e, v + "conv2", v + "bn2", B_STEM, B_X0, B_NONE, 112, 112, 64, 32, 64, 32, 3, 1, 1);
    add_conv(e, v + "conv3", v + "bn3", B_X0, B_X1, B_NONE, 112, 112, 64, 32, 64, 64, 3, 1, 1);
    add_pool(e, B_X1, B_X0, 112, 64);
    e->ops.back().tap = "stem3";
    e->taps["stem3"] = {B_X0, {56, 56, 64, 0}};
    const int nblk[4] = {3, 4, 6, 3};
    int hw = 56, inpl = 64, x = B_X0;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        for (int bi = 0; bi < nblk[li]; ++bi) {
            char pfx[64];
            snprintf(pfx, sizeof pfx, "visual.layer%d.%d", li + 1, bi);
            const std::string p = pfx;
            const int stride = (bi == 0 && li > 0) ? 2 : 1;
            const int ohw = hw / stride;
            const int y = x == B_X0 ? B_X1 : B_X0;
            const bool last = (li == 3 && bi == nblk[li] - 1);
            add_conv(e, p + ".conv1", p + ".bn1", x, B_T1, B_NONE, hw, hw, inpl, inpl, planes, planes, 1, 1, 1);
            add_conv(e, p + ".conv2", p + ".bn2", B_T1, B_T2, B_NONE, hw, hw, planes, planes, planes, planes, 3, 1, 1);
            int c3_in = B_T2, res = x;
            if (stride > 1) { add_pool(e, B_T2, B_T1, hw, planes); c3_in = B_T1; }
            if (stride > 1 || inpl != planes * 4) {
                int ds_in = x;
                if (stride > 1) { add_pool(e, x, B_T2, hw, inpl); ds_in = B_T2; }
                add_conv(e, p + ".downsample.0", p + ".downsample.1", ds_in, B_DS, B_NONE, ohw, ohw, inpl, inpl, planes * 4, planes * 4, 1, 1, 0);
                res = B_DS;
            }
            add_conv(e, p + ".conv3", p + ".bn3", c3_in, last ? B_F32 : y, res, ohw, ohw, planes, planes, planes * 4, planes * 4, 1, 1, 1, last ? 1 : 0);
            if (bi == nblk[li] - 1) {
                char tn[16]; snprintf(tn, sizeof tn, "layer%d", li + 1);
                e->ops.back().tap = tn;
                e->taps[tn] = {last ? B_F32 : y, {ohw, ohw, planes * 4, last ? 1 : 0}};
            }
            x = y; hw = ohw; inpl = planes * 4;
        }
    }
    e->out_size = 1024; e->final_hw = 49; e->final_c = 2048; e->final_creal = 2048;
}

const HostTensor *enc_find(pvr_encoder *e, const std::string &name) {
    auto it = e->weights.find(name);
    return it == e->weights.end() ? nullptr : &it->second;
}

pvr_status enc_need(pvr_encoder *e, const std::string &name, const HostTensor **out, size_t numel) {
    const HostTensor *t = enc_find(e, name);
    if (!t) { set_error("missing weight: %s", name.c_str()); return PVR_ERR_MISSING_WEIGHT; }
    if (t->data.size() != numel) {
        set_error("weight %s has %zu elements, expected %zu", name.c_str(), t->data.size(), numel);
        return PVR_ERR_INVALID;
    }
    *out = t;
    return PVR_OK;
}

static pvr_status bn_fold(pvr_encoder *e, const std::string &bn, int c, std::vector<float> &scale,
                          std::vector<float> &shift) {
    const HostTensor *g, *b, *m, *v;
    pvr_status s;
    if ((s = enc_need(e, bn + ".weight", &g, c))) return s;
    if ((s = enc_need(e, bn + ".bias", &b, c))) return s;
    if ((s = enc_need(e, bn + ".running_mean", &m, c))) return s;
    if ((s = enc_need(e, bn + ".running_var", &v, c))) return s;
    scale.resize(c); shift.resize(c);
    for (int i = 0; i < c; ++i) {
        scale[i] = g->data[i] / sqrtf(v->data[i] + 1e-5f);
        shift[i] = b->data[i] - m->data[i] * scale[i];
    }
    return PVR_OK;
}


static pvr_status finalize_conv(pvr_encoder *e, ConvOp &op) {
    const int k = op.k, cr = op.cin_real, cor = op.cout_real;
    const HostTensor *w;
    pvr_status s;
    if ((s = enc_need(e, op.conv + ".weight", &w, (size_t)cor * cr * k * k))) return s;
    std::vector<float> scale, shift;
    if ((s = bn_fold(e, op.bn, cor, scale, shift))) return s;
    if (const HostTensor *cb = enc_find(e, op.conv + ".bias")) {
        if ((int)cb->data.size() != cor) { set_error("bad bias size for %s", op.conv.c_str()); return PVR_ERR_INVALID; }
        for (int i = 0; i < cor; ++i) shift[i] += scale[i] * cb->data[i];
    }
    const int cout_pad = (op.cout + 63) / 64 * 64;
    const size_t K = (size_t)k * k * op.cin;
    std::vector<u16> hw(cout_pad * K, 0);
    for (int co = 0; co < cor; ++co)
        for (int ci = 0; ci < cr; ++ci)
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) {
                    const float v = w->data[(((size_t)co * cr + ci) * k + a) * k + b] * scale[co];
                    hw[co * K + ((size_t)a * k + b) * op.cin + ci] = f32_to_h(v, e->desc.dtype);
                }
    std::vector<float> hb(cout_pad, 0.f);
    for (int co = 0; co < cor; ++co) hb[co] = shift[co];
    if (e->desc.dtype == PVR_F32 || op.f32op || op.from32) {   // reference-precision mode / fp32 head of a 16-bit plan / a 16-bit conv reading fp32: same layout, fp32 values
        std::vector<float> hf(cout_pad * K, 0.f);
        for (int co = 0; co < cor; ++co)
            for (int ci = 0; ci < cr; ++ci)
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b)
                        hf[co * K + ((size_t)a * k + b) * op.cin + ci] = w->data[(((size_t)co * cr + ci) * k + a) * k + b] * scale[co];
        if ((s = enc_upload(&op.d_wf, hf))) return s;
        if ((op.f32op || op.from32) && e->desc.dtype == PVR_F16 && e->sw.split16 && conv_split16_supported(op.cin, op.cout, op.k)) {
            // the fp32 stage / head of the parity plan on the 16-bit matrix pipe: (hi, lo) f16 pairs of the same fp32 weights (conv_split16.hip); for a from32
            // convolution only the hi half is used: f16(w), the 16-bit plan's own weight.  (d_wf stays until the schedules are built: the head's pair image)
            PVR_HIP_TRY(hipMalloc((void **)&op.d_wsp, (size_t)cout_pad * K * 4));
            if ((s = launch_split16_pack(op.d_wf, op.d_wsp, cout_pad, (int)K, nullptr))) return s;
        }
        if (op.from32 && !op.d_wsp) { set_error("%s: a convolution that reads the fp32 stream needs the conv_split16 weight image (cin %d, cout %d)", op.conv.c_str(), op.cin, op.cout); return PVR_ERR_INVALID; }
        op.h_b = hb;
        return enc_upload(&op.d_b, hb);
    }
    if ((s = enc_upload(&op.d_w, hw))) return s;
    op.h_w = std::move(hw);
    op.h_b = hb;
    return enc_upload(&op.d_b, hb);
}

static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// Split-K plan (conv_igemm.hip::launch_conv_splitk): long narrow convolutions (K >= 16384 against Cout <= 64: the 3x3 compression
// head of the *_l4 PVRs, 98 pixel tiles of 288 K-slices; the *_l3 head has 392 tiles and is bound by its im2col reads instead,
// measured) get 8 K ranges and, as scratch for the fp32 partial planes, a 16-bit ping-pong buffer that is
// dead at that point of the plan (not read by this or any later op before it is overwritten).  PVR_SPLITK=0 turns it off.
static void plan_splitk(pvr_encoder *e) {
    if (!e->sw.splitk || e->desc.dtype == PVR_F32) return;
    const int n = (int)e->ops.size();
    for (int i = 0; i < n; ++i) {
        ConvOp &op = e->ops[i];
        if (op.kind != 0 || op.f32op || op.from32 || op.cout > 64 || op.k * op.k * op.cin < 16384 || (op.out_f32 & 2)) continue;
        const int ho = (op.h + 2 * op.pad - op.k) / op.stride + 1;
        const size_t need = (size_t)8 * e->desc.chunk * ho * ho * op.cout * sizeof(float);
        if (need > e->buf_elems * 2) continue;
        for (int b = 0; b < B_F32 && op.ks_buf == B_NONE; ++b) {      // (16-bit ping-pong buffers only)
            if (b == op.in_buf || b == op.out_buf || b == op.res_buf) continue;
            bool dead = true;
            for (int j = i + 1; j < n; ++j) {
                if (e->ops[j].in_buf == b || e->ops[j].res_buf == b) { dead = false; break; }
                if (e->ops[j].out_buf == b) break;
            }
            if (dead) { op.ks_buf = b; op.ksplit = 8; }
        }

    }
}

// Build both launch schedules.  Fused: every bottleneck of width 64 / 128 (layer1, layer2) runs as
// [conv1 unless the previous chain already produced it] [downsample] [chain: conv2 -> conv3 (+res) -> next conv1].
static pvr_status build_schedules(pvr_encoder *e) {
    const int n = (int)e->ops.size();
    for (int i = 0; i < n; ++i) { Launch l; l.conv2 = i; e->sched_plain.push_back(l); }
    if (e->desc.dtype == PVR_F32 || e->desc.arch == PVR_ARCH_CLIP_RN50) { e->sched_fused = e->sched_plain; return PVR_OK; }   // (CLIP: pools between the convolutions)
    int cur_t1 = B_T1;
    bool conv1_done = false;
    int conv1_frame_out = -1;                                   // t1 buffer the previous per-frame launch wrote the next conv1's output to
    for (int i = 0; i < n;) {
        ConvOp &op = e->ops[i];
        if (ends_with(op.conv, ".conv1") && conv1_done) { conv1_done = false; ++i; continue; }
        // index of the conv3 that closes the chain starting at conv2 `j`, or -1 when that bottleneck does not run as a chain
        auto chain_end = [&](int j) -> int {
            if (j < 0 || j >= n) return -1;
            const ConvOp &o2 = e->ops[j];
            if (!(ends_with(o2.conv, ".conv2") && o2.k == 3 && o2.cin == o2.cout && o2.cin_real == o2.cin && o2.cout_real == o2.cout) || o2.f32op) return -1;
            int c = j + 1;
            if (c < n && ends_with(e->ops[c].conv, ".downsample.0")) ++c;
            if (!(c < n && ends_with(e->ops[c].conv, ".conv3") && e->ops[c].cout == 4 * o2.cout && e->ops[c].relu &&
                  !e->ops[c].out_f32 && e->ops[c].res_buf != B_NONE && chain_supported(o2.cout, 0)))
                return -1;
            return c;
        };
        // layer3's stride-1 bottlenecks as ONE launch per block: conv1 -> conv2 -> conv3 + identity of one 14 x 14 image per workgroup
        // (bneck_frame.hip with the block's own conv1 in front; PVR_FRAME_FRONT1=0: conv1 keeps its launch)
        {
            const bool front_on = e->sw.frame_front1 && !e->sw.frame_next1;
            if (front_on && ends_with(op.conv, ".conv1") && op.k == 1 && op.stride == 1 && op.relu == 1 && !op.f32op && !op.out_f32 && op.tap.empty() && i + 2 < n &&
                op.cin_real == op.cin && op.cout_real == op.cout) {
                const ConvOp &o2 = e->ops[i + 1], &o3 = e->ops[i + 2];
                if (ends_with(o2.conv, ".conv2") && o2.k == 3 && !o2.f32op && o2.relu == 1 && o2.cin == o2.cout && o2.cin_real == o2.cin && o2.in_buf == op.out_buf &&
                    ends_with(o3.conv, ".conv3") && !o3.f32op && !o3.out_f32 && o3.relu == 1 && o3.res_buf == op.in_buf && o3.cout == op.cin && o3.cout_real == o3.cout &&
                    o3.in_buf == o2.out_buf && op.cout == o2.cin && bneck_frame_supported(e->desc.chunk, o2.h, o2.w, o2.cout, o3.cout, o2.stride)) {
                    Launch l;
                    l.conv1 = i; l.conv2 = i + 1; l.conv3 = i + 2; l.frame = 1; l.t1_in = op.out_buf;
                    e->sched_fused.push_back(l);
                    for (int oi : {l.conv1, l.conv2, l.conv3}) {
                        ConvOp &o = e->ops[oi];
                        if (o.d_wfb) continue;
                        const size_t K = (size_t)o.k * o.k * o.cin;
                        PVR_HIP_TRY(hipMalloc((void **)&o.d_wfb, (size_t)o.cout * K * 2));
                        pvr_status s = launch_pack_frag_weights(o.d_w, o.d_wfb, o.cout, (int)K, nullptr);
                        if (s) return s;
                    }
                    i += 3;
                    continue;
                }
            }
        }
        // layer3's stride-1 bottlenecks: conv2 -> conv3 + residual of one 14 x 14 image per workgroup (bneck_frame.hip); with PVR_FRAME_NEXT1=1 the
        // next block's conv1 rides in the same launch (it then reads / writes the two t1 buffers in turns, as the layer1 / layer2 chains do)
        if (ends_with(op.conv, ".conv2") && op.k == 3 && !op.f32op && i + 1 < n && ends_with(e->ops[i + 1].conv, ".conv3") && !e->ops[i + 1].f32op &&
            !e->ops[i + 1].out_f32 && e->ops[i + 1].relu == 1 && e->ops[i + 1].res_buf != B_NONE && op.relu == 1 && op.cin == op.cout && op.cin_real == op.cin &&
            e->ops[i + 1].cout_real == e->ops[i + 1].cout && bneck_frame_supported(e->desc.chunk, op.h, op.w, op.cout, e->ops[i + 1].cout, op.stride)) {
            Launch l;
            l.conv2 = i; l.conv3 = i + 1; l.frame = 1; l.t1_in = conv1_frame_out >= 0 ? conv1_frame_out : op.in_buf;
            conv1_frame_out = -1;
            const bool next1_on = e->sw.frame_next1 != 0;
            const int nx = i + 2;
            if (next1_on && nx + 2 < n && ends_with(e->ops[nx].conv, ".conv1") && e->ops[nx].k == 1 && e->ops[nx].stride == 1 && e->ops[nx].relu == 1 && !e->ops[nx].f32op &&
                e->ops[nx].cin == e->ops[i + 1].cout && e->ops[nx].cout == op.cout && e->ops[nx].in_buf == e->ops[i + 1].out_buf && e->ops[nx].cout_real == e->ops[nx].cout &&
                e->ops[i + 1].tap.empty() && ends_with(e->ops[nx + 1].conv, ".conv2") && ends_with(e->ops[nx + 2].conv, ".conv3") &&
                bneck_frame_supported(e->desc.chunk, e->ops[nx + 1].h, e->ops[nx + 1].w, e->ops[nx + 1].cout, e->ops[nx + 2].cout, e->ops[nx + 1].stride)) {
                l.next1 = nx;
                l.t1_out = l.t1_in == B_T1 ? B_T2 : B_T1;
                conv1_frame_out = l.t1_out;
                conv1_done = true;                              // (the loop skips that conv1: it ran inside this launch)
            }
            e->sched_fused.push_back(l);
            for (int oi : {l.conv2, l.conv3, l.next1}) {
                if (oi < 0 || e->ops[oi].d_wfb) continue;
                ConvOp &o = e->ops[oi];
                const size_t K = (size_t)o.k * o.k * o.cin;
                PVR_HIP_TRY(hipMalloc((void **)&o.d_wfb, (size_t)o.cout * K * 2));
                pvr_status s = launch_pack_frag_weights(o.d_w, o.d_wfb, o.cout, (int)K, nullptr);
                if (s) return s;
            }
            i += 2;
            continue;
        }
        const int c3 = chain_end(i);
        if (c3 < 0) {
            // A stride-2 bottleneck outside the chains (layer3.0, layer4.0): its 1 x 1 downsample and the conv3 that adds it run as ONE two-operand
            // launch (conv_pp256 DUAL: K = conv3's channels, then the block input's) - the identity branch is accumulated in fp32 and never exists in
            // HBM (- 2 x 103 MB at layer3.0, - 2 x 51 MB at layer4.0 per 256 frames, one launch less).  PVR_DUAL_DS=0: separate launches.
            const bool dual_on = e->sw.dual_ds != 0;
            if (dual_on && ends_with(op.conv, ".downsample.0") && op.kind == 0 && !op.f32op && op.k == 1 && op.pad == 0 && !op.relu && !op.out_f32 &&
                op.res_buf == B_NONE && op.tap.empty() && op.cin_real == op.cin && op.cout_real == op.cout && op.cin % 64 == 0 && i + 1 < n) {
                ConvOp &o3 = e->ops[i + 1];
                if (ends_with(o3.conv, ".conv3") && o3.kind == 0 && !o3.f32op && o3.k == 1 && o3.stride == 1 && o3.pad == 0 && o3.relu == 1 && !o3.out_f32 &&
                    o3.res_buf == op.out_buf && o3.cout == op.cout && o3.cout_real == o3.cout && o3.cin_real == o3.cin && o3.cin % 64 == 0 &&
                    (op.h - 1) / op.stride + 1 == o3.h && (op.w - 1) / op.stride + 1 == o3.w && o3.cout >= 256 && o3.ksplit <= 1 && op.ksplit <= 1) {
                    const size_t K1 = (size_t)o3.cin, K2 = (size_t)op.cin, cp = ((size_t)o3.cout + 63) / 64 * 64;
                    std::vector<u16> wc(cp * (K1 + K2), 0);
                    for (int r = 0; r < o3.cout; ++r) {
                        memcpy(&wc[(size_t)r * (K1 + K2)], &o3.h_w[(size_t)r * K1], K1 * 2);
                        memcpy(&wc[(size_t)r * (K1 + K2) + K1], &op.h_w[(size_t)r * K2], K2 * 2);
                    }
                    std::vector<float> bs(cp, 0.f);
                    for (int c = 0; c < o3.cout; ++c) bs[c] = o3.h_b[c] + op.h_b[c];
                    pvr_status s = enc_upload(&o3.d_wcat, wc);
                    if (!s) s = enc_upload(&o3.d_bsum, bs);
                    if (s) return s;
                    Launch l; l.conv2 = i + 1; l.ds = i;
                    e->sched_fused.push_back(l);
                    i += 2;
                    continue;
                }
            }
            // the compression head: conv1 (+ ReLU) and the downsample convolution read the SAME fp32 tensor with the same geometry: one conv_split16 launch over
            // [W1 ; Wd] (128 couts), two outputs - the 205 / 103 MB input is read once
            if (op.f32op && op.d_wsp && op.d_wf && i + 1 < n) {
                ConvOp &od = e->ops[i + 1];
                if (od.f32op && od.d_wsp && od.d_wf && od.in_buf == op.in_buf && od.k == op.k && od.stride == op.stride && od.pad == op.pad && od.cin == op.cin && od.h == op.h &&
                    op.cout == 64 && od.cout == 64 && op.relu == 1 && od.relu == 0 && op.res_buf == B_NONE && od.res_buf == B_NONE && op.tap.empty() && od.tap.empty() &&
                    ends_with(op.conv, ".conv1") && ends_with(od.conv, ".downsample.0")) {
                    const size_t K = (size_t)op.k * op.k * op.cin;
                    float *cat = nullptr;
                    PVR_HIP_TRY(hipMalloc((void **)&cat, 128 * K * 4));
                    PVR_HIP_TRY(hipMemcpy(cat, op.d_wf, 64 * K * 4, hipMemcpyDeviceToDevice));
                    PVR_HIP_TRY(hipMemcpy(cat + 64 * K, od.d_wf, 64 * K * 4, hipMemcpyDeviceToDevice));
                    PVR_HIP_TRY(hipMalloc((void **)&op.d_wsp_pair, 128 * K * 4));
                    pvr_status s = launch_split16_pack(cat, op.d_wsp_pair, 128, (int)K, nullptr);
                    PVR_HIP_TRY(hipDeviceSynchronize());
                    (void)hipFree(cat);
                    if (s) return s;
                    std::vector<float> bb(128, 0.f);
                    for (int c = 0; c < 64; ++c) { bb[c] = op.h_b[c]; bb[64 + c] = od.h_b[c]; }
                    if ((s = enc_upload(&op.d_b_pair, bb))) return s;
                    Launch l; l.conv2 = i; l.pair = i + 1;
                    e->sched_fused.push_back(l);
                    i += 2;
                    continue;
                }
            }
            Launch l; l.conv2 = i;
            e->sched_fused.push_back(l);
            // a stand-alone convolution with few pixels and a deep K (layer3 / layer4's 1 x 1 and 3 x 3 at 14 x 14 and 7 x 7): conv_wfrag.hip may take it at
            // run time (conv_wfrag_preferred: by the batch) - it reads the fragment-blocked copy of the weights
            if (op.kind == 0 && !op.f32op && !op.from32 && !op.d_wfb && op.h == op.w && op.h <= 14 && op.cout_real == op.cout && (int64_t)op.k * op.k * op.cin >= 512 &&
                conv_wfrag_supported(1, 1, op.cin, op.cout, op.k, op.k, op.pad, op.relu, op.out_f32)) {
                const size_t K = (size_t)op.k * op.k * op.cin;
                PVR_HIP_TRY(hipMalloc((void **)&op.d_wfb, (size_t)op.cout * K * 2));
                pvr_status s = launch_pack_frag_weights(op.d_w, op.d_wfb, op.cout, (int)K, nullptr);
                if (s) return s;
            }
            if (ends_with(op.conv, ".conv1")) cur_t1 = B_T1;
            ++i;
            continue;
        }
        Launch l;
        l.conv2 = i; l.conv3 = c3; l.t1_in = cur_t1;
        const int nx = c3 + 1;
        // layer1's block 0: its 64-channel stride-1 downsample is accumulated inside the chain's conv3 (PVR_CHAIN_DS=0: own launch)
        const bool ds_on = e->sw.chain_ds != 0;
        if (c3 == i + 2 && ds_on && nx < n) {
            const ConvOp &d = e->ops[i + 1];
            if (d.k == 1 && d.pad == 0 && !d.relu && !d.out_f32 && !d.f32op && d.cin_real == d.cin && d.cout == 4 * op.cout &&
                d.out_buf == e->ops[c3].res_buf && chain_ds_supported(op.cout, e->ops[nx].cout, d.cin, d.stride) && op.stride == 1)
                l.ds = i + 1;
        }
        if (l.ds < 0)
            for (int d = i + 1; d < c3; ++d) { Launch l2; l2.conv2 = d; e->sched_fused.push_back(l2); }   // the downsample runs first
        // the next block's conv1 rides in this chain only if that block is a chain itself: the chain leaves t1' in the OTHER of the two
        // t1 buffers (it reads one while it writes the next), which only a following chain knows to read (l.t1_in); a plain conv2 launch
        // reads its own in_buf.  (Round 3: with the fp32 residual stream of the compressed PVRs' parity plan starting at layer2, layer1's
        // last chain is followed by plain launches.)
        if (nx < n && ends_with(e->ops[nx].conv, ".conv1") && e->ops[nx].k == 1 && e->ops[nx].stride == 1 && e->ops[nx].relu &&
            e->ops[nx].cin == 4 * op.cout && e->ops[nx].in_buf == e->ops[c3].out_buf && e->ops[nx].cout_real == e->ops[nx].cout &&
            chain_supported(op.cout, e->ops[nx].cout) && chain_end(nx + 1) >= 0) {
            l.next1 = nx;
            l.t1_out = cur_t1 == B_T1 ? B_T2 : B_T1;
            cur_t1 = l.t1_out;
            conv1_done = true;
        }
        if (l.ds >= 0 && l.next1 < 0) {               // (the DS instance carries a next conv1)
            for (int d = i + 1; d < c3; ++d) { Launch l2; l2.conv2 = d; e->sched_fused.push_back(l2); }
            l.ds = -1;
        }
        e->sched_fused.push_back(l);
        if (l.ds >= 0) {
            ConvOp &o3 = e->ops[c3];
            std::vector<float> bs(o3.h_b);
            for (size_t c = 0; c < bs.size(); ++c) bs[c] += e->ops[l.ds].h_b[c];
            pvr_status s = enc_upload(&o3.d_bsum, bs);
            if (s) return s;
        }
        // row-permuted copies of the chain's 1x1 weights
        for (int which = 0; which < 3; ++which) {
            const int oi = which == 0 ? c3 : which == 1 ? l.next1 : l.ds;
            if (oi < 0) continue;
            ConvOp &o = e->ops[oi];
            if (o.d_wp) continue;
            const size_t K = (size_t)o.cin;
            std::vector<u16> hp((size_t)o.cout * K);
            for (int r = 0; r < o.cout; ++r) memcpy(&hp[(size_t)r * K], &o.h_w[(size_t)chain_row_source(r) * K], K * 2);
            pvr_status s = enc_upload(&o.d_wp, hp);
            if (s) return s;
            if (which != 1) {                        // W3 / Wd once more, in the blocked layout the wave form reads its L2-resident pieces in
                std::vector<u16> hb(hp.size());
                for (int r = 0; r < o.cout; ++r)
                    for (size_t c = 0; c < K; ++c) hb[(((size_t)(r >> 4) * (K / 8) + (c >> 3)) * 16 + (r & 15)) * 8 + (c & 7)] = hp[(size_t)r * K + c];
                if ((s = enc_upload(&o.d_wpb, hb))) return s;
            }
        }
        i = c3 + 1;
    }
    // Two consecutive wave-form tails hand y (the second one's residual) and t1' (its conv2 input) over in the blocked layout
    // (chain_wave.hip): only when nothing else reads those two buffers in between - no tap, no other launch - and the geometry allows it.
    for (Launch &l : e->sched_fused)
        if (l.conv3 >= 0 && !l.frame) {
            const ConvOp &c2 = e->ops[l.conv2];
            l.wave = chain_uses_wave_form(c2.cout, l.next1 >= 0 ? e->ops[l.next1].cout : 0, c2.stride, l.ds >= 0);
        }
    // layer2's stride-1 tails on their wave form (chain_wave128.hip, round 6).  That kernel reads t1 and the residual blocked and writes t1' blocked, so a
    // launch can take it only if (i) the launch in front is a chain that carries this block's conv1 and hands y and t1' over untapped - the block form (it can
    // write both blocked: out_blk 1 | 2) or another launch of this form - and (ii) the launch behind it, if it carries the next conv1 here, takes this form too.
    {
        std::vector<Launch> &sc = e->sched_fused;
        const int ns = (int)sc.size();
        auto linked = [&](int a, int b) {               // sc[a] hands y (as the residual) and t1' straight to sc[b]
            if (a < 0 || b >= ns) return false;
            const Launch &A = sc[a], &B = sc[b];
            if (A.conv3 < 0 || B.conv3 < 0 || A.next1 < 0 || A.frame || B.frame || B.ds >= 0) return false;
            const ConvOp &a2 = e->ops[A.conv2], &a3 = e->ops[A.conv3], &b2 = e->ops[B.conv2], &b3 = e->ops[B.conv3];
            return B.t1_in == A.t1_out && b3.res_buf == a3.out_buf && a3.tap.empty() && b2.h == a2.h / a2.stride && b2.w == a2.w / a2.stride && A.next1 + 1 == B.conv2;
        };
        auto eligible = [&](int b) {
            const Launch &B = sc[b];
            if (B.conv3 < 0 || B.frame || B.ds >= 0 || B.wave) return false;
            const ConvOp &b2 = e->ops[B.conv2];
            return (b2.h * b2.w) % 16 == 0 && e->sw.chain_blocked && chain_uses_wave128(b2.cout, B.next1 >= 0 ? e->ops[B.next1].cout : 0, b2.stride, (int64_t)b2.h * b2.w);
        };
        std::vector<char> can(ns, 0);
        for (int b = ns - 1; b >= 1; --b)
            can[b] = eligible(b) && linked(b - 1, b) && (sc[b].next1 < 0 || (b + 1 < ns && can[b + 1] && linked(b, b + 1)));
        for (int b = 1; b < ns; ++b) {
            if (!can[b] || !(sc[b - 1].wave == 0 || sc[b - 1].wave == 2)) continue;
            Launch &B = sc[b];
            B.wave = 2;
            ConvOp &c2 = e->ops[B.conv2];
            if (!c2.d_wpk) {
                PVR_HIP_TRY(hipMalloc((void **)&c2.d_wpk, chain_wave128_pack_bytes()));
                pvr_status s = launch_chain_wave128_pack(c2.d_w, e->ops[B.conv3].d_wp, B.next1 >= 0 ? e->ops[B.next1].d_wp : nullptr, c2.d_wpk, nullptr);
                if (s) return s;
            }
        }
    }
    for (size_t a = 0; a + 1 < e->sched_fused.size(); ++a) {
        Launch &A = e->sched_fused[a], &B = e->sched_fused[a + 1];
        if (A.conv3 < 0 || B.conv3 < 0 || A.next1 < 0 || B.ds >= 0 || A.frame || B.frame) continue;
        const ConvOp &a2 = e->ops[A.conv2], &a3 = e->ops[A.conv3], &b2 = e->ops[B.conv2], &b3 = e->ops[B.conv3];
        const int a_cmn = e->ops[A.next1].cout;
        if (!e->sw.chain_blocked) continue;
        if (B.t1_in != A.t1_out || b3.res_buf != a3.out_buf || !a3.tap.empty() || b2.h != a2.h / a2.stride || b2.w != a2.w / a2.stride || b2.stride != 1) continue;
        if (B.wave == 2) {                            // a layer2 wave-form tail: everything it reads arrives blocked (its producer: block form or this form)
            A.out_blk = A.wave == 2 ? 1 : 3; B.in_blk = 1;
            continue;
        }
        if (!A.wave && !B.wave) {
            // two block-form tails (layer2): y = the next residual travels blocked (16-byte accesses of a lane land in 512-byte runs);
            // t1' stays NHWC (the halo DMA wants contiguous pixel rows)
            if ((b2.h * b2.w) % 16 == 0) { A.out_blk = 1; B.in_blk = 1; }
            continue;
        }
        if (!A.wave || !B.wave) continue;
        if (!chain_wave_blocked_ok(a_cmn, a2.h, a2.w)) continue;
        A.out_blk = 1; B.in_blk = 1;
        // ... and when A is layer1's first tail (downsample inside), its conv2 input t1 can arrive blocked too: from conv1's own launch,
        // which directly precedes it (conv_expand.hip writes either layout; whether THAT kernel runs is known per forward: batch size)
        if (A.ds >= 0 && a > 0 && a2.w == 56 && chain_wave_halo_enabled()) {   // (the downsample tail reads a blocked t1 through the halo form only)
            Launch &C = e->sched_fused[a - 1];
            if (C.conv3 < 0 && C.conv2 >= 0) {
                const ConvOp &c1 = e->ops[C.conv2];
                if (c1.kind == 0 && !c1.f32op && c1.k == 1 && c1.stride == 1 && c1.out_buf == A.t1_in && c1.tap.empty() && c1.res_buf == B_NONE && !c1.out_f32 &&
                    c1.ksplit <= 1) { C.out_blk = 1; A.in_blk = 1; }
            }
        }
    }
    // layer1.0.conv1 (1 x 1, 64 -> 64 on the pooled stem output) inside the fused stem (stem.hip, StemC1; round 6): the launch leaves the fused schedule; the
    // forward hands the stem its weights, or - where the stem's register-pooling form does not run - launches the convolution itself in front of the plan
    if (e->sw.stem_conv1 && !e->sched_fused.empty() && (e->desc.arch == PVR_ARCH_RESNET50 || e->desc.arch == PVR_ARCH_RESNET50_L3 || e->desc.arch == PVR_ARCH_RESNET50_L4)) {
        const Launch &L0 = e->sched_fused[0];
        if (L0.conv3 < 0 && L0.ds < 0 && L0.pair < 0 && !L0.frame && L0.conv2 == 0) {
            const ConvOp &c1 = e->ops[0];
            if (c1.kind == 0 && !c1.f32op && !c1.from32 && c1.k == 1 && c1.stride == 1 && c1.cin == 64 && c1.cout == 64 && c1.cin_real == 64 && c1.cout_real == 64 && c1.relu == 1 &&
                c1.in_buf == B_X0 && c1.out_buf == B_T1 && c1.res_buf == B_NONE && !c1.out_f32 && c1.tap.empty() && c1.ksplit <= 1 && c1.h == 56 && (int)c1.h_w.size() == 64 * 64) {
                std::vector<u16> img(64 * 64);
                stem_c1_pack(c1.h_w.data(), img.data());
                pvr_status s = enc_upload(&e->d_stem_c1w, img);
                if (s) return s;
                e->stem_c1 = 0; e->stem_c1_blk = L0.out_blk;
                e->sched_fused.erase(e->sched_fused.begin());
            }
        }
    }
    return PVR_OK;
}

// conv_name (cout_real, 3, ks, ks) with ks = 7 (torchvision) or 3 (CLIP: stride 2, pad 1 == the centre 3x3 taps of a 7x7 / pad 3)
static pvr_status finalize_stem(pvr_encoder *e, const std::string &conv_name = "conv1.weight", const std::string &bn_name = "bn1",
                                int cout_real = 64, int ks = 7) {
    const HostTensor *w;
    pvr_status s;
    if ((s = enc_need(e, conv_name, &w, (size_t)cout_real * 3 * ks * ks))) return s;
    std::vector<float> scale, shift;
    if ((s = bn_fold(e, bn_name, cout_real, scale, shift))) return s;
    scale.resize(64, 0.f); shift.resize(64, 0.f);
    const int t0 = (7 - ks) / 2;
    auto wat = [&](int co, int c, int a, int b) -> double {       // 7x7 view of the (possibly smaller) filter
        const int aa = a - t0, bb = b - t0;
        if (co >= cout_real || aa < 0 || aa >= ks || bb < 0 || bb >= ks) return 0.0;
        return (double)w->data[(((size_t)co * 3 + c) * ks + aa) * ks + bb];
    };
    if (e->desc.dtype == PVR_F32) {               // normalisation stays a separate fp32 kernel, exactly as torch applies it
        std::vector<float> hf(64 * 49 * 4, 0.f);
        for (int co = 0; co < 64; ++co)
            for (int c = 0; c < 3; ++c)
                for (int t = 0; t < 49; ++t) hf[((size_t)co * 49 + t) * 4 + c] = (float)wat(co, c, t / 7, t % 7) * scale[co];
        if ((s = enc_upload(&e->d_stem_wf, hf))) return s;
        return enc_upload(&e->d_stem_b, shift);
    }
    std::vector<u16> hw(64 * 224, 0);
    for (int co = 0; co < 64; ++co)
        for (int a = 0; a < 7; ++a)
            for (int b = 0; b < 7; ++b) {
                double vsum = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double wv = wat(co, c, a, b) * scale[co];
                    // (x/255 - mean)/std with x = xc + 128:  xc/(255 std) + (128 - 255 mean)/(255 std)
                    hw[co * 224 + (a * 8 + b) * 4 + c] = f32_to_h((float)(wv / (255.0 * e->desc.std_[c])), e->desc.dtype);
                    vsum += wv * (128.0 - 255.0 * e->desc.mean[c]) / (255.0 * e->desc.std_[c]);
                }
                hw[co * 224 + (a * 8 + b) * 4 + 3] = f32_to_h((float)vsum, e->desc.dtype);
            }
    if ((s = enc_upload(&e->d_stem_w, hw))) return s;
    return enc_upload(&e->d_stem_b, shift);
}

// CLIP AttentionPool2d parameters: q/k/v projections concatenated row-wise (one GEMM), c_proj, positional embedding
static pvr_status finalize_attnpool(pvr_encoder *e) {
    const int C = 2048, O = 1024, dt = e->desc.dtype;
    const std::string a = "visual.attnpool.";
    pvr_status s;
    std::vector<u16> wq((size_t)3 * C * C);
    std::vector<float> bq((size_t)3 * C);
    const char *nm[3] = {"q_proj", "k_proj", "v_proj"};
    for (int i = 0; i < 3; ++i) {
        const HostTensor *w, *b;
        if ((s = enc_need(e, a + nm[i] + ".weight", &w, (size_t)C * C))) return s;
        if ((s = enc_need(e, a + nm[i] + ".bias", &b, (size_t)C))) return s;
        for (size_t k = 0; k < (size_t)C * C; ++k) wq[(size_t)i * C * C + k] = f32_to_h(w->data[k], dt);
        for (int k = 0; k < C; ++k) bq[(size_t)i * C + k] = b->data[k];
    }
    const HostTensor *wc, *bc, *pos;
    if ((s = enc_need(e, a + "c_proj.weight", &wc, (size_t)O * C))) return s;
    if ((s = enc_need(e, a + "c_proj.bias", &bc, (size_t)O))) return s;
    if ((s = enc_need(e, a + "positional_embedding", &pos, (size_t)50 * C))) return s;
    std::vector<u16> wch((size_t)O * C);
    for (size_t k = 0; k < wch.size(); ++k) wch[k] = f32_to_h(wc->data[k], dt);
    if ((s = enc_upload(&e->ap_wqkv, wq)) || (s = enc_upload(&e->ap_bqkv, bq)) || (s = enc_upload(&e->ap_wc, wch)) ||
        (s = enc_upload(&e->ap_bc, bc->data)) || (s = enc_upload(&e->ap_pos, pos->data))) return s;
    PVR_HIP_TRY(hipMalloc((void **)&e->ap_out, (size_t)e->desc.chunk * O * sizeof(float) * PVR_MAX_LANES));
    return PVR_OK;
}

}  // namespace pvr

static bool pooled_head(const pvr_encoder *enc) {   // global average pool of the fp32 last activation (vs C-major flatten of the compression heads)
    return enc->desc.arch == PVR_ARCH_RESNET50 || enc->desc.arch == PVR_ARCH_RESNET18 || enc->desc.arch == PVR_ARCH_RESNET34;
}

static void *bufp(pvr_encoder *enc, int id) { return id == B_STEM ? (void *)enc->d_stem : enc->d_buf[id]; }

// Low-latency plan (pvr_encoder_set_low_latency; the online pattern of EmbeddingWrapper: N = 2 frames per environment step).
// A forward of <= 4 frames has 1-7 pixel tiles in layer3 / layer4, so every deep convolution is a handful of blocks each
// walking its whole K range alone: 21 such launches x 27 us were 65 % of a 0.88 ms N = 2 forward (rocprofv3,
// profiles/r02_small_batch_kernel_stats.csv).  Here K is cut into ranges of ~4 slices over blockIdx.y (conv_igemm split-K: fp32
// partial planes + a fixed-order reduce with bias / residual / ReLU).  The split depends on the layer's K only, so results do
// not depend on N within the plan; against the unsplit plan they differ by fp32 regrouping (<= 1 ulp of the storage type), which
// is why the plan is opt-in and batch-size independence of the default plan stays bit-exact.
constexpr size_t SMALLK_BYTES = (size_t)32 << 20;
static int small_batch_ksplit(const pvr_encoder *enc, const ConvOp &op, int nb) {
    if (!enc->low_latency || nb > 4 || op.kind != 0 || op.f32op || op.from32 || op.ksplit > 1 || op.relu > 1 || (op.out_f32 & 2)) return 0;
    const int K = op.k * op.k * op.cin, nk = K / 64;
    if (nk < 8) return 0;                                        // K < 512: nothing to share
    const int ho = (op.h + 2 * op.pad - op.k) / op.stride + 1;
    const long long M = (long long)nb * ho * ho, blocks = ((M + 127) / 128) * ((op.cout + 127) / 128);
    if (blocks > 64) return 0;
    const int div = enc->sw.smallk_div;                          // K slices per block (PVR_SMALLK_DIV, read at create)
    int ks = nk / div;
    if (ks > (div >= 4 ? 16 : 32)) ks = div >= 4 ? 16 : 32;
    if ((size_t)ks * M * op.cout * sizeof(float) > SMALLK_BYTES) return 0;
    return ks;
}

// The low-latency plan's scratch, for every lane that has a workspace: allocated when the plan is switched on (pvr_encoder_set_low_latency), at
// finalize when it was switched on before, and when a lane's workspace is first made - never inside a forward (no allocation on the
// forward path after finalize).
static pvr_status ensure_smallk(pvr_encoder *enc) {
    if (!enc->low_latency || enc->vit || enc->rnd || enc->host) return PVR_OK;
    for (int l = 0; l < PVR_MAX_LANES; ++l)
        if (enc->lane_ws[l].valid && !enc->d_smallk[l]) PVR_HIP_TRY(hipMalloc((void **)&enc->d_smallk[l], SMALLK_BYTES));
    return PVR_OK;
}

// Which kernel launch `li` of the plan runs as for a forward of nb frames (allow_pool = false: the caller's output rows cannot take the
// pooled epilogue's 16-byte stores).  A pure function of the plan, the switches and nb: tabulated by resolve_kinds, off the hot path.
static uint8_t resolve_kind(const pvr_encoder *enc, const std::vector<Launch> &plan, size_t li, int nb, bool allow_pool = true) {
    const Launch &l = plan[li];
    const ConvOp &op = enc->ops[l.conv3 >= 0 ? l.conv3 : l.conv2];
    const bool ll = enc->low_latency && nb <= 4;                 // (the low-latency plan covers forwards of <= 4 frames: small_batch_ksplit)
    const bool autoalgo = conv_algo() == -1;
    if (l.frame) {
        // small batches (a frame per workgroup leaves most CUs idle): the member convolutions as their own launches - bit-identical
        if (nb >= enc->sw.frame_min_n && !enc->low_latency) return l.conv1 >= 0 ? LK_FRAME_FRONT1 : LK_FRAME;
        return LK_FRAME_MEMBERS;
    }
    if (l.conv3 < 0 && l.ds >= 0) return (!ll && autoalgo) ? LK_DUAL : LK_DUAL_MEMBERS;
    if (l.conv3 >= 0) return LK_CHAIN;
    if (op.kind == 2) return LK_CAST;
    if (l.pair >= 0) return LK_SPLIT16_PAIR;
    if (op.f32op) return op.d_wsp ? LK_SPLIT16 : LK_F32;
    if (op.from32) return LK_SPLIT16_IN32;
    if (small_batch_ksplit(enc, op, nb)) return LK_SPLITK_SMALL;
    if (op.ksplit > 1) return LK_SPLITK;
    const int ho = (op.h + 2 * op.pad - op.k) / op.stride + 1, wo = (op.w + 2 * op.pad - op.k) / op.stride + 1;
    if (l.out_blk && autoalgo && op.cin == 64 &&                 // (blocked output: the cin = 64 instances of conv_expand only)
        conv_expand_supported((int64_t)nb * op.h * op.w, op.h, op.w, op.cin, op.cout, 1, 1, 1, 0, op.relu, 0, false))
        return LK_EXPAND_BLOCKED;
    if (allow_pool && enc->sw.pool_fuse && li + 1 == plan.size() && op.d_wfb && pooled_head(enc) && enc->final_hw == 49 && op.h == 7 && op.w == 7 && op.k == 1 &&
        op.stride == 1 && op.relu == 1 && (op.out_f32 & 1) && !(op.out_f32 & 2) && op.res_buf != B_NONE && op.out_buf == B_F32 && enc->final_c == op.cout &&
        autoalgo && !ll)
        return LK_WFRAG_POOL;
    if (op.d_wfb && autoalgo && !ll && conv_wfrag_preferred((int64_t)nb * ho * wo, op.cin, op.cout, op.k, op.k) &&
        conv_wfrag_supported((int64_t)nb * ho * wo, (int64_t)nb * op.h * op.w * op.cin * 2, op.cin, op.cout, op.k, op.k, op.pad, op.relu, op.out_f32))
        return LK_WFRAG;
    return LK_CONV;
}

static const std::vector<Launch> &cur_plan(const pvr_encoder *enc) { return enc->fuse ? enc->sched_fused : enc->sched_plain; }

// kinds[(nb - 1) * launches + i] for nb = 1 .. chunk: rebuilt whenever something it depends on changes (finalize, set_low_latency,
// debug_set_fusion, debug_set_switch; pvr_debug_set_conv_algo is process-wide, so the forward compares kinds_algo first)
static void resolve_kinds(pvr_encoder *enc) {
    const std::vector<Launch> &plan = cur_plan(enc);
    const int chunk = enc->desc.chunk;
    enc->kinds_stride = plan.size();
    enc->kinds.assign((size_t)chunk * plan.size(), LK_CONV);
    if (enc->desc.dtype == PVR_F32 || enc->desc.arch == PVR_ARCH_CLIP_RN50 || enc->vit || enc->rnd || enc->host) { enc->kinds_algo = conv_algo(); return; }
    for (int nb = 1; nb <= chunk; ++nb)
        for (size_t i = 0; i < plan.size(); ++i) enc->kinds[(size_t)(nb - 1) * plan.size() + i] = resolve_kind(enc, plan, i, nb);
    // consecutive whole-bottleneck frame launches, each reading its predecessor's output (layer3.1 .. 3.5): one launch for the run (bneck_frame.hip RUN)
    if (enc->sw.frame_run)
        for (int nb = 1; nb <= chunk; ++nb) {
            uint8_t *k = enc->kinds.data() + (size_t)(nb - 1) * plan.size();
            for (size_t i = 0; i + 1 < plan.size(); ++i) {
                if (k[i] != LK_FRAME_FRONT1) continue;
                size_t j = i;
                while (j + 1 < plan.size() && j + 1 - i < 6 && k[j + 1] == LK_FRAME_FRONT1 &&
                       enc->ops[plan[j + 1].conv3].res_buf == enc->ops[plan[j].conv3].out_buf && enc->ops[plan[j + 1].conv1].in_buf == enc->ops[plan[j].conv3].out_buf) ++j;
                if (j > i) { k[i] = LK_FRAME_RUN; for (size_t t = i + 1; t <= j; ++t) k[t] = LK_FRAME_RUN_TAIL; }
                i = j;
            }
        }
    enc->kinds_algo = conv_algo();
}

// activation workspace of the current lane (ResNet50 family)
static pvr_status alloc_workspace(pvr_encoder *enc) {
    const int C = enc->desc.chunk, crop = enc->desc.crop;
    const bool f32 = enc->desc.dtype == PVR_F32;
    const size_t esz = f32 ? 4 : 2;                               // activation element size
    const size_t img = (size_t)C * (crop + 6) * (crop + 8) * 4;
    PVR_HIP_TRY(hipMalloc((void **)&enc->d_img, img * 2));
    PVR_HIP_TRY(hipMemset(enc->d_img, 0, img * 2));            // zero border = conv1 padding, written once
    PVR_HIP_TRY(hipMalloc((void **)&enc->d_stem, (size_t)C * 112 * 112 * 64 * esz));
    if (f32) PVR_HIP_TRY(hipMalloc((void **)&enc->d_imgf, (size_t)C * crop * crop * 4 * sizeof(float)));
    enc->buf_elems = (size_t)C * 56 * 56 * 256;                 // largest activation (layer1 output)
    for (int b = 0; b < B_COUNT; ++b) {
        size_t bytes = enc->buf_elems * esz;
        if (b == B_F32) bytes = (size_t)C * enc->final_hw * enc->final_c * 4;
        if ((b == B_Y0 || b == B_Y1) && !enc->resid32) continue;             // fp32 residual stream: parity plan of the compressed PVRs only
        PVR_HIP_TRY(hipMalloc(&enc->d_buf[b], bytes));
    }
    return PVR_OK;
}
static void save_lane(pvr_encoder *enc, int lane) {
    auto &l = enc->lane_ws[lane];
    l.d_img = enc->d_img; l.d_stem = enc->d_stem; l.d_imgf = enc->d_imgf;
    for (int b = 0; b < B_COUNT; ++b) l.d_buf[b] = enc->d_buf[b];
    l.valid = true; enc->cur_lane = lane;
}
static pvr_status use_lane(pvr_encoder *enc, int lane) {
    if (lane == enc->cur_lane) return PVR_OK;
    if (!enc->lane_ws[lane].valid) {                            // first use: allocate, off the hot path
        // allocate into the encoder's current-pointer slots, but keep the previous lane's pointers aside: if any hipMalloc fails
        // (a ~3 GB workspace can), free what was allocated and put the previous lane back, so the encoder stays usable
        const int prev = enc->cur_lane;
        enc->d_img = nullptr; enc->d_stem = nullptr; enc->d_imgf = nullptr;
        for (int b = 0; b < B_COUNT; ++b) enc->d_buf[b] = nullptr;
        pvr_status s = alloc_workspace(enc);
        if (!s && hipDeviceSynchronize() != hipSuccess) { set_error("use_lane: device sync failed"); s = PVR_ERR_HIP; }
        if (s) {
            for (int b = 0; b < B_COUNT; ++b) if (enc->d_buf[b]) (void)hipFree(enc->d_buf[b]);
            if (enc->d_img) (void)hipFree(enc->d_img);
            if (enc->d_stem) (void)hipFree(enc->d_stem);
            if (enc->d_imgf) (void)hipFree(enc->d_imgf);
            const auto &l = enc->lane_ws[prev];
            enc->d_img = l.d_img; enc->d_stem = l.d_stem; enc->d_imgf = l.d_imgf;
            for (int b = 0; b < B_COUNT; ++b) enc->d_buf[b] = l.d_buf[b];
            return s;
        }
        save_lane(enc, lane);
        return ensure_smallk(enc);
    }
    const auto &l = enc->lane_ws[lane];
    enc->d_img = l.d_img; enc->d_stem = l.d_stem; enc->d_imgf = l.d_imgf;
    for (int b = 0; b < B_COUNT; ++b) enc->d_buf[b] = l.d_buf[b];
    enc->cur_lane = lane;
    return PVR_OK;
}

extern "C" {

pvr_status pvr_encoder_create(const pvr_encoder_desc *desc, pvr_encoder **out) {
    PVR_REQUIRE(desc && out, "pvr_encoder_create: null argument");
    PVR_REQUIRE(desc->arch >= PVR_ARCH_RESNET50 && desc->arch <= PVR_ARCH_RESNET34, "unknown arch %d", desc->arch);
    PVR_REQUIRE(desc->dtype == PVR_BF16 || desc->dtype == PVR_F16 || (desc->dtype == PVR_F32 && (desc->arch <= PVR_ARCH_RESNET50_L3 || desc->arch == PVR_ARCH_RESNET18 || desc->arch == PVR_ARCH_RESNET34)),
                "dtype must be PVR_BF16 or PVR_F16 (PVR_F32 is built for the ResNet50 family only)");
    PVR_REQUIRE(desc->max_batch > 0, "max_batch must be positive");
    PVR_REQUIRE(desc->crop == 224, "crop must be 224 (reference embeddings.py:82; CLIP input_resolution 224)");
    PVR_REQUIRE(desc->resize >= desc->crop, "resize must be >= crop");
    pvr_encoder *e = new pvr_encoder();
    e->desc = *desc;
    read_switches(e->sw);
    if (e->desc.chunk <= 0 || e->desc.chunk > e->desc.max_batch) e->desc.chunk = e->desc.max_batch;
    if (e->desc.arch == PVR_ARCH_RANDOM5) {
        random5_create(e);
    } else if (e->desc.arch == PVR_ARCH_RESNET18 || e->desc.arch == PVR_ARCH_RESNET34) {
        build_basic_resnet(e);
    } else if (e->desc.arch == PVR_ARCH_CLIP_RN50) {
        build_clip_rn50(e);
        resizer_create(e);
    } else if (e->desc.arch >= PVR_ARCH_CLIP_VIT_B32 && e->desc.arch <= PVR_ARCH_MAE_VIT_H14) {
        pvr_status s = vit_create(e);
        if (s) { delete e; return s; }
    } else {
        build_resnet50(e);
    }
    *out = e;
    return PVR_OK;
}

pvr_status pvr_encoder_load_weights(pvr_encoder *enc, const char *name, const float *host_data, const int64_t *shape,
                                    int32_t ndim) {
    PVR_REQUIRE(enc && name && host_data, "pvr_encoder_load_weights: null argument");
    PVR_REQUIRE(!enc->finalized, "encoder already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + n);
    enc->weights[name] = std::move(t);
    return PVR_OK;
}

pvr_status pvr_encoder_set_host_backend(pvr_encoder *enc, int32_t on) {
    PVR_REQUIRE(enc, "null encoder");
    PVR_REQUIRE(!enc->finalized, "pvr_encoder_set_host_backend: call between create and finalize");
    enc->host = on != 0;
    return PVR_OK;
}

pvr_status pvr_encoder_finalize(pvr_encoder *enc) {
    PVR_REQUIRE(enc, "null encoder");
    PVR_REQUIRE(!enc->finalized, "encoder already finalized");
    pvr_status s;
    if (enc->host) return host_finalize(enc);                   // CPU plan: no HIP call
    if (enc->vit || enc->rnd) {
        if ((s = enc->vit ? vit_finalize(enc) : random5_finalize(enc))) return s;
        PVR_HIP_TRY(hipDeviceSynchronize());
        enc->weights.clear();
        enc->finalized = true;
        return PVR_OK;
    }
    const bool rn50c = enc->desc.arch == PVR_ARCH_CLIP_RN50;
    if ((s = rn50c ? finalize_stem(enc, "visual.conv1.weight", "visual.bn1", 32, 3) : finalize_stem(enc))) return s;
    if (rn50c && (s = finalize_attnpool(enc))) return s;
    for (auto &op : enc->ops)
        if (op.kind == 0 && (s = finalize_conv(enc, op))) return s;
    if ((s = build_schedules(enc))) return s;
    PVR_HIP_TRY(hipDeviceSynchronize());                        // (the weight-packing launches above)
    for (auto &op : enc->ops) {
        op.h_w.clear(); op.h_w.shrink_to_fit(); op.h_b.clear(); op.h_b.shrink_to_fit();
        if (op.d_wsp && op.d_wf) { (void)hipFree(op.d_wf); op.d_wf = nullptr; }     // fp32 weights that were only the source of a split image
    }
    enc->fuse = enc->sw.fuse != 0;
    pvr_status ws = alloc_workspace(enc);
    if (ws) return ws;
    plan_splitk(enc);
    PVR_HIP_TRY(hipMalloc((void **)&enc->d_zero, PVR_ZERO_BYTES));      // zero page: padding rows, and the all-zero bias of split-K launches
    PVR_HIP_TRY(hipMemset(enc->d_zero, 0, PVR_ZERO_BYTES));
    save_lane(enc, 0);
    if ((s = ensure_smallk(enc))) return s;
    resolve_kinds(enc);
    // the memsets above run on the null stream; forwards run on the caller's stream (torch's current
    // stream need not be ordered against it), so drain the device once here, off the hot path
    PVR_HIP_TRY(hipDeviceSynchronize());
    enc->weights.clear();                                       // host copies no longer needed
    enc->finalized = true;
    return PVR_OK;
}

int32_t pvr_encoder_out_size(const pvr_encoder *enc) { return enc ? enc->out_size : 0; }

}  // extern "C"

// One chunk of the CLIP RN50 tower: Resize(224, bicubic, antialias) + CenterCrop -> stem image -> conv1 (stem kernel) -> plan
// (convolutions and 2x2 average pools) -> attention pool (tokens, fused q/k/v GEMM, attention core, c_proj of token 0).

static pvr_status clip_rn50_chunk(pvr_encoder *enc, const uint8_t *fr, int nb, int h, int w, float *out, int64_t out_stride, hipStream_t st) {
    const int dt = enc->desc.dtype, crop = enc->desc.crop;
    pvr_status s;
    const uint8_t *u8; int oh, ow;
    if ((s = resizer_run(enc, enc->cur_lane, fr, nb, h, w, st, &u8, &oh, &ow))) return s;
    // short side == crop here, so this only centre-crops and converts to the stem's centred 4-channel image
    if ((s = launch_preprocess(u8, nb, oh, ow, enc->desc.resize, crop, enc->d_img, dt, st))) return s;
    enc->last_n = nb;
    if (enc->stop_after == "pre") return PVR_OK;
    if ((s = launch_stem(enc->d_img, enc->d_stem_w, enc->d_stem_b, enc->d_stem, nb, crop, dt, st))) return s;
    for (const ConvOp &op : enc->ops) {
        if (op.kind == 1) s = launch_avgpool2(bufp(enc, op.in_buf), bufp(enc, op.out_buf), nb, op.h, op.w, op.cin, dt, st);
        else s = launch_conv(bufp(enc, op.in_buf), op.d_w, op.d_b, op.res_buf == B_NONE ? nullptr : bufp(enc, op.res_buf), bufp(enc, op.out_buf),
                             enc->d_zero, nb, op.h, op.w, op.cin, op.cout, op.k, op.k, op.stride, op.pad, op.relu, op.out_f32, dt, st);
        if (s) return s;
        if (!enc->stop_after.empty() && op.tap == enc->stop_after) return PVR_OK;
    }
    // AttentionPool2d(7, 2048, 32 heads, 1024)
    const int C = 2048, T = 50;
    if ((s = launch_attnpool_tokens((const float *)enc->d_buf[B_F32], enc->ap_pos, enc->d_buf[B_T1], nb, 49, C, dt, st))) return s;
    if ((s = launch_conv(enc->d_buf[B_T1], enc->ap_wqkv, enc->ap_bqkv, nullptr, enc->d_buf[B_X0], enc->d_zero, nb * T, 1, 1, C, 3 * C, 1, 1, 1, 0, 0, 0, dt, st))) return s;
    if ((s = launch_attention(enc->d_buf[B_X0], enc->d_buf[B_T2], T, C, 32, nb, dt, st))) return s;
    // c_proj of token 0 only: a 1x1 "convolution" over (n, T, 1) with stride T picks row 0 of every image; fp32 out
    float *dense = enc->ap_out + (size_t)enc->cur_lane * enc->desc.chunk * 1024;
    if ((s = launch_conv(enc->d_buf[B_T2], enc->ap_wc, enc->ap_bc, nullptr, dense, enc->d_zero, nb, T, 1, C, 1024, 1, 1, T, 0, 0, 1, dt, st))) return s;
    PVR_HIP_TRY(hipMemcpy2DAsync(out, (size_t)out_stride * 4, dense, 1024 * 4, 1024 * 4, nb, hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

// ev != nullptr: record one event before the first launch and one after every launch of the FIRST chunk
static pvr_status forward_impl(pvr_encoder *enc, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out,
                               int64_t out_stride, void *hip_stream, std::vector<hipEvent_t> *ev) {
    PVR_REQUIRE(enc && frames && out, "pvr_encoder_forward: null argument");
    if (!enc->finalized) { set_error("encoder not finalized"); return PVR_ERR_STATE; }
    PVR_REQUIRE(n > 0 && n <= enc->desc.max_batch, "n=%d outside 1..max_batch=%d", n, enc->desc.max_batch);
    PVR_REQUIRE(out_stride >= enc->out_size, "out_stride %lld < out_size %d", (long long)out_stride, enc->out_size);
    hipStream_t st = (hipStream_t)hip_stream;
    const int dt = enc->desc.dtype;
    pvr_status s;
    if (enc->vit) return vit_forward(enc, frames, n, h, w, out, out_stride, st);
    if (enc->rnd) return random5_forward(enc, frames, n, h, w, out, out_stride, st);
    for (int f0 = 0; f0 < n; f0 += enc->desc.chunk) {
        const int nb = (n - f0 < enc->desc.chunk) ? n - f0 : enc->desc.chunk;
        const uint8_t *fr = frames + (size_t)f0 * h * w * 3;
        auto mark = [&]() -> pvr_status {
            if (ev && f0 == 0) {
                hipEvent_t e;
                PVR_HIP_TRY(hipEventCreate(&e));
                // span mode (pvr_encoder_profile_span): only the two marks that bracket the span are recorded, the others are placeholders
                const int idx = (int)ev->size();
                if (enc->span_first < 0 || idx == enc->span_first || idx == enc->span_last) PVR_HIP_TRY(hipEventRecord(e, st));
                ev->push_back(e);
            }
            return PVR_OK;
        };
        if ((s = mark())) return s;
        if (dt == PVR_F32) {
            // reference-precision plan: integer transforms (exact, via the bf16 image) -> fp32 /255, Normalize ->
            // fp32 conv1 -> fp32 maxpool -> fp32 implicit-GEMM convs (f32 MFMA) -> fp32 pool / flatten
            const int crop = enc->desc.crop;
            if ((s = launch_preprocess(fr, nb, h, w, enc->desc.resize, crop, enc->d_img, PVR_BF16, st, enc->crop_pos))) return s;
            if ((s = launch_normalize_nhwc4(enc->d_img, enc->d_imgf, nb, crop, enc->desc.mean, enc->desc.std_, PVR_BF16, st))) return s;
            if ((s = mark())) return s;
            if ((s = launch_stem_f32(enc->d_imgf, enc->d_stem_wf, enc->d_stem_b, (float *)enc->d_stem, nb, crop, st))) return s;
            if ((s = mark())) return s;
            if ((s = launch_maxpool_f32((const float *)enc->d_stem, (float *)enc->d_buf[B_X0], nb, 112, 112, 64, st))) return s;
            if ((s = mark())) return s;
            enc->last_n = nb;
            for (auto &op : enc->ops) {
                const float *res = op.res_buf == B_NONE ? nullptr : (const float *)enc->d_buf[op.res_buf];
                if ((s = launch_conv_f32((const float *)enc->d_buf[op.in_buf], op.d_wf, op.d_b, res, (float *)enc->d_buf[op.out_buf], nb,
                                         op.h, op.w, op.cin, op.cout, op.k, op.stride, op.pad, op.relu, st)))
                    return s;
                if ((s = mark())) return s;
            }
            float *o32 = out + (size_t)f0 * out_stride;
            if (pooled_head(enc))
                s = launch_avgpool(enc->d_buf[B_F32], o32, out_stride, nb, enc->final_hw, enc->final_c, 1, dt, st);
            else
                s = launch_nhwc_to_chw((const float *)enc->d_buf[B_F32], o32, out_stride, nb, enc->final_hw, enc->final_c,
                                       enc->final_creal, st);
            if (s) return s;
            if ((s = mark())) return s;
            continue;
        }
        if (enc->desc.arch == PVR_ARCH_CLIP_RN50) {
            if ((s = clip_rn50_chunk(enc, fr, nb, h, w, out + (size_t)f0 * out_stride, out_stride, st))) return s;
            continue;
        }
        // frames that need no resize (the bench configuration: 256 x 256 frames, Resize(256) is the identity): the fused stem reads the
        // uint8 frames itself - no preprocess launch, no padded 16-bit image in HBM
        int fused_u8 = 0;
        // layer1.0.conv1 has no launch in the fused schedule (build_schedules): the stem's register-pooling form runs it
        const bool c1_pending = enc->fuse && enc->stem_c1 >= 0;
        const bool c1_in_stem = c1_pending && enc->stop_after.empty() && stem_conv1_capable();
        if (enc->sw.stem_u8 && enc->sw.stem_lds && enc->stop_after.empty() && enc->desc.crop == 224 && enc->crop_pos >= 0 && enc->crop_pos <= 4) {
            int rn = 1, top = 0, left = 0;
            preprocess_geometry(h, w, enc->desc.resize, enc->desc.crop, enc->crop_pos, &rn, &top, &left);
            if (!rn && stem_pool_u8_ok(fr, h, w, top, left)) {
                if ((s = mark())) return s;                  // (launch index of the preprocess stays: pvr_encoder_profile)
                if ((s = launch_stem_pool_u8(fr, nb, h, w, top, left, enc->d_stem_w, enc->d_stem_b, enc->d_buf[B_X0], dt, st, c1_in_stem ? enc->d_stem_c1w : nullptr,
                                             c1_in_stem ? enc->ops[enc->stem_c1].d_b : nullptr, c1_in_stem ? enc->d_buf[B_T1] : nullptr, enc->stem_c1_blk))) return s;
                fused_u8 = 1;
            }
        }
        if (!fused_u8) {
        if ((s = launch_preprocess(fr, nb, h, w, enc->desc.resize, enc->desc.crop, enc->d_img, dt, st, enc->crop_pos))) return s;
        if ((s = mark())) return s;
        }
        enc->last_n = nb;
        if (enc->stop_after == "pre") return PVR_OK;
        if (enc->stop_after == "stem") {             // debug tap of the un-pooled conv1 output: unfused kernel
            if ((s = launch_stem(enc->d_img, enc->d_stem_w, enc->d_stem_b, enc->d_stem, nb, enc->desc.crop, dt, st))) return s;
            return PVR_OK;
        }
        // conv1 + bn1 + relu + maxpool fused: the 112x112x64 activation stays in LDS
        if (!fused_u8 && (s = launch_stem_pool(enc->d_img, enc->d_stem_w, enc->d_stem_b, enc->d_buf[B_X0], nb, enc->desc.crop, dt, st, c1_in_stem ? enc->d_stem_c1w : nullptr,
                                               c1_in_stem ? enc->ops[enc->stem_c1].d_b : nullptr, c1_in_stem ? enc->d_buf[B_T1] : nullptr, enc->stem_c1_blk))) return s;
        if (enc->range_flags) {                      // pvr_encoder_check_range: the pooled stem output (flag slot behind the plan's launches)
            const size_t n8 = (size_t)nb * 56 * 56 * 64 / 8;
            hipLaunchKernelGGL(range_flag_kernel<false>, dim3(2048), dim3(256), 0, st, enc->d_buf[B_X0], n8, dt, enc->range_flags, (int)cur_plan(enc).size());
        }
        if ((s = mark())) return s;
        if ((s = mark())) return s;                  // (keeps the launch indices of pvr_encoder_profile stable)
        if (enc->stop_after == "pool") return PVR_OK;
        bool stopped = false, t1_blocked = false;   // t1_blocked: the conv1 launch in front of layer1's first tail wrote t1 in the blocked layout
        if (c1_in_stem) t1_blocked = enc->stem_c1_blk != 0;
        else if (c1_pending && enc->stop_after != "pool") {
            // ... or, where that stem form did not run (debug stops, PVR_STEM_REGPOOL=0), as its own launch in front of the plan - blocked t1 when the tail wants it
            const ConvOp &c1 = enc->ops[enc->stem_c1];
            if (enc->stem_c1_blk && conv_algo() == -1 && conv_expand_supported((int64_t)nb * c1.h * c1.w, c1.h, c1.w, c1.cin, c1.cout, 1, 1, 1, 0, c1.relu, 0, false)) {
                s = launch_conv_expand(enc->d_buf[c1.in_buf], c1.d_w, c1.d_b, nullptr, enc->d_buf[c1.out_buf], nb, c1.h, c1.w, c1.cin, c1.cout, 1, c1.relu, dt, st, 1);
                t1_blocked = true;
            } else
                s = launch_conv(enc->d_buf[c1.in_buf], c1.d_w, c1.d_b, nullptr, enc->d_buf[c1.out_buf], enc->d_zero, nb, c1.h, c1.w, c1.cin, c1.cout, 1, 1, 1, 0, c1.relu, 0, dt, st);
            if (s) return s;
        }
        bool pooled = false;                         // the plan's last convolution wrote the average pool itself (conv_wfrag's pooled form)
        const std::vector<Launch> &plan_ = cur_plan(enc);
        if (enc->kinds_algo != conv_algo() || enc->kinds_stride != plan_.size()) resolve_kinds(enc);   // (pvr_debug_set_conv_algo is process-wide; same size: no allocation)
        const uint8_t *kinds = enc->kinds.data() + (size_t)(nb - 1) * plan_.size();
        float *smallk = enc->d_smallk[enc->cur_lane];
        // the pooled epilogue stores 16-byte pieces of the caller's rows: a property of this call's arguments, not of the plan
        const bool pool_args_ok = enc->stop_after.empty() && out_stride % 4 == 0 && (((size_t)(out + (size_t)f0 * out_stride)) & 15) == 0;
        int launch_idx = 0;                          // debug: stop_after = "#k" ends the forward after conv launch k of the plan
        const bool run_ok = enc->stop_after.empty() && !enc->range_flags;
        const int stop_idx = enc->stop_after.size() > 1 && enc->stop_after[0] == '#' ? atoi(enc->stop_after.c_str() + 1) : -1;
        // a member convolution of a launch that runs as its members (small forwards): split-K in the low-latency plan, else the shape's kernel
        auto member = [&](const ConvOp &o, const void *in, const void *r_, void *out_) -> pvr_status {
            if (const int ks = small_batch_ksplit(enc, o, nb)) {
                if (!smallk) { set_error("low-latency plan without its scratch (pvr_encoder_set_low_latency allocates it)"); return PVR_ERR_STATE; }
                return launch_conv_splitk(in, o.d_w, o.d_b, r_, out_, enc->d_zero, smallk, ks, nb, o.h, o.w, o.cin, o.cout, o.k, o.k, o.stride, o.pad, o.relu, o.out_f32, dt, st);
            }
            return launch_conv(in, o.d_w, o.d_b, r_, out_, enc->d_zero, nb, o.h, o.w, o.cin, o.cout, o.k, o.k, o.stride, o.pad, o.relu, o.out_f32, dt, st);
        };
        for (size_t li = 0; li < plan_.size(); ++li) {
            const Launch &l = plan_[li];
            const ConvOp &op = enc->ops[l.conv3 >= 0 ? l.conv3 : l.conv2];
            const void *res = op.res_buf == B_NONE ? nullptr : enc->d_buf[op.res_buf];
            int kind = kinds[li];
            if (kind == LK_WFRAG_POOL && !pool_args_ok) kind = resolve_kind(enc, plan_, li, nb, false);
            if ((kind == LK_FRAME_RUN || kind == LK_FRAME_RUN_TAIL) && !run_ok) kind = LK_FRAME_FRONT1;     // (taps, debug stops, range validation: one launch per bottleneck)
            switch (kind) {
            case LK_FRAME_RUN: {
                BFBlk blks[6];
                int nblk = 0;
                for (size_t t = li; t < plan_.size() && nblk < 6 && (t == li || kinds[t] == LK_FRAME_RUN_TAIL); ++t, ++nblk) {
                    const Launch &lt = plan_[t];
                    const ConvOp &o3 = enc->ops[lt.conv3], &o2 = enc->ops[lt.conv2], &o1 = enc->ops[lt.conv1];
                    blks[nblk] = BFBlk{(const u16 *)o1.d_wfb, (const u16 *)o2.d_wfb, (const u16 *)o3.d_wfb, (const u16 *)enc->d_buf[o3.res_buf], o1.d_b, o2.d_b, o3.d_b,
                                       (u16 *)enc->d_buf[o3.out_buf]};
                }
                s = launch_bneck_frame_run(blks, nblk, nb, dt, st, enc->sw.frame_stagger);
                break;
            }
            case LK_FRAME_RUN_TAIL:
                s = PVR_OK;                                   // (inside the run's launch)
                break;
            case LK_FRAME_FRONT1: {
                const ConvOp &c2 = enc->ops[l.conv2], &cf = enc->ops[l.conv1];
                s = launch_bneck_frame(nullptr, c2.d_wfb, c2.d_b, op.d_wfb, op.d_b, res, enc->d_buf[op.out_buf], nullptr, nb, 3 | 8, dt, st,
                                       nullptr, nullptr, nullptr, nullptr, cf.d_wfb, cf.d_b);
                break;
            }
            case LK_FRAME: {
                const ConvOp &c2 = enc->ops[l.conv2];
                const ConvOp *c1 = l.next1 >= 0 ? &enc->ops[l.next1] : nullptr;
                s = launch_bneck_frame(enc->d_buf[l.t1_in], c2.d_wfb, c2.d_b, op.d_wfb, op.d_b, res, enc->d_buf[op.out_buf], nullptr, nb, c1 ? 7 : 3, dt, st,
                                       nullptr, c1 ? c1->d_wfb : nullptr, c1 ? c1->d_b : nullptr, c1 ? enc->d_buf[l.t1_out] : nullptr);
                break;
            }
            case LK_FRAME_MEMBERS: {
                // t2 goes to the t1 buffer this launch does not read
                const ConvOp &c2 = enc->ops[l.conv2];
                const ConvOp *c1 = l.next1 >= 0 ? &enc->ops[l.next1] : nullptr;
                const ConvOp *cf = l.conv1 >= 0 ? &enc->ops[l.conv1] : nullptr;
                const int t2b = l.t1_in == B_T1 ? B_T2 : B_T1;
                s = PVR_OK;
                if (cf) s = member(*cf, enc->d_buf[cf->in_buf], nullptr, enc->d_buf[l.t1_in]);
                if (!s) s = member(c2, enc->d_buf[l.t1_in], nullptr, enc->d_buf[t2b]);
                if (!s) s = member(op, enc->d_buf[t2b], res, enc->d_buf[op.out_buf]);
                if (!s && c1) s = member(*c1, enc->d_buf[op.out_buf], nullptr, enc->d_buf[l.t1_out]);
                break;
            }
            case LK_DUAL: {
                // conv3 & downsample as one two-operand launch (layer3.0 / layer4.0)
                const ConvOp &cd = enc->ops[l.ds];
                s = launch_conv_pp256(enc->d_buf[op.in_buf], op.d_wcat, op.d_bsum, nullptr, enc->d_buf[op.out_buf], nb, op.h, op.w, op.cin, op.cout, 1, 1, 1, 0,
                                      op.relu, 0, 0, dt, 224, st, enc->d_buf[cd.in_buf], cd.h, cd.w, cd.cin, cd.stride);
                break;
            }
            case LK_DUAL_MEMBERS: {
                const ConvOp &cd = enc->ops[l.ds];
                s = member(cd, enc->d_buf[cd.in_buf], nullptr, enc->d_buf[cd.out_buf]);
                if (!s) s = member(op, enc->d_buf[op.in_buf], res, enc->d_buf[op.out_buf]);
                break;
            }
            case LK_CHAIN: {
                const ConvOp &c2 = enc->ops[l.conv2];
                const ConvOp *c1 = l.next1 >= 0 ? &enc->ops[l.next1] : nullptr;
                const ConvOp *cd = l.ds >= 0 ? &enc->ops[l.ds] : nullptr;
                s = launch_bottleneck_chain(enc->d_buf[l.t1_in], c2.d_w, c2.d_b, op.d_wp, cd ? op.d_bsum : op.d_b, res, enc->d_buf[op.out_buf],
                                            c1 ? c1->d_wp : nullptr, c1 ? c1->d_b : nullptr, c1 ? enc->d_buf[l.t1_out] : nullptr, nb,
                                            c2.h, c2.w, c2.cout, c1 ? c1->cout : 0, c2.stride, dt, st,
                                            cd ? enc->d_buf[cd->in_buf] : nullptr, cd ? cd->d_wp : nullptr, op.d_wpb, cd ? cd->d_wpb : nullptr,
                                            l.wave, cd ? (l.in_blk && t1_blocked) : l.in_blk, l.out_blk, c2.d_wpk);
                t1_blocked = false;
                break;
            }
            case LK_CAST:
                s = launch_f32_to_h((const float *)enc->d_buf[op.in_buf], enc->d_buf[op.out_buf], (size_t)nb * op.h * op.w * op.cin, dt, st);
                break;
            case LK_F32:
                s = launch_conv_f32((const float *)enc->d_buf[op.in_buf], op.d_wf, op.d_b, (const float *)res, (float *)enc->d_buf[op.out_buf], nb,
                                    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.pad, op.relu, st);
                break;
            case LK_SPLIT16:
                s = launch_conv_split16((const float *)enc->d_buf[op.in_buf], op.d_wsp, op.d_b, (const float *)res, (float *)enc->d_buf[op.out_buf], nb,
                                        op.h, op.w, op.cin, op.cout, op.k, op.stride, op.pad, op.relu, st);
                break;
            case LK_SPLIT16_PAIR: {
                const ConvOp &od = enc->ops[l.pair];
                s = launch_conv_split16((const float *)enc->d_buf[op.in_buf], op.d_wsp_pair, op.d_b_pair, nullptr, (float *)enc->d_buf[op.out_buf], nb, op.h, op.w, op.cin,
                                        128, op.k, op.stride, op.pad, 1, st, (float *)enc->d_buf[od.out_buf], 64);
                break;
            }
            case LK_SPLIT16_IN32:
                // a 16-bit convolution whose operand is the fp32 residual stream (rounded to f16 in the kernel's staging pass): t1 leaves 16-bit, a downsample fp32
                s = launch_conv_split16((const float *)enc->d_buf[op.in_buf], op.d_wsp, op.d_b, nullptr, (op.out_f32 & 1) ? (float *)enc->d_buf[op.out_buf] : nullptr, nb,
                                        op.h, op.w, op.cin, op.cout, op.k, op.stride, op.pad, op.relu, st, nullptr, 0, (op.out_f32 & 1) ? nullptr : enc->d_buf[op.out_buf], 1);
                break;
            case LK_SPLITK_SMALL:
                // low-latency plan: the few pixel tiles of a <= 4-frame forward share each K loop between `ks` blocks
                s = member(op, enc->d_buf[op.in_buf], res, enc->d_buf[op.out_buf]);
                break;
            case LK_SPLITK:
                s = launch_conv_splitk(enc->d_buf[op.in_buf], op.d_w, op.d_b, res, enc->d_buf[op.out_buf], enc->d_zero, (float *)enc->d_buf[op.ks_buf],
                                       op.ksplit, nb, op.h, op.w, op.cin, op.cout, op.k, op.k, op.stride, op.pad, op.relu, op.out_f32, dt, st);
                break;
            case LK_EXPAND_BLOCKED:
                // layer1.0.conv1 in front of a wave-form tail: t1 in the blocked layout
                s = launch_conv_expand(enc->d_buf[op.in_buf], op.d_w, op.d_b, nullptr, enc->d_buf[op.out_buf], nb, op.h, op.w, op.cin, op.cout, 1, op.relu, dt, st, 1);
                t1_blocked = true;
                break;
            case LK_WFRAG_POOL:
                // the trunk's last conv3 + identity + ReLU with AdaptiveAvgPool2d(1) in its epilogue: the (n,7,7,2048) fp32 activation is never written
                s = launch_conv_wfrag(enc->d_buf[op.in_buf], op.d_wfb, op.d_b, res, nullptr, nb, op.h, op.w, op.cin, op.cout, 1, 1, 1, 0, 1, 1, dt, st,
                                      out + (size_t)f0 * out_stride, out_stride);
                pooled = true;
                break;
            case LK_WFRAG:
                // few pixels, deep K (layer4 at batch 256): 112 x 256 tiles, weights as L2 fragments
                s = launch_conv_wfrag(enc->d_buf[op.in_buf], op.d_wfb, op.d_b, res, enc->d_buf[op.out_buf], nb, op.h, op.w, op.cin, op.cout, op.k, op.k,
                                      op.stride, op.pad, op.relu, op.out_f32, dt, st);
                break;
            default:
                s = launch_conv(enc->d_buf[op.in_buf], op.d_w, op.d_b, res, enc->d_buf[op.out_buf], enc->d_zero, nb, op.h, op.w,
                                op.cin, op.cout, op.k, op.k, op.stride, op.pad, op.relu, op.out_f32, dt, st);
            }
            if (s) return s;
            if (enc->range_flags && kind != LK_WFRAG_POOL) {         // pvr_encoder_check_range: the launch's output, all of it
                const int ho_ = (op.h + 2 * op.pad - op.k) / op.stride + 1, wo_ = (op.w + 2 * op.pad - op.k) / op.stride + 1;
                const size_t n8 = (size_t)nb * ho_ * wo_ * op.cout / 8;
                const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
                if ((op.out_f32 & 1) || op.f32op) hipLaunchKernelGGL(range_flag_kernel<true>, dim3(blocks), dim3(256), 0, st, enc->d_buf[op.out_buf], n8, dt, enc->range_flags, (int)li);
                else hipLaunchKernelGGL(range_flag_kernel<false>, dim3(blocks), dim3(256), 0, st, enc->d_buf[op.out_buf], n8, dt, enc->range_flags, (int)li);
            }
            if ((s = mark())) return s;
            if (!enc->stop_after.empty() && op.tap == enc->stop_after) { stopped = true; break; }
            if (launch_idx++ == stop_idx) { stopped = true; break; }
        }
        enc->last_pooled = pooled;
        if (stopped) return PVR_OK;
        float *o = out + (size_t)f0 * out_stride;
        if (pooled)
            s = PVR_OK;                               // (the last launch wrote the pooled rows)
        else if (pooled_head(enc))
            s = launch_avgpool(enc->d_buf[B_F32], o, out_stride, nb, enc->final_hw, enc->final_c, 1, dt, st);
        else
            s = launch_nhwc_to_chw((const float *)enc->d_buf[B_F32], o, out_stride, nb, enc->final_hw, enc->final_c,
                                   enc->final_creal, st);
        if (s) return s;
        if ((s = mark())) return s;
    }
    return PVR_OK;
}

// Same-lane forwards issued on DIFFERENT streams are ordered here, not by the caller: every forward records lane_done[lane] on its
// stream and the next forward on that lane waits for it first (a device-side event wait, no host synchronisation), so a lane's
// workspace is never shared by two forwards in flight.  Different lanes stay independent.
static pvr_status lane_wait(pvr_encoder *enc, int lane, void *hip_stream) {
    hipStream_t st = (hipStream_t)hip_stream;
    if (enc->lane_done[lane] && enc->lane_stream[lane] != st) PVR_HIP_TRY(hipStreamWaitEvent(st, enc->lane_done[lane], 0));
    return PVR_OK;
}
static pvr_status lane_mark(pvr_encoder *enc, int lane, void *hip_stream) {
    hipStream_t st = (hipStream_t)hip_stream;
    if (!enc->lane_done[lane]) PVR_HIP_TRY(hipEventCreateWithFlags(&enc->lane_done[lane], hipEventDisableTiming));
    PVR_HIP_TRY(hipEventRecord(enc->lane_done[lane], st));
    enc->lane_stream[lane] = st;
    return PVR_OK;
}
static pvr_status lane_forward(pvr_encoder *enc, int lane, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out,
                               int64_t out_stride, void *hip_stream) {
    pvr_status s = lane_wait(enc, lane, hip_stream);
    if (!s) s = forward_impl(enc, frames, n, h, w, out, out_stride, hip_stream, nullptr);
    if (!s) s = lane_mark(enc, lane, hip_stream);
    return s;
}

extern "C" {

#define PVR_NO_HOST(enc_, what_) PVR_REQUIRE(!((enc_) && (enc_)->host), what_ ": not available on a host-backend encoder")

pvr_status pvr_encoder_forward(pvr_encoder *enc, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out,
                               int64_t out_stride, void *hip_stream) {
    if (enc && enc->host) {
        PVR_REQUIRE(enc->finalized, "encoder not finalized");
        PVR_REQUIRE(n <= enc->desc.max_batch, "n=%d exceeds max_batch=%d", n, enc->desc.max_batch);
        return host_forward(enc, frames, n, h, w, out, out_stride);
    }
    PVR_REQUIRE(enc, "pvr_encoder_forward: null encoder");
    TraceScope trace("pvr_encoder_forward");
    if (enc->finalized && !enc->vit && !enc->rnd) { pvr_status s = use_lane(enc, 0); if (s) return s; }
    if (enc->finalized && enc->vit) { pvr_status s = vit_use_lane(enc, 0); if (s) return s; }
    if (!enc->finalized) { set_error("encoder not finalized"); return PVR_ERR_STATE; }
    return lane_forward(enc, 0, frames, n, h, w, out, out_stride, hip_stream);
}

pvr_status pvr_encoder_forward_lane(pvr_encoder *enc, int32_t lane, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out,
                                    int64_t out_stride, void *hip_stream) {
    if (enc && enc->host) return pvr_encoder_forward(enc, frames, n, h, w, out, out_stride, hip_stream);
    PVR_REQUIRE(enc, "pvr_encoder_forward_lane: null encoder");
    TraceScope trace(lane == 0 ? "pvr_encoder_forward_lane 0" : "pvr_encoder_forward_lane 1+");
    PVR_REQUIRE(lane >= 0 && lane < PVR_MAX_LANES, "pvr_encoder_forward_lane: lane must be 0..%d", PVR_MAX_LANES - 1);
    if (!enc->finalized) { set_error("encoder not finalized"); return PVR_ERR_STATE; }
    if (enc->vit) { pvr_status s = vit_use_lane(enc, lane); if (s) return s; }
    else if (!enc->rnd) { pvr_status s = use_lane(enc, lane); if (s) return s; }            // (the 'random' plan has one workspace)
    else PVR_REQUIRE(lane == 0, "pvr_encoder_forward_lane: the 'random' PVR plan has a single workspace");
    return lane_forward(enc, lane, frames, n, h, w, out, out_stride, hip_stream);
}

// Instrumented forward of ONE chunk: HIP events between launches on the caller's stream (synchronises).
// op_ms[i] = duration of launch i, op_flops[i] = its algorithmic FLOPs (2*M*K_real*Cout_real; 0 for byte kernels),
// launch order: preprocess, stem, maxpool, conv ops..., pool/flatten.
pvr_status pvr_encoder_profile(pvr_encoder *enc, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out,
                               int64_t out_stride, void *hip_stream, float *op_ms, double *op_flops, int32_t cap,
                               int32_t *n_ops) {
    PVR_REQUIRE(enc && op_ms && op_flops && n_ops, "pvr_encoder_profile: null argument");
    PVR_NO_HOST(enc, "pvr_encoder_profile");
    PVR_REQUIRE(n <= enc->desc.chunk, "profile: n=%d must fit one chunk (%d)", n, enc->desc.chunk);
    std::vector<hipEvent_t> ev;
    pvr_status s = (enc->finalized && !enc->vit && !enc->rnd) ? use_lane(enc, 0) : PVR_OK;
    if (!s && enc->finalized && enc->vit) s = vit_use_lane(enc, 0);
    if (!s) s = lane_wait(enc, 0, hip_stream);
    if (!s) s = forward_impl(enc, frames, n, h, w, out, out_stride, hip_stream, &ev);
    if (!s && hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) { set_error("profile: sync failed"); s = PVR_ERR_HIP; }
    const int nl = (int)ev.size() - 1;
    if (!s && nl > cap) { set_error("profile: %d launches > cap %d", nl, cap); s = PVR_ERR_INVALID; }
    if (!s) {
        for (int i = 0; i < nl; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            op_ms[i] = ms;
            op_flops[i] = 0.0;
        }
        // stem: 118.0 MMAC/frame = 112*112*64*147
        if (nl > 1) op_flops[1] = 2.0 * n * 112.0 * 112.0 * 64.0 * 147.0;
        if (nl > 1 && enc->fuse && enc->stem_c1 >= 0 && enc->desc.dtype != PVR_F32) op_flops[1] += 2.0 * n * 56.0 * 56.0 * 64.0 * 64.0;   // layer1.0.conv1 runs inside the stem
        int i = 3;
        auto flops = [&](int oi) {
            if (oi < 0) return 0.0;
            const ConvOp &op = enc->ops[oi];
            const double ho = (op.h + 2 * op.pad - op.k) / op.stride + 1;
            return 2.0 * n * ho * ho * (double)op.cout_real * op.k * op.k * op.cin_real;
        };
        const bool fused = enc->fuse && enc->desc.dtype != PVR_F32;
        for (const Launch &l : (fused ? enc->sched_fused : enc->sched_plain)) {
            if (i >= nl) break;
            op_flops[i++] = flops(l.conv1) + flops(l.conv2) + flops(l.conv3) + flops(l.next1) + flops(l.ds) + flops(l.pair);
        }
        *n_ops = nl;
    }
    for (auto e : ev) (void)hipEventDestroy(e);
    return s;
}

// Time from the start of launch first_op to the end of launch last_op (indices as pvr_encoder_profile reports them) of ONE forward that
// carries only those two events: an event between two launches costs a few microseconds of dispatch serialisation, so the sum of
// pvr_encoder_profile's per-launch durations overstates a family of 38 launches by ~4 % against rocprofv3's kernel durations.
pvr_status pvr_encoder_profile_span(pvr_encoder *enc, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out, int64_t out_stride,
                                    void *hip_stream, int32_t first_op, int32_t last_op, float *span_ms) {
    PVR_REQUIRE(enc && span_ms, "pvr_encoder_profile_span: null argument");
    PVR_NO_HOST(enc, "pvr_encoder_profile_span");
    PVR_REQUIRE(n <= enc->desc.chunk, "profile: n=%d must fit one chunk (%d)", n, enc->desc.chunk);
    PVR_REQUIRE(first_op >= 0 && last_op >= first_op, "pvr_encoder_profile_span: bad launch range %d..%d", first_op, last_op);
    std::vector<hipEvent_t> ev;
    pvr_status s = (enc->finalized && !enc->vit && !enc->rnd) ? use_lane(enc, 0) : PVR_OK;
    if (!s && enc->finalized && enc->vit) s = vit_use_lane(enc, 0);
    if (!s) s = lane_wait(enc, 0, hip_stream);
    enc->span_first = first_op; enc->span_last = last_op + 1;
    if (!s) s = forward_impl(enc, frames, n, h, w, out, out_stride, hip_stream, &ev);
    enc->span_first = enc->span_last = -1;
    if (!s && hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) { set_error("profile: sync failed"); s = PVR_ERR_HIP; }
    if (!s && last_op + 1 >= (int)ev.size()) { set_error("profile_span: launch %d past the plan's %d launches", last_op, (int)ev.size() - 1); s = PVR_ERR_INVALID; }
    if (!s && hipEventElapsedTime(span_ms, ev[first_op], ev[last_op + 1]) != hipSuccess) { set_error("profile_span: elapsed time failed"); s = PVR_ERR_HIP; }
    for (auto e : ev) (void)hipEventDestroy(e);
    return s;
}

pvr_status pvr_encoder_set_crop_position(pvr_encoder *enc, int32_t pos) {
    PVR_REQUIRE(enc, "null encoder");
    PVR_REQUIRE(pos >= 0 && pos <= 4, "crop position %d outside 0..4", pos);
    PVR_REQUIRE(pos == 0 || (!enc->vit && !enc->rnd && enc->desc.arch != PVR_ARCH_CLIP_RN50), "corner crops are built for the ResNet50 family only");
    enc->crop_pos = pos;
    return PVR_OK;
}

pvr_status pvr_encoder_set_low_latency(pvr_encoder *enc, int32_t on) {
    PVR_REQUIRE(enc, "null encoder");
    enc->low_latency = on != 0;
    if (!enc->finalized || enc->vit || enc->rnd || enc->host) return PVR_OK;      // (finalize allocates / resolves when the plan was asked for earlier)
    pvr_status s = ensure_smallk(enc);                          // the plan's split-K scratch, here and not in a forward
    if (s) return s;
    resolve_kinds(enc);
    return PVR_OK;
}

pvr_status pvr_encoder_debug_set_fusion(pvr_encoder *enc, int32_t on) {
    PVR_REQUIRE(enc, "null encoder");
    enc->fuse = on != 0;
    if (enc->finalized && !enc->vit && !enc->rnd && !enc->host) resolve_kinds(enc);
    return PVR_OK;
}

// The switches of a finalized encoder that do not shape its plan (PlanSwitches, "live"): pool_fuse, stem_u8, frame_min_n.  The A/B tests flip them
// between two forwards of ONE handle; everything else is fixed by the environment at pvr_encoder_create.
pvr_status pvr_encoder_debug_set_switch(pvr_encoder *enc, const char *name, int32_t value) {
    PVR_REQUIRE(enc && name, "pvr_encoder_debug_set_switch: null argument");
    const std::string nm = name;
    if (nm == "pool_fuse") enc->sw.pool_fuse = value;
    else if (nm == "stem_u8") enc->sw.stem_u8 = value;
    else if (nm == "frame_min_n") enc->sw.frame_min_n = value;
    else if (nm == "frame_run") enc->sw.frame_run = value;
    else if (nm == "frame_stagger") enc->sw.frame_stagger = value;
    else { set_error("pvr_encoder_debug_set_switch: '%s' is not a live switch (pool_fuse, stem_u8, frame_min_n, frame_run, frame_stagger); plan switches are read from the environment at create", name); return PVR_ERR_INVALID; }
    if (enc->finalized && !enc->vit && !enc->rnd && !enc->host) resolve_kinds(enc);
    return PVR_OK;
}

// Range validation of the 16-bit storage types (f16: 5 exponent bits).  A non-finite EMBEDDING is caught by the callers' finite check, but an activation that
// overflows INSIDE the network need not reach the output: +inf times a negative weight is -inf, -inf or NaN through fmaxf(v, 0) is 0 - a wrong, finite
// embedding.  This runs ONE forward of the unfused plan (every convolution's output exists in HBM and is bit-identical to what the fused launches compute
// internally) and checks every launch's output for inf / NaN; *first_bad = index of the first such launch (pvr_encoder_launch_name's numbering of the
// UNFUSED plan, 3 = the first convolution) or -1.  Synchronises; allocates and frees its flag buffer: a load-time check, not part of the forward path.
pvr_status pvr_encoder_check_range(pvr_encoder *enc, const uint8_t *frames, int32_t n, int32_t h, int32_t w, float *out, int64_t out_stride, void *hip_stream,
                                   int32_t *first_bad) {
    PVR_REQUIRE(enc && frames && out && first_bad, "pvr_encoder_check_range: null argument");
    PVR_NO_HOST(enc, "pvr_encoder_check_range");
    PVR_REQUIRE(enc->finalized && !enc->vit && !enc->rnd && enc->desc.arch != PVR_ARCH_CLIP_RN50 && enc->desc.dtype != PVR_F32,
                "pvr_encoder_check_range: built for the 16-bit plans of the torchvision ResNet family");
    PVR_REQUIRE(n > 0 && n <= enc->desc.chunk, "pvr_encoder_check_range: n=%d must fit one chunk (%d)", n, enc->desc.chunk);
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t nl = enc->sched_plain.size();
    pvr_status s = use_lane(enc, 0);
    if (!s) s = lane_wait(enc, 0, hip_stream);
    if (s) return s;
    int *flags = nullptr;
    PVR_HIP_TRY(hipMalloc((void **)&flags, (nl + 1) * sizeof(int)));
    if (hipMemsetAsync(flags, 0, (nl + 1) * sizeof(int), st) != hipSuccess) { (void)hipFree(flags); set_error("check_range: memset failed"); return PVR_ERR_HIP; }
    const bool fuse0 = enc->fuse;
    enc->fuse = false; resolve_kinds(enc);
    enc->range_flags = flags;
    s = forward_impl(enc, frames, n, h, w, out, out_stride, hip_stream, nullptr);
    enc->range_flags = nullptr;
    enc->fuse = fuse0; resolve_kinds(enc);
    std::vector<int> hf(nl + 1, 0);
    if (!s && hipMemcpyAsync(hf.data(), flags, (nl + 1) * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("check_range: copy failed"); s = PVR_ERR_HIP; }
    if (!s && hipStreamSynchronize(st) != hipSuccess) { set_error("check_range: sync failed"); s = PVR_ERR_HIP; }
    (void)hipFree(flags);
    if (s) return s;
    *first_bad = hf[nl] ? 1 : -1;                               // 1 = "stem" (conv1 + bn1 + relu + maxpool)
    for (size_t i = 0; i < nl && *first_bad < 0; ++i) if (hf[i]) *first_bad = (int32_t)i + 3;
    return lane_mark(enc, 0, hip_stream);
}

// name of the kernel family launch `index` (the order pvr_encoder_profile reports) runs as in a forward of n frames; returns its length, 0 past the end
int32_t pvr_encoder_launch_kernel(const pvr_encoder *enc, int32_t n, int32_t index, char *buf, int32_t cap) {
    if (!enc || !buf || cap <= 0 || index < 3 || !enc->finalized || enc->vit || enc->rnd || enc->host || n < 1) return 0;
    const std::vector<Launch> &plan = cur_plan(enc);
    const int i = index - 3;
    if (i >= (int)plan.size()) return 0;
    const int nb = n < enc->desc.chunk ? n : enc->desc.chunk;
    // the forward's own table when it is current (it knows the runs: several plan entries in one launch), else the per-entry rule
    const bool tab = enc->kinds_stride == plan.size() && enc->kinds.size() == (size_t)enc->desc.chunk * plan.size() && enc->kinds_algo == conv_algo();
    const int kind = tab ? enc->kinds[(size_t)(nb - 1) * plan.size() + i] : resolve_kind(enc, plan, (size_t)i, nb);
    const char *nm = enc->desc.dtype == PVR_F32 ? "conv_f32" : launch_kind_name(kind);
    if (enc->desc.dtype != PVR_F32 && kind == LK_CHAIN) nm = plan[i].wave == 2 ? "chain_wave128" : plan[i].wave == 1 ? "chain_wave" : "bottleneck_chain";
    snprintf(buf, (size_t)cap, "%s", nm);
    return (int32_t)strlen(nm);
}

// name of launch `index` of the current plan (the order pvr_encoder_profile reports); returns the name's length, 0 past the end
int32_t pvr_encoder_launch_name(const pvr_encoder *enc, int32_t index, char *buf, int32_t cap) {
    if (!enc || !buf || cap <= 0 || index < 0) return 0;
    std::string nm;
    static const char *head[3] = {"preprocess", "stem", "maxpool"};
    if (index < 3) nm = head[index];
    else {
        const bool fused = enc->fuse && enc->desc.dtype != PVR_F32;
        const std::vector<Launch> &sc = fused ? enc->sched_fused : enc->sched_plain;
        const int i = index - 3;
        if (i < (int)sc.size()) {
            nm = enc->ops[sc[i].conv1 >= 0 ? sc[i].conv1 : sc[i].conv2].conv;
            if (sc[i].conv1 >= 0) nm += "+conv2";
            if (sc[i].conv3 >= 0) nm += "+" + enc->ops[sc[i].conv3].conv.substr(enc->ops[sc[i].conv3].conv.rfind('.') + 1);
            if (sc[i].ds >= 0 || sc[i].pair >= 0) nm += "&downsample";
            if (sc[i].next1 >= 0) nm += "+" + enc->ops[sc[i].next1].conv;
        } else if (i == (int)sc.size() && !enc->vit && !enc->rnd) nm = "pool/flatten";
    }
    snprintf(buf, (size_t)cap, "%s", nm.c_str());
    return (int32_t)nm.size();
}

pvr_status pvr_encoder_debug_stop_after(pvr_encoder *enc, const char *tap) {
    PVR_REQUIRE(enc, "null encoder");
    enc->stop_after = tap ? tap : "";
    return PVR_OK;
}

pvr_status pvr_encoder_tap(pvr_encoder *enc, const char *name, float *out, int64_t cap, int64_t *count, void *hip_stream) {
    PVR_REQUIRE(enc && name && out && count, "pvr_encoder_tap: null argument");
    PVR_NO_HOST(enc, "pvr_encoder_tap");
    if (!enc->finalized || enc->last_n == 0) { set_error("no forward has run"); return PVR_ERR_STATE; }
    hipStream_t st = (hipStream_t)hip_stream;
    if (enc->vit) return vit_tap(enc, name, out, cap, count, st);
    const int n = enc->last_n, crop = enc->desc.crop;
    const std::string nm = name;
    const void *src = nullptr;
    size_t elems = 0;
    int f32 = 0;
    if (nm == "pre") { src = enc->d_img; elems = (size_t)n * (crop + 6) * (crop + 8) * 4; }
    else if (nm == "stem") { src = enc->d_stem; elems = (size_t)n * 112 * 112 * 64; }
    else if (nm == "pool") { src = enc->d_buf[B_X0]; elems = (size_t)n * 56 * 56 * 64; }
    else if (nm.compare(0, 3, "buf") == 0 && nm.find(':') != std::string::npos) {   // debug: "buf<b>:<elems>" = raw workspace buffer b
        const int b = atoi(nm.c_str() + 3);
        PVR_REQUIRE(b >= 0 && b < B_F32, "tap %s: 16-bit workspace buffers are 0..%d", name, B_F32 - 1);
        src = enc->d_buf[b]; elems = (size_t)atoll(nm.c_str() + nm.find(':') + 1);
    } else {
        auto it = enc->taps.find(nm);
        PVR_REQUIRE(it != enc->taps.end(), "unknown tap %s", name);
        // taps alias ping-pong buffers: only the LAST layer's tap is guaranteed intact after a full forward - and not even that one when
        // the forward pooled inside its last convolution (conv_wfrag's pooled form: the (n,7,7,2048) activation was never written)
        const auto &g = it->second.second;
        if (it->second.first == B_F32 && enc->last_pooled) {
            set_error("tap %s: the last forward averaged inside its last convolution and never wrote this activation; run the forward with "
                      "pvr_encoder_debug_stop_after(enc, \"%s\") or pvr_encoder_debug_set_switch(enc, \"pool_fuse\", 0) first", name, name);
            return PVR_ERR_STATE;
        }
        src = enc->d_buf[it->second.first];
        elems = (size_t)n * g[0] * g[1] * g[2];
        f32 = g[3];
    }
    PVR_REQUIRE((int64_t)elems <= cap, "tap %s needs %zu elements, cap %lld", name, elems, (long long)cap);
    *count = (int64_t)elems;
    const bool img16 = nm == "pre";                     // the transformed image stays 16-bit in every mode
    if (f32 || (enc->desc.dtype == PVR_F32 && !img16)) {
        PVR_HIP_TRY(hipMemcpyAsync(out, src, elems * 4, hipMemcpyDeviceToDevice, st));
        return PVR_OK;
    }
    return launch_h_to_f32(src, out, elems, enc->desc.dtype == PVR_F32 ? PVR_BF16 : enc->desc.dtype, st);
}

void pvr_encoder_destroy(pvr_encoder *enc) {
    if (!enc) return;
    if (enc->hplan) host_destroy(enc);
    if (enc->vit) vit_destroy(enc);
    if (enc->rnd) random5_destroy(enc);
    for (auto &op : enc->ops) { if (op.d_w) (void)hipFree(op.d_w); if (op.d_wp) (void)hipFree(op.d_wp); if (op.d_wpb) (void)hipFree(op.d_wpb); if (op.d_wfb) (void)hipFree(op.d_wfb); if (op.d_wcat) (void)hipFree(op.d_wcat); if (op.d_wf) (void)hipFree(op.d_wf); if (op.d_wsp) (void)hipFree(op.d_wsp); if (op.d_wsp_pair) (void)hipFree(op.d_wsp_pair); if (op.d_b_pair) (void)hipFree(op.d_b_pair); if (op.d_wpk) (void)hipFree(op.d_wpk); if (op.d_b) (void)hipFree(op.d_b); if (op.d_bsum) (void)hipFree(op.d_bsum); }
    if (enc->d_stem_wf) (void)hipFree(enc->d_stem_wf);
    if (enc->d_stem_c1w) (void)hipFree(enc->d_stem_c1w);
    bool any_lane = false;
    for (auto &l : enc->lane_ws) {
        if (!l.valid) continue;
        any_lane = true;
        for (int b = 0; b < B_COUNT; ++b) if (l.d_buf[b]) (void)hipFree(l.d_buf[b]);
        if (l.d_img) (void)hipFree(l.d_img);
        if (l.d_stem) (void)hipFree(l.d_stem);
        if (l.d_imgf) (void)hipFree(l.d_imgf);
    }
    if (!any_lane) {
        for (int b = 0; b < B_COUNT; ++b) if (enc->d_buf[b]) (void)hipFree(enc->d_buf[b]);
        if (enc->d_img) (void)hipFree(enc->d_img);
        if (enc->d_stem) (void)hipFree(enc->d_stem);
        if (enc->d_imgf) (void)hipFree(enc->d_imgf);
    }
    if (enc->d_stem_w) (void)hipFree(enc->d_stem_w);
    if (enc->d_stem_b) (void)hipFree(enc->d_stem_b);
    if (enc->d_zero) (void)hipFree(enc->d_zero);
    resizer_destroy(enc);
    for (auto &ev : enc->lane_done) if (ev) (void)hipEventDestroy(ev);
    for (float *q : enc->d_smallk) if (q) (void)hipFree(q);
    for (void *q : {(void *)enc->ap_wqkv, (void *)enc->ap_wc, (void *)enc->ap_bqkv, (void *)enc->ap_bc, (void *)enc->ap_pos, (void *)enc->ap_out}) if (q) (void)hipFree(q);
    delete enc;
}

}  // extern "C"
