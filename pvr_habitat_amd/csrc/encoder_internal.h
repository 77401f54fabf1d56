// Internal declarations shared by encoder.hip (ResNet50 family) and vit.hip (CLIP ViT).
#pragma once
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include "common.h"
#include "chain_params.h"

namespace pvr {

pvr_status launch_preprocess(const uint8_t *, int, int, int, int, int, void *, int, hipStream_t, int crop_pos = 0);
pvr_status launch_stem(const void *, const void *, const float *, void *, int, int, int, hipStream_t);
pvr_status launch_maxpool(const void *, void *, int, int, int, int, int, hipStream_t);
// (c1_*: layer1.0.conv1 inside the stem - stem.hip, StemC1; only when stem_conv1_capable())
pvr_status launch_stem_pool(const void *, const void *, const float *, void *, int, int, int, hipStream_t, const void *c1_w = nullptr, const float *c1_b = nullptr,
                            void *c1_t1 = nullptr, int c1_blk = 0);
bool stem_pool_u8_ok(const void *, int, int, int, int);   // geometry only; the PVR_STEM_U8 / PVR_STEM_LDS switches live in PlanSwitches
pvr_status launch_stem_pool_u8(const uint8_t *, int, int, int, int, int, const void *, const float *, void *, int, hipStream_t, const void *c1_w = nullptr,
                               const float *c1_b = nullptr, void *c1_t1 = nullptr, int c1_blk = 0);
bool stem_conv1_capable();
void stem_c1_pack(const u16 *w, u16 *img);
void preprocess_geometry(int h, int w, int resize, int crop, int crop_pos, int *resize_needed, int *top, int *left);
pvr_status launch_avgpool(const void *, float *, int64_t, int, int, int, int, int, hipStream_t);
pvr_status launch_nhwc_to_chw(const float *, float *, int64_t, int, int, int, int, hipStream_t);
pvr_status launch_h_to_f32(const void *, float *, size_t, int, hipStream_t);
pvr_status launch_f32_to_h(const float *, void *, size_t, int, hipStream_t);
pvr_status launch_conv_splitk(const void *, const void *, const float *, const void *, void *, const void *, float *, int, int, int, int, int, int,
                              int, int, int, int, int, int, int, hipStream_t);
pvr_status launch_conv(const void *, const void *, const float *, const void *, void *, const void *, int, int, int, int,
                       int, int, int, int, int, int, int, int, hipStream_t);

// conv_expand.hip: persistent weight-stationary 1x1 convolutions (out_blk: blocked output layout for chain_wave.hip)
bool conv_expand_supported(int64_t M, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32, bool has_res);
pvr_status launch_conv_expand(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                              int cout, int stride, int relu, int dtype, hipStream_t stream, int out_blk = 0);

// conv_pp256.hip: 256x256-tile ping-pong kernel for deep-K convolutions / linear layers
bool pp256_supported(int64_t M, int cin, int cout, int kh, int kw, int64_t in_bytes, int64_t w_bytes, int64_t out_bytes, int64_t res_bytes);
pvr_status launch_conv_pp256(const void *in, const void *wgt, const float *bias, const void *res, void *out, int n, int h, int w, int cin,
                             int cout, int kh, int kw, int stride, int pad, int act, int out_f32, int res_f32, int dtype, int bm, hipStream_t stream,
                             const void *in2 = nullptr, int h2 = 0, int w2 = 0, int cin2 = 0, int stride2 = 1);
int conv_algo();
void set_conv_algo(int a);

// bneck_frame.hip: per-frame fused tail of the layer3 bottlenecks (conv2 -> conv3 + residual [-> the next block's conv1]); weights in the fragment-blocked layout
bool bneck_frame_supported(int n, int h, int w, int cm, int cout, int stride);
pvr_status launch_pack_frag_weights(const void *w, void *out, int rows, int K, hipStream_t stream);
pvr_status launch_bneck_frame(const void *t1, const void *w2p, const float *b2, const void *w3p, const float *b3, const void *res, void *y,
                              void *t2_out, int n, int phases, int dtype, hipStream_t stream, unsigned long long *stamps = nullptr,
                              const void *w1np = nullptr, const float *b1n = nullptr, void *t1n = nullptr, const void *w1fp = nullptr, const float *b1f = nullptr);
// round 6: consecutive whole bottlenecks of the stage per frame in ONE launch (blocks[k + 1].res == blocks[k].y); odd workgroups start `stagger` x 8128 cycles late
struct BFBlk {
    const unsigned short *w1f, *w2, *w3, *res;
    const float *b1f, *b2, *b3;
    unsigned short *y;
};
pvr_status launch_bneck_frame_run(const BFBlk *blocks, int nblk, int n, int dtype, hipStream_t stream, int stagger);
// bneck_frame64.hip (round 6): the whole bottleneck per frame with ONE wave per SIMD, 64 output channels x 13 pixel tiles per wave (half the LDS reads per MFMA)
pvr_status launch_bneck_frame64(const void *w1p, const float *b1, const void *w2p, const float *b2, const void *w3p, const float *b3, const void *x, void *y, int n,
                                int dtype, hipStream_t stream, unsigned long long *stamps);
bool frame64_on();
void set_frame64(int mode);
long long bneck_frame64_launches();

// conv_wfrag.hip: implicit GEMM in 112-pixel x 256-cout tiles with the weights read from L2 as MFMA fragments (layer4 at batch 256)
bool conv_wfrag_supported(int64_t M, int64_t in_bytes, int cin, int cout, int kh, int kw, int pad, int act, int out_f32);
bool conv_wfrag_preferred(int64_t M, int cin, int cout, int kh, int kw);
pvr_status launch_conv_wfrag(const void *in, const void *wp, const float *bias, const void *res, void *out, int n, int h, int w, int cin, int cout,
                             int kh, int kw, int stride, int pad, int act, int out_f32, int dtype, hipStream_t stream, float *pool_out = nullptr,
                             int64_t pool_stride = 0);

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

constexpr int PVR_MAX_LANES = 4;
// B_Y0 / B_Y1: fp32 residual stream of the compressed PVRs' parity plan (allocated only for that plan); B_STEM: d_stem (112x112x64), not in d_buf
enum BufId { B_NONE = -1, B_X0 = 0, B_X1, B_T1, B_T2, B_DS, B_F32, B_Y0, B_Y1, B_COUNT, B_STEM = B_COUNT };

struct ConvOp {
    std::string conv, bn;          // state_dict prefixes
    int in_buf, out_buf, res_buf;
    int h, w, cin, cin_real, cout, cout_real, k, stride, pad, relu, out_f32;
    int kind = 0;                  // 0 convolution, 1 AvgPool2d(2) on NHWC 16-bit (CLIP ModifiedResNet; cin = channels), 2 fp32 -> 16-bit copy
    bool f32op = false;            // convolution on fp32 buffers with fp32 weights on the f32-input MFMA (conv_f32.hip) inside a 16-bit plan
    bool from32 = false;           // a 16-bit convolution (16-bit weights, one MFMA per product) whose INPUT is the fp32 residual stream: conv_split16's single-term
                                   // form rounds the operand in its staging pass - the fp32 -> 16-bit copy launch of the stream is gone (round 6)
    u16 *d_w = nullptr;
    u16 *d_wp = nullptr;           // row-permuted copy for the fused bottleneck chain (bottleneck_chain.hip)
    u16 *d_wfb = nullptr;          // fragment-blocked copy of d_w for the per-frame layer3 tail (bneck_frame.hip: launch_pack_frag_weights)
    u16 *d_wpb = nullptr;          // ... and that copy in the blocked layout [row >> 4][cin >> 3][row & 15][8] (chain_wave.hip reads W3 / Wd pieces from L2)
    std::vector<u16> h_w;          // host copy, kept until finalize has built the chain copies
    float *d_wf = nullptr;         // fp32 weights (PVR_F32 mode)
    u16 *d_wpk = nullptr;          // conv2 of a layer2 wave-form tail: the launch's 17 weight units as LDS images (chain_wave128.hip: launch_chain_wave128_pack)
    u16 *d_wsp_pair = nullptr;     // compression head: [conv1 ; downsample] rows as ONE split weight image (both read the same fp32 input: one launch, round 6)
    float *d_b_pair = nullptr;
    u16 *d_wsp = nullptr;          // fp32 weights as (hi, lo) f16 fragment pairs (conv_split16.hip: the f32op convolutions of an f16 plan)
    float *d_b = nullptr;
    std::vector<float> h_b;        // host copy of the bias (same lifetime as h_w)
    float *d_bsum = nullptr;       // conv3 of a block whose downsample runs inside the chain / the two-operand launch: b3 + b_downsample
    u16 *d_wcat = nullptr;         // conv3 of a two-operand launch (conv_pp256 DUAL): [W3 | W_downsample] rows, (cout_pad, cin + cin_downsample)
    std::string tap;               // non-empty: output of this op is the named tap
    int ksplit = 0, ks_buf = B_NONE;   // split-K launch (conv_igemm.hip): number of K ranges, workspace buffer that is dead at this op
};

// one launch of the forward plan: a single convolution, or a fused bottleneck tail
// (conv2 3x3 -> conv3 1x1 + residual -> the next block's conv1 1x1; bottleneck_chain.hip)
struct Launch {
    int conv2 = -1, conv3 = -1, next1 = -1;   // chain members (indices into ops); conv3 < 0: single launch of ops[conv2]
    int ds = -1;                              // chain: the block's downsample convolution, accumulated inside conv3 (no launch of its own);
                                              // with conv3 < 0: ops[conv2] is a conv3 that runs as conv_pp256's two-operand launch with ops[ds] (layer3.0 / layer4.0)
    int t1_in = B_NONE, t1_out = B_NONE;      // chain: buffer holding conv2's input / receiving the next block's conv1 output
    int wave = 0;                             // chain: 1 the wave form runs it (chain_wave.hip), 2 the layer2 wave form (chain_wave128.hip)
    int conv1 = -1;                           // per-frame form: the block's own conv1 runs in front, inside the launch (the launch reads the block input)
    int frame = 0;                            // per-frame form (bneck_frame.hip, layer3): conv2 -> conv3 + residual [-> next1] of one 14 x 14 image per workgroup
    int pair = -1;                            // conv_split16 pair form: ops[conv2] and ops[pair] read the same fp32 input and run as one launch (the compression head)
    int in_blk = 0, out_blk = 0;              // chain, wave form: t1 + residual / y + t1' travel in the blocked layout between two such launches (chain_wave.hip);
                                              // block form: out_blk 1 = y blocked, 3 = y and t1' blocked (a layer2 wave-form launch follows)
};

// What one launch of the plan runs as for a forward of nb frames: resolved off the hot path (resolve_kinds: finalize, set_low_latency,
// debug_set_fusion, debug_set_switch), one byte per (nb, launch); the forward is a switch over these.
enum LaunchKind : uint8_t {
    LK_CONV = 0,          // launch_conv (conv_igemm.hip picks igemm / pp256 / expand / halo by shape)
    LK_FRAME_FRONT1,      // bneck_frame: conv1 -> conv2 -> conv3 + identity of one 14 x 14 image per workgroup
    LK_FRAME,             // bneck_frame: conv2 -> conv3 + identity [-> next conv1]
    LK_FRAME_RUN,         // round 6: the first of >= 2 consecutive LK_FRAME_FRONT1 launches - ONE launch takes every frame through all of them
    LK_FRAME_RUN_TAIL,    //          ... and the others (no launch of their own)
    LK_FRAME_MEMBERS,     // the member convolutions of a per-frame launch as their own launches (small forwards)
    LK_DUAL,              // conv_pp256 two-operand launch: conv3 & the stride-2 downsample
    LK_DUAL_MEMBERS,      // ... as two launches (low-latency plan, PVR_CONV_ALGO)
    LK_CHAIN,             // bottleneck_chain / chain_wave: conv2 -> conv3 (+ residual / downsample) -> next conv1
    LK_CAST,              // fp32 -> 16-bit copy
    LK_F32,               // conv_f32: fp32 operands on the f32-input MFMA
    LK_SPLIT16,           // conv_split16: fp32 operands as 16-bit (hi, lo) pairs on the 16-bit MFMA
    LK_SPLIT16_PAIR,      // ... two convolutions of the same input in one launch (the compression head's conv1 & downsample)
    LK_SPLIT16_IN32,      // ... single-term form: a 16-bit convolution that reads the fp32 residual stream itself
    LK_SPLITK_SMALL,      // low-latency plan: split-K over the lane's scratch
    LK_SPLITK,            // planned split-K (the *_l4 compression head)
    LK_EXPAND_BLOCKED,    // conv_expand writing the blocked layout in front of a wave-form tail
    LK_WFRAG_POOL,        // conv_wfrag with AdaptiveAvgPool2d(1) in the epilogue
    LK_WFRAG,             // conv_wfrag
};
const char *launch_kind_name(int k);

// A/B switches of the plan, read from the environment ONCE per encoder (pvr_encoder_create) - never on the forward path.  The ones marked
// (live) can be changed on a finalized encoder with pvr_encoder_debug_set_switch; the others shape the plan and are fixed at finalize.
struct PlanSwitches {
    int pool_fuse = 1;        // PVR_POOL_FUSE (live): the pooled form of conv_wfrag for the trunk's last launch
    int stem_u8 = 1;          // PVR_STEM_U8 (live): the fused stem reads uint8 frames itself when no resize is needed
    int stem_lds = 1;         // PVR_STEM_LDS
    int frame_front1 = 1;     // PVR_FRAME_FRONT1: layer3's per-frame launches carry their own conv1
    int frame_next1 = 0;      // PVR_FRAME_NEXT1: ... carry the NEXT block's conv1 instead (measured slower)
    int dual_ds = 1;          // PVR_DUAL_DS: conv3 & downsample of layer3.0 / layer4.0 as one two-operand launch
    int chain_ds = 1;         // PVR_CHAIN_DS: layer1.0's downsample inside the chain
    int chain_blocked = 1;    // PVR_CHAIN_BLOCKED: blocked hand-off between consecutive tails
    int splitk = 1;           // PVR_SPLITK: planned split-K of the *_l4 head
    int smallk_div = 4;       // PVR_SMALLK_DIV: K slices per block of the low-latency plan
    int frame_run = 0;        // PVR_FRAME_RUN (live, opt-in: measured equal): consecutive whole-bottleneck frame launches (layer3.1 .. 3.5) as one launch
    int frame_stagger = 0;    // PVR_FRAME_RUN_STAGGER: odd workgroups of that launch start this many x 8128 cycles late
    int frame_min_n = 128;    // PVR_FRAME_MIN_N (live): frames per forward from which layer3's per-frame launches run as such
    int stem_conv1 = 1;       // PVR_STEM_CONV1: layer1.0.conv1 runs inside the fused stem (no launch of its own; round 6)
    int split16 = 1;          // PVR_SPLIT16: the fp32 stage / head of the compressed PVRs' parity plan on the 16-bit MFMA (0: f32-input MFMA)
    int resid32 = 1;          // PVR_RESID32: fp32 residual stream of that plan (0: all-16-bit plan)
    int tail_f32 = 1;         // PVR_TAIL_F32: its last trunk stage entirely in fp32
    int fuse = 1;             // PVR_FUSE: the fused schedule (0: one launch per convolution; also pvr_encoder_debug_set_fusion)
};

}  // namespace pvr

using namespace pvr;

namespace pvr { struct HostPlan; }

struct pvr_encoder {
    pvr_encoder_desc desc;
    std::map<std::string, HostTensor> weights;
    std::vector<ConvOp> ops;
    std::vector<Launch> sched_plain, sched_fused;   // one launch per op / with the layer1-layer2 bottleneck tails fused
    bool fuse = true;                               // PVR_FUSE=0 or pvr_encoder_debug_set_fusion(enc, 0) selects sched_plain
    bool low_latency = false;                       // pvr_encoder_set_low_latency: split-K plan for forwards of <= 4 frames
    PlanSwitches sw;                                // environment switches, read once in pvr_encoder_create
    int stem_c1 = -1;                               // fused schedule: ops[stem_c1] = layer1.0.conv1 has no launch - the stem runs it (stem.hip, StemC1) or, where that
    int stem_c1_blk = 0;                            // form does not apply, the forward launches it in front of the plan; _blk: the tail behind it reads t1 blocked
    u16 *d_stem_c1w = nullptr;                      // its weights as the stem's fragment image (stem_c1_pack)
    std::vector<uint8_t> kinds;                     // LaunchKind of launch i for a forward of nb frames: kinds[(nb - 1) * plan.size() + i] (resolve_kinds)
    size_t kinds_stride = 0;
    int kinds_algo = -2;                            // conv_algo() the table was resolved under
    int *range_flags = nullptr;                     // pvr_encoder_check_range: per-launch "output holds inf / NaN" flags of the forward in progress (else null)
    bool last_pooled = false;                       // the last forward wrote the pooled rows from the last convolution: the B_F32 tap does not exist
    float *d_smallk[PVR_MAX_LANES] = {nullptr};     // the low-latency plan's fp32 partial planes, per lane (pvr_encoder_set_low_latency / first use of a lane: never in a forward)
    bool tail32 = false;                            // round 3: + the last trunk stage entirely in fp32 (conv_f32.hip), fp32 stream one stage earlier
    bool resid32 = false;                           // compressed PVRs, f16: fp32 residual stream from layer3 on + fp32 compression head
    bool finalized = false;
    int out_size = 0;
    int final_hw = 0, final_c = 0, final_creal = 0;   // geometry of the last activation
    // device
    u16 *d_img = nullptr, *d_stem = nullptr, *d_pool = nullptr, *d_stem_w = nullptr, *d_zero = nullptr;
    float *d_stem_b = nullptr, *d_stem_wf = nullptr, *d_imgf = nullptr;   // fp32 mode: [64][49][4] stem weights, normalised NHWC4 image
    void *d_buf[B_COUNT] = {nullptr};
    size_t buf_elems = 0;
    // second activation workspace (pvr_encoder_forward_lane, lane 1): lets the caller keep two batches in flight on two
    // streams; allocated on first use.  The members above are the CURRENT lane's pointers (swapped by use_lane).
    struct LaneWs { u16 *d_img = nullptr, *d_stem = nullptr; float *d_imgf = nullptr; void *d_buf[B_COUNT] = {nullptr}; bool valid = false; } lane_ws[PVR_MAX_LANES];
    int cur_lane = 0;
    hipEvent_t lane_done[PVR_MAX_LANES] = {nullptr};   // recorded after each forward on the lane; the next forward on it waits
    hipStream_t lane_stream[PVR_MAX_LANES] = {nullptr};   // stream of that forward (no wait when the stream is the same)
    int crop_pos = 0;                                // 0 centre (reference), 1..4 corner crops (pvr_encoder_set_crop_position)
    int span_first = -1, span_last = -1;             // pvr_encoder_profile_span: the two marks of the forward that are recorded
    int last_n = 0;
    std::string stop_after;                                          // debug: end the forward after this tap
    std::map<std::string, std::pair<int, std::vector<int>>> taps;   // name -> (buf, {h,w,c,is_f32})
    // CLIP RN50 (clip_rn50.hip): antialiased-bicubic resizer (a weight-less pvr_vit), attention-pool parameters
    struct pvr_vit *resizer = nullptr;
    u16 *ap_wqkv = nullptr, *ap_wc = nullptr;
    float *ap_bqkv = nullptr, *ap_bc = nullptr, *ap_pos = nullptr, *ap_out = nullptr;
    struct pvr_vit *vit = nullptr;
    bool host = false;                                               // pvr_encoder_set_host_backend: CPU plan (host_encoder.hip), host pointers in / out
    struct pvr::HostPlan *hplan = nullptr;
    struct pvr_random5 *rnd = nullptr;                               // 'random' 5-conv PVR (random_pvr.hip)                                   // CLIP ViT plan (vit.hip) when arch >= PVR_ARCH_CLIP_VIT_B32
};


namespace pvr {
const HostTensor *enc_find(pvr_encoder *e, const std::string &name);
pvr_status enc_need(pvr_encoder *e, const std::string &name, const HostTensor **out, size_t numel);
template <typename T>
pvr_status enc_upload(T **dptr, const std::vector<T> &h) {
    PVR_HIP_TRY(hipMalloc((void **)dptr, h.size() * sizeof(T)));
    PVR_HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return PVR_OK;
}
// conv_f32.hip (PVR_F32 reference-precision mode) and the fp32 normaliser of random_pvr.hip
pvr_status launch_conv_f32(const float *, const float *, const float *, const float *, float *, int, int, int, int, int, int, int, int, int, hipStream_t);
pvr_status launch_stem_f32(const float *, const float *, const float *, float *, int, int, hipStream_t);
pvr_status launch_maxpool_f32(const float *, float *, int, int, int, int, hipStream_t);
pvr_status launch_normalize_nhwc4(const void *img_h, float *out, int n, int crop, const float *mean, const float *std_, int dtype, hipStream_t);
// random_pvr.hip
pvr_status random5_create(pvr_encoder *e);
pvr_status random5_finalize(pvr_encoder *e);
pvr_status random5_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride, hipStream_t st);
void random5_destroy(pvr_encoder *e);
// host_encoder.hip (CPU plan behind the same ABI)
pvr_status host_finalize(pvr_encoder *e);
pvr_status host_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride);
void host_destroy(pvr_encoder *e);
// vit.hip
pvr_status vit_create(pvr_encoder *e);
pvr_status vit_finalize(pvr_encoder *e);
pvr_status vit_use_lane(pvr_encoder *e, int lane);
// vit.hip pieces shared with the CLIP RN50 plan: Resize(224, bicubic, antialias) + CenterCrop into a (res,res,3) uint8 image, attention core
pvr_status resizer_create(pvr_encoder *e);
pvr_status resizer_run(pvr_encoder *e, int lane, const uint8_t *frames, int nb, int h, int w, hipStream_t st, const uint8_t **u8, int *oh, int *ow);
void resizer_destroy(pvr_encoder *e);
pvr_status launch_attention(const void *qkv, void *out, int T, int W, int heads, int nb, int dtype, hipStream_t st);
// clip_rn50.hip
pvr_status launch_avgpool2(const void *in, void *out, int n, int h, int w, int c, int dtype, hipStream_t st);
pvr_status launch_attnpool_tokens(const float *x, const float *pos, void *tokens, int n, int hw, int c, int dtype, hipStream_t st);
pvr_status launch_stem(const void *, const void *, const float *, void *, int, int, int, hipStream_t);
pvr_status vit_forward(pvr_encoder *e, const uint8_t *frames, int n, int h, int w, float *out, int64_t out_stride, hipStream_t st);
void vit_destroy(pvr_encoder *e);
pvr_status vit_tap(pvr_encoder *e, const char *name, float *out, int64_t cap, int64_t *count, hipStream_t st);
}  // namespace pvr
