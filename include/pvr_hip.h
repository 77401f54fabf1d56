/*
 * pvr_hip.h — C-ABI of libpvr_hip.so: the MI355X (gfx950) implementation of the PVR-embedding
 * and behavioural-cloning hot path of sparisi/pvr_habitat.
 *
 * The reference has no FFI: its boundary for this path is the Python class surface of
 * src/embeddings.py (EmbeddingNet, :339-402) and src/models.py (PolicyNet :13-89,
 * PolicyNetWithConv :96-197) plus the training loop of main_bc_2.py:186-227.  Every entry
 * point below names the reference code it replaces.  Plain C types only: raw device/host
 * pointers, sizes, an opaque hipStream_t passed as void*.  Status return: 0 = OK, non-zero =
 * error (message via pvr_last_error, thread-local).  Nothing throws across the ABI.
 *
 * Ownership: the caller owns every buffer it passes (PyTorch-ROCm tensors on the Python
 * side); the library borrows pointers for the duration of the enqueue and owns only its
 * repacked weights and activation workspace (freed in *_destroy).  All work is enqueued on
 * the caller's stream; no hidden synchronisation on the forward/step paths.
 */
#ifndef PVR_HIP_H
#define PVR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pvr_status;
#define PVR_OK 0
#define PVR_ERR_INVALID 1
#define PVR_ERR_MISSING_WEIGHT 2
#define PVR_ERR_HIP 3
#define PVR_ERR_STATE 4

/* storage / MFMA input type of the encoder ("throughput" = bf16, "parity" = f16); accumulation
 * is always fp32.  PVR_F32 (ResNet50 family only) stores and multiplies in fp32 on the f32-input MFMA: the
 * reference's own arithmetic type, ~1e-6 from the fp32 oracle, at the f32 MFMA rate */
#define PVR_BF16 0
#define PVR_F16 1
#define PVR_F32 2

/* encoder architectures */
#define PVR_ARCH_RESNET50 0      /* torchvision resnet50, fc=Identity  -> 2048 (embeddings.py:118-120, moco.py:6-26) */
#define PVR_ARCH_RESNET50_L4 1   /* + BasicBlock(2048->42), no avgpool  -> 2058 (moco.py:73-113, resnet.py:47-83)   */
#define PVR_ARCH_RESNET50_L3 2   /* layer3 + BasicBlock(1024->11)       -> 2156 (moco.py:29-70,  resnet.py:6-44)    */
#define PVR_ARCH_CLIP_VIT_B32 3  /* openai/CLIP visual ViT-B/32 encode_image -> 512 (embeddings.py:303-304,375-376) */
#define PVR_ARCH_CLIP_VIT_B16 4  /* same block layout, patch 16 (197 tokens) -> 512 (BASELINE config 3)          */
#define PVR_ARCH_RANDOM5 6       /* 'random' PVR: 5 x (conv3x3 s2 + ELU), fp32 -> 1568 (embeddings.py:90-106)                */
#define PVR_ARCH_MAE_VIT_L16 7   /* MAE/timm ViT-L/16 encoder (width 1024, 24 blocks, 16 heads) -> 1024 (mae.py:283-288, embeddings.py:141-144) */
#define PVR_ARCH_MAE_VIT_H14 8   /* MAE/timm ViT-H/14 encoder (width 1280, 32 blocks, 16 heads of 80, 257 tokens) -> 1280 (mae.py:291-296, embeddings.py:145-148) */
#define PVR_ARCH_CLIP_RN50 9     /* openai/CLIP visual ModifiedResNet-50 + attention pool -> 1024 (embeddings.py:305-306, 375-376) */
#define PVR_ARCH_RESNET18 10     /* torchvision resnet18, fc=Identity -> 512 (embeddings.py:112-114) */
#define PVR_ARCH_RESNET34 11     /* torchvision resnet34, fc=Identity -> 512 (embeddings.py:115-117; in the reference's sweeps, slurm_eo.py:100) */
#define PVR_ARCH_MAE_VIT_B16 5   /* MAE/timm ViT-B/16 encoder, CLS token  -> 768 (mae.py:202-222, embeddings.py:137-140,377-379) */

const char *pvr_version(void);
/* 1 if the library was built with its measured-slower experiment kernels (make EXPERIMENTS=1: conv_w4 = pvr_debug_set_conv_algo(4), the
 * split-bf16 GEMM = pvr_debug_set_gemm_mode(1..3), PVR_POLICY_BWD_FUSED, PVR_POLICY_PERSIST_BWD); the shipped build returns 0 and refuses them */
int32_t pvr_has_experiments(void);
/* test hook for the uint8-reading stem (stem.hip): 1 if frames of h x w uint8 pixels at `frames` with the 224 x 224 crop window at (top, left)
 * can take the path without a preprocess launch - the window must lie inside the frame and the rows must be 16-byte aligned; the
 * encoder falls back to preprocess + stem otherwise.  Host-side predicate, no GPU work. */
int32_t pvr_debug_stem_u8_geometry_ok(const void *frames, int32_t h, int32_t w, int32_t top, int32_t left);
/* test hook: the uniform in the OPEN interval (0, 1) that the training-mode action sampler (pvr_policy_set_action_sampling) makes of 32 random
 * bits; never 0 or 1 for any input.  Host arithmetic, no GPU work. */
float pvr_debug_sample_uniform(uint32_t bits);
/* copies the calling thread's last error message; returns its length */
size_t pvr_last_error(char *buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * Frozen encoder: replaces EmbeddingNet.forward (src/embeddings.py:386-402) = H2D'd uint8 NHWC
 * frames -> Resize(256)/CenterCrop(224)/ConvertImageDtype/Normalize (:80-85) -> ResNet50-family
 * eval forward -> flatten, fp32 (N, out_size).
 * ------------------------------------------------------------------------------------------- */
typedef struct pvr_encoder pvr_encoder;

typedef struct pvr_encoder_desc {
    int32_t arch;          /* PVR_ARCH_* */
    int32_t dtype;         /* PVR_BF16, PVR_F16, or PVR_F32 (ResNet50 family) */
    int32_t max_batch;     /* frames per forward call the workspace is sized for */
    int32_t chunk;         /* frames pushed through the layer stack at a time (0 = max_batch) */
    int32_t resize;        /* short-side target, 256 (embeddings.py:81) */
    int32_t crop;          /* centre crop, 224 (embeddings.py:82); must be 224 for ResNet50 */
    float mean[3];         /* Normalize mean (embeddings.py:84) */
    float std_[3];         /* Normalize std */
} pvr_encoder_desc;

pvr_status pvr_encoder_create(const pvr_encoder_desc *desc, pvr_encoder **out);
/* Hand over one fp32 host tensor under its torchvision state_dict name ("conv1.weight",
 * "layer1.0.bn1.running_var", ...; the nesting of moco.py's nn.Sequential(layer, BasicBlock) is
 * accepted as "layer4.0.<i>...." / "layer4.1....").  Unknown names are kept but unused
 * (load_state_dict(strict=False) semantics, moco.py:23). */
pvr_status pvr_encoder_load_weights(pvr_encoder *enc, const char *name, const float *host_data,
                                    const int64_t *shape, int32_t ndim);
/* Fold BN (eps 1e-5) and the input normalisation into the conv weights, repack to K-major
 * 16-bit tiles, upload, allocate the workspace.  Fails with PVR_ERR_MISSING_WEIGHT naming the
 * first absent key (the reference asserts len(missing_keys)==0, moco.py:24). */
pvr_status pvr_encoder_finalize(pvr_encoder *enc);
/* Host (CPU) backend, to be selected between create and finalize - the reference runs on the CPU when disable_cuda is set or no GPU is
 * present (src/embeddings.py:367-370; BASELINE configs[0] is a CPU plumbing run).  on != 0: finalize keeps fp32 BN-folded weights on the
 * host and makes no HIP call; pvr_encoder_forward / _forward_lane then take HOST pointers for frames and output and run plain C++ loops
 * over the same op list on this process's threads (PVR_HOST_THREADS, default all cores); hip_stream is ignored.  torchvision ResNet family
 * (arch RESNET50 / _L3 / _L4 / RESNET18 / RESNET34) with dtype PVR_F32 only; taps, profiling and lanes are GPU-plan features. */
pvr_status pvr_encoder_set_host_backend(pvr_encoder *enc, int32_t on);
int32_t pvr_encoder_out_size(const pvr_encoder *enc);
/* frames_dev: uint8 (n,h,w,3) on the device; out_dev: fp32, row i written at
 * out_dev + i*out_stride (elements), out_size values — pass an offset pointer to build the
 * UberModel concat (embeddings.py:55-57) without a cat kernel. n <= max_batch. */
pvr_status pvr_encoder_forward(pvr_encoder *enc, const uint8_t *frames_dev, int32_t n, int32_t h,
                               int32_t w, float *out_dev, int64_t out_stride, void *hip_stream);
/* debug/parity tap: copy an intermediate activation of the last forward, converted to fp32
 * NHWC, into out_dev. name: "pre","stem","pool","layer1".."layer4". Returns element count via *count. */
pvr_status pvr_encoder_tap(pvr_encoder *enc, const char *name, float *out_dev, int64_t cap,
                           int64_t *count, void *hip_stream);
/* debug / A-B: 0 runs one launch per convolution instead of the fused layer1-layer2 bottleneck tails
 * (bottleneck_chain.hip).  Both plans give bit-identical outputs.  Default on; env PVR_FUSE=0 also disables. */
pvr_status pvr_encoder_debug_set_fusion(pvr_encoder *enc, int32_t on);
/* name of launch `index` in the order pvr_encoder_profile reports; returns its length, 0 past the end */
int32_t pvr_encoder_launch_name(const pvr_encoder *enc, int32_t index, char *buf, int32_t cap);
/* Which 224x224 window of the resized frame the following forwards embed: 0 = centre (torchvision CenterCrop, the reference's
 * transform, embeddings.py:82; default), 1..4 = top-left, top-right, bottom-left, bottom-right (the corner crops of
 * torchvision FiveCrop).  Build-defined extension for BASELINE config 5 ("5-crop"); the reference has only the centre crop.
 * ResNet50 family only. */
pvr_status pvr_encoder_set_crop_position(pvr_encoder *enc, int32_t pos);
/* Same forward on one of up to four activation workspaces ("lanes" 0..3; lanes > 0 are allocated on first use).  Forwards on
 * DIFFERENT lanes may be in flight at once on different streams - e.g. batch k+1 on lane 1 while batch k drains on lane 0,
 * which fills the CUs that tile tails and HBM-bound launches of a single batch-256 forward leave idle (+15 % frames/s
 * measured).  Forwards on the SAME lane issued on different streams are chained by the library (each forward records a
 * per-lane event on its stream, the next forward on that lane waits for it on the device), so a workspace is never shared
 * by two forwards in flight; calls on one encoder handle must still come from one host thread at a time.
 * pvr_encoder_forward == lane 0.
 * replaces: the batch loop of behavioral_cloning/save_embedded_obs.py:151-156, which runs one batch at a time. */
pvr_status pvr_encoder_forward_lane(pvr_encoder *enc, int32_t lane, const uint8_t *frames_dev, int32_t n, int32_t h,
                                    int32_t w, float *out_dev, int64_t out_stride, void *hip_stream);
/* Low-latency plan for the online path - EmbeddingWrapper.observation (src/embeddings.py:441-444) embeds the N = 2 frames of one
 * environment step per call, inside the evaluation loop of src/test_model.py:4-22.  on != 0: forwards of <= 4 frames run their deep
 * convolutions split over K (several blocks per pixel tile), 0.88 -> ~0.5 ms per N = 2 ResNet50 call.  Results are independent of N
 * within the plan and differ from the default plan by fp32 regrouping only (<= 1 ulp of the storage type); larger forwards are
 * unaffected.  Default off (batch-size independence of the default plan is bit-exact).  ResNet50 family. */
pvr_status pvr_encoder_set_low_latency(pvr_encoder *enc, int32_t on);
/* debug: make pvr_encoder_forward return right after the named tap has been produced (NULL/"" = off) */
pvr_status pvr_encoder_debug_stop_after(pvr_encoder *enc, const char *tap);
/* Instrumented forward of one chunk (n <= chunk): HIP events between launches on the caller's stream;
 * synchronises.  op_ms[i] = duration of launch i (preprocess, stem, maxpool, convs..., pool/flatten),
 * op_flops[i] = its algorithmic FLOPs (0 for byte kernels).  Used by bench.py's roofline block. */
pvr_status pvr_encoder_profile(pvr_encoder *enc, const uint8_t *frames_dev, int32_t n, int32_t h, int32_t w,
                               float *out_dev, int64_t out_stride, void *hip_stream, float *op_ms,
                               double *op_flops, int32_t cap, int32_t *n_ops);
/* Wall time (ms) from the start of launch first_op to the end of launch last_op - indices as pvr_encoder_profile reports them - of one
 * forward that carries only those two events.  bench.py's conv-family time (launches 3 .. n_ops-2): the per-launch events of
 * pvr_encoder_profile serialise the dispatcher for a few microseconds each, which this figure does not contain. */
pvr_status pvr_encoder_profile_span(pvr_encoder *enc, const uint8_t *frames_dev, int32_t n, int32_t h, int32_t w, float *out_dev,
                                    int64_t out_stride, void *hip_stream, int32_t first_op, int32_t last_op, float *span_ms);
/* Load-time range validation of the 16-bit storage types (the reference computes in fp32, src/embeddings.py:386-402; f16 has 5 exponent bits): one forward of
 * the UNFUSED plan with every launch's output checked for inf / NaN.  An overflow inside the network does not always reach the embedding (+inf x a negative
 * weight = -inf, and ReLU maps -inf and NaN to 0), so a finite check of the output alone can miss it.  *first_bad = index of the first launch whose output is
 * non-finite (pvr_encoder_launch_name numbering of the unfused plan; its name says which convolution) or -1.  n <= chunk frames; synchronises; the embeddings of
 * this forward are written to out_dev as usual.  ResNet family, 16-bit plans. */
pvr_status pvr_encoder_check_range(pvr_encoder *enc, const uint8_t *frames_dev, int32_t n, int32_t h, int32_t w, float *out_dev, int64_t out_stride,
                                   void *hip_stream, int32_t *first_bad);
/* Debug / A-B: the run-time switches of a finalized encoder - "pool_fuse" (the trunk's last convolution writes the average pool itself), "stem_u8" (the
 * fused stem reads uint8 frames that need no resize), "frame_min_n" (frames per forward from which layer3 runs one workgroup per frame).  Every other
 * PVR_* switch shapes the plan and is read from the environment ONCE, in pvr_encoder_create; nothing reads the environment on the forward path. */
pvr_status pvr_encoder_debug_set_switch(pvr_encoder *enc, const char *name, int32_t value);
/* Which kernel family launch `index` (pvr_encoder_launch_name's indices) runs as in a forward of n frames, e.g. "bneck_frame(front1)", "conv_wfrag(pool)",
 * "conv_pp256(dual)", "chain_wave" / "chain_wave128" / "bottleneck_chain" (the three forms of the fused bottleneck tail), "conv_split16", "conv" (the shape-dispatched implicit GEMMs).  The choice is tabulated per batch size when the encoder is
 * finalized; returns the name's length, 0 past the end of the plan. */
int32_t pvr_encoder_launch_kernel(const pvr_encoder *enc, int32_t n, int32_t index, char *buf, int32_t cap);
void pvr_encoder_destroy(pvr_encoder *enc);

/* ---------------------------------------------------------------------------------------------
 * Single operators (unit-parity entry points; the encoder is built from these kernels).
 * dtype = PVR_BF16 / PVR_F16 for activations and weights.
 * ------------------------------------------------------------------------------------------- */
/* transforms (embeddings.py:80-85) up to and including the uint8 crop; output is the stem's input
 * image: (n, crop+6, crop+8, 4) 16-bit, pixel (y,x) at [y+3][x+3], channels (R-128,G-128,B-128,valid): exact
 * centred uint8 values (normalisation is folded into the stem weights), zero border (valid=0). */
pvr_status pvr_op_preprocess(const uint8_t *frames_dev, int32_t n, int32_t h, int32_t w, int32_t resize,
                             int32_t crop, void *out_dev, int32_t dtype, void *hip_stream);
/* conv1 7x7/2 + folded BN + ReLU on the padded image above. wgt: (64, 7*8*4) 16-bit, bias fp32(64).
 * out: (n,112,112,64) NHWC 16-bit */
pvr_status pvr_op_stem(const void *img_dev, const void *wgt_dev, const float *bias_dev, void *out_dev,
                       int32_t n, int32_t dtype, void *hip_stream);
/* maxpool 3x3/2 pad 1 on NHWC 16-bit */
pvr_status pvr_op_maxpool(const void *in_dev, void *out_dev, int32_t n, int32_t h, int32_t w, int32_t c,
                          int32_t dtype, void *hip_stream);
/* implicit-GEMM convolution, NHWC, + bias (+ residual) (+ ReLU).  wgt: (cout_pad, kh*kw*cin) 16-bit,
 * K index = (kh*KW+kw)*cin + c, cin % 64 == 0, cout_pad = cout rounded up to 64 (zero rows).
 * out: (n,ho,wo,cout) 16-bit, or fp32 if out_f32. */
pvr_status pvr_op_conv2d(const void *in_dev, const void *wgt_dev, const float *bias_dev,
                         const void *residual_dev, void *out_dev, int32_t n, int32_t h, int32_t w,
                         int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                         int32_t relu, int32_t out_f32, int32_t dtype, void *hip_stream);
/* Per-frame fused layer3 bottleneck (torchvision Bottleneck, reference src/embeddings.py:118-120; bneck_frame.hip), one workgroup per 14 x 14 frame:
 * [x (n,14,14,1024) -> the block's own conv1 1x1 (+ b1f, ReLU) ->] t1 (n,14,14,256) -> conv2 3x3 pad 1 (+ b2, ReLU, rounded to the storage type) -> conv3 1x1
 * to 1024 channels (+ b3 + residual (n,14,14,1024), ReLU) -> y (n,14,14,1024) [-> the NEXT block's conv1 1x1 (+ b1n, ReLU) -> t1n (n,14,14,256)].
 * w2: (256, 3*3*256), w3: (1024, 256), w1n / w1f: (256, 1024) in the fragment-blocked layout pvr_op_pack_frag_weights makes of pvr_op_conv2d's layout.
 * phases 1: conv2 only, its output to t2_out; 3: conv2 + conv3 (t2_out optional); 7: + the next conv1; 3 + 8 = 11: the block's own conv1 in front (the
 * launch reads the block input = residual_dev, t1_dev is not used).  Bit-identical to the pvr_op_conv2d calls it replaces. */
pvr_status pvr_op_bneck_frame(const void *t1_dev, const void *w2_dev, const float *b2_dev, const void *w3_dev, const float *b3_dev,
                              const void *residual_dev, void *y_dev, void *t2_out_dev, const void *w1n_dev, const float *b1n_dev, void *t1n_dev,
                              const void *w1f_dev, const float *b1f_dev, int32_t n, int32_t phases, int32_t dtype, void *hip_stream);
/* (rows, k) 16-bit weights in pvr_op_conv2d's layout -> the MFMA-fragment-blocked layout [row tile][k / 8][16 rows][8] with the rows permuted inside
 * every 32-row block (row 16 t + 4 a + c holds output channel 8 a + 4 t + c), same size; rows % 32 == 0, k % 32 == 0.  Device to device, enqueued. */
pvr_status pvr_op_pack_frag_weights(const void *w_dev, void *out_dev, int32_t rows, int32_t k, void *hip_stream);
/* diagnostics: the same launch (phases 3, or 7 when w1n_dev is given) writing s_memtime stamps of workgroup 8 to stamps_dev (20 x uint64: waves 0 and 4, ten phase boundaries each) */
pvr_status pvr_debug_bneck_frame_stamps(const void *t1_dev, const void *w2_dev, const float *b2_dev, const void *w3_dev, const float *b3_dev,
                                        const void *residual_dev, void *y_dev, const void *w1n_dev, const float *b1n_dev, void *t1n_dev,
                                        const void *w1f_dev, const float *b1f_dev, int32_t n, int32_t dtype, uint64_t *stamps_dev, void *hip_stream);
/* launches of that kernel so far (tests: the layer3 plan really took it) */
int64_t pvr_debug_bneck_frame_launches(void);
/* round 6: which kernel runs the whole-bottleneck form (w1f given; torchvision Bottleneck conv1 -> conv2 -> conv3 + identity reached from reference
 * src/embeddings.py:118-120): 1 bneck_frame64.hip (one wave per SIMD, 64 output channels x 13 pixel tiles per wave: half the LDS reads per MFMA), 0 the
 * 32-channel tiling of round 5, -1 back to the environment (PVR_FRAME64, default 0: bit-identical, measured slower).  Both give the same bits.  Process-global. */
pvr_status pvr_debug_set_frame64(int32_t mode);
int64_t pvr_debug_bneck_frame64_launches(void);
/* the 64-channel tiling with s_memtime stamps of workgroup 8, wave 0 (8 x uint64 on the device) - diagnostics only */
pvr_status pvr_debug_bneck_frame64_stamps(const void *w1f_dev, const float *b1f_dev, const void *w2_dev, const float *b2_dev, const void *w3_dev, const float *b3_dev,
                                          const void *x_dev, void *y_dev, int32_t n, int32_t dtype, uint64_t *stamps_dev, void *hip_stream);
/* pvr_op_conv2d's convolution for small pixel counts and deep K (layer4 at batch 256; conv_wfrag.hip): 112-pixel x 256-cout tiles, the weight operand read
 * from L2 as whole MFMA fragments.  wgt_packed: pvr_op_pack_frag_weights of the (cout, kh*kw*cin) matrix; cin % 64 == 0, cout % 256 == 0, kh == kw <= 3,
 * relu 0 / 1, residual 16-bit or NULL, out_f32 0 / 1.  Bit-identical to pvr_op_conv2d. */
pvr_status pvr_op_conv_wfrag(const void *in_dev, const void *wgt_packed_dev, const float *bias_dev, const void *residual_dev, void *out_dev, int32_t n,
                             int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t relu,
                             int32_t out_f32, int32_t dtype, void *hip_stream);
/* pvr_op_conv2d with a SECOND pixel operand appended along K (conv_pp256's two-operand form): out = act(W[:, :kh*kw*cin] . in + W[:, kh*kw*cin:] . in2 +
 * bias) - a stride-2 bottleneck's conv3 and its 1 x 1 downsample (torchvision Bottleneck.downsample, reference src/embeddings.py:118-120) in one fp32
 * accumulation.  in2: (n,h2,w2,cin2) read at (ho*stride2, wo*stride2), (h2-1)/stride2+1 == ho; wgt: (cout_pad, kh*kw*cin + cin2); 16-bit output. */
pvr_status pvr_op_conv2d_dual(const void *in_dev, const void *in2_dev, const void *wgt_dev, const float *bias_dev, void *out_dev, int32_t n, int32_t h,
                              int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t h2, int32_t w2,
                              int32_t cin2, int32_t stride2, int32_t relu, int32_t dtype, void *hip_stream);
/* The trunk's last convolution with AdaptiveAvgPool2d(1) inside (torchvision resnet.avgpool, reference src/embeddings.py:118-120): 1 x 1 convolution on
 * (n,7,7,cin) + bias + 16-bit residual (n,7,7,cout) + ReLU, averaged over each frame's 49 pixels in fp32: pool_out[f * pool_stride + c].  The
 * (n,7,7,cout) activation is never written.  Bit-identical to pvr_op_conv2d (fp32 output) followed by pvr_op_avgpool. */
pvr_status pvr_op_conv_wfrag_pool(const void *in_dev, const void *wgt_packed_dev, const float *bias_dev, const void *residual_dev, float *pool_out_dev,
                                  int64_t pool_stride, int32_t n, int32_t cin, int32_t cout, int32_t dtype, void *hip_stream);
/* launches of that kernel so far (tests: the layer4 plan really took it) */
int64_t pvr_debug_conv_wfrag_launches(void);
/* fp32 convolution on the 16-bit matrix pipe (conv_split16.hip): every fp32 operand as an exact (hi, lo) pair of f16 values, three MFMAs per fragment
 * pair, fp32 accumulation: the last trunk stage and the compression head of the `*_l3` / `*_l4` PVRs' parity plan (reference src/vision_models/moco.py:29-113
 * through src/embeddings.py:195-280), ~4e-7 relative per term against an fp32 dot product.  in / residual / out: fp32 NHWC, (n,h,w,cin) / (n,ho,wo,cout);
 * wgt_split: pvr_op_split16_pack_weights of the fp32 (cout_pad, k*k*cin) matrix (K index = (kh*k+kw)*cin + c; cout_pad = cout rounded up to 64, zero rows),
 * same byte size; cin % 32 == 0, cout % 16 == 0, k in 1..3; bias: cout_pad floats. */
pvr_status pvr_op_split16_pack_weights(const float *w_dev, void *out_dev, int32_t rows, int32_t k, void *hip_stream);
pvr_status pvr_op_conv2d_split16(const float *in_dev, const void *wgt_split_dev, const float *bias_dev, const float *residual_dev, float *out_dev, int32_t n,
                                 int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t relu, void *hip_stream);
/* the same convolution on the f32-input MFMA (conv_f32.hip, the PVR_F32 reference-precision mode's kernel): wgt fp32 (cout_pad, k*k*cin) */
pvr_status pvr_op_conv2d_f32(const float *in_dev, const float *wgt_dev, const float *bias_dev, const float *residual_dev, float *out_dev, int32_t n,
                             int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t relu, void *hip_stream);
int64_t pvr_debug_conv_split16_launches(void);
/* launches of the layer2 wave-form tail (chain_wave128.hip: torchvision Bottleneck conv2 -> conv3 + identity -> the next conv1 at Cm = 128, reference
 * src/embeddings.py:118-120) so far (tests: the layer2 plan really took it; PVR_CHAIN_WAVE_L2=0 keeps the block form) */
int64_t pvr_debug_chain_wave128_launches(void);
/* debug / A-B: which form of the fused stem (conv1 7x7/2 + bn1 + relu + maxpool, torchvision ResNet stem reached from reference src/embeddings.py:118-120)
 * runs: 1 the max pool in registers straight from the MFMA accumulators (stem_pool_reg_kernel, default since round 6), 0 the pooling pass over an LDS tile
 * of rounds 3-5, -1 back to the environment (PVR_STEM_REGPOOL).  Both forms give the same bits.  Process-global. */
pvr_status pvr_debug_set_stem_regpool(int32_t mode);
/* debug / A-B: which implicit-GEMM kernel pvr_op_conv2d and the encoder plans use.  -1 = automatic choice by shape
 * (default), 0 = conv_igemm (128x128 tiles) only, 1 / 2 / 3 = conv_pp256 (ping-pong kernel, 256x256 / 128x256 / 224x256 tiles) whenever it
 * accepts the shape.  All kernels accumulate every output in the same K order and give bit-identical results. Process-global. */
pvr_status pvr_debug_set_conv_algo(int32_t algo);
/* debug: launches of the persistent weight-stationary 1x1 kernel (conv_expand.hip) so far in this process - lets a test assert that the
 * automatic choice really took that kernel for a shape.  PVR_CONV_EXPAND=0 disables the kernel (A/B; bit-identical). */
/* ---- PNG source (reference behavioral_cloning/save_embedded_obs.py:63-64,71-72: cv2.imread of <t>_goal.png and <t>_<s>.png) ----
 * n whole PNG files concatenated in device memory, file i = bytes [offsets[i], offsets[i+1]); every file h x w, bit depth 8,
 * non-interlaced, colour type grey / RGB / grey+alpha / RGBA.  out (n, h, w, 3) uint8 = cv2.imread's B,G,R layout.
 * status[i]: 0 decoded; 1 valid-looking PNG of an unsupported kind and 12 size other than h x w (decode those on the host);
 * anything else = corrupt file (2 signature, 3 truncated, 4 zlib header, 5 block, 6 Huffman code, 7 distance, 8 overrun,
 * 9 short stream, 10 Adler-32, 11 filter type).  scratch: pvr_png_scratch_bytes(n, h, w) bytes of device memory. */
/* host side of the per-file sources: sizes of n files, then their bytes into dst[offsets[i] .. offsets[i+1]) (offsets = prefix sums of
 * the sizes), read by `threads` native threads - one file per frame is a system-call workload (save_embedded_obs.py:71) */
pvr_status pvr_file_sizes(const char *const *paths, int32_t n, int64_t *sizes, int32_t threads);
pvr_status pvr_read_files(const char *const *paths, int32_t n, uint8_t *dst, const int64_t *offsets, int32_t threads);
/* host frames -> a (pinned) staging buffer by `threads` native threads (reference save_embedded_obs.py:148-154 hands torch a pageable slice per
 * batch): dst = rows x row_bytes contiguous bytes; source row r starts at src + r * src_row_stride and consists of runs of run_bytes bytes
 * every run_stride bytes (run_bytes == row_bytes: plain rows; run_bytes 3, run_stride 3F: one 3-channel plane of (H, W, 3F) frames). */
pvr_status pvr_stage_copy(void *dst, const void *src, int64_t rows, int64_t row_bytes, int64_t src_row_stride, int64_t run_bytes,
                          int64_t run_stride, int32_t threads);
int64_t pvr_png_scratch_bytes(int32_t n, int32_t h, int32_t w);
pvr_status pvr_png_decode(const uint8_t *files_dev, const int64_t *offsets_dev, int32_t n, int32_t h, int32_t w, uint8_t *out_dev,
                          uint8_t *scratch_dev, int64_t scratch_bytes, int32_t *status_dev, void *hip_stream);

int64_t pvr_debug_conv_expand_launches(void);
/* launches of conv_pp256's persistent form (tests: the many-tile GEMMs really took it) */
int64_t pvr_debug_pp_persistent_launches(void);
/* global average pool of NHWC (16-bit or fp32) -> fp32 rows at out + i*out_stride */
pvr_status pvr_op_avgpool(const void *in_dev, float *out_dev, int64_t out_stride, int32_t n, int32_t hw,
                          int32_t c, int32_t in_f32, int32_t dtype, void *hip_stream);
/* Finite check of fp32 results on the device: rows x cols values with row stride `stride` (elements); sets *flag_dev (a device int32 the caller
 * zeroed) to 1 if any value is inf or NaN.  Enqueued on `stream`, no synchronisation.  The streaming embedder checks every batch this way and
 * reads the flag once per call (the reference has no such check: its fp32 path cannot overflow a 16-bit storage type). */
pvr_status pvr_op_nonfinite_flag(const float *x_dev, int64_t rows, int64_t cols, int64_t stride, int32_t *flag_dev, void *hip_stream);

/* host-only: the fp32 -> bf16/f16 round-to-nearest-even conversion finalize() applies to weights */
pvr_status pvr_debug_convert(const float *src, uint16_t *dst, int64_t n, int32_t dtype);

/* ---------------------------------------------------------------------------------------------
 * BC policy: PolicyNet / PolicyNetWithConv forward (src/models.py:57-89,159-197) and one training
 * iteration of main_bc_2.py:206-227 (loss, BPTT, grad-norm, clip, RMSprop with LambdaLR).  fp32.
 * Declared in pvr_policy.h.
 * ------------------------------------------------------------------------------------------- */

#ifdef __cplusplus
}
#endif
#endif /* PVR_HIP_H */
