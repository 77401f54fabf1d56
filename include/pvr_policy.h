/*
 * pvr_policy.h — C-ABI of the behavioural-cloning policy path in libpvr_hip.so (fp32).
 *
 * Replaces, for the reference's hot path:
 *   src/models.py:57-89      PolicyNet.forward  ([BatchNorm1d] -> FC+ReLU -> FC+ReLU -> 2-layer LSTM stepped
 *                            one timestep at a time with state *= (1-done) -> policy/baseline heads -> argmax)
 *   main_bc_2.py:206-227     one training iteration: nll_loss(log_softmax) mean, backward (BPTT), sum of squared
 *                            grad norms, clip_grad_norm_(max_norm), RMSprop(momentum=0) with the LambdaLR factor
 *                            already applied by the caller (lr argument)
 * Parameters live in ONE flat fp32 device buffer owned by the caller (PyTorch parameters are views of it);
 * pvr_policy_param_offset gives each tensor's offset under its reference state_dict name
 * ("fc.1.weight", "core.weight_hh_l0", ...).  Trainable tensors come first; the baseline head (which
 * receives no gradient from the BC loss, so torch leaves its .grad None) is last and is excluded from
 * the grad norm and the update, exactly as in the reference.
 */
#ifndef PVR_POLICY_H
#define PVR_POLICY_H

#include "pvr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvr_policy pvr_policy;

typedef struct pvr_policy_desc {
    int32_t obs_size;      /* observation_shape[0] (models.py:22) */
    int32_t hidden;        /* 1024 in the reference; multiple of 1024 */
    int32_t num_actions;   /* <= 16 */
    int32_t batch_norm;    /* BatchNorm1d in front of the MLP (models.py:30-34) */
    int32_t max_t;         /* unroll_length the workspace is sized for */
    int32_t max_b;         /* batch_size the workspace is sized for (<= 64) */
    int32_t conv_frames;   /* 0: vector observations (PolicyNet).  n > 0: PolicyNetWithConv (models.py:96-197) on raw uint8
                              (T,B,64,64,3n) observations; obs_size must be 128*n and `obs` arguments are uint8 */
} pvr_policy_desc;

pvr_status pvr_policy_create(const pvr_policy_desc *desc, pvr_policy **out);
/* The same handle on the HOST (CPU) backend - BASELINE configs[0] runs main_bc_2.py's policy without a GPU, and the reference's model
 * lives wherever flags.device says (main_bc_2.py:64-66).  No HIP call is made; params / square_avg / BatchNorm buffers / obs / done /
 * actions / outputs of pvr_policy_forward and pvr_policy_step are HOST pointers, hip_stream is ignored, and the arithmetic is plain fp32
 * C++ on this process's threads (PVR_HOST_THREADS).  Vector observations only (conv_frames must be 0); the split entry points
 * (backward / apply / backward_dlogits / data parallel) belong to the HIP plan and are refused. */
pvr_status pvr_policy_create_host(const pvr_policy_desc *desc, pvr_policy **out);
void pvr_policy_destroy(pvr_policy *pol);

/* number of fp32 elements of the flat parameter buffer, and of its trainable prefix */
int64_t pvr_policy_param_count(const pvr_policy *pol);
int64_t pvr_policy_trainable_count(const pvr_policy *pol);
/* offset (elements) and size of a tensor by reference state_dict name; returns -1 if unknown */
int64_t pvr_policy_param_offset(const pvr_policy *pol, const char *name, int64_t *numel);

/* BatchNorm1d buffers (device): running_mean, running_var (fp32, obs_size) and num_batches_tracked (int64[1]);
 * ignored when batch_norm == 0 */
typedef struct pvr_policy_bn {
    float *running_mean;
    float *running_var;
    int64_t *num_batches_tracked;
} pvr_policy_bn;

/* Data-parallel finetune (BASELINE config 4; the reference's main_bc_finetune.py:167-208 is single-GPU and src/models.py:132-136
 * is a plain BatchNorm1d).  The caller hands the library ONE collective as a C function pointer:
 *     fn(buf, count, hip_stream, user) = in-place SUM all-reduce over the ranks of `count` fp32 values at device address `buf`,
 *     enqueued on hip_stream (RCCL over xGMI: ncclAllReduce, or torch.distributed.all_reduce under backend "nccl"); returns 0 on
 *     success, non-zero makes the calling entry point fail with PVR_ERR_COMM.
 * With world_size > 1 installed, pvr_policy_backward / pvr_policy_step on every rank
 *   - all-reduce the gradient in four buckets, in the order backward finalises them (LSTM layer 1 + policy head, LSTM layer 0,
 *     the two fc layers, conv stack + BatchNorm affine), each on a library-owned communication stream that waits only for the
 *     event recorded after the bucket's last kernel, so the transfers overlap the rest of the backward pass; the buckets (and
 *     the loss) are divided by world_size there and the compute stream re-joins before the entry point returns: `grads` then
 *     holds the gradient of the GLOBAL batch mean and stats_out[0] the global loss;
 *   - with sync_bn != 0, use BatchNorm statistics of the global batch (world_size x T x B rows, equal rows per rank): per-rank
 *     column sums are all-reduced on the compute stream between the statistic kernels (mean, centred second moment, and - for
 *     PolicyNetWithConv - the two backward sums), so N ranks x B/N sequences equal one rank x B.
 * world_size <= 1 or fn == NULL restores single-rank behaviour. */
typedef int32_t (*pvr_allreduce_fn)(void *buf_dev, int64_t count, void *hip_stream, void *user);
#define PVR_ERR_COMM 5
#define PVR_ERR_TIMEOUT 6
pvr_status pvr_policy_set_data_parallel(pvr_policy *pol, int32_t world_size, int32_t sync_bn, pvr_allreduce_fn fn, void *user);

/* Health of the handle's persistent launches.  The forward recurrence (models.py:66-73) may run as ONE launch per layer whose
 * blocks hand h_t to each other inside the kernel; it is only chosen when the whole grid fits the GPU at once (occupancy x CU count,
 * checked at create), its waits are bounded, and a wait that runs out stores into a pinned status word (and poisons that call's
 * outputs with NaN).  pvr_policy_status returns PVR_ERR_TIMEOUT once for such an event (message via pvr_last_error) and the handle
 * falls back to per-step launches; pvr_policy_forward / _backward / _backward_dlogits / _step perform the same check on entry, so a
 * time-out never goes unnoticed.  It does not synchronise: call it after the stream has been synchronised to learn about the
 * launches just enqueued.  (The reference has no counterpart: nn.LSTM cannot time out.)
 * pvr_policy_recurrence_mode: 0 per-step launches, 1 / 2 persistent (counter / data-as-flag hand-off) - what the next forward will use. */
pvr_status pvr_policy_status(pvr_policy *pol);
int32_t pvr_policy_recurrence_mode(const pvr_policy *pol);
/* test hook: block `block` (>= 0) of every following persistent launch exits at once, so its peers' waits run out and the failure
 * path above can be exercised on a healthy GPU (tests/test_gpu_policy.py); -1 restores normal launches */
pvr_status pvr_policy_debug_drop_block(pvr_policy *pol, int32_t block);

/* PolicyNet.forward (models.py:57-89).  obs (T,B,obs_size) fp32 (uint8 (T,B,64,64,3n) when conv_frames = n > 0), done (T,B) uint8, h0/c0 (2,B,hidden) fp32 are
 * device inputs; logits (T,B,A), baseline (T,B), action (T,B) int64 = argmax (eval branch, :82; a sample in training mode after pvr_policy_set_action_sampling), h_out/c_out
 * (2,B,hidden) are device outputs.  training != 0 uses batch statistics and updates the BN buffers (:31-34). */
/* Training-mode action of PolicyNet.forward (models.py:78-80: torch.multinomial(F.softmax(policy_logits, dim=1), num_samples=1)).  With
 * on != 0 every following training-mode pvr_policy_forward writes ONE SAMPLE of softmax(logits) per row into `action` instead of the
 * argmax, drawn inside the heads kernel (Gumbel-max over a counter-based Philox-4x32-10 stream keyed by `seed` and counted by call and
 * row: reproducible for a seed, independent of the launch geometry; torch's own generator stream is not reproduced - the contract is the
 * distribution).  Calling it again restarts the stream.  Eval-mode forwards and pvr_policy_step are unaffected.  Default off. */
pvr_status pvr_policy_set_action_sampling(pvr_policy *pol, int32_t on, uint64_t seed);
/* Position of that stream = the number of sampling forwards since pvr_policy_set_action_sampling; a caller that replaces a handle (larger
 * T / B, another device) reads it from the old handle and sets it on the new one, so the noise stream continues instead of replaying
 * (the reference's torch.multinomial consumes the global generator, which no module re-arms: src/models.py:78-80). */
uint64_t pvr_policy_action_sampling_call(const pvr_policy *pol);
pvr_status pvr_policy_set_action_sampling_call(pvr_policy *pol, uint64_t call);
pvr_status pvr_policy_forward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs,
                              const uint8_t *done, const float *h0, const float *c0, int32_t T, int32_t B,
                              int32_t training, float *logits, float *baseline, int64_t *action, float *h_out,
                              float *c_out, void *hip_stream);

/* One iteration of main_bc_2.py:206-227 from a zero initial state: forward (training), loss, backward,
 * grad-norm, clip, RMSprop.  params / square_avg: flat device buffers (updated in place).  actions (T,B) int64.
 * lr = learning_rate * LambdaLR factor for this update.  stats_out (device, 2 floats): loss, grad norm (pre-clip).
 * logits_out (optional, may be NULL): (T,B,A) training-mode logits. */
pvr_status pvr_policy_step(pvr_policy *pol, float *params, float *square_avg, const pvr_policy_bn *bn,
                           const void *obs, const uint8_t *done, const int64_t *actions, int32_t T, int32_t B,
                           float lr, float alpha, float eps, float max_grad_norm, float *stats_out,
                           float *logits_out, void *hip_stream);

/* The same iteration in two halves (data-parallel training, autograd bridge).  pvr_policy_backward leaves the UNCLIPPED
 * gradient of the mean loss in grads (device, trainable_count floats, caller-owned) and the loss in stats_out[0] - the local
 * batch's, or the global batch's when pvr_policy_set_data_parallel is installed; pvr_policy_apply computes the norm of whatever
 * grads holds, clips and applies RMSprop; stats_out[1] = that norm. */
pvr_status pvr_policy_backward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs,
                               const uint8_t *done, const int64_t *actions, int32_t T, int32_t B, float *grads,
                               float *stats_out, float *logits_out, void *hip_stream);
pvr_status pvr_policy_apply(pvr_policy *pol, float *params, float *square_avg, const float *grads, float lr,
                            float alpha, float eps, float max_grad_norm, float *stats_out, void *hip_stream);

/* Autograd bridge: the backward half for a caller that computes the loss itself (the reference's own lines, main_bc_2.py:211-220:
 * F.nll_loss(F.log_softmax(logits)) ... loss.backward()).  Call after a training-mode pvr_policy_forward of the same (obs, T, B):
 * dlogits (device, (T,B,A) contiguous) is d(loss)/d(policy_logits); leaves d(loss)/d(params) in grads (trainable_count floats) -
 * all-reduced over the ranks when pvr_policy_set_data_parallel is installed.  The baseline head receives no gradient (the BC loss
 * does not read it; torch leaves its .grad None too).  One backward per forward. */
pvr_status pvr_policy_backward_dlogits(pvr_policy *pol, const float *params, const void *obs, const float *dlogits, int32_t T,
                                       int32_t B, float *grads, void *hip_stream);

/* Other update rules on the same flat buffers (grad norm + clip as in pvr_policy_apply; stats_out[1] = pre-clip norm):
 * torch.optim.RMSprop with momentum != 0 (src/arguments.py:63 exposes --momentum; the reference default 0 is pvr_policy_apply) and
 * torch.optim.Adam(betas, eps; amsgrad off, no weight decay) - an extension, BASELINE.json's north_star names Adam.  step = 1, 2, ...
 * is the update count (bias correction). */
pvr_status pvr_policy_apply_momentum(pvr_policy *pol, float *params, float *square_avg, float *momentum_buf, const float *grads,
                                     float lr, float alpha, float eps, float momentum, float max_grad_norm, float *stats_out,
                                     void *hip_stream);
pvr_status pvr_policy_apply_adam(pvr_policy *pol, float *params, float *exp_avg, float *exp_avg_sq, const float *grads, float lr,
                                 float beta1, float beta2, float eps, int64_t step, float max_grad_norm, float *stats_out,
                                 void *hip_stream);

/* BC batch assembly on the device: replaces the host gather of main_bc_2.py:186-204 (identical in main_bc_1.py:193-211 and
 * main_bc_finetune.py:173-188).  The dataset stays resident in HBM: obs_dev (n_samples rows of row_bytes bytes: fp32 embeddings, or
 * raw uint8 frames for the finetune model), action_dev (int64), done_dev (uint8).  For the B start indices of
 * sample_with_minimum_distance (starts_dev, int64) row (t, b) of every output is dataset row (starts[b] + t) mod n_samples, i.e.
 * out_obs is the (T, B, ...) batch np.stack(..., axis=1) builds.  action / done may both be NULL. */
pvr_status pvr_bc_gather(const void *obs_dev, const int64_t *action_dev, const uint8_t *done_dev, int64_t n_samples,
                         int64_t row_bytes, const int64_t *starts_dev, int32_t T, int32_t B, void *out_obs,
                         int64_t *out_action, uint8_t *out_done, void *hip_stream);

/* parity/debug: copy the flat gradient of the last pvr_policy_step (pre-clip) to grads_out (device, trainable_count) */
pvr_status pvr_policy_last_grads(pvr_policy *pol, float *grads_out, void *hip_stream);

/* fp32-in / fp32-out GEMM as the policy runs it: C[M,N] = op(A) op(B)^T-style contraction over K.
 * a_km != 0: A stored [K][M] else [M][K];  b_kn != 0: B stored [K][N] else [N][K].  Optional bias[N], relu.
 * (unit-parity entry point; the arithmetic is the one pvr_debug_set_gemm_mode selects) */
pvr_status pvr_op_gemm_f32(const float *A, const float *B, const float *bias, float *C, int32_t M, int32_t N,
                           int32_t K, int32_t a_km, int32_t b_kn, int32_t relu, void *hip_stream);
/* How every fp32 GEMM of the policy is computed (process-wide; the reference's counterpart is torch's fp32 matmul, src/models.py:66-73):
 *   0  fp32 MFMA (v_mfma_f32_16x16x4_f32: the ascending-k fma chain; the default, rounds 1-3)
 *   1  bf16 MFMA on an exact three-term split of both operands, six products per pair, fp32 accumulation (round 3, opt-in: 2.7 x closer
 *      to an fp64 product than mode 0 and about as fast); 2 / 3: the same with 128 x 128 / 64 x 64 tiles forced
 *  -1  back to the default (environment PVR_GEMM_BF16X3, else 0) */
pvr_status pvr_debug_set_gemm_mode(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* PVR_POLICY_H */
