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

/* SyncBN for the data-parallel finetune (BASELINE config 4; the reference itself is single-GPU, src/models.py:132-136 is a plain
 * BatchNorm1d): with a callback installed and world_size > 1, training-mode BatchNorm uses the statistics of the GLOBAL batch
 * (world_size x T x B rows, equal rows per rank), which makes N ranks x B/N sequences equal to one rank x B.  The library writes
 * per-rank column sums into sync_buf (device, 2 * obs_size floats, caller-owned) and calls fn(offset, count, user) from inside
 * pvr_policy_backward / _step / _forward(training); fn must enqueue, on the stream of that call, an in-place SUM all-reduce of
 * sync_buf[offset, offset+count) over the ranks (torch.distributed.all_reduce on the current stream: RCCL over xGMI).  Three
 * calls per iteration: mean (obs_size floats), centred second moment (obs_size), backward sums (2 * obs_size).  fn = NULL
 * restores per-rank statistics. */
typedef void (*pvr_policy_sync_fn)(int64_t offset, int64_t count, void *user);
pvr_status pvr_policy_set_bn_sync(pvr_policy *pol, float *sync_buf, int32_t world_size, pvr_policy_sync_fn fn, void *user);

/* PolicyNet.forward (models.py:57-89).  obs (T,B,obs_size) fp32 (uint8 (T,B,64,64,3n) when conv_frames = n > 0), done (T,B) uint8, h0/c0 (2,B,hidden) fp32 are
 * device inputs; logits (T,B,A), baseline (T,B), action (T,B) int64 = argmax (eval branch, :82), h_out/c_out
 * (2,B,hidden) are device outputs.  training != 0 uses batch statistics and updates the BN buffers (:31-34). */
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

/* Data-parallel training (finetune configuration): the same iteration in two halves so the caller can
 * all-reduce the flat gradient (RCCL / torch.distributed) between them.  pvr_policy_backward leaves the UNCLIPPED
 * local gradient of the mean loss in grads (device, trainable_count floats, caller-owned) and the loss in
 * stats_out[0]; pvr_policy_apply computes the norm of whatever grads now holds (e.g. the rank average), clips and
 * applies RMSprop; stats_out[1] = that norm. */
pvr_status pvr_policy_backward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs,
                               const uint8_t *done, const int64_t *actions, int32_t T, int32_t B, float *grads,
                               float *stats_out, float *logits_out, void *hip_stream);
pvr_status pvr_policy_apply(pvr_policy *pol, float *params, float *square_avg, const float *grads, float lr,
                            float alpha, float eps, float max_grad_norm, float *stats_out, void *hip_stream);

/* parity/debug: copy the flat gradient of the last pvr_policy_step (pre-clip) to grads_out (device, trainable_count) */
pvr_status pvr_policy_last_grads(pvr_policy *pol, float *grads_out, void *hip_stream);

/* fp32 GEMM on the f32 MFMA path used by the policy: C[M,N] = op(A) op(B)^T-style contraction over K.
 * a_km != 0: A stored [K][M] else [M][K];  b_kn != 0: B stored [K][N] else [N][K].  Optional bias[N], relu.
 * (unit-parity entry point) */
pvr_status pvr_op_gemm_f32(const float *A, const float *B, const float *bias, float *C, int32_t M, int32_t N,
                           int32_t K, int32_t a_km, int32_t b_kn, int32_t relu, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PVR_POLICY_H */
