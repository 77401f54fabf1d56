// Host (CPU) backend of the BC policy path behind the pvr_policy_* C-ABI (pvr_policy_create_host): BASELINE configs[0] runs
// main_bc_2.py's policy without a GPU.  host_policy.hip implements it as plain fp32 C++ loops; policy.hip only hands it the layout.
#pragma once
#include "common.h"
#include "../../include/pvr_policy.h"

namespace pvr {

struct HostPolicyLayout {
    int O, H, A, bn;
    int64_t n_total, n_train;
    int64_t o_bnw, o_bnb, o_fc1w, o_fc1b, o_fc2w, o_fc2b, o_wih[2], o_whh[2], o_bih[2], o_bhh[2], o_pw, o_pb, o_bw, o_bb;
};

struct HostPolicy;
HostPolicy *host_policy_new(const HostPolicyLayout &lay);
void host_policy_free(HostPolicy *hp);
void host_policy_set_sampling(HostPolicy *hp, int on, unsigned long long seed);
unsigned long long host_policy_sampling_call(const HostPolicy *hp);
void host_policy_set_sampling_call(HostPolicy *hp, unsigned long long call);
pvr_status host_policy_forward(HostPolicy *hp, const float *params, const pvr_policy_bn *bn, const float *obs, const uint8_t *done, const float *h0,
                               const float *c0, int T, int B, int training, float *logits, float *baseline, int64_t *action, float *h_out, float *c_out);
pvr_status host_policy_step(HostPolicy *hp, float *params, float *square_avg, const pvr_policy_bn *bn, const float *obs, const uint8_t *done,
                            const int64_t *actions, int T, int B, float lr, float alpha, float eps, float max_grad_norm, float *stats_out,
                            float *logits_out);
pvr_status host_policy_last_grads(HostPolicy *hp, float *grads_out);

}  // namespace pvr
