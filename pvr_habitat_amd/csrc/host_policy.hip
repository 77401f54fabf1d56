// Host (CPU) backend of the behavioural-cloning policy behind the pvr_policy_* C-ABI (pvr_policy_create_host): BASELINE configs[0] runs
// main_bc_2.py's policy "on 1 Replica scene (plumbing, no GPU)".  Plain fp32 C++ loops on this process's threads - the product's slow
// path for a box without a GPU, never the test oracle (oracle/policy_oracle.py is torch autograd) and never a fallback of the HIP path.
//
//   forward   src/models.py:57-89: [BatchNorm1d (batch statistics + running-stat update when training)] -> Linear + ReLU -> Linear + ReLU
//             -> 2-layer LSTM stepped one timestep at a time with the state multiplied by (1 - done) (:66-72; gate order i, f, g, o)
//             -> policy / baseline heads -> argmax (:82)
//   step      main_bc_2.py:206-227 from a zero state: mean nll_loss(log_softmax) -> backward (BPTT; the baseline head gets no gradient)
//             -> sum of squared gradient norms -> clip_grad_norm_(max_norm) -> RMSprop(momentum 0) with the caller's lr
// Vector observations only (PolicyNet; the conv front end of PolicyNetWithConv is a GPU-plan feature).
#include <cmath>
#include "host_policy.h"
#include "host_math.h"
#include "sample_rng.h"

namespace pvr {

struct HostPolicy {
    HostPolicyLayout L;
    std::vector<float> xhat, invstd, a0, a1, a2, Gx[2], Hs[2], Cs[2], HM[2], dG[2], dX, dA, dB, grads, logits, dlogits, nd;
    bool have_grads = false;
    bool sample_on = false;                    // pvr_policy_set_action_sampling
    unsigned long long sample_seed = 0, sample_call = 0;
};

HostPolicy *host_policy_new(const HostPolicyLayout &lay) {
    HostPolicy *hp = new HostPolicy();
    hp->L = lay;
    hp->grads.assign((size_t)lay.n_train, 0.f);
    return hp;
}
void host_policy_free(HostPolicy *hp) { delete hp; }

static inline float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// the recurrence of one layer over T steps: pre-activations Gx[t] (input projection + both biases, hoisted) + hm . W_hh^T -> gates (stored
// ACTIVATED in Gx), c, h; HM[t] = nd[t] * h[t-1] is kept for the weight gradient
static void lstm_layer(HostPolicy *hp, int l, const float *P, const float *h0, const float *c0, int T, int B, float *h_last, float *c_last) {
    const int H = hp->L.H;
    const float *Whh = P + hp->L.o_whh[l];
    std::vector<float> hm((size_t)B * H), cm((size_t)B * H), rec((size_t)B * 4 * H);
    for (int t = 0; t < T; ++t) {
        float *G = hp->Gx[l].data() + (size_t)t * B * 4 * H;
        for (int b = 0; b < B; ++b) {
            const float ndv = hp->nd[(size_t)t * B + b];
            const float *hp_ = t == 0 ? (h0 ? h0 + (size_t)b * H : nullptr) : hp->Hs[l].data() + ((size_t)(t - 1) * B + b) * H;
            const float *cp_ = t == 0 ? (c0 ? c0 + (size_t)b * H : nullptr) : hp->Cs[l].data() + ((size_t)(t - 1) * B + b) * H;
            for (int u = 0; u < H; ++u) {
                hm[(size_t)b * H + u] = hp_ ? ndv * hp_[u] : 0.f;
                cm[(size_t)b * H + u] = cp_ ? ndv * cp_[u] : 0.f;
            }
        }
        memcpy(hp->HM[l].data() + (size_t)t * B * H, hm.data(), (size_t)B * H * 4);
        host_gemm_nt(hm.data(), Whh, nullptr, rec.data(), B, 4 * H, H, false);
        float *Ht = hp->Hs[l].data() + (size_t)t * B * H, *Ct = hp->Cs[l].data() + (size_t)t * B * H;
        for (int b = 0; b < B; ++b) {
            float *g = G + (size_t)b * 4 * H;
            const float *r = rec.data() + (size_t)b * 4 * H;
            for (int u = 0; u < H; ++u) {
                const float i = sigm(g[u] + r[u]), f = sigm(g[H + u] + r[H + u]), gg = tanhf(g[2 * H + u] + r[2 * H + u]), o = sigm(g[3 * H + u] + r[3 * H + u]);
                const float c = f * cm[(size_t)b * H + u] + i * gg;
                g[u] = i; g[H + u] = f; g[2 * H + u] = gg; g[3 * H + u] = o;
                Ct[(size_t)b * H + u] = c;
                Ht[(size_t)b * H + u] = o * tanhf(c);
            }
        }
    }
    if (h_last) memcpy(h_last, hp->Hs[l].data() + (size_t)(T - 1) * B * H, (size_t)B * H * 4);
    if (c_last) memcpy(c_last, hp->Cs[l].data() + (size_t)(T - 1) * B * H, (size_t)B * H * 4);
}

static pvr_status forward_impl(HostPolicy *hp, const float *P, const pvr_policy_bn *bn, const float *obs, const uint8_t *done, const float *h0,
                               const float *c0, int T, int B, int training, float *logits, float *baseline, int64_t *action, float *h_out,
                               float *c_out) {
    const HostPolicyLayout &L = hp->L;
    const int N = T * B, O = L.O, H = L.H, A = L.A;
    PVR_REQUIRE(P && obs && done && T > 0 && B > 0, "policy (host): null argument");
    PVR_REQUIRE(!L.bn || (bn && bn->running_mean && bn->running_var), "policy (host): batch_norm needs its buffers");
    // torch: "Expected more than 1 value per channel when training" (F.batch_norm on an (N = 1, C) input): the unbiased variance does not exist
    PVR_REQUIRE(!(L.bn && training && N == 1), "policy (host): BatchNorm in training mode needs more than one row (T * B = 1), as torch.nn.BatchNorm1d does");
    hp->nd.resize(N);
    for (int n = 0; n < N; ++n) hp->nd[n] = done[n] ? 0.f : 1.f;
    hp->a0.resize((size_t)N * O);
    if (L.bn) {
        hp->xhat.resize((size_t)N * O); hp->invstd.resize(O);
        const float *gam = P + L.o_bnw, *bet = P + L.o_bnb;
        host_parallel_for(O, host_threads(), [&](int k) {
            float mean, var;
            if (training) {
                double s = 0.0;
                for (int n = 0; n < N; ++n) s += obs[(size_t)n * O + k];
                mean = (float)(s / N);
                double q = 0.0;
                for (int n = 0; n < N; ++n) { const double d = obs[(size_t)n * O + k] - mean; q += d * d; }
                var = (float)(q / N);
                bn->running_mean[k] = 0.9f * bn->running_mean[k] + 0.1f * mean;
                bn->running_var[k] = 0.9f * bn->running_var[k] + 0.1f * (N > 1 ? (float)(q / (N - 1)) : var);
            } else { mean = bn->running_mean[k]; var = bn->running_var[k]; }
            const float is = 1.f / sqrtf(var + 1e-5f);
            hp->invstd[k] = is;
            for (int n = 0; n < N; ++n) {
                const float xh = (obs[(size_t)n * O + k] - mean) * is;
                hp->xhat[(size_t)n * O + k] = xh;
                hp->a0[(size_t)n * O + k] = xh * gam[k] + bet[k];
            }
        });
        if (training && bn->num_batches_tracked) *bn->num_batches_tracked += 1;
    } else memcpy(hp->a0.data(), obs, (size_t)N * O * 4);
    hp->a1.resize((size_t)N * H); hp->a2.resize((size_t)N * H);
    host_gemm_nt(hp->a0.data(), P + L.o_fc1w, P + L.o_fc1b, hp->a1.data(), N, H, O, true);
    host_gemm_nt(hp->a1.data(), P + L.o_fc2w, P + L.o_fc2b, hp->a2.data(), N, H, H, true);
    std::vector<float> bsum((size_t)4 * H);
    for (int l = 0; l < 2; ++l) {
        hp->Gx[l].resize((size_t)N * 4 * H); hp->Hs[l].resize((size_t)N * H); hp->Cs[l].resize((size_t)N * H); hp->HM[l].resize((size_t)N * H);
        for (int k = 0; k < 4 * H; ++k) bsum[k] = P[L.o_bih[l] + k] + P[L.o_bhh[l] + k];
        host_gemm_nt(l == 0 ? hp->a2.data() : hp->Hs[0].data(), P + L.o_wih[l], bsum.data(), hp->Gx[l].data(), N, 4 * H, H, false);
        lstm_layer(hp, l, P, h0 ? h0 + (size_t)l * B * H : nullptr, c0 ? c0 + (size_t)l * B * H : nullptr, T, B,
                   h_out ? h_out + (size_t)l * B * H : nullptr, c_out ? c_out + (size_t)l * B * H : nullptr);
    }
    hp->logits.resize((size_t)N * A);
    host_gemm_nt(hp->Hs[1].data(), P + L.o_pw, P + L.o_pb, hp->logits.data(), N, A, H, false);
    if (logits) memcpy(logits, hp->logits.data(), (size_t)N * A * 4);
    if (baseline) host_gemm_nt(hp->Hs[1].data(), P + L.o_bw, P + L.o_bb, baseline, N, 1, H, false);
    if (action && training && hp->sample_on) {             // models.py:78-80: one sample of softmax(logits) per row
        const unsigned long long call = hp->sample_call++;
        for (int n = 0; n < N; ++n) action[n] = sample_softmax_row(hp->logits.data() + (size_t)n * A, A, hp->sample_seed, call, (unsigned long long)n);
    } else if (action)
        for (int n = 0; n < N; ++n) {
            int best = 0;
            for (int a = 1; a < A; ++a) if (hp->logits[(size_t)n * A + a] > hp->logits[(size_t)n * A + best]) best = a;   // first max on ties (torch.argmax)
            action[n] = best;
        }
    return PVR_OK;
}

void host_policy_set_sampling(HostPolicy *hp, int on, unsigned long long seed) { hp->sample_on = on != 0; hp->sample_seed = seed; hp->sample_call = 0; }
unsigned long long host_policy_sampling_call(const HostPolicy *hp) { return hp->sample_call; }
void host_policy_set_sampling_call(HostPolicy *hp, unsigned long long call) { hp->sample_call = call; }

pvr_status host_policy_forward(HostPolicy *hp, const float *params, const pvr_policy_bn *bn, const float *obs, const uint8_t *done, const float *h0,
                               const float *c0, int T, int B, int training, float *logits, float *baseline, int64_t *action, float *h_out, float *c_out) {
    return forward_impl(hp, params, bn, obs, done, h0, c0, T, B, training, logits, baseline, action, h_out, c_out);
}

// BPTT of layer l: dHext[t] = gradient flowing into h[t] from above (the heads / the next layer's input); fills dG[l] (pre-activation gate
// gradients of every step), accumulating the recurrent path dh[t-1] += nd[t] * dG[t] . W_hh and the cell path on the way
static void lstm_layer_bwd(HostPolicy *hp, int l, const float *P, const float *dHext, int T, int B) {
    const int H = hp->L.H;
    const float *Whh = P + hp->L.o_whh[l];
    hp->dG[l].resize((size_t)T * B * 4 * H);
    std::vector<float> dh_carry((size_t)B * H, 0.f), dc_carry((size_t)B * H, 0.f), dhm((size_t)B * H);
    for (int t = T - 1; t >= 0; --t) {
        const float *G = hp->Gx[l].data() + (size_t)t * B * 4 * H, *Ct = hp->Cs[l].data() + (size_t)t * B * H;
        float *dGt = hp->dG[l].data() + (size_t)t * B * 4 * H;
        for (int b = 0; b < B; ++b) {
            const float ndv = hp->nd[(size_t)t * B + b];
            const float *cp_ = t == 0 ? nullptr : hp->Cs[l].data() + ((size_t)(t - 1) * B + b) * H;
            for (int u = 0; u < H; ++u) {
                const size_t bu = (size_t)b * H + u;
                const float i = G[(size_t)b * 4 * H + u], f = G[(size_t)b * 4 * H + H + u], g = G[(size_t)b * 4 * H + 2 * H + u], o = G[(size_t)b * 4 * H + 3 * H + u];
                const float tc = tanhf(Ct[bu]), dh = dHext[(size_t)t * B * H + bu] + dh_carry[bu];
                const float dc = dc_carry[bu] + dh * o * (1.f - tc * tc);
                const float cm = cp_ ? ndv * cp_[u] : 0.f;
                dGt[(size_t)b * 4 * H + u] = dc * g * i * (1.f - i);
                dGt[(size_t)b * 4 * H + H + u] = dc * cm * f * (1.f - f);
                dGt[(size_t)b * 4 * H + 2 * H + u] = dc * i * (1.f - g * g);
                dGt[(size_t)b * 4 * H + 3 * H + u] = dh * tc * o * (1.f - o);
                dc_carry[bu] = ndv * dc * f;                          // into c[t-1] through cm = nd * c[t-1]
            }
        }
        if (t > 0) {
            host_gemm_nn(dGt, Whh, dhm.data(), B, 4 * H, H, false);   // d(hm) = dG . W_hh
            for (int b = 0; b < B; ++b) {
                const float ndv = hp->nd[(size_t)t * B + b];
                for (int u = 0; u < H; ++u) dh_carry[(size_t)b * H + u] = ndv * dhm[(size_t)b * H + u];
            }
        }
    }
}

pvr_status host_policy_step(HostPolicy *hp, float *P, float *sq, const pvr_policy_bn *bn, const float *obs, const uint8_t *done,
                            const int64_t *actions, int T, int B, float lr, float alpha, float eps, float max_grad_norm, float *stats_out,
                            float *logits_out) {
    const HostPolicyLayout &L = hp->L;
    const int N = T * B, O = L.O, H = L.H, A = L.A;
    PVR_REQUIRE(P && sq && actions && stats_out, "policy step (host): null argument");
    // reject the batch BEFORE the forward touches any state (the BatchNorm running statistics are updated in there), as the HIP plan and torch do
    for (int n = 0; n < N; ++n)
        PVR_REQUIRE(actions[n] >= 0 && actions[n] < A, "policy step (host): action %lld outside 0..%d", (long long)actions[n], A - 1);
    pvr_status s = forward_impl(hp, P, bn, obs, done, nullptr, nullptr, T, B, 1, logits_out, nullptr, nullptr, nullptr, nullptr);
    if (s) return s;
    // loss = mean over rows of -log_softmax(logits)[target]; dlogits = (softmax - onehot) / N
    hp->dlogits.resize((size_t)N * A);
    double loss = 0.0;
    for (int n = 0; n < N; ++n) {
        const float *lg = hp->logits.data() + (size_t)n * A;
        const int64_t tg = actions[n];
        float mx = lg[0];
        for (int a = 1; a < A; ++a) mx = lg[a] > mx ? lg[a] : mx;
        double se = 0.0;
        for (int a = 0; a < A; ++a) se += exp((double)(lg[a] - mx));
        const double lse = (double)mx + log(se);
        loss += lse - (double)lg[tg];
        for (int a = 0; a < A; ++a) hp->dlogits[(size_t)n * A + a] = (float)((exp((double)lg[a] - lse) - (a == tg ? 1.0 : 0.0)) / N);
    }
    float *g = hp->grads.data();
    std::fill(hp->grads.begin(), hp->grads.end(), 0.f);
    // heads (the baseline head receives no gradient from the BC loss)
    host_gemm_tn(hp->dlogits.data(), hp->Hs[1].data(), g + L.o_pw, N, A, H);
    for (int a = 0; a < A; ++a) { double c = 0.0; for (int n = 0; n < N; ++n) c += hp->dlogits[(size_t)n * A + a]; g[L.o_pb + a] = (float)c; }
    hp->dA.resize((size_t)N * H); hp->dB.resize((size_t)N * H);
    host_gemm_nn(hp->dlogits.data(), P + L.o_pw, hp->dA.data(), N, A, H, false);            // dH of layer 1
    // LSTM, layer 1 then layer 0
    for (int l = 1; l >= 0; --l) {
        lstm_layer_bwd(hp, l, P, hp->dA.data(), T, B);
        const float *X = l == 0 ? hp->a2.data() : hp->Hs[0].data();
        host_gemm_tn(hp->dG[l].data(), X, g + L.o_wih[l], N, 4 * H, H);
        host_gemm_tn(hp->dG[l].data(), hp->HM[l].data(), g + L.o_whh[l], N, 4 * H, H);
        for (int k = 0; k < 4 * H; ++k) {
            double c = 0.0;
            for (int n = 0; n < N; ++n) c += hp->dG[l][(size_t)n * 4 * H + k];
            g[L.o_bih[l] + k] = (float)c; g[L.o_bhh[l] + k] = (float)c;
        }
        host_gemm_nn(hp->dG[l].data(), P + L.o_wih[l], hp->dA.data(), N, 4 * H, H, false);   // gradient of the layer's input: dH of layer 0 / d a2
    }
    // fc2, fc1 (ReLU masks from the stored activations), BatchNorm affine
    for (size_t i = 0; i < (size_t)N * H; ++i) hp->dA[i] = hp->a2[i] > 0.f ? hp->dA[i] : 0.f;
    host_gemm_tn(hp->dA.data(), hp->a1.data(), g + L.o_fc2w, N, H, H);
    for (int k = 0; k < H; ++k) { double c = 0.0; for (int n = 0; n < N; ++n) c += hp->dA[(size_t)n * H + k]; g[L.o_fc2b + k] = (float)c; }
    host_gemm_nn(hp->dA.data(), P + L.o_fc2w, hp->dB.data(), N, H, H, false);
    for (size_t i = 0; i < (size_t)N * H; ++i) hp->dB[i] = hp->a1[i] > 0.f ? hp->dB[i] : 0.f;
    host_gemm_tn(hp->dB.data(), hp->a0.data(), g + L.o_fc1w, N, H, O);
    for (int k = 0; k < H; ++k) { double c = 0.0; for (int n = 0; n < N; ++n) c += hp->dB[(size_t)n * H + k]; g[L.o_fc1b + k] = (float)c; }
    if (L.bn) {
        hp->dX.resize((size_t)N * O);
        host_gemm_nn(hp->dB.data(), P + L.o_fc1w, hp->dX.data(), N, H, O, false);
        host_parallel_for(O, host_threads(), [&](int k) {
            double dgam = 0.0, dbet = 0.0;
            for (int n = 0; n < N; ++n) { dgam += (double)hp->dX[(size_t)n * O + k] * hp->xhat[(size_t)n * O + k]; dbet += hp->dX[(size_t)n * O + k]; }
            g[L.o_bnw + k] = (float)dgam; g[L.o_bnb + k] = (float)dbet;
        });
    }
    hp->have_grads = true;
    // sum of squared norms (main_bc_2.py:220-224), clip_grad_norm_ (:226), RMSprop (:227; momentum 0, not centred)
    double n2 = 0.0;
    for (int64_t i = 0; i < L.n_train; ++i) n2 += (double)g[i] * g[i];
    const float norm = (float)sqrt(n2);
    stats_out[0] = (float)(loss / N); stats_out[1] = norm;
    const float coef = max_grad_norm / (norm + 1e-6f) < 1.f ? max_grad_norm / (norm + 1e-6f) : 1.f;
    host_parallel_for(64, host_threads(), [&](int part) {
        const int64_t lo = L.n_train * part / 64, hi = L.n_train * (part + 1) / 64;
        for (int64_t i = lo; i < hi; ++i) {
            const float gi = g[i] * coef;
            sq[i] = alpha * sq[i] + (1.f - alpha) * gi * gi;
            P[i] -= lr * gi / (sqrtf(sq[i]) + eps);
        }
    });
    return PVR_OK;
}

pvr_status host_policy_last_grads(HostPolicy *hp, float *grads_out) {
    PVR_REQUIRE(hp->have_grads && grads_out, "pvr_policy_last_grads (host): no training step has run");
    memcpy(grads_out, hp->grads.data(), (size_t)hp->L.n_train * 4);
    return PVR_OK;
}

}  // namespace pvr
