// BC policy plan (fp32): PolicyNet forward and one fused training iteration as a static sequence of launches
// over a pre-allocated workspace.  C-ABI in include/pvr_policy.h.
//
// Replaces reference src/models.py:57-89 (forward) and main_bc_2.py:206-227 (loss / backward / clip / RMSprop).
// The reference steps nn.LSTM one timestep at a time from Python (models.py:66-73: 100 x cuDNN calls with M = 16);
// here the input projections of all T steps are hoisted into one GEMM per layer and only the recurrent product
// stays sequential (one launch per layer-step, weights served from L2 / Infinity Cache: 16.8 MB per layer).
#include "host_policy.h"
#include <map>
#include <string>
#include <vector>
#include <mutex>
#include <atomic>
#include "policy_kernels.h"
#include "policy_conv.h"
#include "../../include/pvr_policy.h"

using namespace pvr;

namespace {
struct Slot { int64_t off, numel; };
}

// most waves a conv weight-gradient launch is cut over (its partial buffer: 4096 x [32][9*32] floats = 151 MB, PolicyNetWithConv only)
constexpr int CONV_WG_WAVES_MAX = 4096;
constexpr int CONV_BG_BLOCKS_MAX = 2048;      // row blocks of a conv bias-gradient column sum

struct pvr_policy {
    pvr::HostPolicy *hostp = nullptr;       // pvr_policy_create_host: the CPU plan (host_policy.hip); every device member below stays null
    pvr_policy_desc d;
    std::map<std::string, Slot> slots;
    int64_t n_total = 0, n_train = 0;
    // offsets (elements) into the flat parameter / gradient buffers
    int64_t o_cw[5] = {-1, -1, -1, -1, -1}, o_cb[5] = {-1, -1, -1, -1, -1};
    int64_t o_bnw = -1, o_bnb = -1, o_fc1w, o_fc1b, o_fc2w, o_fc2b, o_wih[2], o_whh[2], o_bih[2], o_bhh[2], o_pw, o_pb, o_bw, o_bb;
    // workspace
    float *a0 = nullptr, *bn_mean = nullptr, *bn_invstd = nullptr, *a1 = nullptr, *a2 = nullptr;
    float *G[2] = {nullptr, nullptr}, *Hs[2] = {nullptr, nullptr}, *Cs[2] = {nullptr, nullptr};
    float *hprev = nullptr, *nd = nullptr, *zeros = nullptr, *dc_carry = nullptr, *rec_partial = nullptr;
    float *hprev1 = nullptr, *dc_carry1 = nullptr, *rec_partial1 = nullptr;   // layer-1 copies: the two layers run concurrently
    float *whhT[2] = {nullptr, nullptr};    // W_hh^T [H][4H] of both layers, refreshed at the start of every backward pass (lstm_bwd_step2_kernel)
    int bwd_fused = 0;                       // PVR_POLICY_BWD_FUSED=1: one launch per BPTT step (correct, SLOWER: profiles/experiments/r03_bc_fused_bptt_step.txt)
    // layer pipeline: the recurrences of the two LSTM layers are chains of ~6 us launches; layer 1 of time chunk c only needs
    // layer 0 of chunk c (forward; mirrored in BPTT), so the two chains run on two streams, a quarter of the sequence apart
    hipStream_t lane_a = nullptr, lane_b = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join_a = nullptr, ev_join_b = nullptr, ev_chunk[8] = {nullptr};
    int pipeline = 1;
    // persistent recurrence (lstm_fwd_seq_kernel): one launch per layer (or per (layer, chunk)) instead of one per step; bit-identical.
    // PVR_POLICY_PERSIST=1 = round 1's hand-off (sc1 stores + drain, 256 atomic adds on one agent-scope counter, a block-wide poll):
    // ~13 us per step against ~10 us for a launch (155 vs 208 steps/s).  =2 (round 2, default): the data is its own flag - Hs[t] is
    // pre-filled with 0xFFFFFFFF words, a consumer wave re-reads its slice of h_{t-1} until no word is that pattern; no atomics, no
    // drain, no block-wide poll: ~5 us per step, 208 -> 219 steps/s with one launch per layer for the whole sequence (two persistent
    // kernels side by side on the two lanes measured slower: 182-194).
    // chunked layer wavefront (default): the two recurrences share launches - layer 0 at step t and layer 1 one chunk (T/4 steps)
    // behind run as the two blockIdx.y jobs of one launch, forward and BPTT, with the hoisted projections done per chunk.  Same
    // per-(layer, step) arithmetic as every other mode (bit-identical); (NCH+1)/(2 NCH) of the dependent launches.
    int chunkwave = 1;
    unsigned *seq_counters = nullptr;       // 16 slots of 16 bytes, zeroed before each launch that uses one
    int persist = 2;                        // 0 per-step launches, 1 persistent with the counter hand-off, 2 persistent with the data-as-flag hand-off
    // A persistent launch is only chosen when its whole grid can be resident at once (persist_fits: occupancy x CU count >= H/4 blocks,
    // checked at create); its bounded spins report through `status_host` (one pinned, GPU-visible word) instead of only poisoning h:
    // every entry point looks at the word on the way in (sticky: the error surfaces at the next call, or at pvr_policy_status after the
    // caller's own sync), returns PVR_ERR_TIMEOUT once, and the handle then stays on per-step launches (persist_tripped).
    int persist_fits = 0, persist_tripped = 0, debug_drop_block = -1;
    // persistent BPTT (lstm_bwd_seq_kernel, round 3; OPT-IN, PVR_POLICY_PERSIST_BWD=1): both layers' recurrences of one chunk wave in
    // ONE launch (2 x 256 blocks), two in-kernel hand-offs per step.  dGx = hand-off copy of the gate gradients (pre-filled per launch),
    // Px = armed ring of partials.  Bit-identical to the launches (tests) but SLOWER on MI355X: a cross-XCD hand-off costs ~3.5 us, two
    // per step plus the arithmetic come to ~14.5 us against 11.7 us for the wavefronted pair of launches (212 vs 233 steps/s) - a BPTT
    // step needs two grid-wide exchanges where the forward step needs one (profiles/experiments/r03_bc_persistent_bptt.txt).
    int persist_bwd = 0, persist_bwd_fits = 0;
    float *dGx[2] = {nullptr, nullptr}, *Px[2] = {nullptr, nullptr};
    unsigned *status_host = nullptr, *status_dev = nullptr;
    float *logits = nullptr, *baseline = nullptr, *dlogits = nullptr, *loss_row = nullptr, *stats = nullptr, *partial = nullptr;
    long long *action = nullptr;
    float *dA = nullptr, *dB = nullptr, *da0 = nullptr;   // [N][H] scratch x2, [N][O]
    float *grads = nullptr;
    bool have_grads = false;
    int sample_on = 0;                      // pvr_policy_set_action_sampling: training-mode forwards write a SAMPLE of softmax(logits) as the action
    unsigned long long sample_seed = 0, sample_call = 0;
    int fwd_T = 0, fwd_B = 0;               // shape of the last training-mode forward (pvr_policy_backward_dlogits needs its activations)
    // PolicyNetWithConv front end (conv_frames > 0)
    float *act[5] = {nullptr}, *dact[5] = {nullptr}, *wp[5] = {nullptr}, *wt[5] = {nullptr}, *feat = nullptr, *dfeat = nullptr;
    float *cpartial = nullptr, *cgpacked = nullptr, *bpartial = nullptr;
    std::vector<void *> conv_owned;
    // Optional hipGraph replay of the whole training iteration (pvr_policy_step, PVR_POLICY_GRAPH=1): the ~430 launches of
    // one T=100 iteration are 5-7 us kernels with ~3.8 us between them (rocprofv3 kernel trace).  Inputs are copied into
    // library-owned staging buffers and lr into stats[3] so that the captured graph has fixed addresses; it is re-captured
    // when any other argument changes.  Measured on ROCm 7.2 / MI355X (scripts/bc_graph_ab.py, 50 steps): replay 174.5-174.8
    // steps/s vs eager 173.2-174.2 - the gap is per-dispatch cost in the command processor, which a graph of kernel nodes
    // pays as well - so the default stays eager; fewer, larger launches are what removes it.
    void *in_obs = nullptr; uint8_t *in_done = nullptr; long long *in_act = nullptr;
    struct StepKey {
        const void *params = nullptr, *sq = nullptr, *bn_rm = nullptr, *bn_rv = nullptr, *bn_nbt = nullptr, *stream = nullptr;
        int T = 0, B = 0; float alpha = 0, eps = 0, mgn = 0;
        bool operator==(const StepKey &o) const {
            return params == o.params && sq == o.sq && bn_rm == o.bn_rm && bn_rv == o.bn_rv && bn_nbt == o.bn_nbt && stream == o.stream &&
                   T == o.T && B == o.B && alpha == o.alpha && eps == o.eps && mgn == o.mgn;
        }
    } graph_key, eager_key;
    // Data parallelism (pvr_policy_set_data_parallel): the caller's collective as a C function pointer.  Gradients are all-reduced
    // in four buckets on comm_stream as each becomes final (LSTM l1 + heads, LSTM l0, fc, conv + BatchNorm), overlapping the rest
    // of the backward pass; SyncBN statistics are all-reduced on the compute stream (they are true dependencies).
    int dp_world = 1, dp_sync_bn = 0;
    pvr_allreduce_fn dp_fn = nullptr;
    void *dp_user = nullptr;
    float *sync_buf = nullptr;              // 2 * obs_size floats (library-owned)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready[4] = {nullptr}, ev_comm = nullptr;
    // scratch of the split-K GEMMs and two-stage column reductions: per policy (two policies may step concurrently on different
    // streams / host threads); superseded buffers are kept until destroy (launches already enqueued may still use them)
    struct Scratch { float *splitk = nullptr, *col = nullptr; size_t splitk_elems = 0, col_elems = 0; std::vector<void *> retired; } scratch;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t cap_stream = nullptr;       // capture happens on a private stream (the caller's may be the legacy stream, which cannot capture)
    int use_graph = 0;
};

namespace {

int64_t add_slot(pvr_policy *p, const std::string &name, int64_t numel) {
    const int64_t off = p->n_total;
    p->slots[name] = Slot{off, numel};
    p->n_total += (numel + 3) / 4 * 4;            // every tensor starts on a 16-byte boundary
    return off;
}

template <typename T>
pvr_status dalloc(T **ptr, size_t n) {
    PVR_HIP_TRY(hipMalloc((void **)ptr, n * sizeof(T)));
    PVR_HIP_TRY(hipMemset(*ptr, 0, n * sizeof(T)));
    return PVR_OK;
}

// Scratch owner of the calling thread: every extern "C" entry installs its policy's scratch (ScratchScope); the unit-parity entry
// pvr_op_gemm_f32 has no policy and uses a process-wide one under a mutex.
static thread_local pvr_policy::Scratch *tls_scratch = nullptr;
static pvr_policy::Scratch g_scratch;
static std::mutex g_scratch_mu;
struct ScratchScope {
    pvr_policy::Scratch *prev;
    bool locked = false;
    explicit ScratchScope(pvr_policy *pol) : prev(tls_scratch) {
        if (pol) tls_scratch = &pol->scratch;
        else { g_scratch_mu.lock(); locked = true; tls_scratch = &g_scratch; }
    }
    ~ScratchScope() { tls_scratch = prev; if (locked) g_scratch_mu.unlock(); }
};
static pvr_status scratch_grow(float **buf, size_t *have, size_t need) {
    pvr_policy::Scratch *S = tls_scratch;
    if (!S) { set_error("policy: internal error, no scratch owner"); return PVR_ERR_STATE; }
    if (need > *have) {
        if (*buf) S->retired.push_back(*buf);
        PVR_HIP_TRY(hipMalloc((void **)buf, need * sizeof(float)));
        *have = need;
    }
    return PVR_OK;
}

template <bool ATR, bool BTR, int BM, int BN>
static pvr_status launch_gemm(const GemmP &g, dim3 gd, hipStream_t st) {
    constexpr size_t lds = gemm_f32_lds<BM, BN, ATR, BTR>();
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)gemm_f32_kernel<ATR, BTR, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    hipLaunchKernelGGL((gemm_f32_kernel<ATR, BTR, BM, BN>), gd, dim3(256), lds, st, g);
    return PVR_OK;
}

static std::atomic<int> g_gemm_mode{-1};          // pvr_debug_set_gemm_mode

#ifdef PVR_EXPERIMENTS
template <bool ATR, bool BTR, int BM, int BN>
static pvr_status launch_gemm_x3(const GemmP &g, dim3 gd, hipStream_t st) {
    constexpr size_t lds = gemm_bf16x3_lds<BM, BN>();
    static DeviceOnce attr_done;          // per device: a second GPU of the process needs the attribute too
    if (attr_done.needed()) {
        PVR_HIP_TRY(hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<ATR, BTR, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.mark();
    }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<ATR, BTR, BM, BN>), gd, dim3(256), lds, st, g);
    return PVR_OK;
}
#endif

pvr_status gemm(const float *A, const float *B, const float *bias, const float *mask, float *C, int M, int N, int K,
                bool a_km, bool b_kn, int relu, hipStream_t st) {
    GemmP g;
    g.A = A; g.B = B; g.bias = bias; g.mask = mask; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = a_km ? M : K; g.ldb = b_kn ? N : K; g.ldc = N; g.relu = relu;
    g.a_kstride = 0; g.b_kstride = 0; g.c_stride = 0;
    PVR_REQUIRE(((a_km && b_kn) || K % 4 == 0) && (!a_km || M % 4 == 0) && (!b_kn || N % 4 == 0),
                "gemm_f32: contiguous dims must be multiples of 4 (M=%d N=%d K=%d)", M, N, K);
    // tile choice (the result never depends on it): PVR_GEMM_TILE = 0 64x64 (round 1/2), 1 128x64, 2 64x128, 3 128x128; default: see below
    static const int forced = [] { const char *e = getenv("PVR_GEMM_TILE"); return e ? atoi(e) : -1; }();
    // measured per shape (profiles/experiments/r03_gemm_f32_tile_shapes.txt): 64x64 everywhere except the two 4 M-output weight-gradient
    // products (dW_hh / dW_ih: 4096 x 1024, dW_fc1: 1024 x 4096; K = T*B), where 64x128 is 3-4 % faster
    int tile = forced >= 0 && forced <= 3 ? forced : (a_km && b_kn && (long long)M * N >= (4ll << 20) && N % 128 == 0) ? 2 : 0;
    const int BMs[4] = {64, 128, 64, 128}, BNs[4] = {64, 64, 128, 128};
    const int BM = BMs[tile], BN = BNs[tile];
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // too few output tiles for the chip and a long K: split K over blockIdx.y into fp32 partial products, summed in slice order
    int S = 1;
    if (((M + 63) / 64) * ((N + 63) / 64) < 192 && K >= 2048 && K % 128 == 0) S = 4;
    float *splitk_buf = nullptr;
    if (S > 1) {
        const size_t need = (size_t)S * M * N;
        if (!tls_scratch) { set_error("policy: internal error, no scratch owner"); return PVR_ERR_STATE; }
        { pvr_status gs = scratch_grow(&tls_scratch->splitk, &tls_scratch->splitk_elems, need); if (gs) return gs; }
        splitk_buf = tls_scratch->splitk;
        g.K = K / S;
        g.a_kstride = a_km ? (long long)g.K * g.lda : g.K;
        g.b_kstride = b_kn ? (long long)g.K * g.ldb : g.K;
        g.c_stride = (long long)M * N;
        g.C = splitk_buf; g.bias = nullptr; g.mask = nullptr; g.relu = 0;
    }
    pvr_status ls = PVR_OK;
    // round 3, experiment build only (make EXPERIMENTS=1; PVR_GEMM_BF16X3=1 or pvr_debug_set_gemm_mode): the same product on the bf16
    // matrix pipe (three-term exact split of both operands, gemm_bf16x3_kernel).  2.7 x more accurate than the fp32 chain and no faster
    // (profiles/experiments/r03_gemm_split_bf16.txt), so the fp32 MFMA GEMM is the product's.
#ifdef PVR_EXPERIMENTS
    static const int x3_env = [] { const char *e = getenv("PVR_GEMM_BF16X3"); return e ? atoi(e) : 0; }();
    const int gm = g_gemm_mode.load();
    const int x3 = gm >= 0 ? gm : x3_env;
    const long long a_ext = ((long long)((a_km ? g.K : M) - 1) * g.lda + (a_km ? M : g.K)) * 4, b_ext = ((long long)((b_kn ? g.K : N) - 1) * g.ldb + (b_kn ? N : g.K)) * 4;
    if (x3 && forced < 0 && a_ext < 0x7ffffff0ll && b_ext < 0x7ffffff0ll) {
        g.a_bytes = (unsigned)a_ext; g.b_bytes = (unsigned)b_ext;
        const long long big = (long long)((M + 127) / 128) * ((N + 127) / 128) * S;
        const bool t128 = x3 == 2 || (x3 != 3 && big >= 192);
        const int bm = t128 ? 128 : 64;
        const dim3 gx(((M + bm - 1) / bm) * ((N + bm - 1) / bm), S);
#define PVR_X3_CASE(BM_)                                                                                                \
        if (!a_km && !b_kn) ls = launch_gemm_x3<false, false, BM_, BM_>(g, gx, st);                                     \
        else if (!a_km && b_kn) ls = launch_gemm_x3<false, true, BM_, BM_>(g, gx, st);                                  \
        else if (a_km && b_kn) ls = launch_gemm_x3<true, true, BM_, BM_>(g, gx, st);                                    \
        else ls = launch_gemm_x3<true, false, BM_, BM_>(g, gx, st);
        if (t128) { PVR_X3_CASE(128) } else { PVR_X3_CASE(64) }
#undef PVR_X3_CASE
    } else
#endif
    {
    const dim3 gd(grid, S);
#define PVR_GEMM_CASE(T_, BM_, BN_)                                                                                     \
    case T_:                                                                                                            \
        if (!a_km && !b_kn) ls = launch_gemm<false, false, BM_, BN_>(g, gd, st);                                        \
        else if (!a_km && b_kn) ls = launch_gemm<false, true, BM_, BN_>(g, gd, st);                                     \
        else if (a_km && b_kn) ls = launch_gemm<true, true, BM_, BN_>(g, gd, st);                                       \
        else ls = launch_gemm<true, false, BM_, BN_>(g, gd, st);                                                        \
        break;
    switch (tile) {
        PVR_GEMM_CASE(0, 64, 64)
        PVR_GEMM_CASE(1, 128, 64)
        PVR_GEMM_CASE(2, 64, 128)
        PVR_GEMM_CASE(3, 128, 128)
    }
#undef PVR_GEMM_CASE
    }
    if (ls) return ls;
    if (S > 1) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st,
                           splitk_buf, C, bias, mask, S, n, N, relu);
    }
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

#define TRY(x) do { pvr_status _s = (x); if (_s) return _s; } while (0)
#define TRY_(x) TRY(x)

// scratch of the two-stage column reductions: [2 * row groups][C] floats, grown on demand (first, eager, iteration)
static pvr_status col_scratch(size_t elems, float **out) {
    if (!tls_scratch) { set_error("policy: internal error, no scratch owner"); return PVR_ERR_STATE; }
    pvr_status gs = scratch_grow(&tls_scratch->col, &tls_scratch->col_elems, elems);
    if (gs) return gs;
    *out = tls_scratch->col;
    return PVR_OK;
}

// column reduction of a [R][C] fp32 matrix: MODE 0 sum, 3 centred sum of squares (c.mean_in), 2 BatchNorm backward sums
template <int MODE>
static pvr_status colreduce2(ColP c, hipStream_t st) {
    constexpr int RPG = 64;
    const int G = (c.R + RPG - 1) / RPG;
    float *part;
    TRY_(col_scratch((size_t)2 * G * c.C, &part));
    hipLaunchKernelGGL(colpart_kernel<MODE>, dim3((c.C / 4 + 31) / 32, G), dim3(256), 0, st, c, part, RPG);
    if (MODE == 2) hipLaunchKernelGGL(colfinal_kernel<1>, dim3((c.C + 255) / 256), dim3(256), 0, st, part, c.out0, c.out1, G, c.C);
    else hipLaunchKernelGGL(colfinal_kernel<0>, dim3((c.C + 255) / 256), dim3(256), 0, st, part, c.out0, c.out1, G, c.C);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

pvr_status colsum(const float *X, float *out0, float *out1, int R, int C, hipStream_t st) {
    ColP c = {};
    c.X = X; c.out0 = out0; c.out1 = out1; c.R = R; c.C = C;
    if (R >= 512 && C % 4 == 0) return colreduce2<0>(c, st);
    hipLaunchKernelGGL(colreduce_kernel<0>, dim3((C + 31) / 32), dim3(256), 0, st, c);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

inline int blocks_for(size_t n, int cap = 4096) {
    size_t b = (n + 255) / 256;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}


static inline bool dp_active(const pvr_policy *pol) { return pol->dp_fn && pol->dp_world > 1; }

// one collective through the caller's function pointer
static pvr_status dp_allreduce(pvr_policy *pol, float *buf, int64_t count, hipStream_t s_) {
    if (pol->dp_fn((void *)buf, count, (void *)s_, pol->dp_user) != 0) {
        set_error("policy: the data-parallel all-reduce callback failed (%lld floats)", (long long)count);
        return PVR_ERR_COMM;
    }
    return PVR_OK;
}

// gradient bucket [off, off+count) of Gd is final on `st`: all-reduce it on the communication stream and divide by the world size
static pvr_status dp_bucket(pvr_policy *pol, float *Gd, int64_t off, int64_t count, int idx, hipStream_t st) {
    if (!dp_active(pol) || count <= 0) return PVR_OK;
    PVR_HIP_TRY(hipEventRecord(pol->ev_ready[idx], st));
    PVR_HIP_TRY(hipStreamWaitEvent(pol->comm_stream, pol->ev_ready[idx], 0));
    pvr_status s = dp_allreduce(pol, Gd + off, count, pol->comm_stream);
    if (s) return s;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, pol->comm_stream, Gd + off,
                       1.0f / (float)pol->dp_world, (long long)count);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}


// forward through the workspace; logits/baseline/action land in pol->logits etc.  `target` non-null also
// produces dlogits and per-row losses.
pvr_status forward_core(pvr_policy *pol, const float *P, const pvr_policy_bn *bn, const void *obs_in, const uint8_t *done,
                        const float *h0, const float *c0, int T, int B, int training, const long long *target,
                        hipStream_t st) {
    const auto &d = pol->d;
    const int N = T * B, H = d.hidden, O = d.obs_size;
    // torch: "Expected more than 1 value per channel when training" (nn.BatchNorm1d on one row): no unbiased variance to put in running_var
    PVR_REQUIRE(!(d.batch_norm && training && N == 1), "policy: BatchNorm in training mode needs more than one row (T * B = 1), as torch.nn.BatchNorm1d does");
    hipLaunchKernelGGL(notdone_kernel, dim3((N + 255) / 256), dim3(256), 0, st, done, pol->nd, N);
    const float *obs = (const float *)obs_in;
    if (d.conv_frames > 0) {
        // models.py:163-170: x/255, per-frame transpose(1,3), 5 x (conv3x3 s2 p1 + ELU), cat(-1), view(T*B,-1)
        const int nf = d.conv_frames, F = N * nf;
        const void *in = obs_in;
        int S = 64;
        for (int l = 0; l < 5; ++l) {
            const int So = S / 2;
            if (l == 0) hipLaunchKernelGGL(conv_pack_kernel<3>, dim3(5), dim3(256), 0, st, P + pol->o_cw[l], pol->wp[l], 1);
            else hipLaunchKernelGGL(conv_pack_kernel<32>, dim3(36), dim3(256), 0, st, P + pol->o_cw[l], pol->wp[l], 1);
            ConvFP c;
            c.in = in; c.W = pol->wp[l]; c.bias = P + pol->o_cb[l]; c.out = pol->act[l]; c.F = F; c.Sin = S; c.So = So; c.nf = nf;
            const long long tiles = ((long long)F * So * So + 15) / 16;
            if (l == 0) hipLaunchKernelGGL(conv_s2_fwd_kernel<3>, dim3((unsigned)((tiles + 4 * CONV_TPW - 1) / (4 * CONV_TPW))), dim3(256), 0, st, c);
            else hipLaunchKernelGGL(conv_s2_fwd_kernel<32>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, c);
            in = pol->act[l]; S = So;
        }
        hipLaunchKernelGGL(conv_feat_kernel, dim3(blocks_for((size_t)N * 128 * nf)), dim3(256), 0, st, pol->act[4], pol->feat, N, nf, 1);
        PVR_LAUNCH_CHECK();
        obs = pol->feat;
    }
    const float *x0 = obs;
    if (d.batch_norm) {
        PVR_REQUIRE(bn && bn->running_mean && bn->running_var, "policy: batch_norm=1 needs the BN buffers");
        if (training) {
            PVR_REQUIRE(N > 1, "BatchNorm1d training needs more than one row");
            {
                // mean, then centred second moment (two passes, as torch): per-rank, or over the global batch under SyncBN with an
                // all-reduce of obs_size floats after each pass
                const bool sync = dp_active(pol) && pol->dp_sync_bn;
                const float ng = (float)N * (float)(sync ? pol->dp_world : 1);
                float *sums = sync ? pol->sync_buf : pol->bn_invstd;                // (bn_invstd doubles as scratch until the final kernel)
                float *sq = sync ? pol->sync_buf + O : pol->da0;                    // (da0 is free during the forward pass)
                TRY(colsum(obs, sums, nullptr, N, O, st));
                if (sync) TRY(dp_allreduce(pol, pol->sync_buf, O, st));
                hipLaunchKernelGGL(scale_kernel, dim3((O + 255) / 256), dim3(256), 0, st, pol->bn_mean, sums, 1.0f / ng, O);
                ColP c2 = {};
                c2.X = obs; c2.mean_in = pol->bn_mean; c2.out0 = sq; c2.R = N; c2.C = O;
                if (N >= 512 && O % 4 == 0) TRY(colreduce2<3>(c2, st));
                else hipLaunchKernelGGL(colreduce_kernel<3>, dim3((O + 31) / 32), dim3(256), 0, st, c2);
                PVR_LAUNCH_CHECK();
                if (sync) TRY(dp_allreduce(pol, pol->sync_buf + O, O, st));
                hipLaunchKernelGGL(bn_sync_final_kernel, dim3((O + 255) / 256), dim3(256), 0, st, sq, pol->bn_mean, ng,
                                   pol->bn_invstd, bn->running_mean, bn->running_var, (long long *)bn->num_batches_tracked, O);
            }
            hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for((size_t)N * O / 4)), dim3(256), 0, st, obs, pol->bn_mean, pol->bn_invstd,
                               P + pol->o_bnw, P + pol->o_bnb, pol->a0, (size_t)N * O / 4, O, 0);
        } else {
            hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for((size_t)N * O / 4)), dim3(256), 0, st, obs, bn->running_mean,
                               bn->running_var, P + pol->o_bnw, P + pol->o_bnb, pol->a0, (size_t)N * O / 4, O, 1);
        }
        PVR_LAUNCH_CHECK();
        x0 = pol->a0;
    }
    TRY(gemm(x0, P + pol->o_fc1w, P + pol->o_fc1b, nullptr, pol->a1, N, H, O, false, false, 1, st));
    TRY(gemm(pol->a1, P + pol->o_fc2w, P + pol->o_fc2b, nullptr, pol->a2, N, H, H, false, false, 1, st));
    // LSTM, two layers.  Input projections are hoisted out of the recurrence (one GEMM over all steps of a chunk).
    // effective hand-off mode of this call: none when the grid cannot be co-resident or a spin has run out on this handle before;
    // the data-as-flag form never inside a hipGraph (capture or replay: see below)
    const int persist = (!pol->persist_fits || pol->persist_tripped) ? 0 : (pol->persist == 2 && pol->use_graph) ? 0 : pol->persist;
    hipError_t fill_err = hipSuccess;
    auto fwd_job = [&](int l, int t) {
        LstmFwdP f;
        f.G = pol->G[l] + (size_t)t * B * 4 * H;
        f.h_prev = t == 0 ? h0 + (size_t)l * B * H : pol->Hs[l] + (size_t)(t - 1) * B * H;
        f.c_prev = t == 0 ? c0 + (size_t)l * B * H : pol->Cs[l] + (size_t)(t - 1) * B * H;
        f.nd = pol->nd + (size_t)t * B;
        f.W = P + pol->o_whh[l];
        f.bhh = P + pol->o_bhh[l];
        f.h_out = pol->Hs[l] + (size_t)t * B * H;
        f.c_out = pol->Cs[l] + (size_t)t * B * H;
        f.B = B; f.H = H;
        return f;
    };
    auto fwd_steps = [&](int l, int t0, int t1, hipStream_t s_) {
        if (persist && H == 1024 && B <= 64 && t1 - t0 > 1) {
            // one persistent launch for the whole step range (grid-wide hand-off per step inside the kernel)
            unsigned *ctr = pol->seq_counters + 4 * ((l * 4 + (t0 * 4 / (T > 0 ? T : 1))) & 15);
            LstmSeqP q;
            q.data_flag = persist == 2;
            q.status = pol->status_dev;
            q.drop_block = pol->debug_drop_block;
            const hipError_t me = q.data_flag ? hipMemsetAsync(pol->Hs[l] + (size_t)t0 * B * H, 0xFF, (size_t)(t1 - t0) * B * H * sizeof(float), s_)
                                              : hipMemsetAsync(ctr, 0, 16, s_);
            if (me != hipSuccess) {                              // without the pre-fill the hand-off has no meaning: per-step launches instead
                fill_err = me;
                for (int t = t0; t < t1; ++t) { LstmFwdP f = fwd_job(l, t); hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(H / 4), dim3(256), 0, s_, f); }
                return;
            }
            q.G = pol->G[l];
            q.h_init = t0 == 0 ? h0 + (size_t)l * B * H : pol->Hs[l] + (size_t)(t0 - 1) * B * H;
            q.c_init = t0 == 0 ? c0 + (size_t)l * B * H : pol->Cs[l] + (size_t)(t0 - 1) * B * H;
            q.nd = pol->nd; q.W = P + pol->o_whh[l]; q.bhh = P + pol->o_bhh[l];
            q.Hs = pol->Hs[l]; q.Cs = pol->Cs[l]; q.counter = ctr; q.t0 = t0; q.t1 = t1; q.B = B; q.H = H;
            hipLaunchKernelGGL(lstm_fwd_seq_kernel, dim3(H / 4), dim3(256), 0, s_, q);
            return;
        }
        for (int t = t0; t < t1; ++t) { LstmFwdP f = fwd_job(l, t); hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(H / 4), dim3(256), 0, s_, f); }
    };
    TRY(gemm(pol->a2, P + pol->o_wih[0], P + pol->o_bih[0], nullptr, pol->G[0], N, 4 * H, H, false, false, 0, st));
    const int NCH = (pol->pipeline || pol->chunkwave) && T >= 8 ? 4 : 1, CH = (T + NCH - 1) / NCH;
    // persistent recurrence with the data-as-flag hand-off (default, PVR_POLICY_PERSIST=2): ONE launch per layer for the whole sequence
    // (not inside a hipGraph: replayed nodes run with weaker cache maintenance between them than stream launches - the 0xFF pre-fill of a
    //  memset node was not visible to the other XCDs' sc1 loads in time, which then took the PREVIOUS iteration's h for data; measured
    //  as a 1e-3 drift with PVR_POLICY_GRAPH=1 on the conv model, scripts/debug_bc_modes.py)
    const bool use_persist = persist == 2 && H == 1024 && B <= 64 && T > 1;
    if (use_persist) {
        fwd_steps(0, 0, T, st);
        TRY(gemm(pol->Hs[0], P + pol->o_wih[1], P + pol->o_bih[1], nullptr, pol->G[1], N, 4 * H, H, false, false, 0, st));
        fwd_steps(1, 0, T, st);
        PVR_LAUNCH_CHECK();
    } else if (pol->chunkwave && !persist && NCH > 1) {
        for (int c = 0; c <= NCH; ++c) {
            if (c >= 1) {                                       // input projection of layer 1 for the chunk layer 0 has just finished
                const int t0 = (c - 1) * CH, t1 = c * CH < T ? c * CH : T;
                if (t1 > t0) {
                    const size_t r0 = (size_t)t0 * B;
                    TRY(gemm(pol->Hs[0] + r0 * H, P + pol->o_wih[1], P + pol->o_bih[1], nullptr, pol->G[1] + r0 * 4 * H, (t1 - t0) * B, 4 * H, H,
                             false, false, 0, st));
                }
            }
            for (int s_ = 0; s_ < CH; ++s_) {
                const int ta = c * CH + s_, tb = (c - 1) * CH + s_;
                LstmFwd2P w = {};
                w.active[0] = c < NCH && ta < T && ta < (c + 1) * CH;
                w.active[1] = c >= 1 && tb < T && tb < c * CH;
                if (!w.active[0] && !w.active[1]) continue;
                if (w.active[0]) w.j[0] = fwd_job(0, ta);
                if (w.active[1]) w.j[1] = fwd_job(1, tb);
                hipLaunchKernelGGL(lstm_fwd_step2_kernel, dim3(H / 4, 2), dim3(256), 0, st, w);
            }
        }
        PVR_LAUNCH_CHECK();
    } else if (NCH == 1 || !pol->pipeline) {
        fwd_steps(0, 0, T, st);
        TRY(gemm(pol->Hs[0], P + pol->o_wih[1], P + pol->o_bih[1], nullptr, pol->G[1], N, 4 * H, H, false, false, 0, st));
        fwd_steps(1, 0, T, st);
        PVR_LAUNCH_CHECK();
    } else {
        // lane A: layer 0; lane B: per chunk the input projection of layer 1, then its steps
        hipStream_t sa = pol->lane_a, sb = pol->lane_b;
        PVR_HIP_TRY(hipEventRecord(pol->ev_fork, st));
        PVR_HIP_TRY(hipStreamWaitEvent(sa, pol->ev_fork, 0));
        PVR_HIP_TRY(hipStreamWaitEvent(sb, pol->ev_fork, 0));
        for (int c = 0; c < NCH; ++c) {
            const int t0 = c * CH, t1 = (c + 1) * CH < T ? (c + 1) * CH : T;
            if (t0 >= t1) break;
            fwd_steps(0, t0, t1, sa);
            PVR_HIP_TRY(hipEventRecord(pol->ev_chunk[c], sa));
            PVR_HIP_TRY(hipStreamWaitEvent(sb, pol->ev_chunk[c], 0));
            const size_t r0 = (size_t)t0 * B;
            TRY(gemm(pol->Hs[0] + r0 * H, P + pol->o_wih[1], P + pol->o_bih[1], nullptr, pol->G[1] + r0 * 4 * H, (t1 - t0) * B, 4 * H, H,
                     false, false, 0, sb));
            fwd_steps(1, t0, t1, sb);
        }
        PVR_LAUNCH_CHECK();
        PVR_HIP_TRY(hipEventRecord(pol->ev_join_a, sa));
        PVR_HIP_TRY(hipEventRecord(pol->ev_join_b, sb));
        PVR_HIP_TRY(hipStreamWaitEvent(st, pol->ev_join_a, 0));
        PVR_HIP_TRY(hipStreamWaitEvent(st, pol->ev_join_b, 0));
    }
    HeadP hp;
    hp.out = pol->Hs[1]; hp.Wp = P + pol->o_pw; hp.bp = P + pol->o_pb; hp.Wb = P + pol->o_bw; hp.bb = P + pol->o_bb;
    hp.logits = pol->logits; hp.baseline = pol->baseline; hp.dlogits = pol->dlogits; hp.loss_row = pol->loss_row;
    hp.action = pol->action; hp.target = target; hp.N = N; hp.H = H; hp.A = d.num_actions;
    hp.sample = pol->sample_on && training && !target; hp.seed = pol->sample_seed; hp.call = hp.sample ? pol->sample_call++ : 0;
    hipLaunchKernelGGL(heads_kernel, dim3((N + 3) / 4), dim3(256), 0, st, hp);
    PVR_LAUNCH_CHECK();
    if (fill_err != hipSuccess) {
        // the recurrence ran through per-step launches (results are valid); the failed memset is still an error of this call
        set_error("policy: hipMemsetAsync of the persistent recurrence's hand-off words failed: %s", hipGetErrorString(fill_err));
        return PVR_ERR_HIP;
    }
    return PVR_OK;
}

}  // namespace

// backward of everything after the forward in the workspace; leaves the unclipped gradient in Gd
static pvr_status backward_core(pvr_policy *pol, const float *P, const void *obs_in, int T, int B, float *Gd, hipStream_t st) {
    const auto &d = pol->d;
    const int N = T * B, H = d.hidden, O = d.obs_size, A = d.num_actions;
    const float *obs = d.conv_frames > 0 ? pol->feat : (const float *)obs_in;
    // ---- heads backward --------------------------------------------------------------------------------------------
    if (N >= 512 && H % 4 == 0) {
        constexpr int RPG = 64;
        const int G = (N + RPG - 1) / RPG;
        float *part;
        TRY(col_scratch((size_t)G * A * (H + 4), &part));
        hipLaunchKernelGGL(head_dw_part_kernel, dim3((H / 4 + 31) / 32, G), dim3(256), 0, st, pol->dlogits, pol->Hs[1], part, N, H, A, RPG);
        hipLaunchKernelGGL(head_dw_final_kernel, dim3((A * (H + 1) + 255) / 256), dim3(256), 0, st, part, Gd + pol->o_pw, Gd + pol->o_pb, G, H, A);
    } else {
        hipLaunchKernelGGL(head_dw_kernel, dim3((H + 1 + 31) / 32), dim3(256), 0, st, pol->dlogits, pol->Hs[1], Gd + pol->o_pw, Gd + pol->o_pb, N, H, A);
    }
    hipLaunchKernelGGL(head_dx_kernel, dim3(blocks_for((size_t)N * H)), dim3(256), 0, st, pol->dlogits, P + pol->o_pw, pol->dA, N, H, A);
    PVR_LAUNCH_CHECK();
    // ---- LSTM backward, layer 1 then layer 0 ------------------------------------------------------------------------
    float *scr_dc[2] = {pol->dc_carry, pol->dc_carry1}, *scr_rec[2] = {pol->rec_partial, pol->rec_partial1}, *scr_hp[2] = {pol->hprev, pol->hprev1};
    // opt-in, one launch per step (round 3): W_hh^T of both layers first, then lstm_bwd_step2_kernel per (wavefronted) step
#ifdef PVR_EXPERIMENTS
    const bool fused = pol->bwd_fused && pol->whhT[0] && H % 16 == 0 && (4 * H) % 512 == 0;
    if (fused)
        for (int l = 0; l < 2; ++l)
            hipLaunchKernelGGL(transpose_kernel, dim3(H / 32, 4 * H / 32), dim3(256), 0, st, P + pol->o_whh[l], pol->whhT[l], 4 * H, H);
#endif
#ifdef PVR_EXPERIMENTS
    auto step_job = [&](int l, int t, const float *dh_ext_) {
        const bool has_next = t < T - 1;
        LstmStepP c;
        c.dG_next = has_next ? pol->G[l] + (size_t)(t + 1) * B * 4 * H : nullptr;
        c.WT = pol->whhT[l];
        c.nd_next = has_next ? pol->nd + (size_t)(t + 1) * B : nullptr;
        c.dh_ext = dh_ext_ + (size_t)t * B * H;
        c.dc_carry = scr_dc[l];
        c.G = pol->G[l] + (size_t)t * B * 4 * H;
        c.c_t = pol->Cs[l] + (size_t)t * B * H;
        c.c_prev = t == 0 ? pol->zeros : pol->Cs[l] + (size_t)(t - 1) * B * H;
        c.nd = pol->nd + (size_t)t * B;
        c.B = B; c.H = H;
        return c;
    };
#endif
    auto bwd_steps = [&](int l, int t_hi, int t_lo, const float *dh_ext_, hipStream_t s_) {          // t = t_hi-1 ... t_lo
        for (int t = t_hi - 1; t >= t_lo; --t) {
#ifdef PVR_EXPERIMENTS
            if (fused) {
                LstmStep2P q = {};
                q.active[0] = 1; q.j[0] = step_job(l, t, dh_ext_);
                hipLaunchKernelGGL(lstm_bwd_step2_kernel, dim3(H / 16, 1), dim3(256), 0, s_, q);
                continue;
            }
#endif
            const bool has_next = t < T - 1;
            if (has_next) {
                LstmRecP r;
                r.dG_next = pol->G[l] + (size_t)(t + 1) * B * 4 * H; r.W = P + pol->o_whh[l]; r.partial = scr_rec[l]; r.B = B; r.H = H;
                hipLaunchKernelGGL(lstm_bwd_rec_kernel, dim3(256), dim3(256), 0, s_, r);
            }
            LstmCellBP c;
            c.partial = has_next ? scr_rec[l] : nullptr;
            c.nd_next = has_next ? pol->nd + (size_t)(t + 1) * B : nullptr;
            c.dh_ext = dh_ext_ + (size_t)t * B * H;
            c.dc_carry = scr_dc[l];
            c.G = pol->G[l] + (size_t)t * B * 4 * H;
            c.c_t = pol->Cs[l] + (size_t)t * B * H;
            c.c_prev = t == 0 ? pol->zeros : pol->Cs[l] + (size_t)(t - 1) * B * H;
            c.nd = pol->nd + (size_t)t * B;
            c.B = B; c.H = H;
            hipLaunchKernelGGL(lstm_bwd_cell_kernel, dim3((B * H + 255) / 256), dim3(256), 0, s_, c);
        }
    };
    // dW_hh = dG^T (nd * h_prev), dW_ih = dG^T x_in, db_ih = db_hh = colsum(dG)
    auto weight_grads = [&](int l, hipStream_t s_) -> pvr_status {
        const float *xin = l == 0 ? pol->a2 : pol->Hs[0];
        hipLaunchKernelGGL(hprev_kernel, dim3(blocks_for((size_t)N * H / 4)), dim3(256), 0, s_, pol->Hs[l], pol->zeros, pol->nd, scr_hp[l], T, B, H);
        TRY(gemm(pol->G[l], scr_hp[l], nullptr, nullptr, Gd + pol->o_whh[l], 4 * H, H, N, true, true, 0, s_));
        TRY(gemm(pol->G[l], xin, nullptr, nullptr, Gd + pol->o_wih[l], 4 * H, H, N, true, true, 0, s_));
        return colsum(pol->G[l], Gd + pol->o_bih[l], Gd + pol->o_bhh[l], N, 4 * H, s_);
    };
    float *dh1 = pol->dA, *dh0 = pol->dB;          // d(loss)/d(h) arriving from above: layer 1 <- heads, layer 0 <- layer 1's dx
    // data-parallel gradient buckets = contiguous ranges of the flat layout, in the order backward finalises them
    const int64_t b_l1 = pol->o_wih[1], b_l0 = pol->o_wih[0], b_fc = pol->o_fc1w;
    const int NCH = (pol->pipeline || pol->chunkwave) && T >= 8 ? 4 : 1, CH = (T + NCH - 1) / NCH;
    auto rec_job = [&](int l, int t) {
        LstmRecP r;
        r.dG_next = pol->G[l] + (size_t)(t + 1) * B * 4 * H; r.W = P + pol->o_whh[l]; r.partial = scr_rec[l]; r.B = B; r.H = H;
        return r;
    };
    auto cell_job = [&](int l, int t, const float *dh_ext_) {
        const bool has_next = t < T - 1;
        LstmCellBP c;
        c.partial = has_next ? scr_rec[l] : nullptr;
        c.nd_next = has_next ? pol->nd + (size_t)(t + 1) * B : nullptr;
        c.dh_ext = dh_ext_ + (size_t)t * B * H;
        c.dc_carry = scr_dc[l];
        c.G = pol->G[l] + (size_t)t * B * 4 * H;
        c.c_t = pol->Cs[l] + (size_t)t * B * H;
        c.c_prev = t == 0 ? pol->zeros : pol->Cs[l] + (size_t)(t - 1) * B * H;
        c.nd = pol->nd + (size_t)t * B;
        c.B = B; c.H = H;
        return c;
    };
    // persistent BPTT: the chunk waves below, each as ONE launch (both layers' step ranges as the two jobs of lstm_bwd_seq_kernel)
#ifdef PVR_EXPERIMENTS
    const bool bwd_persist = pol->persist_bwd && pol->persist_bwd_fits && pol->dGx[0] && !pol->persist_tripped && !pol->use_graph &&
                             pol->persist == 2 && pol->chunkwave && H == 1024 && B <= 64 && T > 1;
#else
    const bool bwd_persist = false;
#endif
    if (pol->chunkwave && (NCH > 1 || bwd_persist)) {
        // chunk index c descending; launch pair s: layer 1 at the s-th step (from the top) of chunk c, layer 0 at the s-th of chunk c+1
        for (int c = NCH - 1; c >= -1; --c) {
#ifdef PVR_EXPERIMENTS
            if (bwd_persist) {
                LstmBwdSeqP q = {};
                q.nd = pol->nd; q.T = T; q.B = B; q.H = H; q.drop_block = pol->debug_drop_block; q.status = pol->status_dev;
                const float *dh_of[2] = {dh0, dh1};
                for (int l = 0; l < 2; ++l) {
                    const int cc = l == 1 ? c : c + 1;              // layer 1 works on chunk c, layer 0 on the chunk behind it
                    LstmBwdJob &jb = q.j[l];
                    jb.t_lo = cc * CH; jb.t_hi = (cc + 1) * CH < T ? (cc + 1) * CH : T;
                    jb.active = cc >= 0 && cc < NCH && jb.t_hi > jb.t_lo;
                    if (!jb.active) continue;
                    jb.G = pol->G[l]; jb.dGx = pol->dGx[l]; jb.Px = pol->Px[l]; jb.W = P + pol->o_whh[l]; jb.dh_ext = dh_of[l];
                    jb.Cs = pol->Cs[l]; jb.c0 = pol->zeros; jb.dc_carry = scr_dc[l];
                    PVR_HIP_TRY(hipMemsetAsync(pol->dGx[l] + (size_t)jb.t_lo * B * 4 * H, 0xFF, (size_t)(jb.t_hi - jb.t_lo) * B * 4 * H * sizeof(float), st));
                }
                if (q.j[0].active || q.j[1].active) hipLaunchKernelGGL(lstm_bwd_seq_kernel, dim3(256, 2), dim3(256), 0, st, q);
            } else
#endif
            for (int s_ = 0; s_ < CH; ++s_) {
                const int hi1 = (c + 1) * CH < T ? (c + 1) * CH : T, hi0 = (c + 2) * CH < T ? (c + 2) * CH : T;
                const int ta = hi1 - 1 - s_, tb = hi0 - 1 - s_;          // layer 1 step, layer 0 step
                const bool a1 = c >= 0 && ta >= c * CH && ta >= 0, a0 = c + 1 < NCH && tb >= (c + 1) * CH && tb >= 0;
                if (!a1 && !a0) continue;
#ifdef PVR_EXPERIMENTS
                if (fused) {
                    LstmStep2P q = {};
                    q.active[0] = a0; q.active[1] = a1;
                    if (a0) q.j[0] = step_job(0, tb, dh0);
                    if (a1) q.j[1] = step_job(1, ta, dh1);
                    hipLaunchKernelGGL(lstm_bwd_step2_kernel, dim3(H / 16, 2), dim3(256), 0, st, q);
                    continue;
                }
#endif
                LstmRec2P r = {};
                r.active[0] = a0 && tb < T - 1; r.active[1] = a1 && ta < T - 1;
                if (r.active[0]) r.j[0] = rec_job(0, tb);
                if (r.active[1]) r.j[1] = rec_job(1, ta);
                if (r.active[0] || r.active[1]) hipLaunchKernelGGL(lstm_bwd_rec2_kernel, dim3(256, 2), dim3(256), 0, st, r);
                LstmCellB2P q = {};
                q.active[0] = a0; q.active[1] = a1;
                if (a0) q.j[0] = cell_job(0, tb, dh0);
                if (a1) q.j[1] = cell_job(1, ta, dh1);
                hipLaunchKernelGGL(lstm_bwd_cell2_kernel, dim3((B * H + 255) / 256, 2), dim3(256), 0, st, q);
            }
            PVR_LAUNCH_CHECK();
            if (c >= 0) {                                       // dh of layer 0 for the chunk layer 1 has just finished
                const int t0 = c * CH, t1 = (c + 1) * CH < T ? (c + 1) * CH : T;
                if (t1 > t0) {
                    const size_t r0 = (size_t)t0 * B;
                    TRY(gemm(pol->G[1] + r0 * 4 * H, P + pol->o_wih[1], nullptr, nullptr, dh0 + r0 * H, (t1 - t0) * B, H, 4 * H, false, true, 0, st));
                }
            }
            if (c == 0 && dp_active(pol)) {
                // data parallel: layer 1 has finished its BPTT here - its weight gradients (33.5 MB with the policy head) go out now
                // and travel while layer 0 runs its last chunk and everything below it
                TRY(weight_grads(1, st));
                TRY(dp_bucket(pol, Gd, b_l1, pol->n_train - b_l1, 0, st));
            }
        }
        if (!dp_active(pol)) TRY(weight_grads(1, st));
    } else if (NCH == 1 || !pol->pipeline) {
        bwd_steps(1, T, 0, dh1, st);
        PVR_LAUNCH_CHECK();
        TRY(weight_grads(1, st));
        TRY(dp_bucket(pol, Gd, b_l1, pol->n_train - b_l1, 0, st));
        TRY(gemm(pol->G[1], P + pol->o_wih[1], nullptr, nullptr, dh0, N, H, 4 * H, false, true, 0, st));
        bwd_steps(0, T, 0, dh0, st);
        PVR_LAUNCH_CHECK();
    } else {
        // lane A: layer 1 BPTT chunk by chunk (t descending), then its weight gradients; lane B: per chunk the dx GEMM of layer 1
        // (= dh of layer 0) and layer 0's BPTT steps of that chunk
        hipStream_t sa = pol->lane_a, sb = pol->lane_b;
        PVR_HIP_TRY(hipEventRecord(pol->ev_fork, st));
        PVR_HIP_TRY(hipStreamWaitEvent(sa, pol->ev_fork, 0));
        PVR_HIP_TRY(hipStreamWaitEvent(sb, pol->ev_fork, 0));
        for (int c = NCH - 1; c >= 0; --c) {
            const int t0 = c * CH, t1 = (c + 1) * CH < T ? (c + 1) * CH : T;
            if (t0 >= t1) continue;
            bwd_steps(1, t1, t0, dh1, sa);
            PVR_HIP_TRY(hipEventRecord(pol->ev_chunk[c], sa));
            PVR_HIP_TRY(hipStreamWaitEvent(sb, pol->ev_chunk[c], 0));
            const size_t r0 = (size_t)t0 * B;
            TRY(gemm(pol->G[1] + r0 * 4 * H, P + pol->o_wih[1], nullptr, nullptr, dh0 + r0 * H, (t1 - t0) * B, H, 4 * H, false, true, 0, sb));
            bwd_steps(0, t1, t0, dh0, sb);
        }
        PVR_LAUNCH_CHECK();
        TRY(weight_grads(1, sa));
        PVR_HIP_TRY(hipEventRecord(pol->ev_join_a, sa));
        PVR_HIP_TRY(hipEventRecord(pol->ev_join_b, sb));
        PVR_HIP_TRY(hipStreamWaitEvent(st, pol->ev_join_a, 0));
        PVR_HIP_TRY(hipStreamWaitEvent(st, pol->ev_join_b, 0));
        TRY(dp_bucket(pol, Gd, b_l1, pol->n_train - b_l1, 0, st));
    }
    TRY(weight_grads(0, st));
    TRY(dp_bucket(pol, Gd, b_l0, b_l1 - b_l0, 1, st));
    // dx_in of layer 0 = dG W_ih with the ReLU of fc2 applied as a mask (a2 > 0); dh1's buffer is free again
    TRY(gemm(pol->G[0], P + pol->o_wih[0], nullptr, pol->a2, dh1, N, H, 4 * H, false, true, 0, st));
    float *dh_ext = dh1, *dx = dh0;
    // dh_ext now holds dz2 = d(fc2 pre-activation) [N][H]
    float *dz2 = dh_ext, *dz1 = dx;
    const float *x0 = d.batch_norm ? pol->a0 : obs;
    TRY(gemm(dz2, pol->a1, nullptr, nullptr, Gd + pol->o_fc2w, H, H, N, true, true, 0, st));
    TRY(colsum(dz2, Gd + pol->o_fc2b, nullptr, N, H, st));
    TRY(gemm(dz2, P + pol->o_fc2w, nullptr, pol->a1, dz1, N, H, H, false, true, 0, st));      // masked by a1 > 0
    const bool need_dobs = d.conv_frames > 0;
    // PolicyNet with BatchNorm: the affine gradients come out of the fc1 weight-gradient GEMM itself (bn_fold_grads_kernel), without
    // the [N][O] product dz1 W1 (PVR_POLICY_BN_FOLD=0: round 2's path through da0)
    static const bool bn_fold_on = [] { const char *e = getenv("PVR_POLICY_BN_FOLD"); return !e || atoi(e) != 0; }();
    const bool bn_fold = d.batch_norm && !need_dobs && bn_fold_on && O % 128 == 0 && H % 64 == 0;
    if (bn_fold) {
        float *xhat = pol->da0;                                                    // (free: nothing needs d(loss)/d(a0) on this path)
        hipLaunchKernelGGL(bn_xhat_kernel, dim3(blocks_for((size_t)N * O / 4)), dim3(256), 0, st, obs, pol->bn_mean, pol->bn_invstd, xhat, (size_t)N * O / 4, O);
        TRY(gemm(dz1, xhat, nullptr, nullptr, Gd + pol->o_fc1w, H, O, N, true, true, 0, st));         // S = dz1^T xhat
        TRY(colsum(dz1, Gd + pol->o_fc1b, nullptr, N, H, st));
        constexpr int RPG = 64;
        const int G = (H + RPG - 1) / RPG;
        float *part;
        TRY(col_scratch((size_t)2 * G * O, &part));
        hipLaunchKernelGGL(bn_fold_grads_kernel, dim3(O / 128, G), dim3(256), 0, st, Gd + pol->o_fc1w, P + pol->o_fc1w, Gd + pol->o_fc1b, P + pol->o_bnw,
                           P + pol->o_bnb, part, H, O, RPG);
        hipLaunchKernelGGL(colfinal_kernel<1>, dim3((O + 255) / 256), dim3(256), 0, st, part, Gd + pol->o_bnw, Gd + pol->o_bnb, G, O);
        PVR_LAUNCH_CHECK();
        TRY(dp_bucket(pol, Gd, b_fc, b_l0 - b_fc, 2, st));
    } else {
    TRY(gemm(dz1, x0, nullptr, nullptr, Gd + pol->o_fc1w, H, O, N, true, true, 0, st));
    TRY(colsum(dz1, Gd + pol->o_fc1b, nullptr, N, H, st));
    TRY(dp_bucket(pol, Gd, b_fc, b_l0 - b_fc, 2, st));
    }
    if (!bn_fold && (d.batch_norm || need_dobs)) TRY(gemm(dz1, P + pol->o_fc1w, nullptr, nullptr, pol->da0, N, O, H, false, true, 0, st));
    const float *dfeat = pol->da0;
    if (d.batch_norm && !bn_fold) {
        ColP c = {};
        c.X = obs; c.dY = pol->da0; c.mean_in = pol->bn_mean; c.invstd_in = pol->bn_invstd;
        c.out0 = Gd + pol->o_bnw; c.out1 = Gd + pol->o_bnb; c.R = N; c.C = O;
        if (N >= 512 && O % 4 == 0) TRY(colreduce2<2>(c, st));
        else hipLaunchKernelGGL(colreduce_kernel<2>, dim3((O + 31) / 32), dim3(256), 0, st, c);
        if (need_dobs) {
            const float *dg = Gd + pol->o_bnw, *db = Gd + pol->o_bnb;
            int n_bn = N;
            if (dp_active(pol) && pol->dp_sync_bn) {
                // dx needs the sums over the GLOBAL batch; the gradient buffer keeps the local ones (they are averaged with the rest)
                PVR_HIP_TRY(hipMemcpyAsync(pol->sync_buf, dg, (size_t)O * 4, hipMemcpyDeviceToDevice, st));
                PVR_HIP_TRY(hipMemcpyAsync(pol->sync_buf + O, db, (size_t)O * 4, hipMemcpyDeviceToDevice, st));
                TRY(dp_allreduce(pol, pol->sync_buf, 2 * O, st));
                dg = pol->sync_buf; db = pol->sync_buf + O; n_bn = N * pol->dp_world;
            }
            hipLaunchKernelGGL(bn_dx_kernel, dim3(blocks_for((size_t)N * O)), dim3(256), 0, st, obs, pol->da0, pol->bn_mean, pol->bn_invstd,
                               P + pol->o_bnw, dg, db, pol->dfeat, N, O, n_bn);
            dfeat = pol->dfeat;
        }
        PVR_LAUNCH_CHECK();
    }
    if (need_dobs) {
        // ---- conv stack backward (models.py:107-118) ----------------------------------------------------------------
        const int nf = d.conv_frames, F = N * nf;
        hipLaunchKernelGGL(conv_feat_kernel, dim3(blocks_for((size_t)N * 128 * nf)), dim3(256), 0, st, pol->dact[4], const_cast<float *>(dfeat), N, nf, 0);
        for (int l = 4; l >= 0; --l) {
            const int So = 64 >> (l + 1), Sin = So * 2;
            const size_t ne = (size_t)F * So * So * 32;
            if (l == 4) hipLaunchKernelGGL(elu_bwd_kernel, dim3(blocks_for(ne / 4)), dim3(256), 0, st, pol->dact[l], pol->act[l], ne / 4);   // (lower layers: fused into the input-gradient kernel above them)
            // bias gradient: per-block partial column sums over a row range, then a fixed-order sum of the partials
            {
                const int R = F * So * So, G = R < 1024 ? 1 : R >= (1 << 19) ? CONV_BG_BLOCKS_MAX : 256, per = (R + G - 1) / G;
                ColP c = {};
                c.X = pol->dact[l]; c.out0 = pol->bpartial; c.out1 = nullptr; c.R = per; c.C = 32;
                // one launch per partial would be slow: view [R][32] as G row blocks through a strided colreduce
                hipLaunchKernelGGL(rowblock_colsum_kernel, dim3(G), dim3(256), 0, st, pol->dact[l], pol->bpartial, R, per);
                TRY(colsum(pol->bpartial, Gd + pol->o_cb[l], nullptr, G, 32, st));
            }
            // weight gradient: the pixel range is cut over WAVES waves (each keeps a private [32][9*CP] partial, summed in wave order
            // afterwards).  A wave's loop is one dependent gather -> MFMA round per 4 pixels, so the launch is latency-bound: its
            // time is (pixels per wave) x (one memory latency) until the chip is full - 256 waves left three quarters of the SIMDs
            // idle and put 3200 rounds on each wave of the first layer (4.7 ms per call)
            const long long npix_l = (long long)F * So * So;
            const int WAVES = npix_l >= (1 << 19) ? CONV_WG_WAVES_MAX : npix_l >= (1 << 16) ? 1024 : 256;
            ConvWP w;
            w.in = l == 0 ? obs_in : (const void *)pol->act[l - 1]; w.dpre = pol->dact[l]; w.partial = pol->cpartial;
            w.F = F; w.Sin = Sin; w.So = So; w.nf = nf; w.waves = WAVES;
            const int cols = l == 0 ? 32 * 9 * 4 : 32 * 9 * 32;
            if (l == 0) hipLaunchKernelGGL(conv_s2_wgrad_kernel<3>, dim3(WAVES / 4), dim3(256), 0, st, w);
            else hipLaunchKernelGGL(conv_s2_wgrad_kernel<32>, dim3(WAVES / 4), dim3(256), 0, st, w);
            TRY(colsum(pol->cpartial, pol->cgpacked, nullptr, WAVES, cols, st));
            if (l == 0) hipLaunchKernelGGL(conv_pack_kernel<3>, dim3(5), dim3(256), 0, st, Gd + pol->o_cw[l], pol->cgpacked, 0);
            else hipLaunchKernelGGL(conv_pack_kernel<32>, dim3(36), dim3(256), 0, st, Gd + pol->o_cw[l], pol->cgpacked, 0);
            if (l > 0) {
                hipLaunchKernelGGL(conv_wt_kernel, dim3(36), dim3(256), 0, st, pol->wp[l], pol->wt[l]);
                ConvDP g;
                g.dpre = pol->dact[l]; g.Wt = pol->wt[l]; g.din = pol->dact[l - 1]; g.act_in = pol->act[l - 1]; g.F = F; g.Sin = Sin; g.So = So;
                const long long tiles = 4 * (((long long)F * (Sin / 2) * (Sin / 2) + 15) / 16);     // per parity class
                hipLaunchKernelGGL(conv_s2_dgrad_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, g);
            }
            PVR_LAUNCH_CHECK();
        }
    }
    if (dp_active(pol)) {
        TRY(dp_bucket(pol, Gd, 0, b_fc, 3, st));                 // conv stack + BatchNorm affine (everything in front of fc.1)
        // the loss of the global batch: mean of the per-rank means (equal rows per rank)
        PVR_HIP_TRY(hipEventRecord(pol->ev_ready[3], st));
        PVR_HIP_TRY(hipStreamWaitEvent(pol->comm_stream, pol->ev_ready[3], 0));
        TRY(dp_allreduce(pol, pol->stats, 1, pol->comm_stream));
        hipLaunchKernelGGL(scale_inplace_kernel, dim3(1), dim3(256), 0, pol->comm_stream, pol->stats, 1.0f / (float)pol->dp_world, 1LL);
        PVR_LAUNCH_CHECK();
        PVR_HIP_TRY(hipEventRecord(pol->ev_comm, pol->comm_stream));
        PVR_HIP_TRY(hipStreamWaitEvent(st, pol->ev_comm, 0));   // the compute stream re-joins: grads are the global-batch gradient from here on
    }
    pol->have_grads = true;
    return PVR_OK;
}

// lr is read from pol->stats[3] (set_lr)
static pvr_status apply_core(pvr_policy *pol, float *params, float *square_avg, const float *Gd, float alpha, float eps,
                             float max_grad_norm, hipStream_t st) {
    const size_t nt = (size_t)pol->n_train;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(1024), dim3(256), 0, st, Gd, nt, pol->partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, pol->partial, 1024, max_grad_norm, pol->stats);
    hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks_for(nt / 4, 8192)), dim3(256), 0, st, params, square_avg, Gd, pol->stats, nt / 4, alpha, eps);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

static pvr_status set_lr(pvr_policy *pol, float lr, hipStream_t st) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, pol->stats + 3, lr);
    PVR_LAUNCH_CHECK();
    return PVR_OK;
}

// forward (training mode, zero initial state: main_bc_2.py:207-209) + loss + backward into `grads`
static pvr_status loss_backward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs, const uint8_t *done,
                                const long long *actions, int T, int B, float *grads, hipStream_t st) {
    const int N = T * B;
    TRY(forward_core(pol, params, bn, obs, done, pol->zeros, pol->zeros, T, B, 1, actions, st));
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, pol->loss_row, N, 1.0f / (float)N, pol->stats);
    return backward_core(pol, params, obs, T, B, grads, st);
}

static void drop_graph(pvr_policy *pol) {
    if (pol->graph_exec) (void)hipGraphExecDestroy(pol->graph_exec);
    if (pol->graph) (void)hipGraphDestroy(pol->graph);
    pol->graph_exec = nullptr; pol->graph = nullptr;
}

extern "C" {

static pvr_status policy_create_impl(const pvr_policy_desc *desc, pvr_policy **out, bool host);
pvr_status pvr_policy_create(const pvr_policy_desc *desc, pvr_policy **out) { return policy_create_impl(desc, out, false); }
pvr_status pvr_policy_create_host(const pvr_policy_desc *desc, pvr_policy **out) { return policy_create_impl(desc, out, true); }

static pvr_status policy_create_impl(const pvr_policy_desc *desc, pvr_policy **out, bool host) {
    PVR_REQUIRE(desc && out, "pvr_policy_create: null argument");
    PVR_REQUIRE(!host || desc->conv_frames == 0, "pvr_policy_create_host: PolicyNetWithConv has no CPU plan (vector observations only)");
    PVR_REQUIRE(desc->hidden > 0 && desc->hidden % 1024 == 0, "hidden must be a positive multiple of 1024 (got %d)", desc->hidden);
    PVR_REQUIRE(desc->obs_size > 0 && desc->obs_size % 4 == 0, "obs_size must be a positive multiple of 4 (got %d)", desc->obs_size);
    PVR_REQUIRE(desc->num_actions > 0 && desc->num_actions <= 16, "num_actions must be in 1..16");
    PVR_REQUIRE(desc->max_t > 0 && desc->max_b > 0 && desc->max_b <= 64, "max_t > 0 and 0 < max_b <= 64 required");
    PVR_REQUIRE(desc->conv_frames >= 0 && (desc->conv_frames == 0 || desc->obs_size == 128 * desc->conv_frames),
                "conv_frames=%d needs obs_size=%d", desc->conv_frames, 128 * desc->conv_frames);
    pvr_policy *p = new pvr_policy();
    p->d = *desc;
    const int64_t O = desc->obs_size, H = desc->hidden, A = desc->num_actions;
    const int o = desc->batch_norm ? 1 : 0;
    char nm[64];
    if (desc->conv_frames > 0)
        for (int l = 0; l < 5; ++l) {                  // reference parameter order: feat_extract first (models.py:107-118)
            snprintf(nm, sizeof nm, "feat_extract.%d.weight", 2 * l); p->o_cw[l] = add_slot(p, nm, 32 * (l == 0 ? 3 : 32) * 9);
            snprintf(nm, sizeof nm, "feat_extract.%d.bias", 2 * l); p->o_cb[l] = add_slot(p, nm, 32);
        }
    if (desc->batch_norm) { p->o_bnw = add_slot(p, "fc.0.weight", O); p->o_bnb = add_slot(p, "fc.0.bias", O); }
    snprintf(nm, sizeof nm, "fc.%d.weight", o); p->o_fc1w = add_slot(p, nm, H * O);
    snprintf(nm, sizeof nm, "fc.%d.bias", o); p->o_fc1b = add_slot(p, nm, H);
    snprintf(nm, sizeof nm, "fc.%d.weight", o + 2); p->o_fc2w = add_slot(p, nm, H * H);
    snprintf(nm, sizeof nm, "fc.%d.bias", o + 2); p->o_fc2b = add_slot(p, nm, H);
    for (int l = 0; l < 2; ++l) {
        snprintf(nm, sizeof nm, "core.weight_ih_l%d", l); p->o_wih[l] = add_slot(p, nm, 4 * H * H);
        snprintf(nm, sizeof nm, "core.weight_hh_l%d", l); p->o_whh[l] = add_slot(p, nm, 4 * H * H);
        snprintf(nm, sizeof nm, "core.bias_ih_l%d", l); p->o_bih[l] = add_slot(p, nm, 4 * H);
        snprintf(nm, sizeof nm, "core.bias_hh_l%d", l); p->o_bhh[l] = add_slot(p, nm, 4 * H);
    }
    p->o_pw = add_slot(p, "policy.weight", A * H);
    p->o_pb = add_slot(p, "policy.bias", A);
    p->n_train = p->n_total;                       // the baseline head gets no gradient from the BC loss
    p->o_bw = add_slot(p, "baseline.weight", H);
    p->o_bb = add_slot(p, "baseline.bias", 1);

    if (host) {                                    // CPU plan: the layout above is all it shares with the HIP plan; no HIP call is made
        pvr::HostPolicyLayout lay;
        lay.O = (int)O; lay.H = (int)H; lay.A = (int)A; lay.bn = desc->batch_norm ? 1 : 0; lay.n_total = p->n_total; lay.n_train = p->n_train;
        lay.o_bnw = p->o_bnw; lay.o_bnb = p->o_bnb; lay.o_fc1w = p->o_fc1w; lay.o_fc1b = p->o_fc1b; lay.o_fc2w = p->o_fc2w; lay.o_fc2b = p->o_fc2b;
        for (int l = 0; l < 2; ++l) { lay.o_wih[l] = p->o_wih[l]; lay.o_whh[l] = p->o_whh[l]; lay.o_bih[l] = p->o_bih[l]; lay.o_bhh[l] = p->o_bhh[l]; }
        lay.o_pw = p->o_pw; lay.o_pb = p->o_pb; lay.o_bw = p->o_bw; lay.o_bb = p->o_bb;
        p->hostp = pvr::host_policy_new(lay);
        *out = p;
        return PVR_OK;
    }
    const size_t N = (size_t)desc->max_t * desc->max_b, B = desc->max_b;
    pvr_status s = PVR_OK;
#define A_(ptr, n) if (!s) s = dalloc(&p->ptr, (n))
    A_(a0, N * O); A_(bn_mean, O); A_(bn_invstd, O); A_(a1, N * H); A_(a2, N * H);
    for (int l = 0; l < 2; ++l) { A_(G[l], N * 4 * H); A_(Hs[l], N * H); A_(Cs[l], N * H); }
    A_(hprev, N * H); A_(nd, N); A_(zeros, 2 * B * H); A_(dc_carry, B * H); A_(rec_partial, 16 * B * H);
    A_(hprev1, N * H); A_(dc_carry1, B * H); A_(rec_partial1, 16 * B * H);
    A_(logits, N * 16); A_(baseline, N); A_(dlogits, N * 16); A_(loss_row, N); A_(stats, 4); A_(partial, 1024);
    A_(action, N); A_(dA, N * H); A_(dB, N * H); A_(da0, N * O); A_(grads, (size_t)p->n_train);
    if (desc->conv_frames > 0) {
        const size_t F = N * desc->conv_frames;
        for (int l = 0; l < 5; ++l) {
            const size_t So = 64 >> (l + 1);
            A_(act[l], F * So * So * 32); A_(dact[l], F * So * So * 32); A_(wp[l], 32 * 9 * 32); A_(wt[l], 32 * 9 * 32);
        }
        A_(feat, N * O); A_(dfeat, N * O); A_(cpartial, (size_t)CONV_WG_WAVES_MAX * 32 * 9 * 32); A_(cgpacked, 32 * 9 * 32); A_(bpartial, CONV_BG_BLOCKS_MAX * 32);
    }
    if (!s) {
        const size_t ob = desc->conv_frames > 0 ? N * 64 * 64 * 3 * desc->conv_frames : N * O * 4;
        uint8_t *raw = nullptr;
        s = dalloc(&raw, ob); p->in_obs = raw;
    }
    A_(in_done, N); A_(in_act, N);
    if (const char *e = getenv("PVR_POLICY_GRAPH")) p->use_graph = atoi(e) != 0;
    if (const char *e = getenv("PVR_POLICY_PIPELINE")) p->pipeline = atoi(e) != 0;
    if (const char *e = getenv("PVR_POLICY_PERSIST")) p->persist = atoi(e);      // 1: counter hand-off, 2: data-as-flag hand-off
    if (const char *e = getenv("PVR_POLICY_CHUNKWAVE")) p->chunkwave = atoi(e) != 0;
#ifdef PVR_EXPERIMENTS
    if (const char *e = getenv("PVR_POLICY_BWD_FUSED")) p->bwd_fused = atoi(e) != 0;
    if (p->bwd_fused) { A_(whhT[0], (size_t)4 * H * H); A_(whhT[1], (size_t)4 * H * H); }   // (only that path reads W_hh^T: 32 MB at H = 1024)
#endif
    A_(seq_counters, 64);
    if (!s) {
        // status word of the persistent recurrence: pinned host memory the GPU writes with a system-scope store
        hipError_t he = hipHostMalloc((void **)&p->status_host, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (he == hipSuccess) { memset(p->status_host, 0, 64); he = hipHostGetDevicePointer((void **)&p->status_dev, p->status_host, 0); }
        if (he != hipSuccess) { set_error("policy: status word allocation failed: %s", hipGetErrorString(he)); s = PVR_ERR_HIP; }
    }
    if (!s) {
        // co-residency of the persistent grid (H/4 blocks of 256 threads, all of which spin on each other): blocks per CU x CUs
        int dev = 0, cus = 0, per_cu = 0;
        hipError_t he = hipGetDevice(&dev);
        if (he == hipSuccess) he = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (he == hipSuccess) he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_fwd_seq_kernel, 256, 0);
        p->persist_fits = he == hipSuccess && (long long)per_cu * cus >= desc->hidden / 4;
#ifdef PVR_EXPERIMENTS
        int per_cu_b = 0;
        if (he == hipSuccess) he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, lstm_bwd_seq_kernel, 256, 0);
        p->persist_bwd_fits = he == hipSuccess && (long long)per_cu_b * cus >= 2 * 256 && desc->hidden == 1024;
#endif
        (void)hipGetLastError();
    }
#ifdef PVR_EXPERIMENTS
    if (const char *e = getenv("PVR_POLICY_PERSIST_BWD")) p->persist_bwd = atoi(e) != 0;
#endif
    if (!s && p->persist_bwd && p->persist_bwd_fits && p->persist == 2) {
        for (int l = 0; l < 2 && !s; ++l) {
            const size_t px = (size_t)2 * 16 * B * desc->hidden;
            if (hipMalloc((void **)&p->dGx[l], N * 4 * desc->hidden * sizeof(float)) != hipSuccess ||
                hipMalloc((void **)&p->Px[l], px * sizeof(float)) != hipSuccess ||
                hipMemset(p->Px[l], 0xFF, px * sizeof(float)) != hipSuccess) {              // the ring starts fully armed
                set_error("policy: allocation of the persistent-BPTT hand-off buffers failed"); s = PVR_ERR_HIP;
            }
        }
    }
    if (!s && p->pipeline) {
        hipError_t he = hipStreamCreateWithFlags(&p->lane_a, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&p->lane_b, hipStreamNonBlocking);
        hipEvent_t *evs[3] = {&p->ev_fork, &p->ev_join_a, &p->ev_join_b};
        for (auto ev : evs) if (he == hipSuccess) he = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        for (int i = 0; i < 8; ++i) if (he == hipSuccess) he = hipEventCreateWithFlags(&p->ev_chunk[i], hipEventDisableTiming);
        if (he != hipSuccess) { set_error("policy: stream/event creation failed: %s", hipGetErrorString(he)); s = PVR_ERR_HIP; }
    }
#undef A_
    if (!s && hipDeviceSynchronize() != hipSuccess) { set_error("policy: device sync failed"); s = PVR_ERR_HIP; }
    if (s) { pvr_policy_destroy(p); return s; }
    *out = p;
    return PVR_OK;
}

void pvr_policy_destroy(pvr_policy *p) {
    if (!p) return;
    if (p->hostp) { pvr::host_policy_free(p->hostp); delete p; return; }
    void *ptrs[] = {p->a0, p->bn_mean, p->bn_invstd, p->a1, p->a2, p->G[0], p->G[1], p->Hs[0], p->Hs[1], p->Cs[0], p->Cs[1],
                    p->hprev, p->nd, p->zeros, p->dc_carry, p->rec_partial, p->logits, p->baseline, p->dlogits,
                    p->loss_row, p->stats, p->partial, p->action, p->dA, p->dB, p->da0, p->grads, p->feat, p->dfeat,
                    p->cpartial, p->cgpacked, p->bpartial, p->in_obs, p->in_done, p->in_act, p->hprev1, p->dc_carry1, p->rec_partial1, p->seq_counters, p->whhT[0], p->whhT[1]};
    if (p->lane_a) (void)hipStreamDestroy(p->lane_a);
    if (p->lane_b) (void)hipStreamDestroy(p->lane_b);
    for (hipEvent_t ev : {p->ev_fork, p->ev_join_a, p->ev_join_b}) if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < 8; ++i) if (p->ev_chunk[i]) (void)hipEventDestroy(p->ev_chunk[i]);
    drop_graph(p);
    if (p->status_host) (void)hipHostFree(p->status_host);
    for (int l = 0; l < 2; ++l) { if (p->dGx[l]) (void)hipFree(p->dGx[l]); if (p->Px[l]) (void)hipFree(p->Px[l]); }
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    if (p->comm_stream) (void)hipStreamDestroy(p->comm_stream);
    for (hipEvent_t ev : p->ev_ready) if (ev) (void)hipEventDestroy(ev);
    if (p->ev_comm) (void)hipEventDestroy(p->ev_comm);
    for (void *q : {(void *)p->sync_buf, (void *)p->scratch.splitk, (void *)p->scratch.col}) if (q) (void)hipFree(q);
    for (void *q : p->scratch.retired) (void)hipFree(q);
    for (void *q : ptrs) if (q) (void)hipFree(q);
    for (int l = 0; l < 5; ++l) { void *c[] = {p->act[l], p->dact[l], p->wp[l], p->wt[l]}; for (void *q : c) if (q) (void)hipFree(q); }
    delete p;
}

pvr_status pvr_policy_set_data_parallel(pvr_policy *pol, int32_t world_size, int32_t sync_bn, pvr_allreduce_fn fn, void *user) {
    PVR_REQUIRE(pol, "pvr_policy_set_data_parallel: null policy");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_set_data_parallel: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    const bool on = fn && world_size > 1;
    if (on && !pol->comm_stream) {
        PVR_HIP_TRY(hipStreamCreateWithFlags(&pol->comm_stream, hipStreamNonBlocking));
        for (auto &ev : pol->ev_ready) PVR_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        PVR_HIP_TRY(hipEventCreateWithFlags(&pol->ev_comm, hipEventDisableTiming));
        PVR_HIP_TRY(hipMalloc((void **)&pol->sync_buf, (size_t)2 * pol->d.obs_size * sizeof(float)));
    }
    pol->dp_fn = on ? fn : nullptr; pol->dp_user = on ? user : nullptr; pol->dp_world = on ? world_size : 1;
    pol->dp_sync_bn = on && sync_bn && pol->d.batch_norm;
    return PVR_OK;
}

// The persistent recurrence's spin ran out in an earlier launch of this handle: report it once, stay on per-step launches afterwards.
pvr_status pvr_policy_set_action_sampling(pvr_policy *pol, int32_t on, uint64_t seed) {
    PVR_REQUIRE(pol, "pvr_policy_set_action_sampling: null policy");
    if (pol->hostp) { pvr::host_policy_set_sampling(pol->hostp, on, seed); return PVR_OK; }
    pol->sample_on = on != 0; pol->sample_seed = seed; pol->sample_call = 0;
    return PVR_OK;
}

// the stream's position (number of sampling forwards so far) - read before a handle is destroyed, restored on its replacement, so that a handle
// rebuilt because T or B grew (or after .to()) continues the noise stream instead of replaying it
uint64_t pvr_policy_action_sampling_call(const pvr_policy *pol) {
    if (!pol) return 0;
    return pol->hostp ? pvr::host_policy_sampling_call(pol->hostp) : pol->sample_call;
}
pvr_status pvr_policy_set_action_sampling_call(pvr_policy *pol, uint64_t call) {
    PVR_REQUIRE(pol, "pvr_policy_set_action_sampling_call: null policy");
    if (pol->hostp) { pvr::host_policy_set_sampling_call(pol->hostp, call); return PVR_OK; }
    pol->sample_call = call;
    return PVR_OK;
}

pvr_status pvr_policy_status(pvr_policy *pol) {
    PVR_REQUIRE(pol, "pvr_policy_status: null policy");
    if (pol->hostp) return PVR_OK;
    const unsigned w = pol->status_host ? __atomic_load_n(pol->status_host, __ATOMIC_ACQUIRE) : 0u;
    if (!w) return PVR_OK;
    __atomic_store_n(pol->status_host, 0u, __ATOMIC_RELEASE);
    pol->persist_tripped = 1;
    set_error("policy: the persistent LSTM recurrence gave up waiting for a peer block (status 0x%x): the hidden states, logits and - after "
              "a training step - the parameters of that call are NaN.  Its grid was not co-resident (another process or stream holding "
              "CUs?); this handle uses per-step launches from now on (PVR_POLICY_PERSIST=0 selects them from the start)", w);
    return PVR_ERR_TIMEOUT;
}

pvr_status pvr_policy_debug_drop_block(pvr_policy *pol, int32_t block) {
    PVR_REQUIRE(pol, "pvr_policy_debug_drop_block: null policy");
    pol->debug_drop_block = block;
    return PVR_OK;
}

int32_t pvr_policy_recurrence_mode(const pvr_policy *pol) {
    if (!pol) return -1;
    if (pol->hostp) return 0;
    return (!pol->persist_fits || pol->persist_tripped) ? 0 : (pol->persist == 2 && pol->use_graph) ? 0 : pol->persist;
}

int64_t pvr_policy_param_count(const pvr_policy *p) { return p ? p->n_total : 0; }
int64_t pvr_policy_trainable_count(const pvr_policy *p) { return p ? p->n_train : 0; }

int64_t pvr_policy_param_offset(const pvr_policy *p, const char *name, int64_t *numel) {
    if (!p || !name) return -1;
    auto it = p->slots.find(name);
    if (it == p->slots.end()) return -1;
    if (numel) *numel = it->second.numel;
    return it->second.off;
}

pvr_status pvr_policy_forward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs,
                              const uint8_t *done, const float *h0, const float *c0, int32_t T, int32_t B, int32_t training,
                              float *logits, float *baseline, int64_t *action, float *h_out, float *c_out, void *hip_stream) {
    PVR_REQUIRE(pol && params && obs && done, "pvr_policy_forward: null argument");
    if (pol->hostp)                                              // CPU plan: every pointer is a host pointer, hip_stream is ignored
        return pvr::host_policy_forward(pol->hostp, params, bn, (const float *)obs, done, h0, c0, T, B, training, logits, baseline, action, h_out, c_out);
    TraceScope trace("pvr_policy_forward");
    ScratchScope scratch_scope(pol);
    TRY(pvr_policy_status(pol));                                 // (sticky: a time-out of an earlier launch surfaces here)
    PVR_REQUIRE(T > 0 && T <= pol->d.max_t && B > 0 && B <= pol->d.max_b, "T=%d B=%d outside the workspace (%d,%d)", T, B, pol->d.max_t, pol->d.max_b);
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = T * B, H = pol->d.hidden, A = pol->d.num_actions;
    const float *h_in = h0 ? h0 : pol->zeros, *c_in = c0 ? c0 : pol->zeros;
    TRY(forward_core(pol, params, bn, obs, done, h_in, c_in, T, B, training, nullptr, st));
    pol->fwd_T = training ? T : 0; pol->fwd_B = training ? B : 0;
    if (logits) PVR_HIP_TRY(hipMemcpyAsync(logits, pol->logits, (size_t)N * A * 4, hipMemcpyDeviceToDevice, st));
    if (baseline) PVR_HIP_TRY(hipMemcpyAsync(baseline, pol->baseline, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    if (action) PVR_HIP_TRY(hipMemcpyAsync(action, pol->action, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
    for (int l = 0; l < 2; ++l) {
        if (h_out) PVR_HIP_TRY(hipMemcpyAsync(h_out + (size_t)l * B * H, pol->Hs[l] + (size_t)(T - 1) * B * H, (size_t)B * H * 4, hipMemcpyDeviceToDevice, st));
        if (c_out) PVR_HIP_TRY(hipMemcpyAsync(c_out + (size_t)l * B * H, pol->Cs[l] + (size_t)(T - 1) * B * H, (size_t)B * H * 4, hipMemcpyDeviceToDevice, st));
    }
    return PVR_OK;
}

pvr_status pvr_policy_backward(pvr_policy *pol, const float *params, const pvr_policy_bn *bn, const void *obs, const uint8_t *done,
                               const int64_t *actions, int32_t T, int32_t B, float *grads, float *stats_out, float *logits_out,
                               void *hip_stream) {
    PVR_REQUIRE(pol && params && obs && done && actions && grads, "pvr_policy_backward: null argument");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_backward: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    TraceScope trace("pvr_policy_backward");
    ScratchScope scratch_scope(pol);
    TRY(pvr_policy_status(pol));                                 // (sticky: a time-out of an earlier launch surfaces here)
    PVR_REQUIRE(T > 0 && T <= pol->d.max_t && B > 0 && B <= pol->d.max_b, "T=%d B=%d outside the workspace (%d,%d)", T, B, pol->d.max_t, pol->d.max_b);
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = T * B, A = pol->d.num_actions;
    TRY(loss_backward(pol, params, bn, obs, done, (const long long *)actions, T, B, grads, st));
    if (logits_out) PVR_HIP_TRY(hipMemcpyAsync(logits_out, pol->logits, (size_t)N * A * 4, hipMemcpyDeviceToDevice, st));
    if (stats_out) PVR_HIP_TRY(hipMemcpyAsync(stats_out, pol->stats, sizeof(float), hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

pvr_status pvr_policy_apply(pvr_policy *pol, float *params, float *square_avg, const float *grads, float lr, float alpha, float eps,
                            float max_grad_norm, float *stats_out, void *hip_stream) {
    PVR_REQUIRE(pol && params && square_avg && grads, "pvr_policy_apply: null argument");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_apply: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    PVR_REQUIRE(((uintptr_t)grads & 15) == 0, "pvr_policy_apply: grads must be 16-byte aligned (the norm kernel reads float4s)");
    ScratchScope scratch_scope(pol);
    hipStream_t st = (hipStream_t)hip_stream;
    TRY(set_lr(pol, lr, st));
    TRY(apply_core(pol, params, square_avg, grads, alpha, eps, max_grad_norm, st));
    if (stats_out) PVR_HIP_TRY(hipMemcpyAsync(stats_out + 1, pol->stats + 1, sizeof(float), hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

pvr_status pvr_policy_backward_dlogits(pvr_policy *pol, const float *params, const void *obs, const float *dlogits, int32_t T, int32_t B,
                                       float *grads, void *hip_stream) {
    PVR_REQUIRE(pol && params && obs && dlogits && grads, "pvr_policy_backward_dlogits: null argument");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_backward_dlogits: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    ScratchScope scratch_scope(pol);
    TRY(pvr_policy_status(pol));
    if (pol->fwd_T != T || pol->fwd_B != B) {
        set_error("pvr_policy_backward_dlogits: needs the activations of a training-mode pvr_policy_forward with T=%d, B=%d (last: %d, %d)", T, B,
                  pol->fwd_T, pol->fwd_B);
        return PVR_ERR_STATE;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = T * B;
    hipLaunchKernelGGL(dlogits_pad_kernel, dim3((N * 16 + 255) / 256), dim3(256), 0, st, dlogits, pol->dlogits, N, pol->d.num_actions);
    PVR_LAUNCH_CHECK();
    pol->fwd_T = pol->fwd_B = 0;                                  // (backward reuses activation buffers as scratch: one backward per forward)
    return backward_core(pol, params, obs, T, B, grads, st);
}

pvr_status pvr_policy_apply_momentum(pvr_policy *pol, float *params, float *square_avg, float *momentum_buf, const float *grads, float lr,
                                     float alpha, float eps, float momentum, float max_grad_norm, float *stats_out, void *hip_stream) {
    PVR_REQUIRE(pol && params && square_avg && momentum_buf && grads, "pvr_policy_apply_momentum: null argument");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_apply_momentum: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    PVR_REQUIRE(((uintptr_t)grads & 15) == 0, "pvr_policy_apply_momentum: grads must be 16-byte aligned (the norm kernel reads float4s)");
    hipStream_t st = (hipStream_t)hip_stream;
    TRY(set_lr(pol, lr, st));
    const size_t nt = (size_t)pol->n_train;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(1024), dim3(256), 0, st, grads, nt, pol->partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, pol->partial, 1024, max_grad_norm, pol->stats);
    hipLaunchKernelGGL(rmsprop_momentum_kernel, dim3(blocks_for(nt / 4, 8192)), dim3(256), 0, st, params, square_avg, momentum_buf, grads, pol->stats,
                       nt / 4, alpha, eps, momentum);
    PVR_LAUNCH_CHECK();
    if (stats_out) PVR_HIP_TRY(hipMemcpyAsync(stats_out + 1, pol->stats + 1, sizeof(float), hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

pvr_status pvr_policy_apply_adam(pvr_policy *pol, float *params, float *exp_avg, float *exp_avg_sq, const float *grads, float lr, float beta1,
                                 float beta2, float eps, int64_t step, float max_grad_norm, float *stats_out, void *hip_stream) {
    PVR_REQUIRE(pol && params && exp_avg && exp_avg_sq && grads && step >= 1, "pvr_policy_apply_adam: null argument or step < 1");
    PVR_REQUIRE(!pol->hostp, "pvr_policy_apply_adam: not part of the host (CPU) plan - it carries pvr_policy_forward and pvr_policy_step");
    PVR_REQUIRE(((uintptr_t)grads & 15) == 0, "pvr_policy_apply_adam: grads must be 16-byte aligned (the norm kernel reads float4s)");
    hipStream_t st = (hipStream_t)hip_stream;
    TRY(set_lr(pol, lr, st));
    const size_t nt = (size_t)pol->n_train;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(1024), dim3(256), 0, st, grads, nt, pol->partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, pol->partial, 1024, max_grad_norm, pol->stats);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(nt / 4, 8192)), dim3(256), 0, st, params, exp_avg, exp_avg_sq, grads, pol->stats, nt / 4, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2));
    PVR_LAUNCH_CHECK();
    if (stats_out) PVR_HIP_TRY(hipMemcpyAsync(stats_out + 1, pol->stats + 1, sizeof(float), hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

pvr_status pvr_policy_step(pvr_policy *pol, float *params, float *square_avg, const pvr_policy_bn *bn, const void *obs,
                           const uint8_t *done, const int64_t *actions, int32_t T, int32_t B, float lr, float alpha, float eps,
                           float max_grad_norm, float *stats_out, float *logits_out, void *hip_stream) {
    PVR_REQUIRE(pol && params && square_avg && obs && done && actions, "pvr_policy_step: null argument");
    if (pol->hostp)
        return pvr::host_policy_step(pol->hostp, params, square_avg, bn, (const float *)obs, done, actions, T, B, lr, alpha, eps, max_grad_norm, stats_out, logits_out);
    TraceScope trace("pvr_policy_step");
    ScratchScope scratch_scope(pol);
    TRY(pvr_policy_status(pol));                                 // (sticky: a time-out of an earlier launch surfaces here)
    PVR_REQUIRE(T > 0 && T <= pol->d.max_t && B > 0 && B <= pol->d.max_b, "T=%d B=%d outside the workspace (%d,%d)", T, B, pol->d.max_t, pol->d.max_b);
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = T * B, A = pol->d.num_actions;
    TRY(set_lr(pol, lr, st));
    if (!pol->use_graph || dp_active(pol)) {    // (the collective callback cannot run inside a stream capture)
        TRY(loss_backward(pol, params, bn, obs, done, (const long long *)actions, T, B, pol->grads, st));
        TRY(apply_core(pol, params, square_avg, pol->grads, alpha, eps, max_grad_norm, st));
    } else {
        // fixed addresses for the captured launches
        const size_t obs_bytes = pol->d.conv_frames > 0 ? (size_t)N * 64 * 64 * 3 * pol->d.conv_frames : (size_t)N * pol->d.obs_size * 4;
        PVR_HIP_TRY(hipMemcpyAsync(pol->in_obs, obs, obs_bytes, hipMemcpyDeviceToDevice, st));
        PVR_HIP_TRY(hipMemcpyAsync(pol->in_done, done, (size_t)N, hipMemcpyDeviceToDevice, st));
        PVR_HIP_TRY(hipMemcpyAsync(pol->in_act, actions, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
        pvr_policy::StepKey key;
        key.params = params; key.sq = square_avg;
        if (bn) { key.bn_rm = bn->running_mean; key.bn_rv = bn->running_var; key.bn_nbt = bn->num_batches_tracked; }
        key.T = T; key.B = B; key.alpha = alpha; key.eps = eps; key.mgn = max_grad_norm;
        if (pol->graph_exec && key == pol->graph_key) {
            PVR_HIP_TRY(hipGraphLaunch(pol->graph_exec, st));
        } else if (key == pol->eager_key) {
            // second iteration with these arguments (the first ran eagerly, so every lazy initialisation is done): capture
            drop_graph(pol);
            if (!pol->cap_stream) PVR_HIP_TRY(hipStreamCreateWithFlags(&pol->cap_stream, hipStreamNonBlocking));
            hipStream_t cs = pol->cap_stream;
            PVR_HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            pvr_status s = loss_backward(pol, params, bn, pol->in_obs, pol->in_done, pol->in_act, T, B, pol->grads, cs);
            if (!s) s = apply_core(pol, params, square_avg, pol->grads, alpha, eps, max_grad_norm, cs);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(cs, &g);
            if (s) { if (g) (void)hipGraphDestroy(g); return s; }
            if (e != hipSuccess || !g) { set_error("pvr_policy_step: stream capture failed: %s", hipGetErrorString(e)); return PVR_ERR_HIP; }
            pol->graph = g;
            PVR_HIP_TRY(hipGraphInstantiate(&pol->graph_exec, g, nullptr, nullptr, 0));
            pol->graph_key = key;
            PVR_HIP_TRY(hipGraphLaunch(pol->graph_exec, st));
        } else {
            TRY(loss_backward(pol, params, bn, pol->in_obs, pol->in_done, pol->in_act, T, B, pol->grads, st));
            TRY(apply_core(pol, params, square_avg, pol->grads, alpha, eps, max_grad_norm, st));
            pol->eager_key = key;
        }
    }
    pol->have_grads = true;
    if (logits_out) PVR_HIP_TRY(hipMemcpyAsync(logits_out, pol->logits, (size_t)N * A * 4, hipMemcpyDeviceToDevice, st));
    if (stats_out) PVR_HIP_TRY(hipMemcpyAsync(stats_out, pol->stats, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return PVR_OK;
}

pvr_status pvr_policy_last_grads(pvr_policy *pol, float *grads_out, void *hip_stream) {
    PVR_REQUIRE(pol && grads_out, "pvr_policy_last_grads: null argument");
    if (pol->hostp) return pvr::host_policy_last_grads(pol->hostp, grads_out);
    if (!pol->have_grads) { set_error("no training step has run"); return PVR_ERR_STATE; }
    PVR_HIP_TRY(hipMemcpyAsync(grads_out, pol->grads, (size_t)pol->n_train * 4, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    return PVR_OK;
}

pvr_status pvr_op_gemm_f32(const float *A, const float *B, const float *bias, float *C, int32_t M, int32_t N, int32_t K,
                           int32_t a_km, int32_t b_kn, int32_t relu, void *hip_stream) {
    PVR_REQUIRE(A && B && C, "pvr_op_gemm_f32: null pointer");
    ScratchScope scratch_scope(nullptr);
    return gemm(A, B, bias, nullptr, C, M, N, K, a_km != 0, b_kn != 0, relu, (hipStream_t)hip_stream);
}

pvr_status pvr_debug_set_gemm_mode(int32_t mode) {
    PVR_REQUIRE(mode >= -1 && mode <= 3, "pvr_debug_set_gemm_mode: mode must be -1 (default), 0 (fp32 MFMA), 1 / 2 / 3 (split-bf16 MFMA: automatic / 128 / 64 tiles)");
#ifndef PVR_EXPERIMENTS
    PVR_REQUIRE(mode <= 0, "pvr_debug_set_gemm_mode: the split-bf16 GEMM is an experiment kernel; this library was built without it (make EXPERIMENTS=1)");
#endif
    g_gemm_mode = mode;
    return PVR_OK;
}

}  // extern "C"
