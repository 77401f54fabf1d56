// The data-parallel hook of the BC path driven from C++ with RCCL itself (no Python, no torch.distributed): what a host written against
// include/pvr_policy.h does for BASELINE config 4 - one communicator per rank, ncclAllReduce handed to the library as a plain C function
// pointer through pvr_policy_set_data_parallel (the reference has no counterpart: main_bc_finetune.py:167-208 is single-GPU).
//
//   hipcc -O2 -std=c++17 -I include tests/native/rccl_binding.cpp -L pvr_habitat_amd/lib -lpvr_hip -lrccl -Wl,-rpath,$PWD/pvr_habitat_amd/lib -o tests/native/rccl_binding
//
// One GPU is enough to run it: the communicator then has one rank (all-reduce = identity), the library is told world_size = 2, so the
// gradient of a backward with the hook must be exactly HALF the gradient of the same backward without it, the loss likewise, and the
// thunk must have been called for the library's four gradient buckets (+ the loss).  With two GPUs visible (RANKS=2) two host threads
// run one rank each and the averaged gradients of the two ranks must agree bit for bit.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "pvr_policy.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

struct Rank { ncclComm_t comm; std::atomic<int> calls{0}; std::atomic<long long> floats{0}; };

// the hook: in-place SUM all-reduce of `count` fp32 values on the library's stream
static int32_t allreduce_thunk(void *buf, int64_t count, void *hip_stream, void *user) {
    Rank *r = (Rank *)user;
    r->calls++; r->floats += count;
    return ncclAllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, r->comm, (hipStream_t)hip_stream) == ncclSuccess ? 0 : 1;
}

static std::vector<float> run_rank(int dev, Rank *rk, int world_for_library, bool hook, float *loss_out) {
    CHECK(hipSetDevice(dev) == hipSuccess);
    const int T = 6, B = 4, OBS = 256, A = 3;
    pvr_policy_desc d = {OBS, 1024, A, 1, T, B, 0};
    pvr_policy *pol = nullptr;
    CHECK(pvr_policy_create(&d, &pol) == PVR_OK);
    const int64_t np = pvr_policy_param_count(pol), nt = pvr_policy_trainable_count(pol);
    std::vector<float> hp(np), hobs((size_t)T * B * OBS);
    unsigned s = 12345u;                                   // same parameters and (per rank different) observations everywhere
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : hp) v = rnd() * 0.05f;
    s += 77u * dev;
    for (auto &v : hobs) v = rnd();
    std::vector<long long> hact((size_t)T * B);
    for (auto &v : hact) v = (long long)(fabsf(rnd()) * 5.9f) % A;
    std::vector<unsigned char> hdone((size_t)T * B, 0);
    float *p, *g, *obs, *stats, *rm, *rv; long long *act, *nbt; unsigned char *done;
    CHECK(hipMalloc(&p, np * 4) == hipSuccess && hipMalloc(&g, nt * 4) == hipSuccess && hipMalloc(&obs, hobs.size() * 4) == hipSuccess);
    CHECK(hipMalloc(&stats, 8) == hipSuccess && hipMalloc(&rm, OBS * 4) == hipSuccess && hipMalloc(&rv, OBS * 4) == hipSuccess);
    CHECK(hipMalloc(&act, hact.size() * 8) == hipSuccess && hipMalloc(&nbt, 8) == hipSuccess && hipMalloc(&done, hdone.size()) == hipSuccess);
    hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice); hipMemcpy(obs, hobs.data(), hobs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(act, hact.data(), hact.size() * 8, hipMemcpyHostToDevice); hipMemcpy(done, hdone.data(), hdone.size(), hipMemcpyHostToDevice);
    hipMemset(rm, 0, OBS * 4); hipMemset(nbt, 0, 8);
    std::vector<float> ones(OBS, 1.f); hipMemcpy(rv, ones.data(), OBS * 4, hipMemcpyHostToDevice);
    pvr_policy_bn bn = {rm, rv, (int64_t *)nbt};
    hipStream_t st; CHECK(hipStreamCreate(&st) == hipSuccess);
    if (hook) CHECK(pvr_policy_set_data_parallel(pol, world_for_library, 0, allreduce_thunk, rk) == PVR_OK);
    const pvr_status rc = pvr_policy_backward(pol, p, &bn, obs, done, (const int64_t *)act, T, B, g, stats, nullptr, st);
    if (rc != PVR_OK) { char msg[512]; pvr_last_error(msg, sizeof msg); fprintf(stderr, "pvr_policy_backward: %s\n", msg); exit(1); }
    CHECK(hipStreamSynchronize(st) == hipSuccess);
    std::vector<float> hg(nt); float hs[2];
    hipMemcpy(hg.data(), g, nt * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, stats, 8, hipMemcpyDeviceToHost);
    *loss_out = hs[0];
    pvr_policy_destroy(pol);
    hipStreamDestroy(st);
    for (void *q : {(void *)p, (void *)g, (void *)obs, (void *)stats, (void *)rm, (void *)rv, (void *)act, (void *)nbt, (void *)done}) hipFree(q);
    return hg;
}

int main() {
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev >= 1);
    const char *e = getenv("RANKS");
    const int ranks = e ? atoi(e) : 1;
    CHECK(ranks >= 1 && ranks <= ndev && ranks <= 8);
    int devs[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    ncclComm_t comms[8];
    CHECK(ncclCommInitAll(comms, ranks, devs) == ncclSuccess);
    Rank rk[8];
    for (int r = 0; r < ranks; ++r) rk[r].comm = comms[r];
    float loss_plain = 0.f, loss[8];
    std::vector<float> plain = run_rank(0, nullptr, 1, false, &loss_plain);     // no hook: the local gradient
    std::vector<float> g[8];
    const int world = ranks == 1 ? 2 : ranks;              // one rank: pretend to be half of a world of two (sum of one, divided by two)
    std::vector<std::thread> th;
    for (int r = 0; r < ranks; ++r) th.emplace_back([&, r]() { g[r] = run_rank(r, &rk[r], world, true, &loss[r]); });
    for (auto &t : th) t.join();
    CHECK(rk[0].calls >= 4);                               // four gradient buckets (+ the loss)
    CHECK(rk[0].floats >= (long long)plain.size());
    double n2 = 0;
    for (float v : plain) n2 += (double)v * v;
    CHECK(n2 > 0 && std::isfinite(n2) && std::isfinite(loss_plain));
    if (ranks == 1) {
        for (size_t i = 0; i < plain.size(); ++i) CHECK(g[0][i] == 0.5f * plain[i]);
        CHECK(loss[0] == 0.5f * loss_plain);
    } else {
        for (int r = 1; r < ranks; ++r) { CHECK(g[r] == g[0]); CHECK(loss[r] == loss[0]); }
    }
    for (int r = 0; r < ranks; ++r) ncclCommDestroy(comms[r]);
    printf("rccl_binding: ok (%d rank(s), %d all-reduce calls, %lld floats, |g| = %.4e, loss %.6f)\n", ranks, rk[0].calls.load(), rk[0].floats.load(), sqrt(n2), loss_plain);
    return 0;
}
