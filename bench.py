#!/usr/bin/env python3
"""Headline benchmark: frames/sec embedded by the frozen ResNet50 (MoCo-v2 layout) PVR encoder.

BASELINE.json metric "frames/sec embedded (ResNet50, 256x256)", workload = configs[1]:
ResNet50 MoCo-v2 frozen, 256x256 uint8 frames, batch 256, bf16, synthetic frames + synthetic weights
(no network).  A step = one pass of the hot path (resize/crop/normalise + 53 convs + pool) over one
batch of 256 frames that is already resident in HBM.  Frames shard across ranks with no collective
(SURVEY 8e): weak scaling, value = frames all ranks embedded / max-over-ranks time.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_FRAME = 8.174           # 2 x 4.0871 GMAC, 53 convs @ 224x224
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16/f16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3           # f32-input MFMA (= the fp32 vector rate), reference-precision mode only


def conv_algorithmic_bytes(n, names=None):
    """Activation bytes (16-bit) the conv launches of one n-frame chunk must move: input + output (+ residual) of every
    launch of the plan `names` (HipResNet50.op_names(); a fused bottleneck tail 'a.conv2+conv3+b.conv1' reads t1 and the
    residual and writes y and the next block's t1, nothing else).  names=None: one launch per convolution."""
    def geom(layer, block):
        planes = 64 << (layer - 1)
        hw_out = 56 >> (layer - 1)
        hw_in = hw_out * 2 if (block == 0 and layer > 1) else hw_out
        inpl = (64 if layer == 1 else planes * 2) if block == 0 else planes * 4
        return planes, hw_in, hw_out, inpl
    if names is None:
        names = []
        for li, nb in enumerate((3, 4, 6, 3)):
            for bi in range(nb):
                names += ['layer%d.%d.conv1' % (li + 1, bi), 'layer%d.%d.conv2' % (li + 1, bi)]
                if bi == 0:
                    names.append('layer%d.%d.downsample.0' % (li + 1, bi))
                names.append('layer%d.%d.conv3' % (li + 1, bi))
    tot = 0
    for nm in names:
        parts = nm.split('+')
        head = parts[0].split('.')
        layer, block, kind = int(head[0][5:]), int(head[1]), head[2]
        planes, hi, ho, inpl = geom(layer, block)
        if len(parts) > 1:                                   # conv2 -> conv3 (+ residual) [-> next conv1]
            tot += hi * hi * planes + 2 * ho * ho * planes * 4
            if len(parts) > 2:
                nh = parts[2].split('.')
                tot += ho * ho * geom(int(nh[0][5:]), int(nh[1]))[0]
        elif kind == 'conv1':
            tot += hi * hi * inpl + hi * hi * planes
        elif kind == 'conv2':
            tot += hi * hi * planes + ho * ho * planes
        elif kind == 'downsample':
            tot += hi * hi * inpl + ho * ho * planes * 4
        elif kind == 'conv3':
            tot += ho * ho * planes + 2 * ho * ho * planes * 4
    return tot * 2 * n


def cpu_baseline(sd, frames_u8, budget_s=12.0):
    """The oracle (torch fp32 eager restatement of the reference path) timed on this box's host cores,
    batch 64 like the reference's 32 obs x 2 frames (save_embedded_obs.py:151-153)."""
    from oracle import encoder_oracle as eo
    # torch CPU convs stop scaling (and regress) at very high thread counts: probe a few, keep the fastest
    best, best_t = 1, float('inf')
    for th in sorted({min(os.cpu_count() or 1, t) for t in (16, 32, 64, 128)}):
        torch.set_num_threads(th)
        eo.embed(sd, frames_u8[:8], 'conv5')                  # warm-up at this thread count
        t0 = time.perf_counter(); eo.embed(sd, frames_u8[:16], 'conv5'); dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = th, dt
    torch.set_num_threads(best)
    done, t0 = 0, time.perf_counter()
    while True:
        eo.embed(sd, frames_u8[:64], 'conv5')
        done += 64
        el = time.perf_counter() - t0
        if el > budget_s or done >= 64 * 8:
            break
    return dict(value=round(done / el, 2), unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample='%d synthetic 256x256 frames in batches of 64, torch fp32 eager oracle, %.1f s' % (done, el))


BC_GFLOP_PER_STEP = 211.4         # PolicyNet T=100,B=16,obs 4096: 3 x fwd of 22.02 MMAC x 1600


def bc_bench(steps, warmup, with_cpu):
    """Second half of the BASELINE metric: BC steps/sec (main_bc_2.py:186-227 iteration, slurm_bc.py:121-128
    configuration T=100, B=16, obs 4096, BatchNorm on, RMSprop) on one GPU, batches resident in HBM."""
    from pvr_habitat_amd import synth
    from pvr_habitat_amd.models import PolicyNet, HipRMSprop
    T, B, O, A = 100, 16, 4096, 3
    m = PolicyNet((O,), A, True, max_unroll=T, max_batch=B)
    sd = synth.policy_state_dict(1, O, A, True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to('cuda').train()
    opt = HipRMSprop(m, max_epochs=10 ** 6)
    obs, done, act = synth.bc_batches(1, T, B, O, A, 2)
    obs_d, done_d, act_d = torch.from_numpy(obs).cuda(), torch.from_numpy(done).cuda(), torch.from_numpy(act).cuda()
    for i in range(warmup):
        opt.scheduler_step(); opt.step(obs_d[i % 2], done_d[i % 2], act_d[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        opt.scheduler_step(); loss, gn = opt.step(obs_d[i % 2], done_d[i % 2], act_d[i % 2])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res = {'metric': 'BC steps/sec (PolicyNet T=100 B=16 obs=4096 BN, fp32)', 'value': round(steps / el, 2), 'unit': 'steps/s',
           'ms_per_step': round(el / steps * 1e3, 3), 'tflops': round(BC_GFLOP_PER_STEP * steps / el / 1e3, 2),
           'dtype': 'f32', 'final_loss': round(float(loss), 5)}
    if with_cpu:
        from oracle import policy_oracle as po
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        p = po.to_params(sd)
        o = po.RMSpropState(p, max_epochs=10 ** 6)
        po.bc_step(p, o, torch.from_numpy(obs[0]), torch.from_numpy(done[0]), torch.from_numpy(act[0]), True)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 8.0 and n < 5:
            po.bc_step(p, o, torch.from_numpy(obs[n % 2]), torch.from_numpy(done[n % 2]), torch.from_numpy(act[n % 2]), True); n += 1
        res['cpu_baseline'] = {'value': round(n / (time.perf_counter() - t0), 3), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
                               'kind': 'port', 'sample': '%d oracle steps (torch fp32 autograd restatement)' % n}
    return res


def finetune_bench(steps, warmup):
    """BASELINE config 4 on one GPU: the end-to-end BC iteration of main_bc_finetune.py - PolicyNetWithConv (5 x (conv3x3 s2 + ELU) over
    every frame of the (T, B) batch, forward and backward) + the PolicyNet step; T = 100, B = 16, 64x64x6 uint8 observations, BN."""
    from pvr_habitat_amd.models import PolicyNetWithConv, HipRMSprop
    T, B = 100, 16
    torch.manual_seed(0)
    net = PolicyNetWithConv((64, 64, 6), 4, True, max_unroll=T, max_batch=B).to(device='cuda')
    opt = HipRMSprop(net, max_epochs=10 ** 6)
    g = torch.Generator().manual_seed(1)
    o = torch.randint(0, 256, (T, B, 64, 64, 6), dtype=torch.uint8, generator=g).cuda()
    d = (torch.rand((T, B), generator=g) < 0.02).cuda()
    a = torch.randint(0, 4, (T, B), generator=g).cuda()
    for _ in range(warmup):
        opt.scheduler_step(); opt.step(o, d, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.scheduler_step(); loss, gn = opt.step(o, d, a)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {'metric': 'finetune steps/sec (PolicyNetWithConv T=100 B=16, 64x64x6 uint8, BN, fp32)', 'value': round(steps / el, 2), 'unit': 'steps/s',
            'ms_per_step': round(el / steps * 1e3, 3), 'frames_per_s_through_conv_stack': round(steps * T * B * 2 / el), 'dtype': 'f32',
            'final_loss': round(float(loss), 5)}


VIT_GFLOP = {'clip_b32': 8.82, 'clip_b16': 35.13}      # per frame


def vit_bench(variant, batch, steps, warmup, dtype):
    """BASELINE config 3: CLIP-layout ViT frozen, 224x224 frames resident in HBM, frames/s on one GPU."""
    from pvr_habitat_amd import synth
    from pvr_habitat_amd.embeddings import HipResNet50
    sd = synth.clip_vit_state_dict(1, patch=16 if variant == 'clip_b16' else 32)
    m = HipResNet50(sd, variant, compute_dtype=dtype, max_batch=batch)
    fr = torch.from_numpy(synth.frames(3, batch, 224, 224)).cuda()
    outs = [torch.empty((batch, 512), dtype=torch.float32, device='cuda') for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]     # two batches in flight, as in the headline loop

    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % 2]):
                m.forward_into(fr, outs[i % 2], lane=i % 2)
    torch.cuda.synchronize()
    run(max(warmup, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.equal(outs[0], outs[1])
    fps = steps * batch / el
    return {'metric': 'frames/sec embedded (%s, 224x224)' % variant, 'value': round(fps, 1), 'unit': 'frames/s', 'dtype': dtype,
            'ms_per_step': round(el / steps * 1e3, 3), 'batch': batch, 'tflops': round(fps * VIT_GFLOP[variant] / 1e3, 1),
            'frac_of_mfma_peak': round(fps * VIT_GFLOP[variant] / 1e3 / PEAK_BF16_TFLOPS, 4)}


def pcie_bench(model_sd, batch, frame, dtype, nbatches=8):
    """PCIe-inclusive rate (never the headline `value`): host-resident uint8 frames -> pinned staging -> H2D ->
    encoder -> D2H fp32 embeddings, overlapped on separate HIP streams (embeddings.stream_embed)."""
    from pvr_habitat_amd import synth
    from pvr_habitat_amd.embeddings import HipResNet50, stream_embed

    class _Net:                                           # minimal EmbeddingNet-like holder
        pass
    net = _Net()
    net.embedding = HipResNet50(model_sd, 'conv5', compute_dtype=dtype, max_batch=batch)
    net.out_size = net.embedding.out_size
    fr = torch.from_numpy(synth.frames(9, batch, frame, frame)).repeat(nbatches, 1, 1, 1)
    stream_embed(net, fr[:2 * batch], batch)              # warm-up (allocations, first launches)
    res = {}
    for kind, src in (('pageable_source', fr), ('pinned_source', fr.pin_memory())):
        stream_embed(net, src, batch)                     # first pass touches / page-locks the host pages
        t0 = time.perf_counter()
        out = stream_embed(net, src, batch)
        el = time.perf_counter() - t0
        assert np.isfinite(out).all()
        res[kind] = {'value': round(fr.shape[0] / el, 1), 'unit': 'frames/s', 'h2d_GBps': round(fr.numel() / el / 1e9, 2)}
    res['frames'] = int(fr.shape[0])
    res['note'] = 'host uint8 frames -> (pinned double buffer ->) H2D -> encode -> D2H fp32, copies overlapped with compute on separate HIP streams'
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--frame', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--lanes', type=int, default=2, help='batches in flight per GPU (1 = strictly one forward at a time)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bc', action='store_true', help='skip the BC steps/sec leg')
    ap.add_argument('--no-pcie', action='store_true', help='skip the PCIe-inclusive streaming leg')
    ap.add_argument('--no-vit', action='store_true', help='skip the CLIP ViT legs (BASELINE config 3)')
    ap.add_argument('--no-fuse', action='store_true', help='one launch per convolution (A/B against the fused bottleneck tails)')
    ap.add_argument('--per-op', action='store_true', help='print per-launch ms / TFLOP/s of one chunk to stderr')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
    # PVR_BENCH_ONE_GPU=1 (test aid for 1-GPU boxes): every rank on cuda:0 with the gloo backend, to exercise the N > 1 path
    one_gpu = os.environ.get('PVR_BENCH_ONE_GPU', '0') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from pvr_habitat_amd import synth, _lib
    from pvr_habitat_amd.embeddings import HipResNet50
    sd = synth.resnet50_state_dict(1, 'conv5')               # MoCo-v2 encoder_q layout == torchvision resnet50
    model = HipResNet50(sd, 'conv5', compute_dtype=args.dtype, max_batch=args.batch, chunk=args.chunk)
    if args.no_fuse:
        model.set_fusion(False)
    # each rank owns its own shard of the frame stream (different seed = different frames)
    frames_np = synth.frames(1 + rank, args.batch, args.frame, args.frame)
    frames = torch.from_numpy(frames_np).cuda()
    out = torch.empty((args.batch, model.out_size), dtype=torch.float32, device='cuda')
    # Batches in flight per GPU: consecutive steps alternate between two activation workspaces ("lanes") on two streams, as the
    # streaming path (stream_embed / save_embedded_obs) does, so batch k+1 starts while batch k drains; every step is a full
    # forward of its own 256-frame batch into its own output buffer and all K steps finish inside the timed region.
    lanes = max(1, min(args.lanes, model.lanes))
    outs = [out] + [torch.empty_like(out) for _ in range(lanes - 1)]
    streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [torch.cuda.current_stream()]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        if lanes == 1:
            for _ in range(k):
                model.forward_into(frames, out)
            return
        for i in range(k):
            with torch.cuda.stream(streams[i % lanes]):
                model.forward_into(frames, outs[i % lanes], lane=i % lanes)

    torch.cuda.synchronize()                                     # default-stream setup work done before the side streams start
    run_steps(max(args.warmup, lanes))
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if lanes > 1:
        assert torch.equal(outs[0], outs[1])                     # same frames on both lanes: identical embeddings
    assert torch.isfinite(out).all()

    # roofline of the dominant kernel (conv_igemm): HIP events between launches on the launch stream
    chunk = args.chunk if args.chunk else args.batch
    cap = 128
    op_ms = (C.c_float * cap)(); op_fl = (C.c_double * cap)(); n_ops = C.c_int32()
    conv_ms, conv_fl, other_ms = 0.0, 0.0, 0.0
    plan_names = [nm for nm in model.op_names()]
    algo_bytes = conv_algorithmic_bytes(chunk, plan_names if args.dtype != 'f32' else None)
    grp = {}                                                 # per ResNet stage: [ms, flops, algorithmic bytes] of its conv launches
    reps = 3
    for _ in range(reps):
        _lib.check(_lib.lib().pvr_encoder_profile(model._handle, C.c_void_p(frames.data_ptr()), chunk, args.frame, args.frame,
                                                  C.c_void_p(out.data_ptr()), out.stride(0), _lib.stream_ptr(), op_ms, op_fl,
                                                  cap, C.byref(n_ops)))
        for i in range(n_ops.value):
            if i >= 3 and op_fl[i] > 0:
                conv_ms += op_ms[i]; conv_fl += op_fl[i]
                if args.dtype != 'f32' and i - 3 < len(plan_names):
                    g = grp.setdefault(plan_names[i - 3].split('.')[0], [0.0, 0.0, 0.0])
                    g[0] += op_ms[i]; g[1] += op_fl[i]; g[2] += conv_algorithmic_bytes(chunk, [plan_names[i - 3]])
            else:
                other_ms += op_ms[i]
    # every stage against BOTH roofs: layer1 / layer2 (fused tails, 16-bit NHWC activations) sit on the HBM roof,
    # layer3 / layer4 (deep-K implicit GEMMs) on the MFMA roof
    stages = {k: {'ms': round(v[0] / reps, 3), 'TFLOPs': round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                  'frac_mfma': round(v[1] / (v[0] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 3),
                  'algorithmic_GBps': round(v[2] / (v[0] * 1e-3) / 1e9, 1), 'frac_hbm': round(v[2] / (v[0] * 1e-3) / 8e12, 3)}
              for k, v in sorted(grp.items())}
    n_conv = n_ops.value - 4
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'pmc_conv_traffic.json')
    if os.path.isfile(tf):                                   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
        traffic = round(json.load(open(tf))['avg_hbm_bytes_per_launch'])
    if args.per_op and rank == 0:
        names = ['preprocess', 'stem', 'maxpool'] + [op for op in model.op_names()] + ['pool/flatten']
        for i in range(n_ops.value):
            tf = op_fl[i] / (op_ms[i] * 1e-3) / 1e12 if op_ms[i] > 0 else 0.0
            print('%-28s %8.3f ms %8.1f TFLOP/s' % (names[i] if i < len(names) else '?', op_ms[i], tf), file=sys.stderr)
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12
    peak = PEAK_F32_TFLOPS if args.dtype == 'f32' else PEAK_BF16_TFLOPS
    if args.dtype == 'f32':
        traffic = None                                       # the PMC passes were taken on the bf16 kernel
    barrier()

    if rank == 0:
        fps = world * args.steps * args.batch / el
        line = {
            'metric': 'frames/sec embedded (ResNet50, 256x256)', 'value': round(fps, 1), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(el / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: ResNet50 (MoCo-v2 layout) frozen, %dx%d uint8 frames resident in HBM, batch %d/GPU, '
                                   'random-init synthetic weights' % (args.frame, args.frame, args.batch),
                       'global_batch': world * args.batch, 'frame': args.frame, 'chunk': chunk, 'batches_in_flight': lanes,
                       'parallelism': 'frame shards, no collective (dp%d)' % world},
            'tflops_whole_net': round(fps * GFLOP_PER_FRAME / 1e3, 2),
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'traffic_note': 'avg HBM bytes per conv launch, rocprofv3 PMC passes (profiles/pmc_conv_traffic.json); algorithmic '
                                         'in+out+residual bytes per launch average %.0f' % (algo_bytes / max(n_conv, 1)),
                         'kernel': '%s (all' % ('conv_f32_kernel' if args.dtype == 'f32' else 'implicit-GEMM conv family: conv_igemm_kernel + conv_pp256_kernel + bottleneck_chain_kernel') + ' %d conv launches of one %d-frame chunk, HIP events)' % (n_conv, chunk),
                         'avg_launch_ms': round(conv_ms / reps / max(n_conv, 1), 4),
                         # the whole step against the HBM roof: PMC bytes of the conv launches of one batch / wall time of one step
                         # (with two batches in flight the step is shorter than the sum of its launches)
                         'hbm_step': None if traffic is None else {
                             'achieved': round(traffic * n_conv / (el / args.steps) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': round(traffic * n_conv / (el / args.steps) / 8e12, 4),
                             'note': 'conv-launch HBM bytes per batch (PMC) / step time; layer1-2 launches alone run at 3.2-4.0 TB/s'},
                         'conv_ms_per_chunk': round(conv_ms / reps, 3), 'other_ms_per_chunk': round(other_ms / reps, 3),
                         'stages': stages},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd, frames_np)
        if world == 1 and not args.no_pcie:
            line['pcie_inclusive'] = pcie_bench(sd, args.batch, args.frame, args.dtype)
        if world == 1 and not args.no_vit:
            vdt = 'f16' if args.dtype == 'f32' else args.dtype   # the fp32 mode covers the ResNet50 family only
            line['vit'] = [vit_bench('clip_b16', args.batch, 5, 2, vdt), vit_bench('clip_b32', args.batch, 5, 2, vdt)]
        if world == 1 and not args.no_bc:
            line['bc'] = bc_bench(max(args.steps, 10), args.warmup, not args.no_cpu_baseline)
            line['bc_finetune'] = finetune_bench(max(args.steps, 10), args.warmup)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
