"""Frames/s of the other encoder families at the headline's conditions (batch 256, frames resident in HBM, two batches in flight):
python scripts/variant_rates.py [dtype]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
from pvr_habitat_amd.embeddings import HipResNet50, lane_streams
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
for variant, hw in (('r18', 256), ('r34', 256), ('conv3', 256), ('conv4', 256), ('clip_rn50', 224)):
    try:
        sd = synth.clip_rn50_state_dict(1) if variant == 'clip_rn50' else synth.resnet50_state_dict(2, variant)
    except Exception as e:
        print(variant, 'skipped:', e); continue
    m = HipResNet50(sd, variant, compute_dtype=dt, max_batch=256)
    pool = [torch.from_numpy(synth.frames(3 + i, 256, hw, hw)).cuda() for i in range(4)]
    outs = [torch.empty((256, m.out_size), device='cuda') for _ in range(2)]
    streams = lane_streams()
    def run(steps):
        for s_ in streams: s_.wait_stream(torch.cuda.current_stream())
        for i in range(steps):
            with torch.cuda.stream(streams[i & 1]):
                m.forward_into(pool[i % 4], outs[i & 1], lane=i & 1)
        torch.cuda.synchronize()
    run(8)
    t0 = time.perf_counter(); run(80); el = time.perf_counter() - t0
    print('%-10s %s: %8.0f frames/s (%.3f ms per batch of 256)' % (variant, dt, 80 * 256 / el, el / 80 * 1e3), flush=True)
    del m
