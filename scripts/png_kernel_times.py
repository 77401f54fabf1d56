"""Durations of the two PNG kernels (run under rocprofv3 --kernel-trace --stats): python3 scripts/png_kernel_times.py [files per call]"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from pvr_habitat_amd import synth, png_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = tempfile.mkdtemp(prefix='pngk_')
fr = synth.smooth_frames(3, 512, 64, 64)
names = []
for i in range(n):
    p = os.path.join(d, '%d.png' % i); Image.fromarray(fr[i % 512]).save(p); names.append(p)
png_gpu.decode_files(names[:32])
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); png_gpu.decode_files(names); torch.cuda.synchronize()
    print('%d files: %.2f ms per call' % (n, (time.perf_counter() - t0) * 1e3))
