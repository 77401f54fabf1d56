#!/usr/bin/env python3
"""How reproducible is the reference's own uint8 Resize off the dyadic sizes?  (round-4 verdict, weak 3)

torchvision 0.10's tensor Resize (reference src/embeddings.py:80-85) is float32 F.interpolate(bilinear, align_corners=False) followed by
round() and a cast back to uint8.  ATen's float result is not a single function of its inputs: the CPU kernel takes different code paths
(and fp32 operation orders / FMA contractions) for different intra-op thread counts, and the CUDA kernel the reference actually runs on
a GPU box is a third order (nvcc contracts a*b + c*d into FMAs its own way).  This script counts, per source size, the uint8 pixels
that differ between ATen with 1 thread and ATen with 8 threads - the reference's own run-to-run spread - next to the bound the HIP
kernel (csrc/preprocess.hip, the CUDA kernel's formula with contraction off) is tested to against the oracle:
<= 1 LSB on < 0.1 % of the pixels (tests/test_gpu_encoder.py::test_preprocess_*).  Sizes where the scale is a power of two (64, 128,
256 squares: Habitat's frames and the bench's) are exact everywhere.  Runs on the CPU."""
import numpy as np
import torch
import torch.nn.functional as F


def resized(h, w, size=256):
    sh, lg = (w, h) if w <= h else (h, w)
    if sh == size:
        return h, w
    nl = int(size * lg / sh)
    return (nl, size) if w <= h else (size, nl)


def main():
    g = torch.Generator().manual_seed(0)
    print('%-12s %-12s %12s %12s %10s' % ('source', 'resized', 'float diffs', 'uint8 diffs', 'max |d|'))
    for h, w in ((64, 64), (128, 128), (96, 64), (100, 75), (75, 100), (130, 97), (300, 256), (480, 640), (720, 1280)):
        x = torch.randint(0, 256, (16, 3, h, w), dtype=torch.uint8, generator=g)
        nh, nw = resized(h, w)
        outs = []
        for th in (1, 8):
            torch.set_num_threads(th)
            outs.append(F.interpolate(x.float(), size=(nh, nw), mode='bilinear', align_corners=False))
        a, b = outs
        ua, ub = a.round().to(torch.uint8), b.round().to(torch.uint8)
        nd = int((ua != ub).sum())
        print('%-12s %-12s %12d %12d %10d   (%.5f %% of %d pixels)' % ('%dx%d' % (h, w), '%dx%d' % (nh, nw), int((a != b).sum()), nd,
              int((ua.int() - ub.int()).abs().max()), 100.0 * nd / ua.numel(), ua.numel()))


if __name__ == '__main__':
    main()
