"""Probe: raw pinned H2D / D2H bandwidth on the box (one stream, several sizes), for reading stream_embed's PCIe-inclusive rate against."""
import time, torch
for mb in (48, 192, 1024):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(mb << 20, dtype=torch.uint8, device='cuda')
    for name, fn in (('H2D', lambda: d.copy_(h, non_blocking=True)), ('D2H', lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
        print('%s %5d MB: %.1f GB/s' % (name, mb, (mb << 20) / el / 1e9))
# two streams, both directions at once
h1 = torch.empty(192 << 20, dtype=torch.uint8).pin_memory(); d1 = torch.empty(192 << 20, dtype=torch.uint8, device='cuda')
h2 = torch.empty(192 << 20, dtype=torch.uint8).pin_memory(); d2 = torch.empty(192 << 20, dtype=torch.uint8, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
print('two H2D streams at once: %.1f GB/s total' % (2 * (192 << 20) / el / 1e9))
import os
try:
    print('numa nodes of GPUs:', [open(p).read().strip() for p in sorted(__import__('glob').glob('/sys/class/drm/card*/device/numa_node'))][:8])
except Exception as e:
    print(e)
