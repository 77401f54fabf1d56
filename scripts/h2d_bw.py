import torch, time
x = torch.empty(256 * 256 * 256 * 3, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device='cuda')
for n in (1, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        d.copy_(x, non_blocking=True)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('H2D pinned 50MB x10: %.1f GB/s' % (10 * x.numel() / el / 1e9))
o = torch.empty(256 * 2048, dtype=torch.float32, device='cuda'); h = torch.empty(256 * 2048, dtype=torch.float32).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): h.copy_(o, non_blocking=True)
torch.cuda.synchronize(); print('D2H 2MB x50: %.1f GB/s' % (50 * o.numel() * 4 / (time.perf_counter() - t0) / 1e9))
