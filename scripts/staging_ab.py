"""Host staging strategies for a pageable uint8 source (50 MB batches): which one feeds H2D fastest on this box?"""
import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n, bs = 16, 256 * 256 * 256 * 3
src = torch.randint(0, 255, (n, bs), dtype=torch.uint8)            # pageable
pin = [torch.empty(bs, dtype=torch.uint8).pin_memory() for _ in range(2)]
dev = torch.empty(bs, dtype=torch.uint8, device='cuda')
def t(fn, label):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('%-60s %.1f GB/s' % (label, n * bs / el / 1e9), flush=True)
def a():
    for i in range(n): pin[i & 1].copy_(src[i])
t(a, 'torch copy_ pageable->pinned (one call per batch)')
for th in (2, 4, 8, 16):
    pool = ThreadPoolExecutor(th)
    def b():
        for i in range(n):
            d, s = pin[i & 1].numpy(), src[i].numpy()
            step = (bs + th - 1) // th
            list(pool.map(lambda o: np.copyto(d[o:o + step], s[o:o + step]), range(0, bs, step)))
    t(b, 'numpy copyto in %d threads' % th)
    pool.shutdown()
def c():
    for i in range(n): dev.copy_(src[i], non_blocking=False)
t(c, 'direct dev.copy_(pageable) (driver staging)')
rt = torch.cuda.cudart()
def d():
    for i in range(n):
        p = src[i].data_ptr()
        assert rt.cudaHostRegister(p, bs, 0) == 0 or True
        dev.copy_(src[i], non_blocking=True); torch.cuda.synchronize()
        rt.cudaHostUnregister(p)
t(d, 'hipHostRegister per batch + H2D + unregister')
torch.set_num_threads(4)
t(a, 'torch copy_ with set_num_threads(4)')
