"""HBM write ceiling of the box: torch fill / copy of a 16 GB tensor (what a pure store / load+store kernel reaches)"""
import time, torch
dev = torch.device("cuda:0")
free, total = torch.cuda.mem_get_info()
x = torch.empty(int(free * 0.9) // 8, dtype=torch.int64, device=dev); x.fill_(1); del x; torch.cuda.empty_cache()   # first touch
n = 16 * 2**30 // 8
a = torch.empty(n, dtype=torch.int64, device=dev)
b = torch.empty(n, dtype=torch.int64, device=dev)
for name, fn, bytes_ in (("fill", lambda: a.fill_(3), 8 * n), ("zero", lambda: a.zero_(), 8 * n), ("copy", lambda: b.copy_(a), 16 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.3f} ms, {bytes_/dt/1e12:.2f} TB/s")
