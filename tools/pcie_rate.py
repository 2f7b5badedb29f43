"""Rate of the copies a draw makes over PCIe, by size: page-locked host memory to HBM and back (hipMemcpyAsync through torch)."""
import time
import torch
dev = torch.device("cuda:0")
for mb in (1, 4, 12, 32, 44, 128):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for what, fn in (("to the device", lambda: d.copy_(h, non_blocking=True)), ("to the host", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{mb:4d} MiB {what}: {best*1e3:.3f} ms, {n/best/1e9:.1f} GB/s", flush=True)
