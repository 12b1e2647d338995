"""GPU-box probe: host-to-device bandwidth from pinned memory for the copy sizes the streamed bench run uses."""
import time, torch
for mb in (4, 20, 40, 160, 640):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 2000 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%4d MB copies: %.1f GB/s" % (mb, reps * n / dt / 1e9))
