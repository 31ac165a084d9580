#!/usr/bin/env python3
"""What the host link gives: pinned host -> device copies of several sizes, one and two streams, and the
rate at which host threads can fill a pinned buffer from pageable memory (the copy every text makes)."""
import time
import numpy as np
import torch

dev = torch.device("cuda", 0)
for mb in (16, 64, 256, 1024):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(2, 2048 // mb)
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("H2D pinned %5d MiB x %3d: %.1f GB/s" % (mb, reps, n * reps / dt / 1e9))
    t0 = time.perf_counter()
    for _ in range(reps):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("D2H pinned %5d MiB x %3d: %.1f GB/s" % (mb, reps, n * reps / dt / 1e9))
# two streams, two buffers
n = 256 << 20
hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
ds = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
ss = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(8):
    with torch.cuda.stream(ss[r & 1]):
        ds[r & 1].copy_(hs[r & 1], non_blocking=True)
torch.cuda.synchronize()
print("H2D two streams 8 x 256 MiB: %.1f GB/s" % (n * 8 / (time.perf_counter() - t0) / 1e9))
# host memcpy pageable -> pinned with k threads (numpy releases the GIL in copyto)
import threading
src = np.random.randint(0, 255, size=1 << 30, dtype=np.uint8)
dst = torch.empty(1 << 30, dtype=torch.uint8).pin_memory().numpy()
for k in (1, 2, 4, 8, 16, 32):
    part = (1 << 30) // k
    def work(i):
        np.copyto(dst[i * part:(i + 1) * part], src[i * part:(i + 1) * part])
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    print("host memcpy pageable -> pinned, %2d threads: %.1f GB/s" % (k, (1 << 30) / (time.perf_counter() - t0) / 1e9))
