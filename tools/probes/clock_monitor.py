"""The shader clock the chip grants while the verifier runs (tools only).  A one-lane kernel on its own stream samples
s_memtime (shader cycles) against s_memrealtime (100 MHz) every 100 us; meanwhile bench.py's pipeline verifies the metric
batch.  Prints the clock idle, under the Merkle stage, and under the register-only SHA-256 calibration kernel.
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/clock_monitor.hip -o build/libclock_monitor.so
    GPU_MAX_HW_QUEUES=24 python tools/probes/clock_monitor.py [proofs per step]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch

import bench
from stark_symphony_amd import verifier


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mon = C.CDLL(os.path.join(ROOT, "build", "libclock_monitor.so"))
    mon.cm_launch.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    _, _, proofs, _ = bench.load_workload("stwo_2p20")
    ver = verifier.Verifier(0)
    batch = verifier.StwoDeviceBatch(ver, proofs[0].cfg, verifier.MODE_FIXTURE, [verifier.stwo_record(p) for p in proofs],
                                     index=[i % len(proofs) for i in range(n)])
    slots = [batch, batch.sibling(), batch.sibling()]
    pipe = verifier.Pipeline(slots, tail_streams=2 if n < 65536 else 1)
    for _ in range(6):
        pipe.submit()
    pipe.synchronize()
    ms = torch.cuda.Stream()
    N, PERIOD = 6000, 10000          # 6000 samples x 100 us = 0.6 s
    samples = torch.zeros(2 * N, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert mon.cm_launch(samples.data_ptr(), N, PERIOD, ms.cuda_stream) == 0
    time.sleep(0.1)                   # 0.1 s idle
    steps = max(4, int(0.3 / (18.2e-3 * n / 65536)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_host0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit()
    pipe.synchronize()
    busy_s = time.perf_counter() - t_host0
    ms.synchronize()
    s = samples.cpu().numpy().astype(np.uint64).reshape(N, 2)
    dr = np.diff(s[:, 0].astype(np.int64)).astype(np.float64)
    dc = np.diff(s[:, 1].astype(np.int64)).astype(np.float64)
    mhz = dc / dr * 100.0
    t = (s[1:, 0] - s[0, 0]).astype(np.float64) / 1e8  # seconds since the first sample
    idle = mhz[t < 0.08]
    # the busy window: from 0.1 s + a margin to the end of the verify loop
    busy = mhz[(t > 0.13) & (t < 0.1 + busy_s - 0.03)]
    after = mhz[t > 0.1 + busy_s + 0.05]
    q = lambda a: (np.percentile(a, 5), np.median(a), np.percentile(a, 95)) if a.size else (0, 0, 0)
    print("%d proofs per step, %d steps in %.3f s (%.3f ms per step)" % (n, steps, busy_s, busy_s / steps * 1e3))
    print("shader clock, MHz (5 %% / median / 95 %% of 100 us samples): idle before %.0f / %.0f / %.0f;  while the verifier runs "
          "%.0f / %.0f / %.0f (%d samples);  idle after %.0f / %.0f / %.0f" % (*q(idle), *q(busy), busy.size, *q(after)))


if __name__ == "__main__":
    main()
