#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (records in host memory -> verdicts):
ss_stwo_verify_records and ss_stwo_verify_shared_records (every distinct sibling once, expanded on the GPU), timed
around the single C call.  Not the metric of bench.py."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_symphony_amd import binding as B, records, verifier  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
distinct = len(sys.argv) > 2 and sys.argv[2] == "distinct"   # every record its own buffer: the staging copy reads host DRAM
proofs = records.load_stwo_npz(os.path.join(ROOT, "tests", "golden", "stwo_trace20.npz"))
recs = [verifier.stwo_record(p) for p in proofs]
cfg = verifier.stwo_cfg_struct(proofs[0].cfg, verifier.MODE_FIXTURE)
ver = verifier.Verifier(0)
rec_batch = [recs[i % len(recs)].copy() if distinct else recs[i % len(recs)] for i in range(n)]
ptrs = verifier._ptr_array(rec_batch)
status = np.zeros(n, dtype=np.uint32)
lib = B.lib()
for rep in range(3):
    t0 = time.perf_counter()
    B.check(lib.ss_stwo_verify_records(ver.ctx, C.byref(cfg), n, ptrs, status.ctypes.data))
    dt = time.perf_counter() - t0
    assert (status == 0).all()
    print(("distinct buffers; " if distinct else "") + "host path: %d proofs in %.3f s = %.0f proofs/s, %.2f GB/s of records"
          % (n, dt, n / dt, n * recs[0].nbytes / dt / 1e9))

from stark_symphony_amd import formats  # noqa: E402
shared = [verifier.stwo_shared_record(p) for p in proofs]
batch = [shared[i % len(shared)].copy() if distinct else shared[i % len(shared)] for i in range(n)]
sptrs = verifier._ptr_array(batch)
words = (C.c_size_t * n)(*[int(r.size) for r in batch])
total = sum(int(r.nbytes) for r in batch)
for rep in range(3):
    status[:] = 0xFFFFFFFF
    t0 = time.perf_counter()
    B.check(lib.ss_stwo_verify_shared_records(ver.ctx, C.byref(cfg), n, sptrs, words, status.ctypes.data))
    dt = time.perf_counter() - t0
    assert (status == 0).all()
    print("shared records: %d proofs in %.3f s = %.0f proofs/s, %.2f GB/s on the link (%.1f %% of the per-query bytes)"
          % (n, dt, n / dt, total / dt / 1e9, 100.0 * total / (n * recs[0].nbytes)))
