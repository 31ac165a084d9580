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
REPS = 7  # the first call allocates the scratch; best and median of the others
times = []
for rep in range(REPS):
    t0 = time.perf_counter()
    B.check(lib.ss_stwo_verify_records(ver.ctx, C.byref(cfg), n, ptrs, status.ctypes.data))
    times.append(time.perf_counter() - t0)
    assert (status == 0).all()
times = sorted(times[1:])
best, med = times[0], times[len(times) // 2]
print(("distinct buffers; " if distinct else "") + "records:        %6d proofs  best %.0f proofs/s (%.2f GB/s on the link)  median %.0f proofs/s (%.2f GB/s)"
      % (n, n / best, n * recs[0].nbytes / best / 1e9, n / med, n * recs[0].nbytes / med / 1e9))

from stark_symphony_amd import formats  # noqa: E402
shared = [verifier.stwo_shared_record(p) for p in proofs]
batch = [shared[i % len(shared)].copy() if distinct else shared[i % len(shared)] for i in range(n)]
sptrs = verifier._ptr_array(batch)
words = (C.c_size_t * n)(*[int(r.size) for r in batch])
total = sum(int(r.nbytes) for r in batch)
times = []
for rep in range(REPS):
    status[:] = 0xFFFFFFFF
    t0 = time.perf_counter()
    B.check(lib.ss_stwo_verify_shared_records(ver.ctx, C.byref(cfg), n, sptrs, words, status.ctypes.data))
    times.append(time.perf_counter() - t0)
    assert (status == 0).all()
times = sorted(times[1:])
best, med = times[0], times[len(times) // 2]
print(("distinct buffers; " if distinct else "") + "shared records: %6d proofs  best %.0f proofs/s (%.2f GB/s on the link)  median %.0f proofs/s (%.2f GB/s)   %.1f %% of the per-query bytes"
      % (n, n / best, total / best / 1e9, n / med, total / med / 1e9, 100.0 * total / (n * recs[0].nbytes)))

# ---- the same inputs from ONE caller-pinned buffer: no staging copy, no host thread per byte (csrc/ss_pinned.hip).  Run
# with SS_STAGE_THREADS=1 / 2 to see what the staged paths above make of one or two host threads (a rank's share of a
# 16-core grant at eight ranks per host); the pinned paths do not depend on it.
minimal = [verifier.stwo_minimise_record(proofs[0].cfg, r, formats.stwo_queries(p)) for p, r in zip(proofs, recs)]
mbatch = [minimal[i % len(minimal)].copy() if distinct else minimal[i % len(minimal)] for i in range(n)]
times = []
words = (C.c_size_t * n)(*[int(r.size) for r in mbatch])
mptrs = verifier._ptr_array(mbatch)
mtotal = sum(int(r.nbytes) for r in mbatch)
for rep in range(REPS):
    status[:] = 0xFFFFFFFF
    t0 = time.perf_counter()
    B.check(lib.ss_stwo_verify_minimal_records(ver.ctx, C.byref(cfg), n, mptrs, words, status.ctypes.data))
    times.append(time.perf_counter() - t0)
    assert (status == 0).all()
times = sorted(times[1:])
print(("distinct buffers; " if distinct else "") + "minimal records: %6d proofs  best %.0f proofs/s (%.2f GB/s on the link)  median %.0f proofs/s (%.2f GB/s)   %.1f %% of the per-query bytes"
      % (n, n / times[0], mtotal / times[0] / 1e9, n / times[len(times) // 2], mtotal / times[len(times) // 2] / 1e9,
         100.0 * mtotal / (n * recs[0].nbytes)))
for kind, src in (("records", rec_batch), ("shared", batch), ("minimal", mbatch)):
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([r.size for r in src])
    flat = ver.pinned_buffer(int(offs[-1]))
    flat[:] = np.concatenate(src)
    times = []
    for rep in range(REPS):
        t0 = time.perf_counter()
        st = ver.verify_stwo_pinned(proofs[0].cfg, flat, None if kind == "records" else offs, kind)
        times.append(time.perf_counter() - t0)
        assert (st == 0).all()
    times = sorted(times[1:])
    best, med = times[0], times[len(times) // 2]
    print("caller-pinned %-8s %6d proofs  best %.0f proofs/s (%.2f GB/s on the link)  median %.0f proofs/s (%.2f GB/s)   stage threads: none (SS_STAGE_THREADS=%s for the staged rows)"
          % (kind + ":", n, n / best, flat.nbytes / best / 1e9, n / med, flat.nbytes / med / 1e9, os.environ.get("SS_STAGE_THREADS", "default 8")))
    del flat
