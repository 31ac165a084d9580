#!/usr/bin/env python3
"""Minimal records RESIDENT in HBM (ss_stwo_verify_minimal_dev): time per pass over n proofs of the 2^20 shape, kernel by
kernel, beside the per-query batch of the same proofs.  Not the metric of bench.py (whose batch is the per-query one)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stark_symphony_amd import formats, records, verifier  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
proofs = records.load_stwo_npz(os.path.join(ROOT, "tests", "golden", "stwo_trace20.npz"))
cfg = proofs[0].cfg
ver = verifier.Verifier(0)
recs = [verifier.stwo_record(p) for p in proofs]
mins = [verifier.stwo_minimise_record(cfg, r, formats.stwo_queries(p)) for p, r in zip(proofs, recs)]
idx = [i % len(proofs) for i in range(n)]
for kind in ("minimal", "per-query"):
    b = ver.stwo_minimal_batch(cfg, mins, index=idx) if kind == "minimal" else verifier.StwoDeviceBatch(ver, cfg, verifier.MODE_FIXTURE, recs, index=idx)
    b.run()
    torch.cuda.synchronize()
    assert (b.status() == 0).all()
    ver.set_timing(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = ver.collect_timing()
    ver.set_timing(False)
    inb = b.record_bytes if kind == "minimal" else len(recs[0]) * 4 * n
    print("%s: %d proofs resident, %.3f ms per pass (one stream, HEAD and TAIL in sequence) = %.3f M proofs/s; input %.2f GB (%.0f B / proof)"
          % (kind, n, dt * 1e3, n / dt / 1e6, inb / 1e9, inb / n))
    for k, (ms, cnt) in sorted(t.items(), key=lambda kv: -kv[1][0]):
        print("   %-20s %8.3f ms per pass" % (k, ms / reps))
    del b
    torch.cuda.empty_cache()
