#!/usr/bin/env python3
"""Kernel times of the minimal-record host path (ss_stwo_verify_minimal_records) per launch, 4096 records of the 2^20 shape."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stark_symphony_amd import formats, records, verifier  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
proofs = records.load_stwo_npz(os.path.join(ROOT, "tests", "golden", "stwo_trace20.npz"))
cfg = proofs[0].cfg
ver = verifier.Verifier(0)
mins = [verifier.stwo_minimise_record(cfg, verifier.stwo_record(p), formats.stwo_queries(p)) for p in proofs]
batch = [mins[i % len(mins)].copy() for i in range(n)]
ver.verify_stwo_minimal_records(cfg, batch)
for kind in ("minimal", "records"):
    ver.set_timing(True)
    t0 = time.perf_counter()
    if kind == "minimal":
        st = ver.verify_stwo_minimal_records(cfg, batch)
    else:
        st = ver.verify_stwo_records(cfg, np.stack([verifier.stwo_record(proofs[i % len(proofs)]) for i in range(n)]))
    dt = time.perf_counter() - t0
    t = ver.collect_timing()
    ver.set_timing(False)
    assert (st == 0).all()
    print("%s: %d proofs in %.2f ms (timing on)" % (kind, n, dt * 1e3))
    for k, (ms, cnt) in sorted(t.items(), key=lambda kv: -kv[1][0]):
        print("   %-22s %8.3f ms total  %4d launches  %7.3f ms each" % (k, ms, cnt, ms / cnt))
