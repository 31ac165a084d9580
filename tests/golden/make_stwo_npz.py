#!/usr/bin/env python3
"""Writes the prover-made stwo fixtures tests/golden/stwo_*.npz (records, include/ss_verify.h).

    python tests/golden/make_stwo_npz.py trace16_blake2s        # BASELINE.json configs[2] as named

Proofs come from tools/stwo_prover.py, the numpy restatement of the external stwo prover that
reproduces the reference's own two proofs byte for byte (tests/test_prover.py).  The reference has
no Blake2s, so the *_blake2s fixtures are parity-unpinned by construction."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import stwo_prover  # noqa: E402

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import records  # noqa: E402

CASES = {
    # name: (prove() keywords, seeds)
    "trace16": (dict(n_cols=4, trace_log=16, log_blowup=4, n_queries=32, pow_bits=5), (0, 1)),
    "trace16_blake2s": (dict(n_cols=4, trace_log=16, log_blowup=4, n_queries=32, pow_bits=5, hash="blake2s"), (0, 1)),
    "wide256": (dict(n_cols=256, trace_log=14, log_blowup=4, n_queries=16, pow_bits=5), (0,)),
    "wide256_blake2s": (dict(n_cols=256, trace_log=14, log_blowup=4, n_queries=16, pow_bits=5, hash="blake2s"), (0,)),
    "trace20": (dict(n_cols=4, trace_log=20, log_blowup=4, n_queries=16, pow_bits=5), (0,)),
    "trace20_blake2s": (dict(n_cols=4, trace_log=20, log_blowup=4, n_queries=16, pow_bits=5, hash="blake2s"), (0,)),
}

if __name__ == "__main__":
    for name in sys.argv[1:]:
        kw, seeds = CASES[name]
        proofs = [ss.stwo_from_json(stwo_prover.prove(seed=s, verbose=True, **kw)) for s in seeds]
        out = os.path.join(ROOT, "tests", "golden", "stwo_%s.npz" % name)
        records.save_stwo_npz(out, proofs)
        print("wrote", out, os.path.getsize(out), "bytes")
