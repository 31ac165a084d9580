#!/usr/bin/env python3
"""GPU stark101 prover timing: prove() of the FibonacciSq statement on one
MI355X, compared byte for byte with the proof the reference's own Python prover wrote
(tests/golden/stark101_proof.json; that prover needs ~16 s of CPU here).
`python tools/prover101_bench.py [reps]`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import prover101, verifier  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ver = verifier.Verifier(0)
gp = prover101.Stark101GpuProver(ver)
want = json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json")))
times = []
for r in range(reps):
    t0 = time.perf_counter()
    got = gp.prove()
    times.append(time.perf_counter() - t0)
assert got == want
print("identical to the reference prover's proof.json:", got == want)
print("verify on GPU:", ver.verify_stark101([ss.stark101_from_json(got)]).tolist())
times.sort()
print(json.dumps({"metric": "stark101 prove time (trace 1023, coset 8192, 10 FRI layers)",
                  "value": times[len(times) // 2], "unit": "s", "best": times[0], "reps": reps}))
