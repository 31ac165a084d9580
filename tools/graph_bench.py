#!/usr/bin/env python3
"""Launch-bound small batches: eager `Pipeline.submit` per step against two alternating
`GraphedPipeline`s (one hipGraph launch per 8 steps).  `python tools/graph_bench.py [steps]`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import verifier  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 960
SLOTS = 8
ver = verifier.Verifier(0)
s101 = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
stwo = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))


def timed(fn, n_calls, steps_per_call):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n_calls * steps_per_call)


for name, make, n in (("stark101 x 1024", lambda n: ver.stark101_batch([s101], replicate=n), 1024),
                      ("stark101 x 4096", lambda n: ver.stark101_batch([s101], replicate=n), 4096),
                      ("stark101 x 16384", lambda n: ver.stark101_batch([s101], replicate=n), 16384),
                      ("stwo fixture x 512", lambda n: ver.stwo_batch([stwo], replicate=n), 512),
                      ("stwo fixture x 4096", lambda n: ver.stwo_batch([stwo], replicate=n), 4096)):
    batch = make(n)
    slots = [batch] + [batch.sibling() for _ in range(2 * SLOTS - 1)]
    pipe = verifier.Pipeline(slots[:4])
    eager = timed(lambda: pipe.submit(), steps, 1)
    assert all(s.accepted() == n for s in slots[:4])
    for s in slots:
        s.accept_dev.zero_()
    ga, gb = verifier.GraphedPipeline(slots[:SLOTS]), verifier.GraphedPipeline(slots[SLOTS:])

    def both():
        ga.replay()
        gb.replay()
    graphed = timed(both, max(1, steps // (2 * SLOTS)), 2 * SLOTS)
    assert all(s.accepted() == n for s in slots)
    print(json.dumps({"workload": name, "eager_us_per_step": round(eager * 1e6, 1),
                      "graph_us_per_step": round(graphed * 1e6, 1),
                      "eager_proofs_per_s": round(n / eager), "graph_proofs_per_s": round(n / graphed)}), flush=True)
