#!/usr/bin/env python3
"""Per-kernel time of one resident stwo batch, one pass at a time (no overlap between kernels).

    python tools/top_probe.py [workload] [proofs] [passes]      (workloads: bench.py's stwo ones)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from stark_symphony_amd import formats, prover, verifier  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "stwo_2p20"
wname, family, proofs, note = bench.load_workload(wl)
ver = verifier.Verifier(0)
if not proofs:
    c = formats.StwoConfig(**bench.GEN_ONLY[wname])
    proofs = [prover.GpuProver(ver).prove_proof(c.n_cols, c.trace_log, c.log_blowup, c.n_queries, c.pow_bits,
                                                seed=0, hash=c.hash)]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (65536 if proofs[0].cfg.lde_log >= 24 else 32768)
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
batch = ver.stwo_batch(proofs, verifier.MODE_FIXTURE, replicate=(n + len(proofs) - 1) // len(proofs))
for _ in range(2):
    batch.run()
torch.cuda.synchronize()
ver.set_timing(True)
ver.collect_timing()
for _ in range(passes):
    batch.run()
    torch.cuda.synchronize()
t = ver.collect_timing()
times = {k: round(ms / max(c, 1), 3) for k, (ms, c) in sorted(t.items())}
print(wname, "accepted", batch.accepted(), "of", batch.n, times, "sum", round(sum(times.values()), 3), flush=True)
