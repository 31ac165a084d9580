#!/usr/bin/env python3
"""Per-kernel time of one resident stwo_2p20 batch, one pass at a time (no overlap between kernels).
    python tools/top_probe.py [proofs] [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stark_symphony_amd import verifier

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wname, family, proofs, note = bench.load_workload("stwo_2p20")
ver = verifier.Verifier(0)
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
print("accepted", batch.accepted(), "of", batch.n,
      {k: round(ms / max(c, 1), 3) for k, (ms, c) in sorted(t.items())}, flush=True)
