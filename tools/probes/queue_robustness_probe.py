"""Does the records path keep its upload / verify overlap whatever else the process has done with streams?
The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); two streams on one
queue serialize.  For k = 0 .. the caller makes and uses k streams of its own BEFORE the library creates its two, then
times ss_stwo_verify_records on n records (best of 5 after a warm-up), each k in a fresh process.
    python tools/probes/queue_robustness_probe.py [n]            (run on a GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from stark_symphony_amd import records, verifier
k, n = int(sys.argv[1]), int(sys.argv[2])
proofs = records.load_stwo_npz(os.path.join(%r, "tests", "golden", "stwo_trace20.npz"))
ver = verifier.Verifier(0)
mine = [torch.cuda.Stream() for _ in range(k)]
x = torch.zeros(1 << 20, device="cuda")
for s in mine:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
recs = [verifier.stwo_record(p) for p in proofs]
batch = np.stack([recs[i %% len(recs)] for i in range(n)])
cfg = proofs[0].cfg
ver.verify_stwo_records(cfg, batch)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    st = ver.verify_stwo_records(cfg, batch)
    best = min(best, time.perf_counter() - t0)
    assert (st == 0).all()
print("caller streams %%2d: %%7.0f proofs/s  %%5.1f GB/s on the link" %% (k, n / best, batch.nbytes / best / 1e9), flush=True)
""" % (ROOT, ROOT)

n = sys.argv[1] if len(sys.argv) > 1 else "4096"
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"))
for k in (0, 1, 2, 3, 4, 5, 6, 7, 9, 13):
    r = subprocess.run([sys.executable, "-c", CHILD, str(k), n], capture_output=True, text=True, timeout=600)
    print(r.stdout.strip() or r.stderr[-400:], flush=True)
