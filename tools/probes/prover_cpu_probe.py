import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from stark_symphony_amd import prover, verifier
ver = verifier.Verifier(0)
gp = prover.GpuProver(ver)
kw = dict(n_cols=4, trace_log=20, log_blowup=4, n_queries=16, pow_bits=5, hash="sha256")
gp.prove_many(list(range(8)), workers=4, **kw)
for w in (4, 8, 12, 16, 8, 12):
    c0, t0 = time.process_time(), time.perf_counter()
    gp.prove_many(list(range(48)), workers=w, **kw)
    c1, t1 = time.process_time(), time.perf_counter()
    print("workers %d: %.1f proofs/s, wall %.2f ms per proof, process CPU %.2f ms per proof" % (w, 48 / (t1 - t0), (t1 - t0) / 48 * 1e3, (c1 - c0) / 48 * 1e3), flush=True)
