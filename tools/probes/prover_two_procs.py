"""Is prove_many bound by the GPU or by its host process?  P prover processes on ONE GPU, each keeping W proofs of 2^20 rows in
flight (GpuProver.prove_many), started together; the aggregate rate against one process with the same total in flight.
    python tools/probes/prover_two_procs.py            (parent: runs the combinations)
If two processes x 3 in flight beat one process x 6, the single process is host-bound (one interpreter lock for all its
worker threads); if not, the GPU is the limit and no host-side change (launch batching, a C++ orchestrator) can help."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from stark_symphony_amd import prover, verifier
    workers, n, start_at = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    gp = prover.GpuProver(verifier.Verifier(0))
    kw = dict(n_cols=4, trace_log=20, log_blowup=4, n_queries=16, pow_bits=5, hash="sha256")
    gp.prove_many(list(range(2 * workers)), workers=workers, **kw)  # warm: twiddles, allocator pools, worker streams
    while time.time() < start_at:
        time.sleep(0.001)
    t0 = time.time()
    gp.prove_many(list(range(n)), workers=workers, **kw)
    print("%f %f %d" % (t0, time.time(), n), flush=True)
    sys.exit(0)

COMBOS = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1, 4), (1, 6), (2, 2), (2, 3), (2, 4), (3, 2), (1, 4)]
for procs, workers in COMBOS:
    n = 96 // procs if procs != 5 else 20
    start_at = time.time() + 25.0
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(workers), str(n), str(start_at)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
    rows = [p.communicate()[0].strip().split() for p in ps]
    if any(len(r) != 3 for r in rows):
        print("%d process(es) x %d in flight: a child failed" % (procs, workers), flush=True)
        continue
    t0 = min(float(r[0]) for r in rows)
    t1 = max(float(r[1]) for r in rows)
    total = sum(int(r[2]) for r in rows)
    print("%d process(es) x %d in flight: %d proofs in %.3f s = %.1f proofs/s (%.2f ms per proof)" % (
        procs, workers, total, t1 - t0, total / (t1 - t0), (t1 - t0) / total * 1e3), flush=True)
