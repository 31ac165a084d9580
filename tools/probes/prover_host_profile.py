"""Where does the HOST side of one 2^20-row proof spend its time?  cProfile of GpuProver.prove_proof (wall time per function, GPU
waits included) -- what the interpreter lock serialises when several proofs are in flight (tools/probes/prover_two_procs.py:
one process 108-112 proofs/s, two processes 118)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stark_symphony_amd import prover, verifier  # noqa: E402

gp = prover.GpuProver(verifier.Verifier(0))
kw = dict(n_cols=4, trace_log=20, log_blowup=4, n_queries=16, pow_bits=5, hash="sha256")
gp.prove_proof(seed=0, **kw)
gp.prove_proof(seed=1, **kw)
pr = cProfile.Profile()
pr.enable()
for s in range(2, 10):
    gp.prove_proof(seed=s, **kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
