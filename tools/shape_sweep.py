#!/usr/bin/env python3
"""Random stwo shapes through GPU prover -> GPU verifier against the oracle (run on the GPU box).

    python tools/shape_sweep.py [shapes] [seed]

For every random configuration (columns, trace size, blow-up, queries incl. non powers of two up to 64,
PoW bits, hash family) the GPU prover makes a proof; the valid proof and seeded mutants of it
(tools/fuzz_parity.mutate_stwo) must get the oracle's status word from the GPU verifier in both
modes, with the pair memoisation on and off.  Since round 4 the same batch also goes through the SHARED forms (every
distinct Merkle sibling once): as shared records through ss_stwo_verify_shared_records and, the honest proof, as
shared-path proof.json through the GPU reader -- the status words of the per-query form.  Since round 5 also through the
MINIMAL forms: minimal records against the oracle's walk, and their minimal proof.json texts through the GPU reader."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import fuzz_parity as fz  # noqa: E402
import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import formats, prover, verifier  # noqa: E402
from oracle import oracle as O  # noqa: E402

shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ver = verifier.Verifier(0)
plain = verifier.Verifier(0)
plain.stwo_flags = verifier.FLAG_NO_DEDUP
topchk = verifier.Verifier(0)
topchk.stwo_flags = verifier.FLAG_TOP_CHECKS
gp = prover.GpuProver(ver)
bad = 0
for i in range(shapes):
    q_choices = [1, 2, 3, 5, 7, 8, 13, 16, 17, 24, 31, 32, 33, 48, 63, 64]
    kw = dict(n_cols=int(rng.integers(3, 41)), trace_log=int(rng.integers(1, 15)), log_blowup=int(rng.integers(1, 5)),
              n_queries=int(q_choices[int(rng.integers(len(q_choices)))]), pow_bits=int(rng.integers(0, 9)),
              seed=int(rng.integers(0, 1000)), hash=("sha256", "blake2s")[int(rng.integers(2))])
    proof = gp.prove_proof(**kw)
    batch = [proof] * 3 + [fz.mutate_stwo(proof, rng) for _ in range(61)]
    for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
        want = O.stwo_verify_batch(batch, mode)
        if mode == verifier.MODE_FIXTURE and want[:3].any():  # the prover's own proof must be accepted by the oracle
            print("PROVER", kw, "honest proof rejected by the oracle:", hex(int(want[0])), flush=True)
            bad += 1
        for v, name in ((ver, "memo"), (plain, "full"), (topchk, "memo, compares in the top kernel")):
            got = v.verify_stwo(batch, mode, cfg=proof.cfg)
            m = int((got != want).sum())
            if m:
                j = int(np.nonzero(got != want)[0][0])
                print("MISMATCH", kw, "mode", mode, name, "at", j, hex(got[j]), hex(want[j]), flush=True)
                bad += m
    # the shared forms of the same batch (query counts that do not divide 64, one query, many columns ...)
    qs = formats.stwo_queries(proof)
    keep, shared = [], []
    for j, p in enumerate(batch):
        try:
            shared.append(verifier.stwo_shared_record(p, qs))
            keep.append(j)
        except ValueError:
            pass
    want = O.stwo_verify_batch(batch, verifier.MODE_FIXTURE)
    got = ver.verify_stwo_shared_records(proof.cfg, shared, verifier.MODE_FIXTURE)
    m = int((got != want[keep]).sum())
    text = json.dumps(ss.stwo_to_json(proof, shared=True, queries=qs), separators=(",", ":")).encode()
    st, stats = ver.verify_stwo_texts(proof.cfg, [text, text + b"\n"])
    if st.tolist() != [0, 0] or stats["host_parsed"] != 0:
        print("SHARED TEXT", kw, st.tolist(), stats["host_parsed"], flush=True)
        m += 1
    if m:
        print("MISMATCH (shared forms)", kw, m, flush=True)
        bad += m
    # the MINIMAL forms (round 5): the batch's records that have one through ss_stwo_verify_minimal_records against the
    # oracle's walk, and as minimal proof.json through the GPU reader (template of the full-length text, landmarks, gaps:
    # every shape has its own) -- the same status words, none of the writers' texts left to the host readers
    minimal, mwant = [], []
    for p in batch:
        try:
            minimal.append(verifier.stwo_minimise_record(proof.cfg, verifier.stwo_record(p), qs))
            mwant.append(O.stwo_verify_minimal(proof.cfg, minimal[-1], verifier.MODE_FIXTURE))
        except ValueError:
            pass
    got = ver.verify_stwo_minimal_records(proof.cfg, minimal, verifier.MODE_FIXTURE)
    m = int((got != np.array(mwant, dtype=np.uint32)).sum())
    texts = [verifier.write_stwo_minimal_text(proof.cfg, r, python_separators=bool(k & 1)) for k, r in enumerate(minimal)]
    st, stats = ver.verify_stwo_minimal_texts(proof.cfg, texts)
    host = sum(1 for t in texts if not verifier.stwo_minimal_text_is_canonical(proof.cfg, t)[0])
    if st.tolist() != mwant or stats["host_parsed"] != host or mwant[0] != 0 or host > len(texts) // 2:
        print("MINIMAL TEXT", kw, int((st != np.array(mwant, dtype=np.uint32)).sum()), stats["host_parsed"], host, flush=True)
        m += 1
    if m:
        print("MISMATCH (minimal forms)", kw, m, flush=True)
        bad += m
    if mode == verifier.MODE_LITERAL and i % 10 == 9:
        print("%d shapes done, last %s" % (i + 1, kw), flush=True)
print("shapes %d, total mismatches %d" % (shapes, bad))
sys.exit(1 if bad else 0)
