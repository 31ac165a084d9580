#!/usr/bin/env python3
"""Differential fuzzing of the GPU verifiers against the CPU oracle (run on the GPU box).

    python tools/fuzz_parity.py [mutations_per_case] [seed] [shared-record structure mutants per case]

For every proof family / configuration it mutates valid proofs (single bit flips, words
replaced by 0 / P / P+v / 2^32-1, whole siblings swapped or zeroed, values copied between
queries) and requires the GPU status word to equal the oracle's for every mutant, in both
stwo modes and both hash families.  Exit status 0 = no mismatch."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import formats, records, verifier  # noqa: E402
from oracle import oracle as O  # noqa: E402
import stwo_prover  # noqa: E402

_cli = __name__ == "__main__"
N = int(sys.argv[1]) if _cli and len(sys.argv) > 1 else 2000
SEED = int(sys.argv[2]) if _cli and len(sys.argv) > 2 else 20251003
N_STRUCT = int(sys.argv[3]) if _cli and len(sys.argv) > 3 else min(N, 400)  # structure mutants of a shared record (one oracle walk each)
GOLDEN = os.path.join(ROOT, "tests", "golden")
P31 = 2147483647
P101 = 3221225473


def special(rng, v, p):
    k = int(rng.integers(6))
    return [0, p, (int(v) + p) & 0xFFFFFFFF, 0xFFFFFFFF, p - 1, int(v) ^ (1 << int(rng.integers(32)))][k]


def mutate_stwo(p, rng):
    q = p.copy()
    kind = int(rng.integers(8))
    Q, K = p.cfg.n_queries, p.cfg.n_layers
    if kind == 0:
        return formats.stwo_corrupt(p, rng)[0]
    if kind == 1:  # special word in a field-element section
        arr = [q.oods_trace, q.oods_cp, q.trace_vals, q.cp_vals, q.last_layer, q.fri_witness][int(rng.integers(6))]
        flat = arr.reshape(-1)
        i = int(rng.integers(flat.size))
        flat[i] = special(rng, flat[i], P31)
    elif kind == 2:  # swap two siblings of one path
        paths = [q.trace_paths, q.cp_paths] + q.fri_paths
        pl = paths[int(rng.integers(len(paths)))]
        pth = pl[int(rng.integers(Q))]
        if len(pth) >= 2:
            i, j = rng.integers(len(pth), size=2)
            pth[[i, j]] = pth[[j, i]]
    elif kind == 3:  # copy a query's values to another query
        a, b = rng.integers(Q, size=2)
        q.trace_vals[a] = q.trace_vals[b]
    elif kind == 4:  # zero one FRI witness
        q.fri_witness[int(rng.integers(K + 1)), int(rng.integers(Q))] = 0
    elif kind == 5:  # nonce tweaks
        q.pow_nonce = (q.pow_nonce + int(rng.integers(1, 5))) & 0xFFFFFFFFFFFFFFFF
    elif kind == 6:  # drop / duplicate a path node (the record's path_len trailer)
        paths = [q.trace_paths, q.cp_paths] + q.fri_paths
        pl = paths[int(rng.integers(len(paths)))]
        j = int(rng.integers(Q))
        if len(pl[j]) > 1:
            pl[j] = pl[j][:-1] if rng.integers(2) else np.concatenate([pl[j], pl[j][:1]])
    else:  # two mutations at once
        q = mutate_stwo(mutate_stwo(p, rng), rng)
    return q


def mutate_s101(p, rng):
    q = p.copy()
    kind = int(rng.integers(6))
    if kind == 0:
        return formats.stark101_corrupt(p, rng)[0]
    if kind == 1:
        li = int(rng.integers(len(q.layers)))
        which = int(rng.integers(3))
        l = q.layers[li]
        if which == 0: l.beta = special(rng, l.beta, P101)
        if which == 1: l.cpa.ev = special(rng, l.cpa.ev, P101)
        if which == 2: l.cpb.ev = special(rng, l.cpb.ev, P101)
    elif kind == 2:
        q.evals[int(rng.integers(3))].ev = special(rng, 0, P101)
    elif kind == 3:
        q.last = special(rng, q.last, P101)
    elif kind == 4:  # ragged shapes
        li = int(rng.integers(len(q.layers)))
        e = q.layers[li].cpa if rng.integers(2) else q.layers[li].cpb
        if len(e.path) > 1:
            e.path = e.path[:-1] if rng.integers(2) else np.concatenate([e.path, e.path[:1]])
    else:
        q.layers = q.layers[:int(rng.integers(len(q.layers) + 1))]
    return q


def main():
    ver = verifier.Verifier(0)
    rng = np.random.default_rng(SEED)
    bad = 0
    s101 = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    muts = [s101] + [mutate_s101(s101, rng) for _ in range(N)]
    got, want = ver.verify_stark101(muts), O.s101_verify_batch(muts)
    mism = int((got != want).sum())
    print("stark101: %d mutants, %d distinct codes, accepts %d, mismatches %d"
          % (N, len(set(want.tolist())), int((want == 0).sum()), mism), flush=True)
    bad += mism
    cases = [("fixture LDE 2^13", ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))),
             ("test config", ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))),
             ("wide256", records.load_stwo_npz(os.path.join(GOLDEN, "stwo_wide256.npz"))[0]),
             ("trace16 Q32", records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace16.npz"))[0]),
             ("blake2s 2^7 N8", ss.stwo_from_json(stwo_prover.prove(n_cols=8, trace_log=7, log_blowup=3, n_queries=16,
                                                                     pow_bits=4, seed=3, hash="blake2s"))),
             ("N=3 Q=2", ss.stwo_from_json(stwo_prover.prove(n_cols=3, trace_log=2, log_blowup=1, n_queries=2,
                                                             pow_bits=0, seed=2)))]
    for name, base in cases:
        muts = [base] + [mutate_stwo(base, rng) for _ in range(N)]
        for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
            got, want = ver.verify_stwo(muts, mode, cfg=muts[0].cfg), O.stwo_verify_batch(muts, mode)
            mism = int((got != want).sum())
            print("stwo %-16s mode %d: %d mutants, %d distinct codes, accepts %d, mismatches %d"
                  % (name, mode, N, len(set(want.tolist())), int((want == 0).sum()), mism), flush=True)
            if mism:
                i = int(np.nonzero(got != want)[0][0])
                print("   first mismatch at %d: gpu %#x oracle %#x" % (i, got[i], want[i]))
            bad += mism
        # the same mutants as SHARED records (every distinct sibling once, expanded on the GPU: round 4) wherever a mutant
        # still has a shared form, + structure mutants of the honest shared record (hints, counts, sizes, bits anywhere)
        if base.cfg.n_queries > 1:
            qs = formats.stwo_queries(base)
            keep, shared = [], []
            for i, m in enumerate(muts):
                try:
                    shared.append(verifier.stwo_shared_record(m, qs))
                    keep.append(i)
                except ValueError:
                    pass
            want_all = O.stwo_verify_batch(muts, verifier.MODE_FIXTURE)
            got = ver.verify_stwo_shared_records(base.cfg, shared, verifier.MODE_FIXTURE)
            mism = int((got != want_all[keep]).sum())
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from test_shared_records import shared_mutants
            sm = shared_mutants(base.cfg, shared[0], rng, N_STRUCT)
            exp = []
            for m in sm:
                orc, orec = O.shared_expand(base.cfg, m)
                exp.append(2 if orc else int(O.stwo_verify_batch([records.stwo_from_record(base.cfg, orec)], verifier.MODE_FIXTURE)[0]))
            got2 = ver.verify_stwo_shared_records(base.cfg, sm, verifier.MODE_FIXTURE)
            mism += int((got2 != np.array(exp, dtype=np.uint32)).sum())
            print("stwo %-16s shared records: %d of the mutants have a shared form, %d structure mutants (%d refused), mismatches %d"
                  % (name, len(keep), len(sm), sum(1 for e in exp if e == 2), mism), flush=True)
            bad += mism
        # MINIMAL records (one sorted, deduplicated decommitment per tree, verified without expansion: round 5): the mutants
        # that still have a minimal form (a mutation of a sibling the form drops leaves the honest record: accepted), and
        # mutants of the honest minimal record itself (bit flips, lists one element short / long, sizes); both modes,
        # against the oracle's layer-by-layer walk
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_minimal import corrupt_minimal
        qs = formats.stwo_queries(base)
        mins = []
        for m in muts[:max(1, N // 4)]:
            try:
                mins.append(verifier.stwo_minimise_record(base.cfg, verifier.stwo_record(m), formats.stwo_queries(m)))
            except ValueError:
                pass
        honest = verifier.stwo_minimise_record(base.cfg, verifier.stwo_record(base), qs)
        mins += [corrupt_minimal(honest, base.cfg, rng)[0] for _ in range(N_STRUCT)]
        mism = 0
        for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
            got = ver.verify_stwo_minimal_records(base.cfg, mins, mode)
            want = np.array([O.stwo_verify_minimal(base.cfg, r, mode) for r in mins], dtype=np.uint32)
            mism += int((got != want).sum())
        print("stwo %-16s minimal records: %d mutants (%d accepted, %d refused), both modes, mismatches %d"
              % (name, len(mins), int((want == 0).sum()), int((want == 2).sum()), mism), flush=True)
        bad += mism
    print("TOTAL mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
