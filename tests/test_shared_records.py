"""Shared RECORDS (include/ss_verify.h, ABI 2.2): the binary twin of the shared-path proof.json -- every distinct
Merkle sibling of a tree once + the query positions as an untrusted hint; the reference presents one full path per
query (stwo-verifier/src/fri/queries.simf:41, scripts/generate_wit.py:36-42).  The library computes the first-use
order in closed form (csrc/ss_shared.h); here it is compared with the DEFINITION of the order, the walk, stated
twice independently: oracle/ss_oracle_shared.c and formats.shared_path_order.  CPU only; the GPU expansion is in
tests/test_gpu_shared.py."""
import json
import os

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import formats, records, verifier
from oracle import oracle as O

from conftest import GOLDEN


def fixtures():
    out = [ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))),
           ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))]
    for npz in ("stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace16_blake2s.npz", "stwo_trace20.npz"):
        out.append(records.load_stwo_npz(os.path.join(GOLDEN, npz))[0])
    return out


def _random_queries(rng, L, Q, kind):
    if kind == 0:    # uniform
        return rng.integers(0, 1 << L, size=Q)
    if kind == 1:    # clustered: many shared prefixes, duplicates
        base = int(rng.integers(0, 1 << L))
        return np.array([(base ^ int(rng.integers(0, 1 << int(rng.integers(0, min(L, 6) + 1))))) for _ in range(Q)])
    if kind == 2:    # all equal
        return np.full(Q, int(rng.integers(0, 1 << L)))
    return np.array([(i * 2 + int(rng.integers(0, 2))) % (1 << L) for i in range(Q)])  # neighbours: siblings of each other


def test_closed_form_equals_the_walk():
    """counts and node indices: library (closed form) == oracle walk == formats.shared_path_order."""
    rng = np.random.default_rng(0x5EED2025 + 41)
    for case in range(160):
        L = int(rng.integers(2, 25))
        K = int(rng.integers(0, L - 1))
        Q = int(rng.choice([1, 2, 3, 7, 16, 24, 32, 64]))
        cfg = ss.StwoConfig(4, max(1, L - 1), L, Q, K, 5)
        qs = _random_queries(rng, L, Q, case % 4).astype(np.uint32)
        counts = verifier.stwo_shared_counts(cfg, qs)
        py = formats.shared_path_order(L, K, [int(q) for q in qs])
        for t in range(K + 3):
            plan, count = O.shared_walk(L, t, qs)
            ln, rows, pcount = py[t]
            assert count == pcount == int(counts[t]), (case, t)
            assert plan.tolist() == rows, (case, t)


def _random_record(rng, cfg, qs):
    """A per-query record whose paths agree wherever the positions `qs` make them meet (random node bytes)."""
    N, L, Q, K = cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers
    plans = formats.shared_path_order(L, K, [int(q) for q in qs])
    nodes = [rng.integers(0, 256, size=(count, 32), dtype=np.uint8) for _, _, count in plans]

    def paths(t):
        ln, rows, _ = plans[t]
        return [nodes[t][row] for row in rows]
    u32 = lambda *shape: rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
    return ss.StwoProof(cfg, rng.integers(0, 256, size=(3, 32), dtype=np.uint8), u32(N, 4), u32(16, 4), u32(Q, N), u32(Q, 16),
                        paths(0), paths(1), rng.integers(0, 256, size=(K + 1, 32), dtype=np.uint8), u32(4), u32(K + 1, Q, 4),
                        [paths(2 + l) for l in range(K + 1)], int(rng.integers(0, 1 << 62)))


def test_share_then_unshare_is_the_identity():
    rng = np.random.default_rng(0x5EED2025 + 42)
    for case in range(60):
        L = int(rng.integers(3, 22))
        K = int(rng.integers(0, L - 1))
        Q = int(rng.choice([1, 2, 5, 16, 24, 64]))
        cfg = ss.StwoConfig(int(rng.choice([1, 3, 4, 9])), max(1, L - 1), L, Q, K, 5)
        qs = _random_queries(rng, L, Q, case % 4).astype(np.uint32)
        p = _random_record(rng, cfg, qs)
        rec = verifier.stwo_record(p)
        sh = verifier.stwo_shared_record(p, qs)
        fixed = ss_fixed = 24 + 4 * cfg.n_cols + 64 + 8 * (K + 1) + 6 + Q * (cfg.n_cols + 16) + 4 * Q * (K + 1) + Q + K + 3
        counts = verifier.stwo_shared_counts(cfg, qs)
        assert sh.size == fixed + 8 * int(counts.sum()) and sh.size <= rec.size + Q + K + 3
        rc, back = verifier.stwo_unshare_record(cfg, sh)
        assert rc == 0 and np.array_equal(back, rec), case
        orc, oback = O.shared_expand(cfg, sh)
        assert orc == 0 and np.array_equal(oback, rec), case


def test_fixtures_as_shared_records():
    """The six fixtures: the shared record expands to the record (library and oracle), holds the node lists of the
    shared-path proof.json, and is 9-21 % smaller at 16 / 32 queries."""
    for p in fixtures():
        rec = verifier.stwo_record(p)
        qs = formats.stwo_queries(p)
        sh = verifier.stwo_shared_record(p)
        rc, back = verifier.stwo_unshare_record(p.cfg, sh)
        orc, oback = O.shared_expand(p.cfg, sh)
        assert rc == 0 and orc == 0 and np.array_equal(back, rec) and np.array_equal(oback, rec)
        if p.cfg.n_queries >= 16:
            assert 0.65 < sh.size / rec.size < 0.93, (p.cfg, sh.size / rec.size)  # 28 % smaller at LDE 2^13, 12 % at 2^24
        obj = ss.stwo_to_json(p, shared=True)
        lists = [obj["decommitments"][1]["hash_witness"], obj["decommitments"][2]["hash_witness"]]
        lists += [l["decommitment"]["hash_witness"] for l in [obj["fri_proof"]["first_layer"]] + obj["fri_proof"]["inner_layers"]]
        flat = np.array([b for lst in lists for node in lst for b in node], dtype=np.uint8)
        K, Q, N = p.cfg.n_layers, p.cfg.n_queries, p.cfg.n_cols
        fixed = sh.size - flat.size // 4
        assert np.array_equal(sh[fixed:], flat.view(">u4").astype(np.uint32))
        assert sh[fixed - (K + 3) - Q:fixed - (K + 3)].tolist() == qs and obj["queries"] == qs
        assert sh[fixed - (K + 3):fixed].tolist() == [len(x) for x in lists]


def test_records_without_a_shared_form():
    p = fixtures()[0]
    qs = formats.stwo_queries(p)
    d = {}
    for i, q in enumerate(qs):
        d.setdefault(q >> (p.cfg.lde_log - 1), []).append(i)
    a, b = next(v for v in d.values() if len(v) > 1)[:2]  # two queries in the same half of the tree: they share the top sibling
    bad = p.copy()
    bad.trace_paths[b] = bad.trace_paths[b].copy()
    bad.trace_paths[b][-1, 0] ^= 1                        # the top sibling: shared with query a
    with pytest.raises(ValueError):
        verifier.stwo_shared_record(bad, qs)
    short = p.copy()
    short.cp_paths[3] = short.cp_paths[3][:-1]
    with pytest.raises(ValueError):
        verifier.stwo_shared_record(short, qs)
    from stark_symphony_amd import binding
    with pytest.raises(binding.SsError):  # a position outside the LDE domain is the caller's error (SS_ERR_ARG), not "no shared form"
        verifier.stwo_shared_record(p, [1 << p.cfg.lde_log] + qs[1:])


def shared_mutants(cfg, sh, rng, n):
    """Mutants of a shared record around its structure: hints, counts, size; bit flips anywhere."""
    N, L, Q, K = cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers
    fixed = 24 + 4 * N + 64 + 8 * (K + 1) + 6 + Q * (N + 16) + 4 * Q * (K + 1) + Q + K + 3
    qry, cnt = fixed - (K + 3) - Q, fixed - (K + 3)
    out = []
    for i in range(n):
        m = sh.copy()
        kind = i % 8
        if kind == 0:
            m[qry + int(rng.integers(0, Q))] = int(rng.integers(0, 1 << L))          # another position inside the domain
        elif kind == 1:
            m[qry + int(rng.integers(0, Q))] = (1 << L) + int(rng.integers(0, 1 << 8))  # outside it
        elif kind == 2:
            j = cnt + int(rng.integers(0, K + 3)); m[j] = (int(m[j]) + int(rng.choice([1, -1, 7]))) & 0xFFFFFFFF      # count that no longer matches
        elif kind == 3:
            m = m[:int(rng.integers(0, m.size))]                                      # truncated
        elif kind == 4:
            m = np.concatenate([m, rng.integers(0, 1 << 32, size=int(rng.integers(1, 40)), dtype=np.uint64).astype(np.uint32)])
        elif kind == 5:
            a, b = rng.integers(0, Q, size=2)
            m[qry + a], m[qry + b] = m[qry + b], m[qry + a]                           # hints swapped
        elif kind == 6:
            w = int(rng.integers(0, m.size))
            m[w] ^= np.uint32(1 << int(rng.integers(0, 32)))                          # a bit anywhere
        else:
            m[qry:qry + Q] = m[qry]                                                   # every hint the same
        out.append(np.ascontiguousarray(m))
    return out


def test_malformed_shared_records_host_equals_oracle():
    rng = np.random.default_rng(0x5EED2025 + 43)
    for p in fixtures()[:3]:
        sh = verifier.stwo_shared_record(p)
        bad = 0
        for m in shared_mutants(p.cfg, sh, rng, 160):
            rc, rec = verifier.stwo_unshare_record(p.cfg, m)
            orc, orec = O.shared_expand(p.cfg, m)
            assert rc == orc and np.array_equal(rec, orec)
            bad += rc != 0
        assert 40 < bad < 150


def test_closed_form_property_based():
    """Hypothesis over positions and shapes: the library's counts are the walk's, a random record that agrees wherever
    its positions meet survives share -> unshare, and expanding with the ORACLE's walk gives the same record."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=120, deadline=None)
    @given(st.integers(2, 20), st.data())
    def prop(L, data):
        K = data.draw(st.integers(0, L - 2))
        Q = data.draw(st.sampled_from([1, 2, 3, 6, 16, 17, 48, 64]))
        # positions with many shared prefixes: a few random bases XOR small offsets
        bases = data.draw(st.lists(st.integers(0, (1 << L) - 1), min_size=1, max_size=3))
        qs = np.array([(bases[data.draw(st.integers(0, len(bases) - 1))] ^ data.draw(st.integers(0, min((1 << L) - 1, 31)))) for _ in range(Q)],
                      dtype=np.uint32)
        cfg = ss.StwoConfig(3, max(1, L - 1), L, Q, K, 5)
        counts = verifier.stwo_shared_counts(cfg, qs)
        for t in range(K + 3):
            assert O.shared_walk(L, t, qs)[1] == int(counts[t])
        rng = np.random.default_rng(int(qs.sum()) + L + K + Q)
        p = _random_record(rng, cfg, qs)
        rec = verifier.stwo_record(p)
        sh = verifier.stwo_shared_record(p, qs)
        rc, back = verifier.stwo_unshare_record(cfg, sh)
        orc, oback = O.shared_expand(cfg, sh)
        assert rc == 0 and orc == 0 and np.array_equal(back, rec) and np.array_equal(oback, rec)
    prop()
