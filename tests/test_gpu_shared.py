"""Shared records on the GPU (csrc/ss_shared.hip, ABI 2.2): the expansion kernel against the oracle's walk
(oracle/ss_oracle_shared.c) and the host twin, and verdicts of shared records against the per-query records they
stand for and against the oracle -- the reference's witness is one full path per query
(stwo-verifier/src/fri/queries.simf:41, scripts/generate_wit.py:36-42), so "identical to the reference" means
"identical to the per-query form".  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from stark_symphony_amd import formats, records, verifier
from oracle import oracle as O

from test_shared_records import _random_queries, _random_record, fixtures, shared_mutants
import stark_symphony_amd as ss

pytestmark = pytest.mark.gpu
SEED = 0x5EED2025


@pytest.fixture(scope="module")
def ver():
    return verifier.Verifier(0)


def test_expansion_kernel_equals_the_walk(ver):
    """Random shapes and positions (uniform, clustered, all equal, neighbours): GPU expansion == oracle == host twin."""
    rng = np.random.default_rng(SEED + 51)
    for case in range(40):
        L = int(rng.integers(3, 25))
        K = int(rng.integers(0, L - 1))
        Q = int(rng.choice([1, 2, 5, 16, 24, 32, 64]))
        cfg = ss.StwoConfig(int(rng.choice([1, 3, 4, 9])), max(1, L - 1), L, Q, K, 5)
        shared, want = [], []
        for i in range(7):
            qs = _random_queries(rng, L, Q, (case + i) % 4).astype(np.uint32)
            p = _random_record(rng, cfg, qs)
            shared.append(verifier.stwo_shared_record(p, qs))
            want.append(verifier.stwo_record(p))
        recs, outcome = ver.expand_shared_on_device(cfg, shared)
        assert outcome.tolist() == [0] * 7, case
        for i in range(7):
            orc, orec = O.shared_expand(cfg, shared[i])
            assert orc == 0 and np.array_equal(recs[i], orec) and np.array_equal(orec, want[i]), (case, i)


def test_expansion_kernel_on_malformed_records(ver):
    """Wrong hints, counts and sizes: outcome and record (zeroed when refused) equal the oracle's, record by record."""
    rng = np.random.default_rng(SEED + 52)
    for p in fixtures()[:4]:
        sh = verifier.stwo_shared_record(p)
        ms = [sh] + shared_mutants(p.cfg, sh, rng, 120) + [np.zeros(0, np.uint32), sh[:1].copy()]
        recs, outcome = ver.expand_shared_on_device(p.cfg, ms)
        refused = 0
        for m, rec, out in zip(ms, recs, outcome):
            orc, orec = O.shared_expand(p.cfg, m)
            hrc, hrec = verifier.stwo_unshare_record(p.cfg, m)
            assert int(out) == orc == hrc and np.array_equal(rec, orec) and np.array_equal(hrec, orec)
            refused += orc != 0
        assert outcome[0] == 0 and 30 < refused < 120


@pytest.mark.parametrize("mode", [verifier.MODE_FIXTURE, verifier.MODE_LITERAL])
def test_shared_records_verify_as_the_per_query_records(ver, mode):
    """The six fixtures + 40 corruptions each that still have a shared form + wrong hints / counts / sizes:
    status(shared record) == status(per-query record) == oracle, both modes."""
    rng = np.random.default_rng(SEED + 53 + mode)
    for base in fixtures():
        cfg = base.cfg
        qs = formats.stwo_queries(base)
        proofs, shared = [base], [verifier.stwo_shared_record(base, qs)]
        tries = 0
        while len(proofs) < 41 and tries < 400:
            tries += 1
            p = formats.stwo_corrupt(base, rng)[0]
            try:
                s = verifier.stwo_shared_record(p, qs)   # the hint stays the honest prover's
            except ValueError:
                continue                                 # queries disagree about a node / ragged path: no shared form
            proofs.append(p)
            shared.append(s)
        assert len(proofs) >= (41 if cfg.n_queries > 1 else 20), (cfg, len(proofs))
        want = O.stwo_verify_batch(proofs, mode)
        got_rec = ver.verify_stwo_records(cfg, [verifier.stwo_record(p) for p in proofs], mode)
        got_sh = ver.verify_stwo_shared_records(cfg, shared, mode)
        assert np.array_equal(got_rec, want) and np.array_equal(got_sh, want), cfg
        offs = np.zeros(len(shared) + 1, dtype=np.uint64)   # the same records back to back + offsets: no per-record Python work
        offs[1:] = np.cumsum([s.size for s in shared])
        assert np.array_equal(ver.verify_stwo_shared_records(cfg, (np.concatenate(shared), offs), mode), want), cfg
        assert (want == 0).sum() >= (1 if mode == verifier.MODE_FIXTURE else 0)
        # structure mutants (hints, counts, sizes, bits anywhere -- a flipped node is seen by every query that shares
        # it): whatever still expands is verified as the record it expands to; the rest is malformed
        ms = shared_mutants(cfg, shared[0], rng, 64)
        exp = []
        for m in ms:
            orc, orec = O.shared_expand(cfg, m)
            exp.append(verifier.B.STATUS_MALFORMED if orc else
                       int(O.stwo_verify_batch([records.stwo_from_record(cfg, orec)], mode)[0]))
        got = ver.verify_stwo_shared_records(cfg, ms, mode)
        assert got.tolist() == exp, cfg
        assert len(set(exp)) >= 3


def test_shared_records_across_chunks_and_sizes(ver):
    """Several upload chunks (records of the 2^20 shape, > 64 MiB in total), a refused record in the middle of a
    chunk, an over-long one, an empty one."""
    import os
    from conftest import GOLDEN
    ps = records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace20.npz"))
    rng = np.random.default_rng(SEED + 54)
    distinct, want = [], []
    for p in ps[:2]:
        qs = formats.stwo_queries(p)
        distinct.append(verifier.stwo_shared_record(p, qs))
        want.append(0)
        for _ in range(3):
            while True:
                c = formats.stwo_corrupt(p, rng)[0]
                try:
                    distinct.append(verifier.stwo_shared_record(c, qs))
                    break
                except ValueError:
                    pass
            want.append(int(O.stwo_verify_batch([c], O.MODE_FIXTURE)[0]))
    n = 1400   # x 138 KB = 190 MB: the doubling first chunks and three full ones
    idx = rng.integers(0, len(distinct), size=n)
    batch = [distinct[i] for i in idx]
    exp = [want[i] for i in idx]
    for j, bad in ((5, distinct[0][:-8].copy()), (700, np.concatenate([distinct[0], np.zeros(1 << 18, np.uint32)])),
                   (1399, np.zeros(0, np.uint32))):
        batch[j] = bad
        exp[j] = verifier.B.STATUS_MALFORMED
    got = ver.verify_stwo_shared_records(ps[0].cfg, batch, verifier.MODE_FIXTURE)
    assert got.tolist() == exp


# ------------------------------------------------------------------------------- shared-path proof.json on the GPU
import json
import random

from stark_symphony_amd import binding
from test_ingest import _text_mutant
from test_shared_paths import _canonical_shared, _write_shared_text

SHARED = binding.TEXT_JSON_SHARED


def _agree_shared(ver, cfg, texts):
    """GPU outcome == scalar rule (csrc/ss_text.cpp shared_text_scan_reference) for every text; same record where taken."""
    recs, outcome = ver.read_stwo_texts(cfg, texts, SHARED)
    taken = 0
    for i, t in enumerate(texts):
        want, rec = _canonical_shared(cfg, t)
        assert (outcome[i] == 0) == want, (i, int(outcome[i]), want, t[:120], t[-120:])
        if want:
            assert np.array_equal(recs[i], rec), (i, np.nonzero(recs[i] != rec)[0][:8])
            taken += 1
    return taken


def test_gpu_reader_takes_honest_shared_texts(ver):
    """Six shapes (1 to 32 queries, 4 to 256 columns, both hashes, LDE 2^4 to 2^24), both separator styles, trailing
    newline, indentation: read into shared records and expanded on the GPU; the record is the per-query record."""
    for p in fixtures():
        rec = verifier.stwo_record(p)
        obj = ss.stwo_to_json(p, shared=True)
        texts = [json.dumps(obj, separators=(",", ":")).encode(), json.dumps(obj).encode(), json.dumps(obj).encode() + b"\n",
                 json.dumps(obj, indent=1).encode(), _write_shared_text(p.cfg, verifier.stwo_shared_record(p), 0)]
        recs, outcome = ver.read_stwo_texts(p.cfg, texts, SHARED)
        assert outcome.tolist() == [0] * len(texts), p.cfg
        assert all(np.array_equal(r, rec) for r in recs), p.cfg
        # the per-query text is not of this format: left to the host reader
        _, out2 = ver.read_stwo_texts(p.cfg, [json.dumps(ss.stwo_to_json(p)).encode()], SHARED)
        assert out2.tolist() == [1] or p.cfg.n_queries == 1


def test_gpu_reader_equals_the_scalar_rule_on_shared_mutants(ver):
    """Byte-level mutants, other hints, other numbers, texts shifted against the 1 KiB window / 16-byte lane grid (the
    gap maps cut the template at arbitrary byte offsets), tails of every alignment."""
    rnd = random.Random(SEED + 61)
    total = 0
    for p, n in zip(fixtures()[:3], (500, 2500, 300)):
        obj = ss.stwo_to_json(p, shared=True)
        base = json.dumps(obj, separators=(",", ":")).encode()
        texts = [base] + [b" " * k + base for k in range(1, 40)] + [base + b" " * k for k in range(1, 40)]
        texts += [base[:-1] + b" " * k + b"}" for k in (1, 15, 16, 17, 1000, 1023, 1024, 1025)]   # blanks inside the tail
        for i in range(n):
            kind = i % 4
            if kind < 2:
                t = _text_mutant(rnd, base)
            elif kind == 2:
                o = json.loads(base)
                o["queries"][rnd.randrange(p.cfg.n_queries)] = rnd.randrange(1 << p.cfg.lde_log)
                t = json.dumps(o, separators=(",", ":")).encode()
            else:
                import re
                ms = list(re.finditer(rb"\d+", base))
                m = ms[rnd.randrange(len(ms))]
                t = base[:m.start()] + str(rnd.choice([0, 1, 255, 256, 2 ** 31, 2 ** 32 - 1, 2 ** 32])).encode() + base[m.end():]
            if i % 7 == 0:
                t = b"\n" * rnd.randrange(1, 30) + t
            texts.append(t)
        texts += [b"", b"}", b"]}", b"[1]}", b"1" * 3000, b" " * 5000, base[:len(base) // 2], base[len(base) // 2:]]
        total += _agree_shared(ver, p.cfg, texts)
    assert total > 400


def test_shared_texts_through_the_entry_point(ver):
    """ss_stwo_verify_texts on one batch of shared texts (GPU reader + expansion), per-query texts and .wit (GPU reader),
    shared texts with another member order or a hint that does not fit the lists (host reader), corrupted proofs,
    garbage: the status words of the oracle on the per-query proofs / the stage-0 codes."""
    base = fixtures()[0]
    cfg = base.cfg
    qs = formats.stwo_queries(base)
    rng = np.random.default_rng(SEED + 62)
    proofs = [base]
    while len(proofs) < 9:
        c = formats.stwo_corrupt(base, rng)[0]
        try:
            ss.stwo_to_json(c, shared=True, queries=qs)
            proofs.append(c)
        except ss.MalformedProof:
            pass
    want_d = O.stwo_verify_batch(proofs).tolist()
    variants, want, host = [], [], []
    for p, w in zip(proofs, want_d):
        obj = ss.stwo_to_json(p, shared=True, queries=qs)
        wrong = dict(obj)
        wrong["queries"] = [qs[1]] + qs[1:]              # two equal positions: the lists no longer have the lengths the hint implies
        variants += [json.dumps(obj).encode(), json.dumps(obj, separators=(",", ":")).encode(), json.dumps(ss.stwo_to_json(p)).encode(),
                     ss.stwo_to_wit(p).encode(), json.dumps(dict(reversed(list(obj.items())))).encode(), json.dumps(wrong).encode()]
        want += [w, w, w, w, w, 2]
        host += [0, 0, 0, 0, 1, 1]
    variants += [b"{\"queries\":[1]}", b"nonsense", b""]
    want += [2, 2, 2]
    host += [1, 1, 1]
    for n in (len(variants), 1500):
        batch = [variants[i % len(variants)] for i in range(n)]
        status, stats = ver.verify_stwo_texts(cfg, batch)
        assert status.tolist() == [want[i % len(variants)] for i in range(n)]
        assert stats["host_parsed"] == sum(host[i % len(variants)] for i in range(n))
    status, stats = ver.verify_stwo_texts(cfg, variants[:2], fmt=SHARED)
    assert status.tolist() == want[:2] and stats["host_parsed"] == 0


def test_full_size_shared_texts_end_to_end(ver):
    """2^20-row proofs as shared-path text (0.5 MB each), valid and corrupted, 400 texts over several chunks."""
    import os
    from conftest import GOLDEN
    p = records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace20.npz"))[0]
    qs = formats.stwo_queries(p)
    rng = np.random.default_rng(SEED + 63)
    proofs = [p]
    while len(proofs) < 4:
        c = formats.stwo_corrupt(p, rng)[0]
        try:
            verifier.stwo_shared_record(c, qs)
            proofs.append(c)
        except ValueError:
            pass
    want = O.stwo_verify_batch(proofs).tolist()
    texts = [_write_shared_text(q.cfg, verifier.stwo_shared_record(q, qs), i % 2) for i, q in enumerate(proofs)]
    full = len(json.dumps(ss.stwo_to_json(p), separators=(",", ":")))
    assert len(texts[0]) < 0.86 * full
    batch = [texts[i % 4] for i in range(400)]
    status, stats = ver.verify_stwo_texts(p.cfg, batch)
    assert status.tolist() == [want[i % 4] for i in range(400)] and stats["host_parsed"] == 0 and want[0] == 0


def test_many_short_records_do_not_size_the_scratch(ver):
    """20 000 records of a few words each against the 2^20 config: every one is SS_STATUS_MALFORMED, and the call's
    device scratch follows the 256 MiB of expanded records a chunk may hold, not 20 000 x 170 KB."""
    import os
    from conftest import GOLDEN
    cfg = records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace20.npz"))[0].cfg
    rng = np.random.default_rng(SEED + 55)
    batch = [rng.integers(0, 1 << 32, size=int(rng.integers(0, 40)), dtype=np.uint64).astype(np.uint32) for _ in range(20000)]
    import torch
    before = torch.cuda.mem_get_info(0)[0]
    got = ver.verify_stwo_shared_records(cfg, batch, verifier.MODE_FIXTURE)
    assert (got == verifier.B.STATUS_MALFORMED).all()
    assert before - torch.cuda.mem_get_info(0)[0] < (3 << 30)
