"""Minimal records on the GPU (csrc/ss_minimal.hip; SURVEY.md 8f row 4 "sorted multi-proof Merkle (real stwo format)"):
status(minimal record) == status(the per-query record R(M) it corresponds to) == the oracle's walk -- on the committed
fixtures, on seeded corruptions incl. lists that are too short / too long, on query counts that do not divide 64 and on
duplicate / neighbouring queries, in both modes, with the pair memoisation on and off.  Anchors:
stwo-verifier/src/fri/queries.simf:41, scripts/generate_wit.py:36-42, merkle.simf:22-44.  Parity unpinned (no bytes of the
form in the reference): the GPU is held to the oracle's restatement of upstream stwo's walk AND to the per-query path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import formats, records, verifier  # noqa: E402
from oracle import oracle as O  # noqa: E402

from test_minimal import corrupt_minimal, fixtures, per_query_status  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ver():
    return verifier.Verifier(0)


def minimal_of(p):
    return verifier.stwo_minimise_record(p.cfg, verifier.stwo_record(p), formats.stwo_queries(p))


def check(ver, cfg, recs, mode):
    """GPU(minimal) == oracle walk == GPU(per-query R(M)) == oracle(per-query R(M)) for every record of `recs`."""
    got = ver.verify_stwo_minimal_records(cfg, recs, mode)
    want = np.array([O.stwo_verify_minimal(cfg, r, mode) for r in recs], dtype=np.uint32)
    assert got.tolist() == want.tolist(), [(i, hex(int(g)), hex(int(w))) for i, (g, w) in enumerate(zip(got, want)) if g != w][:8]
    full, idx = [], []
    for i, r in enumerate(recs):
        st, back = O.stwo_minimal_expand(cfg, r, mode)
        assert st == want[i]
        if st != 2:
            full.append(back)
            idx.append(i)
            assert per_query_status(cfg, back, mode) == st
    if full:
        std = ver.verify_stwo_records(cfg, np.stack(full), mode)
        assert std.tolist() == want[idx].tolist()
    return want


@pytest.mark.parametrize("i", range(6))
def test_fixtures_accept_and_corruptions_match(ver, i):
    """The six committed fixtures: accepted as minimal records in FIXTURE mode, the per-query status in LITERAL mode;
    40 corruptions each (bit flips, lists one element short / long, truncated, extended), both modes."""
    p = fixtures()[i]
    cfg = p.cfg
    m = minimal_of(p)
    rng = np.random.default_rng(0x5EED2025 + 80 + i)
    muts = [corrupt_minimal(m, cfg, rng)[0] for _ in range(40)]
    for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
        want = check(ver, cfg, [m] + muts, mode)
        assert int(want[0]) == O.stwo_verify(p, mode)
        if mode == verifier.MODE_FIXTURE:
            assert want[0] == 0 and (want[1:] != 0).sum() >= 30


@pytest.mark.parametrize("flags", [verifier.FLAG_NO_DEDUP, verifier.FLAG_TOP_CHECKS])
def test_flags_do_not_change_the_verdicts(flags, ver):
    """SS_FLAG_NO_DEDUP: no top kernel, every omitted sibling comes from a lane of the merkle kernel.  SS_FLAG_TOP_CHECKS has
    nothing to act on (a minimal record holds every sibling once)."""
    v2 = verifier.Verifier(0)
    v2.stwo_flags = flags
    for i in (0, 2):
        p = fixtures()[i]
        m = minimal_of(p)
        rng = np.random.default_rng(0x5EED2025 + 90 + i)
        recs = [m] + [corrupt_minimal(m, p.cfg, rng)[0] for _ in range(24)]
        a = v2.verify_stwo_minimal_records(p.cfg, recs, verifier.MODE_FIXTURE)
        b = ver.verify_stwo_minimal_records(p.cfg, recs, verifier.MODE_FIXTURE)
        assert a.tolist() == b.tolist() == [O.stwo_verify_minimal(p.cfg, r, verifier.MODE_FIXTURE) for r in recs]
        assert a[0] == 0


@pytest.mark.parametrize("kw", [
    dict(n_cols=4, trace_log=5, log_blowup=2, n_queries=3, pow_bits=5, seed=0, hash="sha256"),   # 3 queries: padded to 4
    dict(n_cols=8, trace_log=4, log_blowup=1, n_queries=9, pow_bits=3, seed=7, hash="sha256"),   # 9 of 32 positions: siblings, duplicates
    dict(n_cols=3, trace_log=2, log_blowup=1, n_queries=24, pow_bits=0, seed=2, hash="sha256"),  # 24 queries on 8 positions
    dict(n_cols=3, trace_log=1, log_blowup=2, n_queries=64, pow_bits=1, seed=4, hash="sha256"),  # 64 queries on 8 positions
    dict(n_cols=32, trace_log=6, log_blowup=3, n_queries=5, pow_bits=8, seed=1, hash="blake2s"),
    dict(n_cols=5, trace_log=12, log_blowup=2, n_queries=11, pow_bits=10, seed=5, hash="blake2s"),
    dict(n_cols=4, trace_log=10, log_blowup=3, n_queries=48, pow_bits=4, seed=9, hash="sha256"),
])
def test_any_query_count_duplicates_and_neighbours(ver, kw):
    """Prover-made proofs of shapes whose queries collide: query counts that do not divide 64 (the kernels give a proof
    the next power of two of chains, the extra ones repeating query 0), more queries than positions, sibling positions."""
    from stark_symphony_amd import prover
    p = ss.stwo_from_json(prover.GpuProver(ver).prove(**kw))
    cfg = p.cfg
    qs = formats.stwo_queries(p)
    m = minimal_of(p)
    assert np.array_equal(m, verifier.stwo_minimal_record(formats.stwo_minimise(p, qs)))
    rng = np.random.default_rng(0x5EED2025 + 100 + kw["seed"])
    recs = [m] + [corrupt_minimal(m, cfg, rng)[0] for _ in range(30)]
    for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
        want = check(ver, cfg, recs, mode)
        if mode == verifier.MODE_FIXTURE:
            assert want[0] == 0


def test_batch_of_mixed_records_across_chunks(ver):
    """2 500 minimal records of the reference's production shape in one call (several upload chunks): the accepted
    fixture, corruptions, malformed ones; statuses in input order."""
    p = fixtures()[0]
    m = minimal_of(p)
    rng = np.random.default_rng(0x5EED2025 + 120)
    pool = [m] + [corrupt_minimal(m, p.cfg, rng)[0] for _ in range(48)]
    want_pool = [O.stwo_verify_minimal(p.cfg, r, verifier.MODE_FIXTURE) for r in pool]
    order = rng.integers(0, len(pool), size=2500)
    got = ver.verify_stwo_minimal_records(p.cfg, [pool[i] for i in order], verifier.MODE_FIXTURE)
    assert got.tolist() == [want_pool[i] for i in order]
    assert (got == 0).sum() > 20 and (got == 2).sum() > 20


def test_resident_minimal_batch_and_phases(ver):
    """ss_stwo_verify_minimal_dev on records resident in HBM: HEAD and TAIL enqueued separately give the one-call result;
    the minimal records are smaller than the per-query ones by the share the shapes predict."""
    import torch
    p = fixtures()[5]  # 2^20 shape
    m = minimal_of(p)
    full_words = verifier.stwo_record(p).size
    assert 0.6 < m.size / full_words < 0.8
    rng = np.random.default_rng(0x5EED2025 + 130)
    recs = [m] + [corrupt_minimal(m, p.cfg, rng)[0] for _ in range(7)]
    recs = [r for r in recs if O.stwo_verify_minimal(p.cfg, r, verifier.MODE_FIXTURE) != 2 or r.size >= 64]
    idx = [int(i) for i in rng.integers(0, len(recs), size=512)]
    b = ver.stwo_minimal_batch(p.cfg, recs, verifier.MODE_FIXTURE, index=idx)
    b.run()
    one = b.status()
    b.run(phases=verifier.PHASE_HEAD)
    b.run(phases=verifier.PHASE_TAIL)
    torch.cuda.synchronize()
    two = b.status()
    want = [O.stwo_verify_minimal(p.cfg, r, verifier.MODE_FIXTURE) for r in recs]
    assert one.tolist() == two.tolist() == [want[i] for i in idx]
    assert b.accepted() == sum(1 for i in idx if want[i] == 0)


def test_minimal_texts_and_cli(ver, tmp_path):
    """The minimal proof.json as text: ss_stwo_verify_minimal_texts (GPU reader / host readers, then the minimal-record path) gives the
    records' verdicts and the stage-0 codes for texts that are no witness / of another config; `cli verify
    --minimal-proof` keeps the exit-status contract of simfony-cli/src/main.rs:254-257; `cli convert --to json-minimal`
    writes the text the library's writer writes."""
    import json
    import subprocess
    p = fixtures()[0]
    cfg = p.cfg
    m = minimal_of(p)
    rng = np.random.default_rng(0x5EED2025 + 150)
    recs = [m] + [r for r in (corrupt_minimal(m, cfg, rng)[0] for _ in range(40)) if O.stwo_verify_minimal(cfg, r, 1) != 2][:12]
    texts = [verifier.write_stwo_minimal_text(cfg, r) for r in recs]
    other = formats.stwo_minimal_to_json(formats.stwo_minimise(fixtures()[1]))
    texts += [b"not a witness", json.dumps(other).encode()]
    got, stats = ver.verify_stwo_minimal_texts(cfg, texts)
    want = [O.stwo_verify_minimal(cfg, r, 1) for r in recs] + [2, 1]
    assert got.tolist() == want and got[0] == 0 and stats["host_parsed"] == 2  # (the canonical texts are read on the GPU)
    good, bad = tmp_path / "good.json", tmp_path / "bad.json"
    good.write_bytes(texts[0])
    bad.write_bytes(texts[1])
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, "-m", "stark_symphony_amd.cli", *a], env=env, capture_output=True, text=True)
    r = run("verify", "--family", "stwo", "--minimal-proof", str(good))
    assert r.returncode == 0 and "ACCEPT" in r.stdout
    r = run("verify", "--family", "stwo", "--minimal-proof", str(good), str(bad))
    assert r.returncode == 1 and "REJECT" in r.stderr
    r = run("convert", "--family", "stwo", "--to", "json-minimal", os.path.join(ROOT, "tests", "golden", "stwo_proof.json"))
    assert r.returncode == 0 and r.stdout.strip().encode() == texts[0]


def test_pinned_entry_points_give_the_staged_verdicts(ver):
    """ss_stwo_verify_*_pinned (csrc/ss_pinned.hip): records, shared records and minimal records lying back to back in ONE
    page-locked buffer -- torch's pinned allocator, or ordinary memory passed to ss_host_register -- are read by the DMA
    engine directly; verdicts are those of the staged entry points on the same inputs; ordinary memory is refused."""
    from stark_symphony_amd import binding as B
    p = fixtures()[0]
    cfg = p.cfg
    rng = np.random.default_rng(0x5EED2025 + 160)
    full = [verifier.stwo_record(p)] + [verifier.stwo_record(formats.stwo_corrupt(p, rng)[0]) for _ in range(700)]
    m = minimal_of(p)
    mins = [m] + [corrupt_minimal(m, cfg, rng)[0] for _ in range(300)]
    qs = formats.stwo_queries(p)
    shared = []
    for r in full[:300]:
        try:
            shared.append(verifier.stwo_shared_record(records.stwo_from_record(cfg, r), qs))
        except ValueError:  # (a corruption made two queries disagree: no shared form)
            pass
    shared.append(shared[0][:40].copy())  # not a shared record at all

    def flat_of(recs, pinned):
        offs = np.zeros(len(recs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([r.size for r in recs])
        buf = ver.pinned_buffer(int(offs[-1])) if pinned else np.zeros(int(offs[-1]), dtype=np.uint32)
        buf[:] = np.concatenate(recs)
        return buf, offs
    for pinned in (True, False):
        f, _ = flat_of(full, pinned)
        s, so = flat_of(shared, pinned)
        mm, mo = flat_of(mins, pinned)
        if not pinned:
            with pytest.raises(B.SsError):  # pageable memory: refused, not silently staged
                ver.verify_stwo_pinned(cfg, f)
            for a in (f, s, mm):
                ver.register_host(a)
        try:
            assert ver.verify_stwo_pinned(cfg, f).tolist() == ver.verify_stwo_records(cfg, np.stack(full)).tolist()
            assert ver.verify_stwo_pinned(cfg, s, so, "shared").tolist() == ver.verify_stwo_shared_records(cfg, shared).tolist()
            got = ver.verify_stwo_pinned(cfg, mm, mo, "minimal")
            assert got.tolist() == ver.verify_stwo_minimal_records(cfg, mins).tolist()
            assert got[0] == 0 and (got == 2).any()
        finally:
            if not pinned:
                for a in (f, s, mm):
                    ver.unregister_host(a)


def test_pinned_texts_give_the_staged_verdicts(ver):
    """ss_stwo_verify_texts_pinned: proof.json / proof.wit / shared-path texts lying in ONE page-locked buffer (each at a
    multiple of 16) are read by the DMA engine where they are -- canonical texts by the GPU reader, the others (another member
    order, another config, garbage) by the host reader from the same buffer; verdicts and the host reader's share are those
    of ss_stwo_verify_texts.  Misaligned offsets and pageable memory are refused."""
    import json
    from stark_symphony_amd import binding as B
    p = fixtures()[0]
    cfg = p.cfg
    rng = np.random.default_rng(0x5EED2025 + 170)
    proofs = [p] + [formats.stwo_corrupt(p, rng)[0] for _ in range(40)]
    obj = ss.stwo_to_json(p)
    texts = []
    for i, q in enumerate(proofs * 8):
        texts.append(json.dumps(ss.stwo_to_json(q), separators=(",", ":")).encode() if i % 3 == 0 else
                     ss.stwo_to_wit(q).encode() if i % 3 == 1 else json.dumps(ss.stwo_to_json(q)).encode())
    texts += [json.dumps(ss.stwo_to_json(p, shared=True), separators=(",", ":")).encode(),
              json.dumps(dict(reversed(list(obj.items())))).encode(), b"not a witness", b"",
              json.dumps(ss.stwo_to_json(fixtures()[1])).encode()]
    want, wstats = ver.verify_stwo_texts(cfg, texts)
    blob, offs, lens = ver.pinned_text_blob(texts)
    got, stats = ver.verify_stwo_texts_pinned(cfg, blob, offs, lens)
    assert got.tolist() == want.tolist() and stats["host_parsed"] == wstats["host_parsed"] >= 3
    assert got[0] == 0 and (got == 2).sum() >= 2 and (got == 1).sum() >= 1
    bad = offs.copy()
    bad[3] += 8
    with pytest.raises(B.SsError):
        ver.verify_stwo_texts_pinned(cfg, blob, bad, lens)
    pageable = np.array(blob)
    with pytest.raises(B.SsError):
        ver.verify_stwo_texts_pinned(cfg, pageable, offs, lens)


@pytest.mark.parametrize("i", [0, 1, 2, 3, 5])
def test_gpu_reader_of_the_minimal_text_equals_its_scalar_rule(ver, i):
    """ss_stwo_read_texts with SS_TEXT_JSON_MINIMAL (csrc/ss_textdev.hip: summary, scan, text_landmark_kernel,
    text_minhint_kernel, place through the gap maps): outcome == the scalar rule (csrc/ss_text.cpp
    minimal_text_scan_reference, itself held to the host reader in tests/test_minimal.py) for every text, and the same
    minimal record where taken -- fixtures in three spellings, texts of corrupted records (list lengths that are data),
    empty lists, byte-level mutants, another member order."""
    import json
    from stark_symphony_amd import binding as B
    from test_minimal import _min_text_mutants
    p = fixtures()[i]
    cfg = p.cfg
    m = minimal_of(p)
    obj = formats.stwo_minimal_to_json(formats.stwo_minimise(p))
    rng = np.random.default_rng(0x5EED2025 + 180 + i)
    texts = [json.dumps(obj).encode(), json.dumps(obj, separators=(",", ":")).encode(), json.dumps(obj, indent=1).encode()]
    for k in range(24):
        r = corrupt_minimal(m, cfg, rng)[0]
        try:
            texts.append(verifier.write_stwo_minimal_text(cfg, r, python_separators=bool(k & 1)))
        except ValueError:  # (no minimal record of the config any more: no text)
            pass
    texts += [t for t in _min_text_mutants(json.dumps(obj).encode(), rng) if len(t)]
    K = cfg.n_layers
    def variant(edit):
        o = json.loads(json.dumps(obj))
        edit(o)
        return json.dumps(o, separators=(",", ":")).encode()
    layer = lambda o, l: o["fri_proof"]["first_layer"] if l == 0 else o["fri_proof"]["inner_layers"][l - 1]
    def all_empty(o):
        o["queried_values"][1] = []
        o["queried_values"][2] = []
        o["decommitments"][1]["hash_witness"] = []
        o["decommitments"][2]["hash_witness"] = []
        for l in range(K + 1):
            layer(o, l)["fri_witness"] = []
            layer(o, l)["decommitment"]["hash_witness"] = []
    texts += [variant(lambda o: layer(o, 0).__setitem__("fri_witness", [])),
              variant(lambda o: layer(o, K)["decommitment"].__setitem__("hash_witness", [])),
              variant(lambda o: o["decommitments"][2]["hash_witness"].pop()), variant(all_empty),
              variant(lambda o: [o["queried_values"][1].pop() for _ in range(cfg.n_cols)]),
              variant(lambda o: o.__setitem__("hash_witness", [])), json.dumps(dict(reversed(list(obj.items())))).encode()]
    # more landmarks than the per-text table holds (member names in a trailing object, in one window and spread over many)
    many = {"hash_witness_%d" % k: k for k in range(50)}
    texts += [variant(lambda o: o.__setitem__("x", many)), variant(lambda o: o.__setitem__("x", {k + " " * 900: v for k, v in many.items()})),
              variant(lambda o: o.__setitem__("x", {"column_witness_%d" % k: [k] * 40 for k in range(45)})),
              variant(lambda o: o.__setitem__("x", {"proof_of_work_%d" % k: k for k in range(3)}))]
    recs, outcome = ver.read_stwo_texts(cfg, texts, B.TEXT_JSON_MINIMAL)
    taken = 0
    for k, text in enumerate(texts):
        want_taken, want = verifier.stwo_minimal_text_is_canonical(cfg, text)
        assert (outcome[k] == 0) == want_taken, (k, int(outcome[k]), want_taken, text[:60])
        if want_taken:
            taken += 1
            assert np.array_equal(verifier.stwo_minimal_from_capacity(cfg, recs[k]), want), k
    assert outcome[0] == outcome[1] == outcome[2] == 0 and 8 <= taken < len(texts)


def test_minimal_texts_through_the_gpu_reader_give_the_records_verdicts(ver, tmp_path):
    """ss_stwo_verify_minimal_texts end to end at the metric shape (2^20 rows: 0.43-0.54 MB a text, several chunks of the
    pipeline): texts of the fixture and of corrupted records in both spellings go through the GPU reader, texts in another
    member order through the host readers, garbage and another config get the stage-0 codes; every verdict is the oracle's
    for the record; the caller-pinned variant gives the same verdicts."""
    import json
    p = fixtures()[5]
    cfg = p.cfg
    m = minimal_of(p)
    rng = np.random.default_rng(0x5EED2025 + 190)
    recs, texts = [], []
    for r in [m] + [corrupt_minimal(m, cfg, rng)[0] for _ in range(60)]:
        st = O.stwo_verify_minimal(cfg, r, 1)
        if st == 2:
            continue
        recs.append((r, st))
    want = []
    for k in range(360):
        r, st = recs[k % len(recs)]
        texts.append(verifier.write_stwo_minimal_text(cfg, r, python_separators=bool(k % 3 == 0)))
        want.append(st)
    obj = formats.stwo_minimal_to_json(formats.stwo_minimise(p))
    other = formats.stwo_minimal_to_json(formats.stwo_minimise(fixtures()[0]))
    texts += [json.dumps(dict(reversed(list(obj.items())))).encode(), b"not a witness", b"", json.dumps(other).encode(),
              json.dumps(obj, indent=1).encode()]
    want += [0, 2, 2, 1, 0]
    got, stats = ver.verify_stwo_minimal_texts(cfg, texts)
    assert got.tolist() == want, [(k, int(g), w) for k, (g, w) in enumerate(zip(got, want)) if g != w][:8]
    # the host readers' share: exactly the texts the GPU reader's rule does not take (the last five but the indented one,
    # and corrupted records whose two value lists no longer describe the same positions)
    host = sum(1 for t in texts if not verifier.stwo_minimal_text_is_canonical(cfg, t)[0])
    assert stats["host_parsed"] == host and 4 <= host < 40 and len(set(want)) >= 4
    blob, offs, lens = ver.pinned_text_blob(texts)
    got2, stats2 = ver.verify_stwo_minimal_texts_pinned(cfg, blob, offs, lens)
    assert got2.tolist() == want and stats2["host_parsed"] == host
    # the general text entry points take the form by name (never by SS_TEXT_AUTO), files included
    from stark_symphony_amd import binding as B
    got3, _ = ver.verify_stwo_texts(cfg, texts[-40:], fmt=B.TEXT_JSON_MINIMAL)
    assert got3.tolist() == want[-40:]
    paths = []
    for k, t in enumerate(texts[-12:]):
        paths.append(str(tmp_path / ("m%02d.json" % k)))
        open(paths[-1], "wb").write(t)
    paths.append(str(tmp_path / "absent.json"))
    got4, stats4 = ver.verify_stwo_files(cfg, paths, fmt=B.TEXT_JSON_MINIMAL)
    assert got4.tolist() == want[-12:] + [2]
    auto, _ = ver.verify_stwo_texts(cfg, texts[:2])  # (read as a per-query proof.json: its lists are too short for that form)
    assert (auto != 0).all()


def test_committed_minimal_texts_are_accepted_on_the_gpu(ver):
    """tests/golden/formats/*.minimal.json (the reference's two proofs in the minimal form, as this repository's writers
    print it; tests/test_minimal.py holds writers and readers to those bytes): read by the GPU reader, accepted."""
    import json
    from stark_symphony_amd import binding as B
    golden = os.path.join(ROOT, "tests", "golden")
    for name in ("stwo_proof", "stwo_proof_test"):
        text = open(os.path.join(golden, "formats", name + ".minimal.json"), "rb").read()
        cfg = ss.stwo_from_json(json.load(open(os.path.join(golden, name + ".json")))).cfg
        st, stats = ver.verify_stwo_texts(cfg, [text, text], fmt=B.TEXT_JSON_MINIMAL)
        assert st.tolist() == [0, 0] and stats["host_parsed"] == 0
        assert ver.verify_stwo_texts(cfg, [text], mode=verifier.MODE_LITERAL, fmt=B.TEXT_JSON_MINIMAL)[0].tolist() == \
            [O.stwo_verify_minimal(cfg, verifier.parse_stwo_minimal_text(cfg, text)[1], verifier.MODE_LITERAL)]
