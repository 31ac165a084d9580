"""The shared-path variant of proof.json (second half): every distinct Merkle sibling of a tree
once + the query positions, instead of one full path per query (the reference notes that it does not deduplicate:
stwo-verifier/src/fri/queries.simf:41; its adapter splits per-query witnesses, scripts/generate_wit.py:36-40).
No reference bytes exist for such a format, so parity is defined through expansion: a shared text must read, in
formats.py and in the native reader alike, as exactly the per-query proof it was made from -- after which the
verifier (and the oracle) see nothing new.  The second half of the file holds the scalar statement of the GPU reader's rule for
these texts against the host reader.  CPU only; the GPU legs are in tests/test_gpu_text.py and tests/test_gpu_shared.py."""
import json
import os
import random

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import formats, records, verifier
from oracle import oracle as O

from conftest import GOLDEN
from test_ingest import MALFORMED, MISMATCH, OK, _python_outcome, _text_mutant


def _fixtures():
    out = [ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))),
           ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))]
    for npz in ("stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace16_blake2s.npz"):
        out.append(records.load_stwo_npz(os.path.join(GOLDEN, npz))[0])
    return out


def test_query_positions_from_the_public_transcript():
    """formats.stwo_queries replays the Fiat-Shamir transcript with hashlib; the oracle's verifier draws the same."""
    for p in _fixtures():
        st, tr = O.stwo_verify(p, O.MODE_FIXTURE, trace=True)
        assert st == 0 and formats.stwo_queries(p) == [int(q) for q in list(tr.queries)[:p.cfg.n_queries]]


def test_shared_text_reads_as_the_proof_it_was_made_from():
    for p in _fixtures():
        rec = verifier.stwo_record(p)
        obj = ss.stwo_to_json(p, shared=True)
        full = ss.stwo_to_json(p)
        n_shared = sum(len(d["hash_witness"]) for d in obj["decommitments"]) + sum(
            len(l["decommitment"]["hash_witness"]) for l in [obj["fri_proof"]["first_layer"]] + obj["fri_proof"]["inner_layers"])
        n_full = sum(len(d["hash_witness"]) for d in full["decommitments"]) + sum(
            len(l["decommitment"]["hash_witness"]) for l in [full["fri_proof"]["first_layer"]] + full["fri_proof"]["inner_layers"])
        assert n_shared < n_full * (0.97 if p.cfg.n_queries > 1 else 1.01)
        text = json.dumps(obj, separators=(",", ":")).encode()
        back = ss.stwo_from_json(json.loads(text), expect=p.cfg)
        assert back.cfg == p.cfg and np.array_equal(verifier.stwo_record(back), rec)
        got, nrec = verifier.parse_stwo_text(p.cfg, text)
        assert got == OK and np.array_equal(nrec, rec)
        if p.cfg.n_queries > 1:
            assert len(text) < 0.93 * len(json.dumps(full, separators=(",", ":")))


def test_shared_form_of_corrupted_proofs():
    """Seeded corruptions: where the queries still agree about every shared node the shared text expands to the
    same (corrupted) proof; where they do not, there is no shared form and the writer says so."""
    base = _fixtures()[0]
    rng = np.random.default_rng(0x5EED2025 + 404)
    made = refused = 0
    for _ in range(60):
        p = formats.stwo_corrupt(base, rng)[0]
        try:
            obj = ss.stwo_to_json(p, shared=True, queries=formats.stwo_queries(base))
        except ss.MalformedProof:
            refused += 1
            continue
        made += 1
        text = json.dumps(obj).encode()
        got, nrec = verifier.parse_stwo_text(base.cfg, text)
        assert got == OK and np.array_equal(nrec, verifier.stwo_record(p))
    assert made > 20 and refused > 3


def test_wrong_hints_and_wrong_counts():
    p = _fixtures()[0]
    cfg = p.cfg
    obj = ss.stwo_to_json(p, shared=True)

    def outcome(o):
        text = json.dumps(o).encode()
        want = _python_outcome(text, cfg, "json")[0]
        assert verifier.parse_stwo_text(cfg, text)[0] == want
        return want
    import copy
    o = copy.deepcopy(obj); o["queries"][3] ^= 1                       # a wrong position: other sharing pattern
    assert outcome(o) in (OK, MALFORMED)                                # (reads as SOME proof or has the wrong count; never crashes)
    o = copy.deepcopy(obj); o["queries"].pop()
    assert outcome(o) == MALFORMED
    o = copy.deepcopy(obj); o["queries"][0] = 1 << cfg.lde_log
    assert outcome(o) == MALFORMED
    o = copy.deepcopy(obj); o["decommitments"][1]["hash_witness"].pop()
    assert outcome(o) == MALFORMED
    o = copy.deepcopy(obj); o["decommitments"][2]["hash_witness"].append(o["decommitments"][2]["hash_witness"][0])
    assert outcome(o) == MALFORMED
    o = copy.deepcopy(obj); o["fri_proof"]["inner_layers"][2]["decommitment"]["hash_witness"].pop(0)
    assert outcome(o) == MALFORMED
    o = copy.deepcopy(obj); o["queries"] = "0123456789abcdef"
    assert outcome(o) == MALFORMED
    assert outcome(obj) == OK
    other = ss.TESTING_CONFIG
    text = json.dumps(obj).encode()
    assert verifier.parse_stwo_text(other, text)[0] == _python_outcome(text, other, "json")[0] != OK


def test_attacker_sized_shared_texts_are_refused_quickly():
    """the positions and their number come from the untrusted text.  The format allows at most 64
    positions (the ABI's n_queries bound), so the plan is at most 64 x 64 x 31 steps whatever a text claims:
    a text with half a million positions, and one with 64 positions and a hundred thousand hashes, are
    malformed in milliseconds in both readers (the round-3 reader planned O(Q^2 len) before looking at the list)."""
    import time
    p = _fixtures()[0]
    obj = ss.stwo_to_json(p, shared=True)
    big = json.loads(json.dumps(obj))
    big["queries"] = [5] * 500000
    big["config"]["fri_config"]["n_queries"] = 500000
    long_list = json.loads(json.dumps(obj))
    long_list["decommitments"][1]["hash_witness"] = [[7] * 32] * 100000
    q64 = json.loads(json.dumps(obj))
    q64["queries"] = list(range(64))
    q64["config"]["fri_config"]["n_queries"] = 64
    q64["queried_values"][1] = [1] * (64 * p.cfg.n_cols)
    q64["queried_values"][2] = [1] * (64 * 16)
    for o in (big, long_list, q64):
        text = json.dumps(o, separators=(",", ":")).encode()
        t0 = time.perf_counter()
        got = verifier.parse_stwo_text(p.cfg, text)[0]
        dt = time.perf_counter() - t0
        assert got == MALFORMED and dt < 0.5, (got, dt, len(text))
        t0 = time.perf_counter()
        assert _python_outcome(text, p.cfg, "json")[0] == MALFORMED and time.perf_counter() - t0 < 2.0


def test_differential_fuzz_of_shared_texts():
    rnd = random.Random(20261006)
    p = _fixtures()[1]
    base = json.dumps(ss.stwo_to_json(p, shared=True)).encode()
    cfg = ss.TESTING_CONFIG
    seen = {OK: 0, MISMATCH: 0, MALFORMED: 0}
    for i in range(2500):
        text = _text_mutant(rnd, base)
        try:
            want, rec = _python_outcome(text, cfg, "json")
        except (UnicodeDecodeError, RecursionError):
            want, rec = MALFORMED, None
        got, grec = verifier.parse_stwo_text(cfg, text, fmt=1)
        assert got == want, (i, got, want, text[:200])
        if want == OK:
            assert np.array_equal(grec, rec)
        seen[want] += 1
    assert seen[OK] > 50 and seen[MALFORMED] > 500


def test_cli_converts_to_the_shared_form(capsys):
    from stark_symphony_amd import cli
    src = os.path.join(GOLDEN, "stwo_proof.json")
    assert cli.main(["convert", "--family", "stwo", "--to", "json-shared", src]) == 0
    out = capsys.readouterr().out
    p = ss.stwo_from_json(json.loads(out), expect=ss.PRODUCTION_CONFIG)
    want = ss.stwo_from_json(json.load(open(src)))
    assert np.array_equal(verifier.stwo_record(p), verifier.stwo_record(want))


# ---------------------------------------------------------------------- the GPU reader's rule for shared-path texts
def _write_shared_text(cfg, shared, python_separators=0):
    import ctypes as C
    from stark_symphony_amd import binding
    cs = verifier.stwo_cfg_struct(cfg, verifier.MODE_FIXTURE)
    sh = np.ascontiguousarray(shared, dtype=np.uint32)
    n = binding.lib().ss_stwo_write_shared_text(C.byref(cs), sh.ctypes.data, sh.size, python_separators, None, 0)
    if n == 0:
        return None
    buf = C.create_string_buffer(n)
    assert binding.lib().ss_stwo_write_shared_text(C.byref(cs), sh.ctypes.data, sh.size, python_separators, buf, n) == n
    return buf.raw


def _canonical_shared(cfg, text):
    from test_text_fastpath import canonical
    from stark_symphony_amd import binding
    return canonical(cfg, text, binding.TEXT_JSON_SHARED)


def _more_fixtures():
    return _fixtures() + [records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace20.npz"))[0]]


def test_library_writes_the_shared_text_formats_py_writes():
    """ss_stwo_write_shared_text(shared record) == json.dumps(formats.stwo_to_json(p, shared=True)), both separator
    styles: the template of the GPU reader is made by that writer."""
    for p in _more_fixtures():
        sh = verifier.stwo_shared_record(p)
        obj = ss.stwo_to_json(p, shared=True)
        assert _write_shared_text(p.cfg, sh, 0) == json.dumps(obj, separators=(",", ":")).encode()
        assert _write_shared_text(p.cfg, sh, 1) == json.dumps(obj).encode()
        bad = sh.copy()
        bad[-1 - 8 * 3] ^= 1  # a node: still a shared record
        assert _write_shared_text(p.cfg, bad, 0) is not None
        cnt = sh.size - 8 * int(verifier.stwo_shared_counts(p.cfg, formats.stwo_queries(p)).sum()) - 1
        bad = sh.copy()
        bad[cnt] += 1         # a count that its positions do not imply: not a shared record
        assert _write_shared_text(p.cfg, bad, 0) is None and _write_shared_text(p.cfg, sh[:-1], 0) is None


def test_rule_takes_honest_shared_texts():
    """The scalar statement of the GPU reader's rule for format 3 (hint from the tail, template with the gaps the
    hint implies, stored positions against the hint, expansion) takes what honest producers write -- both separator
    styles, trailing newline, indentation -- and yields the per-query record."""
    for p in _more_fixtures():
        rec = verifier.stwo_record(p)
        obj = ss.stwo_to_json(p, shared=True)
        for text in (json.dumps(obj, separators=(",", ":")), json.dumps(obj), json.dumps(obj) + "\n", json.dumps(obj, indent=1)):
            took, got = _canonical_shared(p.cfg, text.encode())
            assert took and np.array_equal(got, rec), p.cfg
        # the per-query text of the same proof is NOT of this format, and the other way round
        from test_text_fastpath import canonical, JSON
        assert not _canonical_shared(p.cfg, json.dumps(ss.stwo_to_json(p)).encode())[0]
        if p.cfg.n_queries > 1:
            assert not canonical(p.cfg, json.dumps(obj).encode(), JSON)[0]


def test_whatever_the_rule_takes_reads_as_the_host_reader_reads_it():
    """Soundness of the fast path for shared texts: 1 500 byte-level and structure-level mutants; every text the rule
    takes is parsed by the host reader (the arbiter) to exactly the same record."""
    rnd = random.Random(0x5EED2025 + 77)
    taken = changed = 0
    for p in _fixtures()[:3]:
        obj = ss.stwo_to_json(p, shared=True)
        base = json.dumps(obj, separators=(",", ":")).encode()
        base_rec = verifier.stwo_record(p)
        Q = p.cfg.n_queries
        for i in range(500):
            kind = i % 5
            if kind < 3:
                text = _text_mutant(rnd, base)
            elif kind == 3:    # another hint inside the domain / swapped hints (the lists then have the wrong lengths)
                o = json.loads(base)
                o["queries"][rnd.randrange(Q)] = rnd.randrange(1 << p.cfg.lde_log)
                text = json.dumps(o, separators=(",", ":")).encode()
            else:              # a number replaced by another canonical number somewhere
                import re
                ms = list(re.finditer(rb"\d+", base))
                m = ms[rnd.randrange(len(ms))]
                text = base[:m.start()] + str(rnd.choice([0, 1, 255, 256, 2 ** 31, 2 ** 32 - 1, 2 ** 32])).encode() + base[m.end():]
            took, rec = _canonical_shared(p.cfg, text)
            if took:
                taken += 1
                got, nrec = verifier.parse_stwo_text(p.cfg, text)
                assert got == OK and np.array_equal(nrec, rec), (p.cfg, i)
                changed += not np.array_equal(rec, base_rec)
    assert taken > 120 and changed > 50, (taken, changed)
