"""The canonical-text rule of the GPU reader (csrc/ss_text.h), on the CPU.

ss_stwo_verify_texts turns a text into a record on the GPU only if it is, byte for byte, what the
reference's producers write for the expected config, numbers and whitespace outside strings aside
(stwo-verifier/scripts/generate_wit.py:218-243; tests/data/proof.json); every other text goes to the host
reader.  Here the scalar statement of that rule (`ss_stwo_text_is_canonical`, the same code the device
kernel restates) is held against the host reader and formats.py:
  * the library's writers print what formats.py / the reference's adapter print, byte for byte;
  * every text the rule takes yields exactly the record the host reader and formats.py yield (soundness:
    a three-way differential over byte-level mutants and number-level mutants);
  * the texts honest producers emit ARE taken (otherwise the fast path would be decoration).
The device kernel itself is compared with this rule under `-m gpu` (tests/test_gpu_text.py).
"""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import binding, records, verifier

from conftest import GOLDEN
from test_ingest import FORMATS, MALFORMED, MISMATCH, OK, _python_outcome, _text_mutant
from test_formats_property import _rand_stwo

JSON, WIT = binding.TEXT_JSON, binding.TEXT_WIT


def write_text(cfg, rec, fmt, python_separators=0):
    cs = verifier.stwo_cfg_struct(cfg, verifier.MODE_FIXTURE)
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    n = binding.lib().ss_stwo_write_text(C.byref(cs), rec.ctypes.data, fmt, python_separators, None, 0)
    if n == 0:
        return None
    buf = C.create_string_buffer(n)
    assert binding.lib().ss_stwo_write_text(C.byref(cs), rec.ctypes.data, fmt, python_separators, buf, n) == n
    return buf.raw


def canonical(cfg, text, fmt):
    """-> (taken on the fast path?, record)"""
    cs = verifier.stwo_cfg_struct(cfg, verifier.MODE_FIXTURE)
    rec = np.zeros(binding.lib().ss_stwo_record_words(C.byref(cs)), dtype=np.uint32)
    got = binding.check(binding.lib().ss_stwo_text_is_canonical(C.byref(cs), text, len(text), fmt, rec.ctypes.data))
    return bool(got), rec


def _fixture_proofs():
    out = [ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))),
           ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))]
    for npz in ("stwo_trace16.npz", "stwo_trace16_blake2s.npz", "stwo_wide256.npz", "stwo_trace20.npz"):
        out.append(records.load_stwo_npz(os.path.join(GOLDEN, npz))[0])
    return out


def test_writers_print_what_the_reference_adapters_print():
    """Record -> proof.wit equals formats.stwo_to_wit (itself byte-identical to generate_wit.py's output,
    tests/golden/formats/), record -> proof.json equals json.dumps(stwo_to_json) in both separator styles,
    and the compact style reproduces the reference's own tests/data/proof.json byte for byte."""
    for p in _fixture_proofs():
        rec = verifier.stwo_record(p)
        assert write_text(p.cfg, rec, WIT) == ss.stwo_to_wit(p).encode()
        assert write_text(p.cfg, rec, JSON, 1) == json.dumps(ss.stwo_to_json(p)).encode()
        assert write_text(p.cfg, rec, JSON, 0) == json.dumps(ss.stwo_to_json(p), separators=(",", ":")).encode()
    for name, cfg in (("stwo_proof", ss.PRODUCTION_CONFIG), ("stwo_proof_test", ss.TESTING_CONFIG)):
        ref_json = open(os.path.join(GOLDEN, name + ".json"), "rb").read()
        ref_wit = open(os.path.join(FORMATS, name + ".wit"), "rb").read()
        rec = verifier.parse_stwo_text(cfg, ref_json)[1]
        assert write_text(cfg, rec, JSON, 0) == ref_json.rstrip()
        assert write_text(cfg, rec, WIT) == ref_wit.rstrip(b"\n")


def test_random_shapes_round_trip_through_the_writers():
    for seed in range(40):
        rnd = random.Random(seed)
        n_cols, lde_log, n_queries = rnd.randrange(1, 9), rnd.randrange(3, 12), rnd.randrange(1, 7)
        n_layers = rnd.randrange(0, min(4, lde_log - 1))
        p = _rand_stwo(seed, n_cols, lde_log, n_queries, n_layers, False)
        rec = verifier.stwo_record(p)
        for fmt, style in ((JSON, 0), (JSON, 1), (WIT, 0)):
            text = write_text(p.cfg, rec, fmt, style)
            taken, got = canonical(p.cfg, text, fmt)
            assert taken and np.array_equal(got, rec), (seed, fmt, style)
            assert np.array_equal(verifier.parse_stwo_text(p.cfg, text, fmt=fmt)[1], rec)
    ragged = _rand_stwo(5, 4, 8, 3, 2, True)  # a path of another length has no fixed-slot text
    assert write_text(ragged.cfg, verifier.stwo_record(ragged), WIT) is None or all(
        len(x) == ragged.cfg.lde_log for x in ragged.trace_paths + ragged.cp_paths)


def test_the_texts_honest_producers_emit_are_taken():
    """The reference's six files, the prover-made fixtures in every style, with and without the
    whitespace a pretty-printer adds outside strings."""
    for name, cfg in (("stwo_proof", ss.PRODUCTION_CONFIG), ("stwo_proof_test", ss.TESTING_CONFIG)):
        j = open(os.path.join(GOLDEN, name + ".json"), "rb").read()
        w = open(os.path.join(FORMATS, name + ".wit"), "rb").read()
        want = verifier.parse_stwo_text(cfg, j)[1]
        for text, fmt in ((j, JSON), (w, WIT), (json.dumps(json.loads(j), indent=2).encode(), JSON),
                          (b"\n  " + j + b"\r\n", JSON), (json.dumps(json.loads(w)).encode(), WIT)):
            taken, rec = canonical(cfg, text, fmt)
            assert taken and np.array_equal(rec, want)
        other = ss.TESTING_CONFIG if cfg is ss.PRODUCTION_CONFIG else ss.PRODUCTION_CONFIG
        assert not canonical(other, j, JSON)[0] and not canonical(other, w, WIT)[0]  # another shape: the host decides
        assert not canonical(cfg, j, WIT)[0] and not canonical(cfg, w, JSON)[0]
    for p in _fixture_proofs()[2:]:
        rec = verifier.stwo_record(p)
        for text, fmt in ((json.dumps(ss.stwo_to_json(p)).encode(), JSON), (ss.stwo_to_wit(p).encode(), WIT)):
            taken, got = canonical(p.cfg, text, fmt)
            assert taken and np.array_equal(got, rec)


def _number_mutant(rnd, text: bytes) -> bytes:
    """Replace one number of the text by another spelling / value."""
    import re
    spans = [m.span() for m in re.finditer(rb"(?<![0-9A-Za-z_])[0-9][0-9A-Za-z_]*", text)]
    if not spans:
        return text
    a, b = spans[rnd.randrange(len(spans))]
    old = text[a:b]
    k = rnd.randrange(12)
    if k == 0: new = b"0" + old
    elif k == 1: new = str(rnd.randrange(0, 300)).encode()
    elif k == 2: new = str(2 ** 32 - 1 + rnd.randrange(3)).encode()
    elif k == 3: new = str(2 ** 64 - 1 + rnd.randrange(2)).encode()
    elif k == 4: new = old + b"0"
    elif k == 5: new = old[:-1] or b"7"
    elif k == 6: new = old.replace(b"0x", b"0X") if old.startswith(b"0x") else b"0x" + old
    elif k == 7: new = old.upper()
    elif k == 8: new = old[:1] + b"_" + old[1:]
    elif k == 9: new = str(rnd.randrange(2 ** 32)).encode()
    elif k == 10: new = b"%de3" % rnd.randrange(10)
    else: new = hex(rnd.getrandbits(256)).encode() if old.startswith(b"0x") else str(rnd.randrange(256)).encode()
    return text[:a] + new + text[b:]


@pytest.mark.parametrize("kind", ["json", "wit"])
def test_whatever_the_rule_takes_is_what_the_readers_read(kind):
    """Soundness, three ways: byte-level mutants (structure damage) and number-level mutants (the only
    thing a canonical text may vary) of the reference's small proof.  Whenever the rule takes a text,
    the host reader and formats.py parse it too and all three records are equal; texts it does not take
    are none of its business -- the host reader's outcome stands (checked against formats.py in
    tests/test_ingest.py)."""
    rnd = random.Random(20261004 + len(kind))
    fmt = JSON if kind == "json" else WIT
    path = os.path.join(GOLDEN, "stwo_proof_test.json") if kind == "json" else os.path.join(FORMATS, "stwo_proof_test.wit")
    base = open(path, "rb").read()
    cfg = ss.TESTING_CONFIG
    taken_n = changed_n = 0
    base_rec = canonical(cfg, base, fmt)[1]
    for i in range(6000):
        text = _text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)
        if i % 7 == 0:
            text = _number_mutant(rnd, text)
        taken, rec = canonical(cfg, text, fmt)
        if not taken:
            continue
        taken_n += 1
        got, nrec = verifier.parse_stwo_text(cfg, text, fmt=fmt)
        assert got == OK and np.array_equal(nrec, rec), (i, text[:200])
        try:
            want, prec = _python_outcome(text, cfg, kind)
        except (UnicodeDecodeError, RecursionError):
            want, prec = MALFORMED, None
        assert want == OK and np.array_equal(prec, rec), (i, text[:200])
        changed_n += not np.array_equal(rec, base_rec)
    assert taken_n > 500 and changed_n > 300  # the rule is exercised, and on texts that differ from the original


def test_structure_deviations_are_left_to_the_host_reader():
    """Every one of these is a text the host reader ACCEPTS with the same record -- so the rule may not
    claim them by accident with another meaning, and in fact it takes none of them: they are rare,
    the host reader is the arbiter, and the rule stays a byte compare."""
    cfg = ss.PRODUCTION_CONFIG
    j = open(os.path.join(GOLDEN, "stwo_proof.json"), "rb").read()
    want = verifier.parse_stwo_text(cfg, j)[1]
    obj = json.loads(j)
    variants = [
        json.dumps(dict(reversed(list(obj.items())))).encode(),                  # member order
        j.replace(b'"config"', b'"\\u0063onfig"', 1),                            # escaped member name
        j.replace(b'"proof_of_work":', b'"extra":[1,2],"proof_of_work":', 1),    # unknown member
        j.replace(b'"pow_bits":5,', b'', 1),                                     # undeclared parameter (the verifier's applies)
        b'{"config":{"pow_bits":5},' + j[1:],                                    # duplicate member, last wins
    ]
    for text in variants:
        got, rec = verifier.parse_stwo_text(cfg, text, fmt=JSON)
        assert got == OK and np.array_equal(rec, want)
        assert not canonical(cfg, text, JSON)[0]
    # and these the host reader refuses or reads as another config; the rule has no opinion
    for text in (j[:-1], j + b"x", j.replace(b"227,", b"227.0,", 1), j.replace(b"227,", b"0227,", 1),
                 j.replace(b"227,", b"256,", 1), j.replace(b'"n_queries":16', b'"n_queries":17', 1),
                 j.replace(b"[227,", b"[ 2 27,", 1), j.replace(b'"commitments"', b'"commit ments"', 1)):
        assert not canonical(cfg, text, JSON)[0]
        assert verifier.parse_stwo_text(cfg, text, fmt=JSON)[0] in (MISMATCH, MALFORMED)


def test_configs_without_a_canonical_text_have_no_fast_path():
    """proof.json declares pow_bits: a pow_target that is no 2^(64-b) - 1 matches no JSON, so there is no JSON
    template (every text goes to the host reader, which reports the mismatch); the .wit declares nothing."""
    p = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))
    cs = verifier.stwo_cfg_struct(p.cfg, verifier.MODE_FIXTURE)
    cs.pow_target = 12345
    rec = verifier.stwo_record(p)
    L = binding.lib()
    assert L.ss_stwo_write_text(C.byref(cs), rec.ctypes.data, JSON, 0, None, 0) == 0
    assert L.ss_stwo_write_text(C.byref(cs), rec.ctypes.data, WIT, 0, None, 0) > 0
    j = open(os.path.join(GOLDEN, "stwo_proof_test.json"), "rb").read()
    assert L.ss_stwo_text_is_canonical(C.byref(cs), j, len(j), JSON, None) == 0


# ---------------------------------------------------------------------------------------- stark101
def s101_write_text(rec, fmt, python_separators=0):
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    n = binding.lib().ss_s101_write_text(rec.ctypes.data, fmt, python_separators, None, 0)
    if n == 0:
        return None
    buf = C.create_string_buffer(n)
    assert binding.lib().ss_s101_write_text(rec.ctypes.data, fmt, python_separators, buf, n) == n
    return buf.raw


def s101_canonical(text, fmt):
    W = binding.lib().ss_s101_record_words(C.byref(binding.S101Shape(10, 13)))
    rec = np.full(W, 0xEEEEEEEE, dtype=np.uint32)
    got = binding.check(binding.lib().ss_s101_text_is_canonical(text, len(text), fmt, rec.ctypes.data))
    return bool(got), rec


def _s101_python(text, kind):
    try:
        p = ss.stark101_from_json(json.loads(text)) if kind == "json" else ss.stark101_from_wit(text.decode())
    except (ss.MalformedProof, ValueError, OverflowError, AttributeError, UnicodeDecodeError, RecursionError, KeyError, TypeError):
        return None
    return p


def test_stark101_writers_and_the_reference_files():
    """stark101: the writers print what prover.py's proof.json / generate_wit.py print (the committed files are the
    reference's own output, tests/golden/make_stark101_golden.py and make_format_golden.py), both files are canonical,
    and the protocol's shape is the only one with a canonical text."""
    p = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    rec = verifier.s101_record(p, 10, 13)
    assert s101_write_text(rec, WIT) == ss.stark101_to_wit(p).encode()
    assert s101_write_text(rec, WIT) == open(os.path.join(FORMATS, "stark101_proof.wit"), "rb").read().rstrip(b"\n")
    assert s101_write_text(rec, JSON, 1) == json.dumps(ss.stark101_to_json(p)).encode()
    assert s101_write_text(rec, JSON, 0) == json.dumps(ss.stark101_to_json(p), separators=(",", ":")).encode()
    for fn, fmt in (("stark101_proof.json", JSON), (os.path.join("formats", "stark101_proof.wit"), WIT)):
        text = open(os.path.join(GOLDEN, fn), "rb").read()
        taken, got = s101_canonical(text, fmt)
        assert taken and np.array_equal(got, rec)
        assert not s101_canonical(text, WIT if fmt == JSON else JSON)[0]
    short = p.copy()
    short.layers = short.layers[:-1]                       # nine layers: a valid text, but not the protocol's shape
    assert verifier.parse_s101_text(json.dumps(ss.stark101_to_json(short)).encode())[0] == 0
    assert not s101_canonical(json.dumps(ss.stark101_to_json(short)).encode(), JSON)[0]
    assert s101_write_text(verifier.s101_record(short, 10, 13), JSON) is None


@pytest.mark.parametrize("kind", ["json", "wit"])
def test_stark101_whatever_the_rule_takes_is_what_the_readers_read(kind):
    """Three-way soundness for stark101 (78-digit decimal hashes): every mutant the rule takes parses in the native
    reader and in formats.py to the same record."""
    rnd = random.Random(20261005 + len(kind))
    fmt = JSON if kind == "json" else WIT
    base = open(os.path.join(GOLDEN, "stark101_proof.json" if kind == "json" else os.path.join("formats", "stark101_proof.wit")), "rb").read()
    base_rec = s101_canonical(base, fmt)[1]
    taken_n = changed_n = 0
    for i in range(2500):
        text = _text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)
        taken, rec = s101_canonical(text, fmt)
        if not taken:
            continue
        taken_n += 1
        rc, shape, nrec = verifier.parse_s101_text(text, fmt=fmt)
        assert rc == 0 and shape == (10, 13) and np.array_equal(nrec, rec), (i, text[:160])
        p = _s101_python(text, kind)
        assert p is not None and np.array_equal(verifier.s101_record(p, 10, 13), rec), (i, text[:160])
        changed_n += not np.array_equal(rec, base_rec)
    assert taken_n > 200 and changed_n > 100
