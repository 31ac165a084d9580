"""Native text readers (csrc/ss_ingest.cpp, C ABI ss_stwo_parse / ss_s101_parse) against the Python
grammar of stark-symphony_amd/formats.py: same records for everything the reference's adapters print
(stwo-verifier/scripts/generate_wit.py, stark101/scripts/generate_wit.py) and for random proofs, same
accept / malformed / other-config outcome for broken inputs.  No GPU needed: parsing is host code."""
import json
import os
import re

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import stark_symphony_amd as ss
from stark_symphony_amd import formats, records, verifier

from conftest import GOLDEN

FORMATS = os.path.join(GOLDEN, "formats")
OK, MISMATCH, MALFORMED = 0, verifier.STATUS_CONFIG_MISMATCH, 2


def _python_outcome(text: bytes, cfg, kind):
    """What the Python path says about `text` when `cfg` is expected: (outcome, record or None)."""
    try:
        if kind == "json":
            p = ss.stwo_from_json(json.loads(text), expect=cfg)
        else:
            p = ss.stwo_from_wit(text.decode(), cfg.trace_log, cfg.pow_bits, cfg.hash)
    except (ss.MalformedProof, ValueError, OverflowError, AttributeError):
        return MALFORMED, None
    if p.cfg != cfg:
        return MISMATCH, None
    return OK, verifier.stwo_record(p)


def _check(text: bytes, cfg, kind):
    want, rec = _python_outcome(text, cfg, kind)
    got, grec = verifier.parse_stwo_text(cfg, text)
    assert got == want, (got, want, text[:120])
    if want == OK:
        assert np.array_equal(grec, rec)
    else:
        assert not grec.any()  # nothing half-written
    return want


@pytest.mark.parametrize("name,cfg", [("stwo_proof", ss.PRODUCTION_CONFIG), ("stwo_proof_test", ss.TESTING_CONFIG)])
def test_reference_files_parse_to_the_same_record(name, cfg):
    j = open(os.path.join(GOLDEN, name + ".json"), "rb").read()
    w = open(os.path.join(FORMATS, name + ".wit"), "rb").read()
    assert _check(j, cfg, "json") == OK and _check(w, cfg, "wit") == OK
    assert np.array_equal(verifier.parse_stwo_text(cfg, j)[1], verifier.parse_stwo_text(cfg, w)[1])
    other = ss.TESTING_CONFIG if cfg is ss.PRODUCTION_CONFIG else ss.PRODUCTION_CONFIG
    assert _check(j, other, "json") == MISMATCH and _check(w, other, "wit") == MISMATCH


@pytest.mark.parametrize("npz", ["stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace20.npz", "stwo_trace16_blake2s.npz"])
def test_prover_made_proofs_through_both_text_formats(npz):
    p = records.load_stwo_npz(os.path.join(GOLDEN, npz))[0]
    j = json.dumps(ss.stwo_to_json(p)).encode()
    w = ss.stwo_to_wit(p).encode()
    assert _check(j, p.cfg, "json") == OK and _check(w, p.cfg, "wit") == OK
    assert np.array_equal(verifier.parse_stwo_text(p.cfg, w)[1], verifier.stwo_record(p))


def test_stark101_files():
    for fn, reader in (("stark101_proof.json", lambda t: ss.stark101_from_json(json.loads(t))),
                       (os.path.join("formats", "stark101_proof.wit"), lambda t: ss.stark101_from_wit(t.decode()))):
        text = open(os.path.join(GOLDEN, fn), "rb").read()
        rc, shape, rec = verifier.parse_s101_text(text)
        p = reader(text)
        assert rc == 0 and shape == verifier.s101_shape_of([p])
        assert np.array_equal(rec, verifier.s101_record(p, *shape))
    for bad in (b"{}", b"[1, 2]", b"{\"p_mt_root\": 1, \"evals\": [], \"fri_layers\": [], \"fri_last_layer\": 0}",
                text[:-40], text.replace(b"list!", b"lisp!", 1)):
        assert verifier.parse_s101_text(bad)[0] == MALFORMED


from test_formats_property import _rand_stwo  # noqa: E402


shapes = st.tuples(st.integers(0, 2 ** 32 - 1), st.integers(1, 9), st.integers(3, 12), st.integers(1, 7),
                   st.integers(0, 3)).filter(lambda t: t[4] + 1 < t[2])


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(shapes, st.booleans())
def test_random_proofs_and_ragged_paths(t, ragged):
    seed, n_cols, lde_log, n_queries, n_layers = t
    p = _rand_stwo(seed, n_cols, lde_log, n_queries, n_layers, ragged)
    w = ss.stwo_to_wit(p).encode()
    back = ss.stwo_from_wit(w.decode(), p.cfg.trace_log, p.cfg.pow_bits)
    import ctypes as C
    from stark_symphony_amd import binding
    cs = verifier.stwo_cfg_struct(back.cfg, verifier.MODE_FIXTURE)
    if binding.lib().ss_stwo_record_words(C.byref(cs)):  # ragged first path = another LDE_LOG_SIZE
        _check(w, back.cfg, "wit")
    else:  # ... possibly one no verifier can be configured for: a loud error, not a verdict
        with pytest.raises(binding.SsError):
            verifier.parse_stwo_text(back.cfg, w)
    _check(w, p.cfg, "wit")
    if not ragged:
        _check(json.dumps(ss.stwo_to_json(p)).encode(), p.cfg, "json")


def _json_mutants(obj):
    """(description, object) pairs, each a small edit of a valid proof.json object"""
    import copy
    def edit(f):
        o = copy.deepcopy(obj)
        f(o)
        return o
    out = []
    out.append(("byte 256", edit(lambda o: o["commitments"][1].__setitem__(5, 256))))
    out.append(("31-byte hash", edit(lambda o: o["commitments"][0].pop())))
    out.append(("u32 overflow", edit(lambda o: o["queried_values"][1].__setitem__(0, 2 ** 32))))
    out.append(("negative", edit(lambda o: o["queried_values"][2].__setitem__(3, -1))))
    out.append(("float", edit(lambda o: o["queried_values"][2].__setitem__(3, 1.5))))
    out.append(("string value", edit(lambda o: o["queried_values"][2].__setitem__(3, "7"))))
    out.append(("missing fri_proof", edit(lambda o: o.pop("fri_proof"))))
    out.append(("missing commitments", edit(lambda o: o.pop("commitments"))))
    out.append(("two coeffs", edit(lambda o: o["fri_proof"]["last_layer_poly"]["coeffs"].append(
        o["fri_proof"]["last_layer_poly"]["coeffs"][0]))))
    out.append(("qm31 with 3 words", edit(lambda o: o["sampled_values"][1][0][0][1].append(1))))
    out.append(("one witness less", edit(lambda o: o["fri_proof"]["first_layer"]["fri_witness"].pop())))
    out.append(("hash_witness not divisible", edit(lambda o: o["decommitments"][1]["hash_witness"].pop())))
    out.append(("one queried value less", edit(lambda o: o["queried_values"][1].pop())))
    out.append(("15 cp columns", edit(lambda o: o["sampled_values"][2].pop())))
    out.append(("extra column = other config", edit(lambda o: (o["sampled_values"][1].append(o["sampled_values"][1][0]),
                                                               o["queried_values"][1].extend([1] * o["config"]["fri_config"]["n_queries"])))))
    out.append(("declares pow_bits 0", edit(lambda o: o["config"].__setitem__("pow_bits", 0))))
    out.append(("declares pow_bits 65", edit(lambda o: o["config"].__setitem__("pow_bits", 65))))
    out.append(("no pow_bits", edit(lambda o: o["config"].pop("pow_bits"))))
    out.append(("no config", edit(lambda o: o.pop("config"))))
    out.append(("declares 1 query", edit(lambda o: o["config"]["fri_config"].__setitem__("n_queries", 1))))
    out.append(("declares blake2s", edit(lambda o: o["config"].__setitem__("hash", "blake2s"))))
    out.append(("declares md5", edit(lambda o: o["config"].__setitem__("hash", "md5"))))
    for falsy in (None, 0, "", False, []):  # a present non-name is not "absent" (ADVICE r2): malformed in both readers
        out.append(("declares hash %r" % (falsy,), edit(lambda o, v=falsy: o["config"].__setitem__("hash", v))))
    out.append(("declares sha256", edit(lambda o: o["config"].__setitem__("hash", "sha256"))))
    out.append(("declares blow-up 3", edit(lambda o: o["config"]["fri_config"].__setitem__("log_blowup_factor", 3))))
    out.append(("one inner layer less", edit(lambda o: o["fri_proof"]["inner_layers"].pop())))
    out.append(("nonce 2^64", edit(lambda o: o.__setitem__("proof_of_work", 2 ** 64))))
    out.append(("nonce 2^64-1", edit(lambda o: o.__setitem__("proof_of_work", 2 ** 64 - 1))))
    out.append(("path node as big integer", edit(lambda o: o["decommitments"][1]["hash_witness"].__setitem__(
        0, int.from_bytes(bytes(o["decommitments"][1]["hash_witness"][0]), "big")))))
    out.append(("path node 2^256", edit(lambda o: o["decommitments"][2]["hash_witness"].__setitem__(1, 2 ** 256))))
    out.append(("commitment as integer", edit(lambda o: o["commitments"].__setitem__(0, 5))))
    out.append(("null member", edit(lambda o: o["fri_proof"].__setitem__("first_layer", None))))
    return out


def test_malformed_and_downgraded_json_get_the_python_outcome():
    obj = json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))
    seen = set()
    for what, o in _json_mutants(obj):
        seen.add((_check(json.dumps(o).encode(), ss.PRODUCTION_CONFIG, "json")))
    assert seen == {OK, MISMATCH, MALFORMED}
    text = json.dumps(obj).encode()
    for cut in (1, 10, len(text) // 2, len(text) - 1):
        assert _check(text[:cut], ss.PRODUCTION_CONFIG, "json") == MALFORMED
    assert _check(text + b" x", ss.PRODUCTION_CONFIG, "json") == MALFORMED
    assert _check(b"  " + text + b"\n\t ", ss.PRODUCTION_CONFIG, "json") == OK
    assert _check(b"[]", ss.PRODUCTION_CONFIG, "json") == MALFORMED


def test_malformed_wit_gets_the_python_outcome():
    cfg = ss.TESTING_CONFIG
    wit = json.load(open(os.path.join(FORMATS, "stwo_proof_test.wit")))

    def with_value(name, f):
        w = json.loads(json.dumps(wit))
        w[name]["value"] = f(w[name]["value"])
        return json.dumps(w).encode()
    cases = [
        with_value("POW_NONCE", lambda v: "0x" + "%X" % int(v)),                 # hex, upper case
        with_value("POW_NONCE", lambda v: "1_8_5"),                              # `_` separators
        with_value("POW_NONCE", lambda v: v + " 1"),                             # trailing characters
        with_value("POW_NONCE", lambda v: "(" + v + ")"),                        # "(x)" is x
        with_value("POW_NONCE", lambda v: "((" + v + "))"),
        with_value("POW_NONCE", lambda v: "(" + v + ",)"),                       # a 1-tuple is not a u64
        with_value("POW_NONCE", lambda v: str(2 ** 64)),
        with_value("POW_NONCE", lambda v: ""),
        with_value("COMMITMENTS", lambda v: v.replace("(", "[", 1)[:-1] + "]"),  # array instead of tuple
        with_value("COMMITMENTS", lambda v: v[:-1] + ", 1)"),                    # four commitments
        with_value("COMMITMENTS", lambda v: v.replace("0x", "0x1" + "0" * 64, 1)),  # 2^256 and more
        with_value("COMMITMENTS", lambda v: v.replace("0x", "0x" + "0" * 20, 1)),   # leading zeros
        with_value("DECOMMITMENTS", lambda v: v.replace("list![", "list! [", 1)),
        with_value("DECOMMITMENTS", lambda v: v.replace("list![", "list [", 1)),
        with_value("DECOMMITMENTS", lambda v: v.replace(", ", " ", 3)),          # commas are optional to formats.py
        with_value("DECOMMITMENTS", lambda v: v.replace("[1]", "[1, 2]", 1)),    # only the first offset is read
        with_value("DECOMMITMENTS", lambda v: v.replace("[1]", "1", 1)),
        with_value("DECOMMITMENTS", lambda v: v[:-1]),                           # unterminated
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "[[((1, 0), (0, 0))]]", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "((1, 0, 0), (0,))", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("(1, 0)", "(4294967296, 0)", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "qm31(1, 0, 0, 0)", 1)),    # .simf constructor
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "qm31 ( 1 0 0 0 )", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "qm31(1, 0, 0)", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "qm31(1, 0, 0, 0, 0)", 1)),
        with_value("OODS_EVALS", lambda v: v.replace("((1, 0), (0, 0))", "qm31[1, 0, 0, 0]", 1)),
        with_value("FRI_COMMITMENTS", lambda v: re.sub(r"\(\((\d+), (\d+)\), \((\d+), (\d+)\)\)\)$", r"qm31(\1, \2, \3, \4))", v)),
        with_value("FRI_COMMITMENTS", lambda v: v.replace("[", "list![", 1)),
        with_value("FRI_DECOMMITMENTS", lambda v: v.replace("list![", "list![" + "1, " * 31, 1)),  # 32+ siblings
        with_value("FRI_DECOMMITMENTS", lambda v: v.replace("list![", "list![7, ", 1)),           # one sibling more
        json.dumps({k: v for k, v in wit.items() if k != "OODS_EVALS"}).encode(),
        json.dumps(dict(wit, COMMITMENTS={"type": "x"})).encode(),
        json.dumps(dict(wit, COMMITMENTS={"value": 5, "type": "x"})).encode(),
    ]
    seen = {_check(c, cfg, "wit") for c in cases}
    assert seen == {OK, MALFORMED}
    assert _check(json.dumps(wit).encode(), ss.PRODUCTION_CONFIG, "wit") == MISMATCH


def test_packed_hash_lists_do_not_change_the_meaning_of_other_lists():
    """The JSON reader keeps a list of exactly 32 byte values as one packed node (that is what a hash
    looks like).  Lists that merely look like one -- 32 small queried values, a hash_witness made of
    32 small integers -- must still read as what they are."""
    p = _rand_stwo(7, 4, 6, 8, 1, ragged=False)          # Q * N = 32 trace values
    p.trace_vals[:] = np.arange(32, dtype=np.uint32).reshape(8, 4) * 7
    _check(json.dumps(ss.stwo_to_json(p)).encode(), p.cfg, "json")
    p = _rand_stwo(8, 3, 5, 2, 1, ragged=False)          # Q * 16 = 32 composition values
    p.cp_vals[:] = (np.arange(32, dtype=np.uint32).reshape(2, 16) * 5) % 251
    assert _check(json.dumps(ss.stwo_to_json(p)).encode(), p.cfg, "json") == OK
    obj = ss.stwo_to_json(p)
    obj["decommitments"][1]["hash_witness"] = list(range(32))      # 16 nodes per query, written as integers
    _check(json.dumps(obj).encode(), p.cfg, "json")
    obj["decommitments"][1]["hash_witness"] = list(range(10))      # 5 per query = the tree depth
    assert _check(json.dumps(obj).encode(), p.cfg, "json") == OK
    obj = ss.stwo_to_json(p)
    obj["fri_proof"]["first_layer"]["fri_witness"] = list(range(32))
    assert _check(json.dumps(obj).encode(), p.cfg, "json") == MALFORMED
    obj = ss.stwo_to_json(p)
    obj["commitments"][2] = [[1] * 32]                              # a hash nested one level too deep
    assert _check(json.dumps(obj).encode(), p.cfg, "json") == MALFORMED


def test_oversized_and_deeply_nested_texts_are_malformed_not_fatal():
    cfg = ss.TESTING_CONFIG
    assert verifier.parse_stwo_text(cfg, b"[" * 100000)[0] == MALFORMED            # nesting bound
    assert verifier.parse_stwo_text(cfg, b"{\"a\": " + b"1," * 10)[0] == MALFORMED
    big = b"{\"x\": [" + b"1," * (17 << 20) + b"1]}"                                  # > 32 MiB of text
    assert len(big) > (32 << 20) and verifier.parse_stwo_text(cfg, big)[0] == MALFORMED
    assert verifier.parse_s101_text(big)[0] == MALFORMED


def test_attacker_sized_lists_parse_in_linear_time():
    """N and Q come from the untrusted text; the schema walkers iterate with cursors, so a
    ~1.5 MB text with one huge list costs milliseconds, not minutes (child(i, k) inside a loop was
    quadratic: 49 s at N = 80 000)."""
    import time
    obj = json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json")))
    big_n = 120000
    obj["sampled_values"][1] = [[[[1, 2], [3, 4]]]] * big_n             # N = 120 000 columns
    obj["queried_values"][1] = [7] * big_n
    text = json.dumps(obj).encode()
    assert len(text) > 1_000_000
    t0 = time.perf_counter()
    got = verifier.parse_stwo_text(ss.TESTING_CONFIG, text)[0]
    assert got == MISMATCH and time.perf_counter() - t0 < 1.0
    # .wit: Q = 60 000 decommitment entries, each of the wrong shape deep inside; N = 60 000 OODS columns
    wit = json.loads(open(os.path.join(FORMATS, "stwo_proof_test.wit")).read())
    one = wit["DECOMMITMENTS"]["value"][1:-1]
    wit["DECOMMITMENTS"]["value"] = "[" + ", ".join([one] * 3000) + "]"
    col = "[((1, 0), (0, 0))]"
    oods = wit["OODS_EVALS"]["value"]
    wit["OODS_EVALS"]["value"] = "([" + ", ".join([col] * 60000) + "], " + oods[oods.index("], [((") + 3:]
    text = json.dumps(wit).encode()
    assert len(text) > 1_000_000
    t0 = time.perf_counter()
    got = verifier.parse_stwo_text(ss.TESTING_CONFIG, text, fmt=2)[0]
    assert got in (MISMATCH, MALFORMED) and time.perf_counter() - t0 < 1.0
    assert got == _python_outcome(text, ss.TESTING_CONFIG, "wit")[0]
    # FRI layer lists as long as the text allows
    wit = json.loads(open(os.path.join(FORMATS, "stwo_proof_test.wit")).read())
    fd = wit["FRI_DECOMMITMENTS"]["value"]
    first = fd[2:fd.index(")], [[") + 1]
    wit["FRI_DECOMMITMENTS"]["value"] = "([" + ", ".join([first] * 8000) + "]" + fd[fd.index(")], [[") + 2:]
    text = json.dumps(wit).encode()
    t0 = time.perf_counter()
    got = verifier.parse_stwo_text(ss.TESTING_CONFIG, text, fmt=2)[0]
    assert got == _python_outcome(text, ss.TESTING_CONFIG, "wit")[0] and time.perf_counter() - t0 < 1.0


def test_native_readers_under_address_and_ub_sanitizers(tmp_path):
    """tests/native/ingest_fuzz.cpp: csrc/ss_ingest.cpp compiled with -fsanitize=address,undefined
    (gcc, CPU only) parses the reference's files and thousands of seeded mutants of them; the run
    aborts on any memory error, integer overflow UB or write past the record."""
    import subprocess
    from conftest import ROOT
    csrc = os.path.join(ROOT, "stark-symphony_amd", "csrc")
    exe = str(tmp_path / "ingest_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + csrc, os.path.join(ROOT, "tests", "native", "ingest_fuzz.cpp"),
                    os.path.join(csrc, "ss_ingest.cpp"), "-o", exe, "-lpthread"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    shared = []  # the shared-path variant (expansion plan from untrusted positions) is part of the corpus
    for name in ("stwo_proof.json", "stwo_proof_test.json"):
        p = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, name))))
        shared.append(str(tmp_path / ("shared_" + name)))
        with open(shared[-1], "w") as f:
            json.dump(ss.stwo_to_json(p, shared=True), f, separators=(",", ":"))
    runs = [("production", [os.path.join(GOLDEN, "stwo_proof.json"), os.path.join(FORMATS, "stwo_proof.wit"), shared[0]], 400),
            ("testing", [os.path.join(GOLDEN, "stwo_proof_test.json"), os.path.join(FORMATS, "stwo_proof_test.wit"), shared[1]], 1500),
            ("s101", [os.path.join(GOLDEN, "stark101_proof.json"), os.path.join(FORMATS, "stark101_proof.wit")], 600)]
    for profile, files, n in runs:
        r = subprocess.run([exe, "20261003", str(n), profile] + files, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, (profile, r.stdout[-500:], r.stderr[-3000:])
        assert "parses:" in r.stdout


def test_duplicate_and_escaped_member_names():
    """Python's json.loads keeps the LAST of duplicate members; so does the native reader (a proof that
    declares its config twice is judged by the same declaration in both).  A member name written with
    an escape is decoded first, in member names and in the literals of a .wit alike."""
    text = open(os.path.join(GOLDEN, "stwo_proof.json"), "rb").read()
    dup_bad_last = text[:-1] + b', "config": {"pow_bits": 0, "fri_config": {"log_blowup_factor": 4, "n_queries": 16}}}'
    dup_good_last = b'{"config": {"pow_bits": 0}, ' + text[1:]
    assert _check(dup_bad_last, ss.PRODUCTION_CONFIG, "json") == MISMATCH
    assert _check(dup_good_last, ss.PRODUCTION_CONFIG, "json") == OK
    escaped = text.replace(b'"config"', b'"\\u0063onfig"', 1).replace(b'"pow_bits":5', b'"pow\\u005Fbits":0', 1)
    assert json.loads(escaped)["config"]["pow_bits"] == 0          # an escaped spelling is still that member
    assert _check(escaped, ss.PRODUCTION_CONFIG, "json") == MISMATCH
    wit = json.load(open(os.path.join(FORMATS, "stwo_proof_test.wit")))
    w = json.dumps(wit).replace('"value": "185"', '"value": "1\\u00385"').encode()   # POW_NONCE 185 with an escaped digit
    assert b"\\u0038" in w and _check(w, ss.TESTING_CONFIG, "wit") == OK
    assert _check(b'{"' + b"k" * 300 + b'": 1, ' + text[1:], ss.PRODUCTION_CONFIG, "json") == OK   # a very long member name


def _text_mutant(rnd, s: bytes) -> bytes:
    m = bytearray(s)
    for _ in range(1 + rnd.randrange(3)):
        if not m:
            break
        at, k = rnd.randrange(len(m)), rnd.randrange(9)
        if k == 0: m[at] ^= 1 << rnd.randrange(8)
        elif k == 1: del m[at:at + 1 + rnd.randrange(40)]
        elif k == 2: m[at:at] = m[at:at + 1 + rnd.randrange(60)]
        elif k == 3: del m[at:]
        elif k == 4: m[at] = ord("0123456789"[rnd.randrange(10)])
        elif k == 5: m[at] = ord("[](){},:\" x_-.eE\\"[rnd.randrange(17)])
        elif k == 6: m[at:at] = bytes(ord("0") + rnd.randrange(10) for _ in range(1 + rnd.randrange(90)))
        elif k == 7: m[at:at] = [b"list![", b"0x", b"qm31(", b"null", b"1e3", b"-1", b"\\u0041", b"\xc3\xa9", b"NaN"][rnd.randrange(9)]
        else: m[at] = rnd.randrange(256)
    return bytes(m)


@pytest.mark.parametrize("kind", ["json", "wit"])
def test_differential_text_fuzz_native_against_python(kind):
    """Seeded byte-level mutants of the reference's small proof in both text formats: the native
    reader and formats.py agree on every one -- parsed (same record), other config, or malformed --
    down to json.loads' strictness (leading zeros, control characters, escapes, UTF-8)."""
    import random
    rnd = random.Random(20261003 + len(kind))
    path = os.path.join(GOLDEN, "stwo_proof_test.json") if kind == "json" else os.path.join(FORMATS, "stwo_proof_test.wit")
    base = open(path, "rb").read()
    cfg = ss.TESTING_CONFIG
    fmt = 1 if kind == "json" else 2
    seen = {OK: 0, MISMATCH: 0, MALFORMED: 0}
    for i in range(3000):
        text = _text_mutant(rnd, base)
        try:
            want, rec = _python_outcome(text, cfg, kind)
        except (UnicodeDecodeError, RecursionError):
            want, rec = MALFORMED, None
        got, grec = verifier.parse_stwo_text(cfg, text, fmt=fmt)
        assert got == want, (i, got, want)
        if want == OK:
            assert np.array_equal(grec, rec), i
        seen[want] += 1
    assert seen[OK] > 100 and seen[MALFORMED] > 1000


@pytest.mark.parametrize("kind", ["json", "wit"])
def test_differential_text_fuzz_stark101(kind):
    """The same for the stark101 readers (shape, record or malformed)."""
    import random
    rnd = random.Random(777 + len(kind))
    path = os.path.join(GOLDEN, "stark101_proof.json") if kind == "json" else os.path.join(FORMATS, "stark101_proof.wit")
    base = open(path, "rb").read()
    parsed = 0
    for i in range(1500):
        text = _text_mutant(rnd, base)
        try:
            p = ss.stark101_from_json(json.loads(text)) if kind == "json" else ss.stark101_from_wit(text.decode())
            want = (0, verifier.s101_shape_of([p]))
        except (ss.MalformedProof, ValueError, UnicodeDecodeError, RecursionError):
            p, want = None, (MALFORMED, None)
        rc, shape, rec = verifier.parse_s101_text(text, fmt=1 if kind == "json" else 2)
        assert (rc, shape) == want, i
        if p is not None:
            assert np.array_equal(rec, verifier.s101_record(p, *shape)), i
            parsed += 1
    assert parsed > 50
