"""The GPU text reader (csrc/ss_textdev.hip) against the scalar rule, the host reader and formats.py.

Third leg of the differential of tests/test_ingest.py / tests/test_text_fastpath.py: for every text, the
kernel's outcome equals `ss_stwo_text_is_canonical`, and where it is 0 the record it wrote equals the host
reader's (and therefore formats.py's).  Then the whole entry point: ss_stwo_verify_texts / _files on
canonical texts, non-canonical texts, other shapes and garbage in one batch gives the status words of the
record path + oracle, whatever the chunking."""
import json
import os
import random

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import binding, formats, records, verifier
from oracle import oracle as O

from conftest import GOLDEN
from test_ingest import FORMATS, _text_mutant
from test_text_fastpath import _number_mutant, canonical, write_text

pytestmark = pytest.mark.gpu
JSON, WIT = binding.TEXT_JSON, binding.TEXT_WIT
SEED = 0x5EED2025


@pytest.fixture(scope="module")
def ver():
    return verifier.Verifier(0)


def _agree(ver, cfg, texts, fmt):
    """GPU outcome == scalar rule for every text; GPU record == scalar rule's record where taken."""
    recs, outcome = ver.read_stwo_texts(cfg, texts, fmt)
    taken = 0
    for i, t in enumerate(texts):
        want, rec = canonical(cfg, t, fmt)
        assert (outcome[i] == 0) == want, (i, int(outcome[i]), want, t[:160])
        if want:
            assert np.array_equal(recs[i], rec), (i, np.nonzero(recs[i] != rec)[0][:8])
            taken += 1
    return taken


@pytest.mark.parametrize("name,cfg", [("stwo_proof", ss.PRODUCTION_CONFIG), ("stwo_proof_test", ss.TESTING_CONFIG)])
def test_reference_files_are_read_by_the_gpu(ver, name, cfg):
    j = open(os.path.join(GOLDEN, name + ".json"), "rb").read()
    w = open(os.path.join(FORMATS, name + ".wit"), "rb").read()
    want = verifier.parse_stwo_text(cfg, j)[1]
    for text, fmt in ((j, JSON), (w, WIT), (json.dumps(json.loads(j)).encode(), JSON),
                      (json.dumps(json.loads(j), indent=1).encode(), JSON)):
        recs, outcome = ver.read_stwo_texts(cfg, [text, text.rstrip()[:-1], text], fmt)
        assert outcome.tolist() == [0, 1, 0]
        assert np.array_equal(recs[0], want) and np.array_equal(recs[2], want)
    other = ss.TESTING_CONFIG if cfg is ss.PRODUCTION_CONFIG else ss.PRODUCTION_CONFIG
    assert ver.read_stwo_texts(other, [j], JSON)[1].tolist() == [1]
    assert ver.read_stwo_texts(cfg, [j], WIT)[1].tolist() == [1] and ver.read_stwo_texts(cfg, [w], JSON)[1].tolist() == [1]


@pytest.mark.parametrize("npz", ["stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace20.npz", "stwo_trace16_blake2s.npz"])
def test_prover_made_proofs_are_read_by_the_gpu(ver, npz):
    """Texts of every size class (a 2^20-row proof is 0.8 MB of JSON: ~780 windows per wave), all styles."""
    p = records.load_stwo_npz(os.path.join(GOLDEN, npz))[0]
    rec = verifier.stwo_record(p)
    for text, fmt in ((write_text(p.cfg, rec, JSON, 0), JSON), (write_text(p.cfg, rec, JSON, 1), JSON),
                      (write_text(p.cfg, rec, WIT), WIT)):
        recs, outcome = ver.read_stwo_texts(p.cfg, [text] * 3, fmt)
        assert outcome.tolist() == [0, 0, 0] and all(np.array_equal(r, rec) for r in recs)


@pytest.mark.parametrize("kind", ["json", "wit"])
def test_gpu_reader_equals_the_scalar_rule_on_mutants(ver, kind):
    """Byte-level and number-level mutants of the reference's small proof and of its production proof
    (texts of every length, cut at every position relative to the 1 KiB windows and 16-byte lanes)."""
    fmt = JSON if kind == "json" else WIT
    rnd = random.Random(SEED + len(kind))
    for name, cfg, n in (("stwo_proof_test", ss.TESTING_CONFIG, 4000), ("stwo_proof", ss.PRODUCTION_CONFIG, 600)):
        path = os.path.join(GOLDEN, name + ".json") if kind == "json" else os.path.join(FORMATS, name + ".wit")
        base = open(path, "rb").read()
        texts = [base]
        for i in range(n):
            t = _text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)
            if i % 5 == 0:
                t = _number_mutant(rnd, t)
            if i % 11 == 0:  # shift everything against the window grid
                t = b" " * rnd.randrange(1, 40) + t
            texts.append(t)
        texts += [b"", b"{", b"0", b"\x00" * 100, b"1" * 5000, b"a" * 3000, b'"' * 2049, b"[" * 1024 + b"1"]
        taken = _agree(ver, cfg, texts, fmt)
        assert taken > n // 10


def test_window_and_lane_boundaries(ver):
    """Numbers, strings and runs that straddle lane (16 B) and window (1 KiB) boundaries: the same text
    shifted by 0..1040 leading blanks (JSON: whitespace outside strings is free) must always be taken
    with the same record; a blank INSIDE a number or a key must never be."""
    cfg = ss.TESTING_CONFIG
    j = open(os.path.join(GOLDEN, "stwo_proof_test.json"), "rb").read()
    want = verifier.parse_stwo_text(cfg, j)[1]
    texts = [b" " * k + j for k in list(range(0, 70)) + list(range(1000, 1045))]
    recs, outcome = ver.read_stwo_texts(cfg, texts, JSON)
    assert not outcome.any() and all(np.array_equal(r, want) for r in recs)
    inner = [j[:k] + b" " + j[k:] for k in range(0, len(j), 7)]
    _agree(ver, cfg, inner, JSON)


def test_verify_texts_mixed_batch_and_chunking(ver, tmp_path):
    """The entry point: canonical texts (GPU reader), non-canonical but valid texts (host reader), other
    shapes, garbage, corrupted proofs -- one batch, verdicts equal the oracle's / the stage-0 codes; the
    count of host-read texts is what the rule predicts; ~400 MB of text so several chunks are in flight."""
    base = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    cfg = base.cfg
    rng = np.random.default_rng(SEED + 90)
    bad = [formats.stwo_corrupt(base, rng)[0] for _ in range(5)]
    proofs = [base] + bad
    want_d = O.stwo_verify_batch(proofs).tolist()
    small = open(os.path.join(GOLDEN, "stwo_proof_test.json"), "rb").read()
    variants, want = [], []
    for p, w in zip(proofs, want_d):
        obj = ss.stwo_to_json(p)
        variants += [json.dumps(obj).encode(), json.dumps(obj, separators=(",", ":")).encode(), ss.stwo_to_wit(p).encode(),
                     json.dumps(dict(reversed(list(obj.items())))).encode()]   # the last one: host reader
        want += [w, w, w, w]
    variants += [small, b"{}", b"nonsense", b""]
    want += [1, 2, 2, 2]
    host_expected = sum(1 for v in variants if not (canonical(cfg, v, JSON)[0] or canonical(cfg, v, WIT)[0]))
    assert host_expected == len(proofs) + 4
    for n in (len(variants), 2500):
        batch = [variants[i % len(variants)] for i in range(n)]
        status, stats = ver.verify_stwo_texts(cfg, batch)
        assert status.tolist() == [want[i % len(variants)] for i in range(n)]
        assert stats["host_parsed"] == sum(1 for i in range(n) if i % len(variants) % 4 == 3 or i % len(variants) >= 4 * len(proofs))
        assert stats["text_bytes"] == sum(len(b) for b in batch)
    # files, with formats forced and sniffed
    paths = []
    for i, v in enumerate(variants[:12]):
        f = tmp_path / ("p%d.txt" % i)
        f.write_bytes(v)
        paths.append(str(f))
    paths.append(str(tmp_path / "absent.json"))
    status, stats = ver.verify_stwo_files(cfg, paths)
    assert status.tolist() == want[:12] + [2]
    # the same files read into ONE page-locked buffer (what a rank of an 8-GPU host uses: distributed.files_verifier)
    empty = tmp_path / "empty.json"
    empty.write_bytes(b"")
    pstatus, pstats = ver.verify_stwo_files_pinned(cfg, paths + [str(empty)])
    assert pstatus.tolist() == want[:12] + [2, 2] and pstats["host_parsed"] == stats["host_parsed"] + 1
    cstatus, cstats = ver.verify_stwo_files_pinned(cfg, paths + [str(empty)], chunk_bytes=200_000)   # several chunks, one buffer
    assert cstatus.tolist() == pstatus.tolist() and cstats["text_bytes"] == pstats["text_bytes"]
    from stark_symphony_amd import distributed
    for world in (1, 8):
        assert distributed.files_verifier(ver, cfg, world=world)(paths).tolist() == want[:12] + [2]
    status, _ = ver.verify_stwo_texts(cfg, [variants[0], variants[2]], fmt=JSON)
    assert status.tolist() == [want[0], 2]  # a .wit read as proof.json is no proof.json


def test_full_size_texts_end_to_end(ver):
    """2^20-row proofs as text, valid and corrupted, both formats, 600 texts (0.4 GB): the statuses of the
    record path."""
    p = records.load_stwo_npz(os.path.join(GOLDEN, "stwo_trace20.npz"))[0]
    rng = np.random.default_rng(SEED + 91)
    proofs = [p] + [formats.stwo_corrupt(p, rng)[0] for _ in range(3)]
    want = O.stwo_verify_batch(proofs).tolist()
    texts = []
    for q in proofs:
        rec = verifier.stwo_record(q)
        texts += [write_text(q.cfg, rec, JSON, 0), write_text(q.cfg, rec, WIT)]
    batch = [texts[i % 8] for i in range(600)]
    status, stats = ver.verify_stwo_texts(p.cfg, batch)
    assert status.tolist() == [want[(i % 8) // 2] for i in range(600)]
    assert stats["host_parsed"] == 0 and want[0] == 0 and any(want[1:])


def test_two_threads_share_one_context(ver):
    """include/ss_verify.h, "threads": the scratch-using entry points serialize inside the context."""
    import threading
    base = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    rng = np.random.default_rng(SEED + 92)
    proofs = [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(3)]
    want = O.stwo_verify_batch(proofs).tolist()
    texts = [json.dumps(ss.stwo_to_json(q)).encode() for q in proofs]
    recs = [verifier.stwo_record(q) for q in proofs]
    errors = []

    def by_text():
        try:
            for _ in range(6):
                st, _ = ver.verify_stwo_texts(base.cfg, texts * 40)
                assert st.tolist() == want * 40
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def by_record():
        try:
            for _ in range(6):
                st = ver.verify_stwo_records(base.cfg, recs * 50)
                assert st.tolist() == want * 50
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=by_text), threading.Thread(target=by_record), threading.Thread(target=by_text)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


# ---------------------------------------------------------------------------------------- stark101
def test_stark101_texts_are_read_by_the_gpu(ver):
    """The reference's stark101 proof.json / proof.wit (78-digit decimal hashes) through the GPU reader: outcome and
    record equal the scalar rule's on the files and on mutants."""
    from test_text_fastpath import s101_canonical
    rnd = random.Random(SEED + 101)
    for fn, fmt in (("stark101_proof.json", JSON), (os.path.join("formats", "stark101_proof.wit"), WIT)):
        base = open(os.path.join(GOLDEN, fn), "rb").read()
        texts = [base] + [(_text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)) for i in range(1500)]
        texts += [b" " * k + base for k in range(1, 40)] if fmt == JSON else []
        recs, outcome = ver.read_stark101_texts(texts, fmt)
        taken = 0
        for i, t in enumerate(texts):
            want, rec = s101_canonical(t, fmt)
            assert (outcome[i] == 0) == want, (i, int(outcome[i]), want, t[:120])
            if want:
                pad = rec == 0xEEEEEEEE   # words the rule leaves alone are zero padding in the GPU's (pre-zeroed) record
                assert np.array_equal(recs[i][~pad], rec[~pad]) and not recs[i][pad].any(), i
                taken += 1
        assert outcome[0] == 0 and taken > 150


def test_stark101_verify_texts_mixed_shapes(ver, tmp_path):
    """ss_s101_verify_texts / _files: the protocol's proof (GPU reader), corrupted ones (GPU reader, rejected by the
    kernels), a proof with a layer less and one with 14-sibling paths (host reader; the latter verified in a batch of
    its own shape), garbage -- statuses equal the oracle's / stage 0, in any mix and across chunks."""
    s101 = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    rng = np.random.default_rng(SEED + 102)
    bad = [formats.stark101_corrupt(s101, rng)[0] for _ in range(6)]
    fewer = s101.copy(); fewer.layers = fewer.layers[:-1]
    longer = s101.copy()
    longer.evals[1].path = np.concatenate([longer.evals[1].path, longer.evals[1].path[:1]])   # 14 siblings
    proofs = [s101] + bad + [fewer, longer]
    want = O.s101_verify_batch(proofs).tolist()
    assert want[0] == 0 and all(want[1:])
    texts = []
    for i, p in enumerate(proofs):
        texts.append(json.dumps(ss.stark101_to_json(p)).encode() if i % 2 else ss.stark101_to_wit(p).encode())
    texts += [b"{}", b"nonsense"]
    want += [2, 2]
    canonical_n = sum(1 for p in proofs[:-2] if True)   # the protocol's shape, whatever the values
    for n in (len(texts), 9000):
        batch = [texts[i % len(texts)] for i in range(n)]
        status, stats = ver.verify_stark101_texts(batch)
        assert status.tolist() == [want[i % len(texts)] for i in range(n)]
        assert stats["host_parsed"] == sum(1 for i in range(n) if i % len(texts) >= canonical_n)
    paths = []
    for i, t in enumerate(texts):
        f = tmp_path / ("s%d.txt" % i)
        f.write_bytes(t)
        paths.append(str(f))
    status, _ = ver.verify_stark101_files(paths + [str(tmp_path / "absent")])
    assert status.tolist() == want + [2]
    # the same texts in one caller-pinned buffer (ss_s101_verify_texts_pinned): the staged verdicts, the same host share
    batch = [texts[i % len(texts)] for i in range(3000)]
    blob, offs, lens = ver.pinned_text_blob(batch)
    status, stats = ver.verify_stark101_texts_pinned(blob, offs, lens)
    assert status.tolist() == [want[i % len(texts)] for i in range(3000)]
    assert stats["host_parsed"] == sum(1 for i in range(3000) if i % len(texts) >= canonical_n)
    with pytest.raises(binding.SsError):
        ver.verify_stark101_texts_pinned(np.array(blob), offs, lens)  # pageable memory


def test_stark101_device_pack_equals_host_pack(ver):
    import torch
    s101 = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    rng = np.random.default_rng(SEED + 103)
    proofs = [s101] + [formats.stark101_corrupt(s101, rng)[0] for _ in range(4)]
    for ml, pm, n in ((10, 13, 131), (12, 20, 5), (31, 31, 64)):
        recs = [verifier.s101_record(proofs[i % 5], ml, pm) for i in range(n)]
        host = verifier.pack_s101(ml, pm, recs)
        sh = binding.S101Shape(ml, pm)
        rec_dev = torch.from_numpy(np.stack(recs).view(np.int32)).to(ver.device)
        out = torch.full((host.size,), 0x55, dtype=torch.int32, device=ver.device)
        import ctypes as C
        binding.check(binding.lib().ss_s101_pack_dev(ver.ctx, C.byref(sh), n, rec_dev.data_ptr(), out.data_ptr(),
                                                     int(torch.cuda.current_stream(ver.device).cuda_stream)))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), host)


def test_shared_path_texts_verify_like_their_per_query_form(ver):
    """The shared-path variant of proof.json (tests/test_shared_paths.py) through ss_stwo_verify_texts: since round 4
    the GPU reader takes canonical ones (tests/test_gpu_shared.py; round 3: the host reader undid the sharing), the
    expansion kernel undoes the sharing, the kernels verify the expanded proof; status words equal the oracle's on the
    per-query form.  A shared text with its members in another order still takes the host reader."""
    base = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    rng = np.random.default_rng(SEED + 93)
    qs = formats.stwo_queries(base)
    proofs, texts = [], []
    for p in [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(40)]:
        try:
            texts.append(json.dumps(ss.stwo_to_json(p, shared=True, queries=qs)).encode())
        except ss.MalformedProof:
            continue  # two queries disagree about a node: no shared form
        proofs.append(p)
    want = O.stwo_verify_batch(proofs).tolist()
    status, stats = ver.verify_stwo_texts(base.cfg, texts)
    assert status.tolist() == want and want[0] == 0 and sum(1 for w in want if w) > 10
    assert stats["host_parsed"] == 0
    odd = [json.dumps(dict(reversed(list(json.loads(t).items())))).encode() for t in texts[:6]]
    status, stats = ver.verify_stwo_texts(base.cfg, odd)
    assert status.tolist() == want[:6] and stats["host_parsed"] == 6


def test_long_runs_and_blank_blocks_across_windows(ver):
    """States that must be carried over many windows: kilobytes of whitespace between two tokens of a canonical
    proof.json (still canonical: same record), and kilobyte runs of digits / letters / quotes dropped into it at
    positions before, inside and behind strings (never canonical; the scalar rule and the kernels agree on every one)."""
    cfg = ss.PRODUCTION_CONFIG
    j = open(os.path.join(GOLDEN, "stwo_proof.json"), "rb").read().rstrip()
    want = verifier.parse_stwo_text(cfg, j)[1]
    rnd = random.Random(SEED + 95)
    commas = [i for i in range(len(j)) if j[i:i + 1] == b","]
    blank, noisy = [], []
    for _ in range(40):
        at = commas[rnd.randrange(len(commas))] + 1
        pad = bytes(rnd.choice(b" \n\t\r") for _ in range(rnd.randrange(1, 5000)))
        blank.append(j[:at] + pad + j[at:])
    recs, outcome = ver.read_stwo_texts(cfg, blank, JSON)
    assert not outcome.any() and all(np.array_equal(r, want) for r in recs)
    for _ in range(120):
        at = rnd.randrange(len(j))
        kind = rnd.randrange(5)
        run = (b"7" * rnd.randrange(1000, 4000) if kind == 0 else b"q" * rnd.randrange(1000, 4000) if kind == 1 else
               b'"' * rnd.randrange(1000, 3000) if kind == 2 else b"0x" + b"f" * rnd.randrange(60, 2100) if kind == 3 else
               b"_" * rnd.randrange(1024, 2100))
        noisy.append(j[:at] + run + j[at:])
    assert _agree(ver, cfg, noisy, JSON) == 0


def test_batches_of_very_different_texts_and_all_host_read(ver):
    """Thousands of tiny texts between full-size ones (one window each against hundreds), and a batch the GPU reader
    takes nothing of (every text has its members in another order): chunking, window bases and the host-reader fix-ups
    hold, verdicts are the record path's."""
    base = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    cfg = base.cfg
    good = json.dumps(ss.stwo_to_json(base), separators=(",", ":")).encode()
    odd = json.dumps(dict(reversed(list(ss.stwo_to_json(base).items())))).encode()
    batch, want = [], []
    for i in range(6000):
        if i % 50 == 0:
            batch.append(good); want.append(0)
        elif i % 50 == 25:
            batch.append(odd); want.append(0)
        else:
            batch.append([b"{}", b"", b"[1]", b"7", b'{"config":{}}'][i % 5]); want.append(2)
    status, stats = ver.verify_stwo_texts(cfg, batch)
    assert status.tolist() == want and stats["host_parsed"] == 6000 - 120
    status, stats = ver.verify_stwo_texts(cfg, [odd] * 700)   # ~150 MB, several chunks, all through the host reader
    assert not status.any() and stats["host_parsed"] == 700


def test_oversized_and_degenerate_texts(ver):
    """Texts far from the template's size: several times its skeleton, megabytes of one byte, the densest possible
    run of numbers (512 per window), a canonical text followed by megabytes of blanks (still canonical) or by a second
    copy.  Nothing but the blank-padded one is taken; the kernels and the scalar rule agree on every one."""
    cfg = ss.PRODUCTION_CONFIG
    j = open(os.path.join(GOLDEN, "stwo_proof.json"), "rb").read().rstrip()
    want = verifier.parse_stwo_text(cfg, j)[1]
    texts = [j + b" " * (3 << 20), j + j, j * 3, b"1," * (2 << 20), b"7" * (4 << 20), b"[" * (4 << 20), b'"' * ((1 << 20) + 1),
             b" " * (5 << 20), j[:len(j) // 2] + b"0," * 300000 + j[len(j) // 2:], b"x", b"", j[:1], b"\n" * 1025 + j]
    recs, outcome = ver.read_stwo_texts(cfg, texts, JSON)
    assert outcome.tolist() == [0] + [1] * (len(texts) - 2) + [0]
    assert np.array_equal(recs[0], want) and np.array_equal(recs[-1], want)
    for t in texts[1:-1]:
        assert not canonical(cfg, t, JSON)[0]
    status, stats = ver.verify_stwo_texts(cfg, texts)
    assert status.tolist() == [0] + [2] * (len(texts) - 2) + [0] and stats["host_parsed"] == len(texts) - 2


def test_template_cache_eviction_keeps_the_templates_in_use(ver):
    """A context keeps the device templates of a dozen (config, format) pairs.  Cycling through eight configs (three
    formats each) must never evict a template of the config being verified: every canonical text stays on the GPU
    reader (host_parsed 0) whatever was verified before (round 3 evicted the oldest entry even when the same call had
    just been handed a view of it)."""
    ps = [ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))),
          ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))]
    for npz in ("stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace16_blake2s.npz", "stwo_wide256_blake2s.npz", "stwo_trace20.npz",
                "stwo_trace20_blake2s.npz"):
        ps.append(records.load_stwo_npz(os.path.join(GOLDEN, npz))[0])
    texts = [[json.dumps(ss.stwo_to_json(p), separators=(",", ":")).encode(), ss.stwo_to_wit(p).encode(),
              json.dumps(ss.stwo_to_json(p, shared=True)).encode()] for p in ps]
    for rnd in range(3):
        for p, ts in zip(ps, texts):
            status, stats = ver.verify_stwo_texts(p.cfg, ts * 2)
            assert status.tolist() == [0] * 6 and stats["host_parsed"] == 0, (rnd, p.cfg)
