"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Every status word must equal the oracle's -- not only accept/reject but the code of the first
failing assert.  Run on the GPU box with `pytest -m gpu`.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import binding, formats, verifier
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
SEED = 0x5EED2025


@pytest.fixture(scope="module")
def ver():
    return verifier.Verifier(0)


# ------------------------------------------------------------------------- primitives
def _mixed_u32(rng, n):
    """Canonical, unreduced and edge-case words."""
    v = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    edges = np.array([0, 1, 2, 2147483646, 2147483647, 2147483648, 2147483649, 3221225472,
                      3221225473, 3221225474, 4294967294, 4294967295], dtype=np.uint32)
    v[:len(edges)] = edges
    return v


def test_sha256_pair(ver):
    rng = np.random.default_rng(SEED)
    msgs = rng.integers(0, 1 << 32, size=(257, 16), dtype=np.uint64).astype(np.uint32)
    msgs[0] = 0
    got = ver.selftest(0, msgs)
    for m, g in zip(msgs, got):
        want = np.frombuffer(O.sha256(m.astype(">u4").tobytes()), dtype=">u4")
        assert np.array_equal(g, want)


def test_m31_ops(ver):
    rng = np.random.default_rng(SEED + 1)
    a, b = _mixed_u32(rng, 4096), np.roll(_mixed_u32(rng, 4096), 7)
    got = ver.selftest(1, np.stack([a, b], 1))
    L = O.lib()
    for x, y, g in zip(a.tolist(), b.tolist(), got.tolist()):
        out = C.c_uint32()
        inv = out.value if L.so_m31_inv(x, C.byref(out)) == 0 else 0xFFFFFFFF
        assert g == [L.so_m31_add(x, y), L.so_m31_sub(x, y), L.so_m31_mul(x, y), inv]


def test_qm31_ops(ver):
    rng = np.random.default_rng(SEED + 2)
    v = _mixed_u32(rng, 8 * 1024).reshape(-1, 8)
    v[3, :4] = 0  # inverse of zero aborts
    got = ver.selftest(2, v)
    L = O.lib()
    for row, g in zip(v.tolist(), got.tolist()):
        a, b = O.qm(row[:4]), O.qm(row[4:])
        inv = O.QM31()
        ok = L.so_qm31_inv(a, C.byref(inv)) == 0
        want = list(L.so_qm31_mul(a, b).t()) + (list(inv.t()) if ok else [0xFFFFFFFF] * 4)
        assert g == want


def test_circle_point(ver):
    rng = np.random.default_rng(SEED + 3)
    idx = _mixed_u32(rng, 2048)
    idx[20:40] = np.arange(20)
    idx[40] = 1389
    got = ver.selftest(3, idx)
    L = O.lib()
    for i, g in zip(idx.tolist(), got.tolist()):
        assert tuple(g) == L.so_circle_point_index_to_m31_point(i).t()


def test_stark101_field(ver):
    rng = np.random.default_rng(SEED + 4)
    a, b = _mixed_u32(rng, 4096), np.roll(_mixed_u32(rng, 4096), 5)
    got = ver.selftest(4, np.stack([a, b], 1))
    L = O.lib()
    for x, y, g in zip(a.tolist(), b.tolist(), got.tolist()):
        out = C.c_uint32()
        d = out.value if L.so_s101_div_mod(x, y, C.byref(out)) == 0 else 0xFFFFFFFF
        assert g == [L.so_s101_add_mod(x, y), L.so_s101_sub_mod(x, y), L.so_s101_mul_mod(x, y), d]


# ---------------------------------------------------------------------------- stark101
def test_stark101_accept(ver, s101_proof):
    assert ver.verify_stark101([s101_proof]).tolist() == [0]
    assert verifier.verify_stark101(s101_proof) is True


def test_stark101_corruptions(ver, s101_proof):
    rng = np.random.default_rng(SEED)
    batch, notes = [s101_proof], ["valid"]
    for _ in range(299):
        p, why = formats.stark101_corrupt(s101_proof, rng)
        batch.append(p)
        notes.append(why)
    got = ver.verify_stark101(batch)
    want = O.s101_verify_batch(batch)
    bad = [(i, notes[i], hex(got[i]), hex(want[i])) for i in range(len(batch)) if got[i] != want[i]]
    assert not bad, bad[:5]
    assert want[0] == 0 and (want[1:] != 0).all()
    assert len(set(want.tolist())) > 5  # the corruptions exercise several stages


def test_stark101_ragged_shapes(ver, s101_proof):
    """Shorter / longer layer lists and Merkle paths (List<_, 32> is data-dependent)."""
    batch = [s101_proof]
    p = s101_proof.copy(); p.layers = p.layers[:7]; batch.append(p)
    p = s101_proof.copy(); p.layers = p.layers[:0]; batch.append(p)
    p = s101_proof.copy(); p.evals[1].path = p.evals[1].path[:9]; batch.append(p)
    p = s101_proof.copy()
    p.layers[3].cpb.path = np.concatenate([p.layers[3].cpb.path, p.layers[3].cpb.path[:2]])
    batch.append(p)
    p = s101_proof.copy(); p.layers[0].cpa.path = p.layers[0].cpa.path[:0]; batch.append(p)
    p = s101_proof.copy(); p.layers = p.layers + p.layers[-3:]; batch.append(p)
    got = ver.verify_stark101(batch)
    want = O.s101_verify_batch(batch)
    assert got.tolist() == want.tolist() and want[0] == 0


def test_stark101_unreduced_words(ver, s101_proof):
    """Field words >= p are legal u32 witnesses; wrap-around semantics must match."""
    P = 3221225473
    batch = []
    for mut in range(6):
        p = s101_proof.copy()
        if mut == 0: p.last = 0xFFFFFFFF
        if mut == 1: p.evals[0].ev = 0xFFFFFFFF
        if mut == 2: p.layers[2].cpb.ev = P
        if mut == 3: p.layers[0].beta = 0xFFFFFFFE
        if mut == 4: p.evals[2].ev = 0
        if mut == 5: p.layers[9].cpa.ev = P + 5
        batch.append(p)
    got = ver.verify_stark101(batch)
    assert got.tolist() == O.s101_verify_batch(batch).tolist()


def test_stark101_batch_4096(ver, s101_proof):
    """BASELINE config 2: 4096 proofs, every fourth one corrupted."""
    rng = np.random.default_rng(SEED + 9)
    distinct = [s101_proof] + [formats.stark101_corrupt(s101_proof, rng)[0] for _ in range(15)]
    idx = [0 if i % 4 else 1 + (i // 4) % 15 for i in range(4096)]
    batch = [distinct[i] for i in idx]
    got = ver.verify_stark101(batch)
    want_d = O.s101_verify_batch(distinct)
    assert got.tolist() == [int(want_d[i]) for i in idx]


# -------------------------------------------------------------------------------- stwo
@pytest.mark.parametrize("which", ["small", "prod"])
def test_stwo_fixtures(ver, stwo_small, stwo_prod, which):
    p = stwo_small if which == "small" else stwo_prod
    assert ver.verify_stwo([p], verifier.MODE_FIXTURE, cfg=p.cfg).tolist() == [0]
    lit = ver.verify_stwo([p], verifier.MODE_LITERAL, cfg=p.cfg).tolist()
    assert lit == [O.stwo_verify(p, O.MODE_LITERAL)] and lit[0] == (7 << 24) | 1


@pytest.mark.parametrize("mode", [verifier.MODE_FIXTURE, verifier.MODE_LITERAL])
def test_stwo_corruptions(ver, stwo_small, stwo_prod, mode):
    for base, n in ((stwo_small, 200), (stwo_prod, 120)):
        rng = np.random.default_rng(SEED + mode)
        batch, notes = [base], ["valid"]
        for _ in range(n - 1):
            p, why = formats.stwo_corrupt(base, rng)
            batch.append(p)
            notes.append(why)
        got = ver.verify_stwo(batch, mode, cfg=base.cfg)
        want = O.stwo_verify_batch(batch, mode)
        bad = [(i, notes[i], hex(got[i]), hex(want[i])) for i in range(n) if got[i] != want[i]]
        assert not bad, bad[:5]
        if mode == verifier.MODE_FIXTURE:
            assert want[0] == 0 and len(set(want.tolist())) > 4


def test_stwo_wrong_path_lengths(ver, stwo_prod):
    batch = [stwo_prod]
    p = stwo_prod.copy(); p.trace_paths[3] = p.trace_paths[3][:-1]; batch.append(p)
    p = stwo_prod.copy()
    p.cp_paths[0] = np.concatenate([p.cp_paths[0], p.cp_paths[0][:1]])
    batch.append(p)
    p = stwo_prod.copy(); p.fri_paths[4][7] = p.fri_paths[4][7][:2]; batch.append(p)
    p = stwo_prod.copy(); p.fri_paths[0][0] = p.fri_paths[0][0][:0]; p.pow_nonce += 1; batch.append(p)
    got = ver.verify_stwo(batch, cfg=stwo_prod.cfg)
    want = O.stwo_verify_batch(batch)
    assert got.tolist() == want.tolist()
    assert [hex(w) for w in want[1:]] == ['0x5000030', '0x5000002', '0x7040070', '0x4000000']


def test_stwo_unreduced_words(ver, stwo_prod):
    P = 2147483647
    batch = []
    for mut in range(8):
        p = stwo_prod.copy()
        if mut == 0: p.fri_witness[0, 0, 1] += P
        if mut == 1: p.fri_witness[3, 5, :] = 0xFFFFFFFF
        if mut == 2: p.oods_trace[1, 2] += P
        if mut == 3: p.oods_cp[5, 0] = 0xFFFFFFFF
        if mut == 4: p.last_layer[0] += P
        if mut == 5: p.trace_vals[2, 1] += P
        if mut == 6: p.cp_vals[7, 9] = 0xFFFFFFFE
        if mut == 7: p.oods_trace[0, 0] = 0x80000001
        batch.append(p)
    for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
        got = ver.verify_stwo(batch, mode, cfg=stwo_prod.cfg)
        assert got.tolist() == O.stwo_verify_batch(batch, mode).tolist()


def test_stwo_batch_replicated(ver, stwo_prod):
    """A 1000-proof batch (not a multiple of 64) mixing a valid proof and corruptions."""
    rng = np.random.default_rng(SEED + 5)
    distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(9)]
    idx = [0 if i % 3 else 1 + (i // 3) % 9 for i in range(1000)]
    got = ver.verify_stwo([distinct[i] for i in idx], cfg=stwo_prod.cfg)
    want_d = O.stwo_verify_batch(distinct)
    assert got.tolist() == [int(want_d[i]) for i in idx]
    b = ver.stwo_batch([distinct[i] for i in idx])
    b.run()
    assert b.accepted() == sum(1 for i in idx if want_d[i] == 0)


# ------------------------------------------------- BASELINE.json configs 3-5 (prover-made proofs)
def _load_npz(name):
    import os
    from stark_symphony_amd import records
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip("%s not generated" % name)
    return records.load_stwo_npz(path)


@pytest.mark.parametrize("name", ["stwo_trace16.npz", "stwo_trace16_blake2s.npz", "stwo_wide256.npz",
                                  "stwo_wide256_blake2s.npz", "stwo_trace20.npz", "stwo_trace20_blake2s.npz"])
def test_stwo_baseline_configs(ver, name):
    """configs[2] (2^16 trace, 32 queries; SHA-256 = the pinned hash, and Blake2s = the config as
    BASELINE.json names it), configs[4] (256 columns, LDE 2^18) and configs[3] (2^20 trace, both
    hashes): valid proofs accept, seeded corruptions match the oracle word for word, both modes."""
    proofs = _load_npz(name)
    rng = np.random.default_rng(SEED + 11)
    batch = list(proofs)
    for _ in range(40):
        batch.append(formats.stwo_corrupt(proofs[0], rng)[0])
    got = ver.verify_stwo(batch, cfg=proofs[0].cfg)
    want = O.stwo_verify_batch(batch)
    assert got.tolist() == want.tolist()
    assert (want[:len(proofs)] == 0).all() and (want[len(proofs):] != 0).any()
    lit = ver.verify_stwo(batch, verifier.MODE_LITERAL, cfg=proofs[0].cfg)
    assert lit.tolist() == O.stwo_verify_batch(batch, O.MODE_LITERAL).tolist()


@pytest.mark.streams
def test_pipeline_matches_single_stream(ver, stwo_prod):
    """HEAD/TAIL halves on two streams with two slots give the same verdicts."""
    rng = np.random.default_rng(SEED + 12)
    distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(7)]
    batch = [distinct[i % 8] for i in range(256)]
    ref = ver.verify_stwo(batch, cfg=stwo_prod.cfg)
    a = ver.stwo_batch(batch)
    pipe = verifier.Pipeline([a, a.sibling()])
    used = [pipe.submit() for _ in range(5)]
    pipe.synchronize()
    assert used == [0, 1, 0, 1, 0]
    for slot in pipe.slots:
        assert slot.status().tolist() == ref.tolist()
        assert slot.accepted() == int((ref == 0).sum())


@pytest.mark.streams
def test_pipeline_callbacks_see_every_pass_once(ver, stwo_prod):
    """Pipeline.submit(after_tail=...) / (on_reuse=...) + flush: the hook a caller hangs its accept reduce on runs exactly
    once per pass, ordered after that pass and before the slot's next one (bench.py's use: all_reduce of the count); with
    two tail streams and a single head stream (head_streams=1) the verdicts are the same."""
    import torch
    rng = np.random.default_rng(SEED + 15)
    distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(5)]
    batch = [distinct[i % 6] for i in range(192)]
    ref = ver.verify_stwo(batch, cfg=stwo_prod.cfg)
    want = int((ref == 0).sum())
    a = ver.stwo_batch(batch)
    for kw in ({}, {"tail_streams": 2}, {"tail_streams": 2, "head_streams": 1}):
        slots = [a, a.sibling(), a.sibling()]
        for how in ("after_tail", "on_reuse"):
            pipe = verifier.Pipeline(slots, **kw)
            # one accumulator per slot: on_reuse hooks of different slots run on different head streams, at the same time
            totals = torch.zeros(len(slots), dtype=torch.int64, device=ver.device)
            pipe.wait_current()  # the fill ran on the current stream; the hooks write `totals` on the pipeline's streams
            calls = []

            def hook(k):
                calls.append(k)
                totals[k:k + 1].add_(slots[k].accept_dev.to(torch.int64))  # on the stream the pipeline made current: ordered after pass k

            for _ in range(7):
                pipe.submit(**{how: hook})
            if how == "on_reuse":
                assert calls == [0, 1, 2, 0]  # passes 0..3 when their slots came round; 4, 5, 6 are still pending
                pipe.flush(hook)
            pipe.synchronize()
            assert sorted(calls) == [0, 0, 0, 1, 1, 2, 2] and totals.tolist() == [3 * want, 2 * want, 2 * want], (kw, how, calls, totals.tolist())
            for slot in slots:
                assert slot.status().tolist() == ref.tolist()


@pytest.mark.parametrize("which,n", [("small", 3), ("prod", 70), ("wide", 5)])
def test_device_pack_equals_host_pack(ver, stwo_small, stwo_prod, which, n):
    """ss_stwo_pack_dev (GPU re-tiling of raw records) writes the same words as ss_stwo_pack."""
    base = {"small": stwo_small, "prod": stwo_prod}.get(which) or _load_npz("stwo_wide256.npz")[0]
    rng = np.random.default_rng(SEED + 13)
    proofs = [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(n - 1)]
    recs = [verifier.stwo_record(p) for p in proofs]
    host = verifier.pack_stwo(base.cfg, verifier.MODE_FIXTURE, recs)
    dev = ver.pack_stwo_on_device(base.cfg, verifier.MODE_FIXTURE, recs)
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), host)


@pytest.mark.streams
def test_host_buffer_entry_point(ver, stwo_prod):
    """ss_stwo_verify_records: pinned chunked upload + GPU packing + verify, several chunks
    (1300 proofs of 54 KB > one 64 MiB staging buffer), called twice to reuse its scratch."""
    rng = np.random.default_rng(SEED + 14)
    distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(6)]
    recs = [verifier.stwo_record(p) for p in distinct]
    want_d = O.stwo_verify_batch(list(distinct))
    for n in (1300, 17):
        idx = [i % 7 for i in range(n)]
        got = ver.verify_stwo_records(stwo_prod.cfg, [recs[i] for i in idx])
        assert got.tolist() == [int(want_d[i]) for i in idx]
        # the same batch as ONE 2-d array (a record per row: no per-record work in the Python wrapper)
        got2 = ver.verify_stwo_records(stwo_prod.cfg, np.stack([recs[i] for i in idx]))
        assert got2.tolist() == got.tolist()


def test_stwo_mixed_shapes_in_one_call(ver, stwo_small, stwo_prod):
    """verify_stwo groups proofs by StwoConfig; statuses come back in input order."""
    rng = np.random.default_rng(SEED + 15)
    wide = _load_npz("stwo_wide256.npz")[0]
    proofs = []
    for k in range(30):
        base = (stwo_small, stwo_prod, wide)[int(rng.integers(3))]
        proofs.append(base if rng.integers(2) else formats.stwo_corrupt(base, rng)[0])
    assert len(verifier.group_by_config(proofs)) == 3
    got = ver.verify_stwo(proofs, cfg=[stwo_small.cfg, stwo_prod.cfg, wide.cfg])
    want = [O.stwo_verify(p, O.MODE_FIXTURE) for p in proofs]
    assert got.tolist() == want and 0 in want and any(want)


def test_stwo_full_size_batch_2p20(ver):
    """BASELINE.json configs[3], one GPU's share of the 65 536-proof batch split over 8 GPUs
    (bench.py itself asserts that all 65 536 resident proofs accept): 8 192 proofs of the 2^20-trace shape (1.4 GB
    resident), half of them seeded corruptions.  Size-independent property: the
    status vector equals the oracle's verdicts of the distinct proofs, gathered through the
    replication map, and the device accept count equals the number of valid entries."""
    proofs = _load_npz("stwo_trace20.npz")
    rng = np.random.default_rng(SEED + 16)
    distinct = list(proofs) + [formats.stwo_corrupt(proofs[i % len(proofs)], rng)[0] for i in range(62)]
    want_d = O.stwo_verify_batch(distinct)
    n = 8192
    idx = [(i // 2) % len(proofs) if i % 2 == 0 else len(proofs) + (i // 2) % 62 for i in range(n)]
    b = ver.stwo_batch([distinct[i] for i in idx])
    b.run()
    got = b.status()
    assert got.tolist() == [int(want_d[i]) for i in idx]
    assert b.accepted() == sum(1 for i in idx if want_d[i] == 0) >= n // 2


def test_stwo_full_size_batch_65536(ver):
    """BASELINE.json configs[3] at its full size on one GPU: ONE batch of 65 536 proofs of the 2^20-trace
    shape (11.2 GB resident) = 64 distinct records (32 valid, 32 seeded corruptions) x 1 024, built through
    the device replication map as bench.py builds its batch.  The status vector must equal the oracle's
    64 verdicts gathered through the map and the accept count must be 32 768 -- with the pair
    memoisation on and with every path hashed in full (SS_FLAG_NO_DEDUP)."""
    import torch
    base = _load_npz("stwo_trace20.npz")[0]
    rng = np.random.default_rng(SEED + 18)
    distinct = [base] * 32 + [formats.stwo_corrupt(base, rng)[0] for _ in range(32)]
    order = rng.permutation(64)
    distinct = [distinct[i] for i in order]
    want_d = O.stwo_verify_batch(distinct)
    assert int((want_d == 0).sum()) == 32
    want = np.tile(want_d, 1024)
    for flags in (0, verifier.FLAG_NO_DEDUP):
        v = verifier.Verifier(0)
        v.stwo_flags = flags
        b = v.stwo_batch(distinct, replicate=1024)
        assert b.n == 65536
        b.run()
        got = b.status()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (flags, int(bad[0]), hex(got[bad[0]]), hex(want[bad[0]]))
        assert b.accepted() == 32768
        del b
        v.close()
        torch.cuda.empty_cache()


@pytest.mark.streams
@pytest.mark.parametrize("concurrent", [False, True])
def test_graphed_pipeline_matches_eager(ver, s101_proof, stwo_prod, concurrent):
    """hipGraph replay of the pipelined passes (fork / join of the head and tail streams captured
    once) leaves the same status words and accept counts as eager submission."""
    rng = np.random.default_rng(SEED + 17)
    for family in ("stark101", "stwo"):
        if family == "stark101":
            distinct = [s101_proof] + [formats.stark101_corrupt(s101_proof, rng)[0] for _ in range(5)]
            want_d = O.s101_verify_batch(distinct)
            idx = [i % 6 if i % 2 else 0 for i in range(257)]
            batch = ver.stark101_batch([distinct[i] for i in idx])
        else:
            distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(5)]
            want_d = O.stwo_verify_batch(distinct)
            idx = [i % 6 if i % 2 else 0 for i in range(130)]
            batch = ver.stwo_batch([distinct[i] for i in idx])
        want = [int(want_d[i]) for i in idx]
        slots = [batch, batch.sibling(), batch.sibling()]
        gp = verifier.GraphedPipeline(slots, concurrent_tails=concurrent)
        assert gp.steps_per_replay == 3
        for _ in range(3):
            for s in slots:
                s.status_dev.fill_(0x55)  # on the current stream; replay() orders the graph after it
            gp.replay()
            gp.synchronize()
            for s in slots:
                assert s.status().tolist() == want
                assert s.accepted() == sum(1 for w in want if w == 0)


def test_lazily_reduced_field_forms(ver):
    """qm31_mul_c / qm31_sqr_c / qm31_mul_im_c / m31_*_c / m31_red64 (ss_fields.h) against the
    oracle's reference-order arithmetic, on words in [0, P] including 0, 1, P - 1 and P."""
    P = 2147483647
    rng = np.random.default_rng(SEED + 18)
    v = rng.integers(0, P, size=(4096, 8), dtype=np.uint32)
    edge = np.array([0, 1, 2, P - 2, P - 1, P], dtype=np.uint32)
    v[:512] = rng.choice(edge, size=(512, 8))
    v[512:1024, ::2] = rng.choice(edge, size=(512, 4))
    got = ver.selftest(5, v)
    L = O.lib()
    for row, g in zip(v.tolist(), got.tolist()):
        a, b = O.qm(row[:4]), O.qm(row[4:])
        im_b = O.qm([0, 0, row[6], row[7]])
        want = list(L.so_qm31_mul(a, b).t()) + list(L.so_qm31_mul(a, a).t()) + list(L.so_qm31_mul(a, im_b).t())
        x, y = row[0] % P, row[4] % P
        want += [(x + y) % P, (x - y) % P, (x * y) % P, ((row[1] << 32) | row[5]) % P]
        assert g == want, (row, g, want)


def test_c_abi_consumer_program(ver, tmp_path, s101_proof, stwo_prod):
    """examples/ss_verify_file.c (plain C over include/ss_verify.h, no Python in the process):
    verdicts and exit status equal the oracle's for a file of raw records."""
    import subprocess
    from test_host import _build_c_example
    exe = _build_c_example(tmp_path)
    rng = np.random.default_rng(SEED + 19)
    proofs = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(5)]
    short = stwo_prod.copy(); short.fri_paths[2][5] = short.fri_paths[2][5][:-1]
    long_ = stwo_prod.copy(); long_.cp_paths[1] = np.concatenate([long_.cp_paths[1], long_.cp_paths[1][:2]])
    proofs += [short, long_]  # wrong path lengths: the record's path_len trailer carries them
    want = O.stwo_verify_batch(proofs)
    c = stwo_prod.cfg
    path = tmp_path / "stwo.bin"
    np.concatenate([verifier.stwo_record(p) for p in proofs]).astype("<u4").tofile(path)
    args = [exe, "stwo", str(c.n_cols), str(c.trace_log), str(c.lde_log), str(c.n_queries), str(c.n_layers),
            str(c.pow_bits), "0", "1", str(path)]
    r = subprocess.run(args, capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(proofs), r.stderr
    for i, w in enumerate(want.tolist()):
        assert lines[i] == ("proof %d: ACCEPT" % i if w == 0 else
                            "proof %d: REJECT (first failing assert 0x%08x)" % (i, w))
    assert r.returncode == (1 if any(want) else 0)
    ok = tmp_path / "one.bin"
    verifier.stwo_record(stwo_prod).astype("<u4").tofile(ok)
    assert subprocess.run(args[:-1] + [str(ok)], capture_output=True).returncode == 0
    # the same from SHARED records (ABI 2.2): the proofs that have a shared form, back to back in one file
    qs = formats.stwo_queries(stwo_prod)
    sh, sh_want = [], []
    for p, w in zip(proofs, want.tolist()):
        try:
            sh.append(verifier.stwo_shared_record(p, qs))
            sh_want.append(w)
        except ValueError:
            pass
    assert len(sh) >= 3 and sh_want[0] == 0
    spath = tmp_path / "shared.bin"
    np.concatenate(sh).astype("<u4").tofile(spath)
    r = subprocess.run([exe, "stwo-shared"] + args[2:-1] + [str(spath)], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(sh), r.stderr
    for i, w in enumerate(sh_want):
        assert lines[i] == ("proof %d: ACCEPT" % i if w == 0 else "proof %d: REJECT (first failing assert 0x%08x)" % (i, w))
    assert r.returncode == (1 if any(sh_want) else 0)
    # ... from MINIMAL records (ABI 2.3), and from text files in three forms (the library reads and parses them)
    mins, min_want = [], []
    for p, w in zip(proofs, want.tolist()):
        try:
            mins.append(verifier.stwo_minimise_record(c, verifier.stwo_record(p), qs))
            min_want.append(O.stwo_verify_minimal(c, mins[-1], 1))
        except ValueError:
            pass
    assert len(mins) >= 3 and min_want[0] == 0
    mpath = tmp_path / "minimal.bin"
    np.concatenate(mins).astype("<u4").tofile(mpath)
    r = subprocess.run([exe, "stwo-minimal"] + args[2:-1] + [str(mpath)], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(mins), r.stderr
    for i, w in enumerate(min_want):
        assert lines[i] == ("proof %d: ACCEPT" % i if w == 0 else "proof %d: REJECT (first failing assert 0x%08x)" % (i, w))
    assert r.returncode == (1 if any(min_want) else 0)
    import json
    files = {"a.json": (json.dumps(formats.stwo_to_json(stwo_prod)).encode(), 1), "a.wit": (formats.stwo_to_wit(stwo_prod).encode(), 2),
             "a.min.json": (verifier.write_stwo_minimal_text(c, mins[0]), 4)}
    for name, (text, fmt) in files.items():
        (tmp_path / name).write_bytes(text)
        r = subprocess.run([exe, "stwo-text"] + args[2:-1] + [str(fmt), str(tmp_path / name)], capture_output=True, text=True)
        assert r.returncode == 0 and "ACCEPT" in r.stdout and "0 of 1 texts went through the host reader" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "stwo-text"] + args[2:-1] + ["4", str(tmp_path / "a.min.json"), str(tmp_path / "absent")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "REJECT (0x00000002)" in r.stdout
    ml, pm = verifier.s101_shape_of([s101_proof])
    p101 = tmp_path / "s101.bin"
    verifier.s101_record(s101_proof, ml, pm).astype("<u4").tofile(p101)
    assert subprocess.run([exe, "stark101", str(ml), str(pm), str(p101)], capture_output=True).returncode == 0


def test_replicated_batch_built_on_the_device_equals_the_host_pack(ver, stwo_prod):
    """Verifier.stwo_batch(replicate=k): distinct records uploaded once, gathered and re-tiled on
    the GPU -- the same words as packing the replicated list on the host, and the same verdicts."""
    rng = np.random.default_rng(SEED + 20)
    distinct = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(4)]
    b = ver.stwo_batch(distinct, verifier.MODE_FIXTURE, replicate=27)
    recs = [verifier.stwo_record(p) for p in distinct]
    host = verifier.pack_stwo(stwo_prod.cfg, verifier.MODE_FIXTURE, recs * 27)
    assert b.n == 135 and np.array_equal(b.batch.cpu().numpy().view(np.uint32), host)
    b.run()
    want = O.stwo_verify_batch(distinct).tolist()
    assert b.status().tolist() == want * 27


# ---------------------------------------------------------- `simfony run` shim
def _cli_here(*args):
    """The same command through cli.main() in THIS process (stdout / stderr captured): the exit status is main()'s return
    value.  A new python process per case costs seconds (twenty each on a slow box); the process-level contract itself is
    checked by the cases that go through _cli."""
    import contextlib
    import io
    import types
    from stark_symphony_amd import cli
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        try:
            rc = cli.main(["verify", *args])
        except SystemExit as e:  # argparse
            rc = e.code
    return types.SimpleNamespace(returncode=rc, stdout=out.getvalue(), stderr=err.getvalue())


def _cli(*args):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-m", "stark_symphony_amd.cli", "verify", *args], cwd=ROOT,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    # the ROCm runtime of the test image prints one line about a missing libdrm id table to stderr
    r.stderr = "".join(l for l in r.stderr.splitlines(True) if "amdgpu.ids" not in l)
    return r


def test_cli_verify_exit_status_contract(tmp_path):
    """The reference's process contract (simfony-cli/src/main.rs:205-206,254-257,273): exit 0 for a
    satisfied program; `Error: Failed to run program ...` on stderr and exit 1 for a failed assert;
    a witness that does not type-check is exit 1 as well (main.rs:77-81,187-190)."""
    import json
    import os
    from conftest import GOLDEN
    F = os.path.join(GOLDEN, "formats")
    r = _cli("--family", "stark101", "--witness", os.path.join(F, "stark101_proof.wit"))
    assert r.returncode == 0 and "ACCEPT" in r.stdout and r.stderr == "", r.stderr
    r = _cli("--family", "stwo", "--witness", os.path.join(F, "stwo_proof.wit"))  # production = default
    assert r.returncode == 0 and "ACCEPT" in r.stdout, r.stderr
    r = _cli_here("--family", "stwo", "--config", "testing", "--witness", os.path.join(F, "stwo_proof_test.wit"))
    assert r.returncode == 0, r.stderr
    r = _cli_here("--family", "stwo", "--proof", os.path.join(GOLDEN, "stwo_proof.json"),
             os.path.join(GOLDEN, "stwo_proof.json"))
    assert r.returncode == 0 and r.stdout.count("ACCEPT") == 2
    # the test-config proof where production is enforced: shape mismatch = typing failure = exit 1
    r = _cli_here("--family", "stwo", "--witness", os.path.join(F, "stwo_proof_test.wit"))
    assert r.returncode == 1 and r.stderr.startswith("Error: Failed to run program") and "ACCEPT" not in r.stdout
    # one flipped bit in a committed root of the stark101 witness
    wit = json.load(open(os.path.join(F, "stark101_proof.wit")))
    v = wit["P_MT_ROOT"]["value"]
    wit["P_MT_ROOT"]["value"] = v[:-1] + ("0" if v[-1] != "0" else "1")
    bad = tmp_path / "bad.wit"
    bad.write_text(json.dumps(wit))
    r = _cli("--family", "stark101", "--witness", str(bad))
    assert r.returncode == 1 and r.stderr.startswith("Error: Failed to run program"), (r.stdout, r.stderr)
    # accepted and rejected inputs together: every verdict is printed, exit 1
    r = _cli_here("--family", "stark101", "--witness", os.path.join(F, "stark101_proof.wit"), str(bad))
    assert r.returncode == 1 and r.stdout.count("ACCEPT") == 1 and "REJECT" in r.stderr
    # literal mode = the .simf text, which rejects the repo's own proof at the first FRI decommitment
    r = _cli_here("--family", "stwo", "--mode", "literal", "--witness", os.path.join(F, "stwo_proof.wit"))
    assert r.returncode == 1 and "0x07000001" in r.stderr
    # malformed witness / missing file: exit 1, nothing verified
    junk = tmp_path / "junk.wit"
    junk.write_text("{\"P_MT_ROOT\": {\"value\": \"(1, 2\", \"type\": \"u256\"}}")
    assert _cli_here("--family", "stark101", "--witness", str(junk)).returncode == 1
    assert _cli("--family", "stwo", "--witness", str(tmp_path / "absent.wit")).returncode == 1


def test_stwo_config_policy_on_the_gpu(ver, stwo_small, stwo_prod):
    """verify_stwo enforces the caller's config: the one-query test proof is status 1 where the
    production config is expected, and verifies where it is allowed."""
    got = ver.verify_stwo([stwo_prod, stwo_small], cfg=stwo_prod.cfg)
    assert got.tolist() == [0, verifier.STATUS_CONFIG_MISMATCH]
    assert got.tolist() == O.stwo_verify_batch([stwo_prod, stwo_small], cfg=stwo_prod.cfg).tolist()
    assert ver.verify_stwo([stwo_prod, stwo_small], cfg=[stwo_prod.cfg, stwo_small.cfg]).tolist() == [0, 0]
    assert verifier.verify_stwo(stwo_prod) is True and verifier.verify_stwo(stwo_small) is False
    with pytest.raises(TypeError):
        ver.verify_stwo([stwo_prod])  # no expected config, no verdict


# ----------------------------------------------------------------------- seeded fuzz slice
def test_fuzz_slice_matches_the_oracle(ver, s101_proof, stwo_small, stwo_prod):
    """A seeded 2 000-mutant slice of tools/fuzz_parity.py (the long runs are summarised in
    profiles/*_fuzz_parity.txt): bit flips, words replaced by 0 / P / P+v / 2^32-1, siblings swapped,
    values copied between queries, ragged path lengths, double mutations; both stwo modes."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz
    rng = np.random.default_rng(SEED + 23)
    muts = [s101_proof] + [fz.mutate_s101(s101_proof, rng) for _ in range(400)]
    assert ver.verify_stark101(muts).tolist() == O.s101_verify_batch(muts).tolist()
    wide = _load_npz("stwo_wide256.npz")[0]
    t16 = _load_npz("stwo_trace16.npz")[0]
    seen = set()
    for base, n in ((stwo_prod, 500), (stwo_small, 300), (wide, 200), (t16, 200)):
        muts = [base] + [fz.mutate_stwo(base, rng) for _ in range(n)]
        for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
            got, want = ver.verify_stwo(muts, mode, cfg=base.cfg), O.stwo_verify_batch(muts, mode)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (mode, int(bad[0]), hex(got[bad[0]]), hex(want[bad[0]]))
            seen |= set(want.tolist())
    assert len(seen) > 12


# ------------------------------------------------- Merkle pair memoisation
def _paths_of(p, kind):
    """kind 0 trace, 1 cp, 2 + l FRI layer l -> the list of Q paths (uint8[len, 32] each)"""
    return p.trace_paths if kind == 0 else p.cp_paths if kind == 1 else p.fri_paths[kind - 2]


def test_pair_memoisation_when_queries_disagree_about_a_node(ver, stwo_prod):
    """The top levels of every tree are hashed once per distinct (left, right) pair.  The status
    word must stay the reference's when queries present DIFFERENT bytes for the same node: one
    query's sibling corrupted at every shared depth (leader and follower alike), every query's
    corrupted alike, a corrupted sibling below the shared levels, a query duplicated."""
    c = stwo_prod.cfg
    rng = np.random.default_rng(SEED + 31)
    batch, notes = [stwo_prod], ["valid"]
    kinds = [0, 1, 2, 2 + c.n_layers // 2, 2 + c.n_layers]
    for kind in kinds:
        length = c.lde_log if kind < 2 else c.fri_path_len(kind - 2)
        for up in range(0, min(length, 8)):  # `up` levels below the root
            lvl = length - 1 - up
            for who in ("q0", "one", "all", "two_differently"):
                p = stwo_prod.copy()
                paths = _paths_of(p, kind)
                if who == "q0":
                    paths[0][lvl, 7] ^= 0x10
                elif who == "one":
                    paths[int(rng.integers(1, c.n_queries))][lvl, 31] ^= 1
                elif who == "all":  # consistent wrong bytes: one shared mismatch with the root
                    for q in range(c.n_queries):
                        paths[q][lvl, 0] ^= 0x80
                else:
                    a, b = rng.choice(c.n_queries, size=2, replace=False)
                    paths[int(a)][lvl, 3] ^= 2
                    paths[int(b)][lvl, 3] ^= 4
                batch.append(p)
                notes.append("kind %d level -%d %s" % (kind, up, who))
    p = stwo_prod.copy()  # two queries made identical (values and paths): pure followers
    for arr in (p.trace_vals, p.cp_vals):
        arr[9] = arr[2]
    p.trace_paths[9], p.cp_paths[9] = p.trace_paths[2].copy(), p.cp_paths[2].copy()
    batch.append(p); notes.append("query 9 := query 2")
    top = verifier.Verifier(0)  # the byte compares in the top kernel (the path of query counts that do not divide 64)
    top.stwo_flags = verifier.FLAG_TOP_CHECKS
    for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
        want = O.stwo_verify_batch(batch, mode)
        for v in (ver, top):  # ver: Q = 16 divides 64, so the merkle kernel makes them
            got = v.verify_stwo(batch, mode, cfg=c)
            bad = [(notes[i], hex(got[i]), hex(want[i])) for i in range(len(batch)) if got[i] != want[i]]
            assert not bad, (v is top, bad[:8])
    top.close()
    assert want[0] != 0 or mode == verifier.MODE_FIXTURE


@pytest.mark.parametrize("name", ["prod", "small", "stwo_trace16.npz", "stwo_trace16_blake2s.npz", "stwo_wide256.npz",
                                  "stwo_wide256_blake2s.npz"])
def test_pair_memoisation_equals_full_hashing(stwo_small, stwo_prod, name):
    """SS_FLAG_NO_DEDUP (every path hashed in full, as the reference does) and the default give the
    same status words, and both equal the oracle's, on valid proofs and seeded mutants."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz
    base = {"prod": stwo_prod, "small": stwo_small}.get(name) or _load_npz(name)[0]
    rng = np.random.default_rng(SEED + 32)
    batch = [base] * 3 + [fz.mutate_stwo(base, rng) for _ in range(150)]
    want = O.stwo_verify_batch(batch)
    results = []
    for flags in (0, verifier.FLAG_NO_DEDUP, verifier.FLAG_TOP_CHECKS):
        v = verifier.Verifier(0)
        v.stwo_flags = flags
        results.append(v.verify_stwo(batch, cfg=base.cfg))
        v.close()
    assert all(r.tolist() == want.tolist() for r in results)
    assert (want[:3] == 0).all() and (want != 0).sum() > 50


def test_pair_memoisation_with_guided_group_sizes(ver, stwo_prod):
    """The top kernel hands its last groups out in quarters (16, 12, 8, 4 proofs at Q = 16).  A batch
    big enough to run through all of those sizes (20 000 proofs of the reference shape: 1 250 groups
    for at most 768 resident blocks), made of valid proofs and mutants in random order plus a ragged
    tail, must get the oracle's status words, and the same ones as with every path hashed in full."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz
    rng = np.random.default_rng(SEED + 33)
    pool = [stwo_prod] + [fz.mutate_stwo(stwo_prod, rng) for _ in range(399)]
    for n in (20000, 16387):
        pick = rng.integers(0, len(pool), size=n)
        batch = [pool[i] for i in pick]
        want_pool = O.stwo_verify_batch(pool)
        want = want_pool[pick]
        got = ver.verify_stwo(batch, cfg=stwo_prod.cfg)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (n, int(bad[0]), hex(got[bad[0]]), hex(want[bad[0]]))
    plain = verifier.Verifier(0)
    plain.stwo_flags = verifier.FLAG_NO_DEDUP
    assert plain.verify_stwo(batch, cfg=stwo_prod.cfg).tolist() == want.tolist()
    plain.close()
    assert (want == 0).sum() > 20 and (want != 0).sum() > n // 2


# --------------------------------------------------------- text ingestion (native readers)
def test_text_ingestion_gives_the_record_path_verdicts(ver, tmp_path, stwo_small, stwo_prod):
    """ss_stwo_verify_texts / _files: proof.json and proof.wit texts parsed by the library, verified
    against the expected config.  Verdicts equal the Python reader + record path + oracle; other
    shapes are status 1, unreadable texts status 2, and nothing of that depends on the batch mix."""
    import json
    import os
    import stark_symphony_amd as ss
    from stark_symphony_amd import binding
    from conftest import GOLDEN
    rng = np.random.default_rng(SEED + 41)
    bad = [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(6)]
    short = stwo_prod.copy(); short.fri_paths[1][3] = short.fri_paths[1][3][:-2]   # ragged: .wit only
    proofs = [stwo_prod] + bad
    texts = [json.dumps(ss.stwo_to_json(p)).encode() for p in proofs] + [ss.stwo_to_wit(p).encode() for p in proofs + [short]]
    want = O.stwo_verify_batch(proofs).tolist() + O.stwo_verify_batch(proofs + [short]).tolist()
    texts += [json.dumps(ss.stwo_to_json(stwo_small)).encode(), ss.stwo_to_wit(stwo_small).encode(),
              b"{\"config\": 1}", b"", texts[0][:5000]]
    want += [1, 1, 2, 2, 2]
    status, stats = ver.verify_stwo_texts(stwo_prod.cfg, texts)
    assert status.tolist() == want
    assert stats["threads"] >= 1 and stats["parse_s"] > 0 and stats["total_s"] >= stats["parse_s"]
    assert stats["text_bytes"] == sum(len(t) for t in texts)
    paths = []
    for i, t in enumerate(texts):
        paths.append(str(tmp_path / ("p%d.txt" % i)))
        open(paths[-1], "wb").write(t)
    paths.append(str(tmp_path / "absent.json"))
    status, _ = ver.verify_stwo_files(stwo_prod.cfg, paths)
    assert status.tolist() == want + [2]
    # forcing the wrong reader on a text is a malformed witness, not a crash
    assert ver.verify_stwo_texts(stwo_prod.cfg, texts[:1], fmt=binding.TEXT_WIT)[0].tolist() == [2]
    # literal mode and the testing profile through the same entry point
    st, _ = ver.verify_stwo_texts(stwo_prod.cfg, texts[:2], verifier.MODE_LITERAL)
    assert st.tolist() == O.stwo_verify_batch(proofs[:2], O.MODE_LITERAL).tolist()
    st, _ = ver.verify_stwo_files(stwo_small.cfg, [os.path.join(GOLDEN, "stwo_proof_test.json"),
                                                   os.path.join(GOLDEN, "formats", "stwo_proof_test.wit"),
                                                   os.path.join(GOLDEN, "stwo_proof.json")])
    assert st.tolist() == [0, 0, 1]


def test_stark101_text_ingestion(ver, tmp_path, s101_proof):
    import json
    import os
    import stark_symphony_amd as ss
    from conftest import GOLDEN
    rng = np.random.default_rng(SEED + 42)
    proofs = [s101_proof] + [formats.stark101_corrupt(s101_proof, rng)[0] for _ in range(9)]
    p = s101_proof.copy(); p.layers = p.layers[:6]; proofs.append(p)                       # ragged shapes
    p = s101_proof.copy(); p.evals[2].path = p.evals[2].path[:5]; proofs.append(p)
    texts = [json.dumps(ss.stark101_to_json(q)).encode() for q in proofs] + [ss.stark101_to_wit(q).encode() for q in proofs]
    want = O.s101_verify_batch(proofs).tolist() * 2
    status, stats = ver.verify_stark101_texts(texts + [b"{}", b"nonsense"])
    assert status.tolist() == want + [2, 2] and stats["threads"] >= 1
    st, _ = ver.verify_stark101_files([os.path.join(GOLDEN, "stark101_proof.json"),
                                       os.path.join(GOLDEN, "formats", "stark101_proof.wit"), str(tmp_path / "none")])
    assert st.tolist() == [0, 0, 2]


def test_simfony_run_shim_in_c(tmp_path):
    """examples/ss_run.c on the GPU: the reference's `make run` with the binary swapped.  Exit 0 / 1 and the
    stderr prefix of simfony-cli/src/main.rs:205-206,254-257, no Python in the verifying process."""
    import json
    import os
    import subprocess
    from conftest import GOLDEN
    from test_host import _build_c_example
    exe = _build_c_example(tmp_path, "ss_run")
    F = os.path.join(GOLDEN, "formats")

    def run(*args):
        r = subprocess.run([exe, "run", *args], capture_output=True, text=True, timeout=300)
        r.stderr = "".join(l for l in r.stderr.splitlines(True) if "amdgpu.ids" not in l)
        return r
    # the family comes from trusted input only: --family, the program's path, or the program's text
    p101 = tmp_path / "main.out.simf"   # what `mcpp -P src/main.simf` leaves in target/ (stark101/Makefile:4-5)
    p101.write_text("fn main() {\n    let root: u256 = witness::P_MT_ROOT;\n}\n")
    pstwo = tmp_path / "other" / "main.simf"
    pstwo.parent.mkdir()
    pstwo.write_text("fn main() {\n    let c: Commitments = witness::COMMITMENTS;\n}\n")
    r = run(str(p101), "--witness", os.path.join(F, "stark101_proof.wit"))                    # family from the program text
    assert r.returncode == 0 and "ACCEPT" in r.stdout and r.stderr == ""
    r = run("stwo-verifier/main.simf", "--witness", os.path.join(F, "stwo_proof.wit"))         # from the path
    assert r.returncode == 0 and "ACCEPT" in r.stdout
    # a witness never selects the statement: a valid stark101 witness handed to the stwo program is a typing
    # failure (exit 1), and a program of unknown family is an error (exit 2), not a verdict
    r = run("stwo-verifier/main.simf", "--witness", os.path.join(F, "stark101_proof.wit"))
    assert r.returncode == 1 and "malformed witness" in r.stderr and "ACCEPT" not in r.stdout
    r = run("stark101/main.simf", "--witness", os.path.join(F, "stwo_proof.wit"))
    assert r.returncode == 1 and "ACCEPT" not in r.stdout
    r = run("main.simf", "--witness", os.path.join(F, "stark101_proof.wit"))
    assert r.returncode == 2 and "--family" in r.stderr and "ACCEPT" not in r.stdout
    r = run(str(pstwo), "--config", "testing", "--witness", os.path.join(F, "stwo_proof_test.wit"))
    assert r.returncode == 0
    r = run(str(pstwo), "--witness", os.path.join(F, "stwo_proof_test.wit"))                  # production program, test witness
    assert r.returncode == 1 and r.stderr.startswith("Error: Failed to run program")
    r = run("main.simf", "--family", "stwo", "--mode", "literal", "--witness", os.path.join(F, "stwo_proof.wit"))
    assert r.returncode == 1 and "0x07000001" in r.stderr
    wit = json.load(open(os.path.join(F, "stwo_proof.wit")))
    v = wit["POW_NONCE"]["value"]
    wit["POW_NONCE"]["value"] = str(int(v) + 1)
    bad = tmp_path / "bad.wit"
    bad.write_text(json.dumps(wit))
    r = run(str(pstwo), "--witness", os.path.join(F, "stwo_proof.wit"), "--witness", str(bad))
    assert r.returncode == 1 and r.stdout.count("ACCEPT") == 1 and "assertion failed" in r.stderr
    assert run(str(pstwo), "--witness", str(tmp_path / "absent.wit")).returncode == 1


@pytest.mark.streams
def test_independent_streams_match_single_stream(ver, s101_proof, stwo_prod):
    """verifier.IndependentStreams (whole passes on their own streams, what bench.py uses for small
    batches): every slot ends with the oracle's status words and accept count, also when a slot is
    reused many times."""
    rng = np.random.default_rng(SEED + 51)
    d101 = [s101_proof] + [formats.stark101_corrupt(s101_proof, rng)[0] for _ in range(5)]
    dstw = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(5)]
    for batch, want_d in ((ver.stark101_batch([d101[i % 6] for i in range(300)]), O.s101_verify_batch(d101)),
                          (ver.stwo_batch([dstw[i % 6] for i in range(150)]), O.stwo_verify_batch(dstw))):
        want = [int(want_d[i % 6]) for i in range(batch.n)]
        slots = [batch.sibling() for _ in range(5)]
        ind = verifier.IndependentStreams(slots)
        for s in slots:
            s.status_dev.fill_(0x55)
        # the fills ran on the current stream and the passes run on non-blocking side streams: without this join a fill
        # may land after its slot's last pass and replace the verifier's status words (GPUTEST_r05, profiles/r06_null_stream_order.txt)
        ind.wait_current()
        used = [ind.submit() for _ in range(23)]
        ind.synchronize()
        assert used == [i % 5 for i in range(23)]
        for s in slots:
            assert s.status().tolist() == want and s.accepted() == sum(1 for w in want if w == 0)


def test_empty_batches_have_empty_answers(ver, stwo_prod):
    """An empty list of witnesses is not an error at the Python boundary (the C ABI itself rejects
    n == 0, include/ss_verify.h): every entry point returns an empty status array."""
    assert ver.verify_stark101([]).shape == (0,)
    assert ver.verify_stwo([], cfg=stwo_prod.cfg).shape == (0,)
    assert ver.verify_stwo_records(stwo_prod.cfg, []).shape == (0,)
    st, stats = ver.verify_stwo_texts(stwo_prod.cfg, [])
    assert st.shape == (0,) and stats["text_bytes"] == 0
    st, stats = ver.verify_stark101_files([])
    assert st.shape == (0,)


@pytest.mark.streams
def test_two_contexts_in_one_process(ver, stwo_prod):
    """the device entry points bind the context's device themselves (csrc/ss_ctx.h DeviceGuard), so a
    process may hold several contexts -- the C / Rust caller of INTEGRATION.md holds one per GPU.  On a one-GPU box:
    two contexts on device 0, used alternately and from two threads at once, each with its own batches and streams,
    give the oracle's status words; the caller's current device is what it was."""
    import threading
    import torch
    rng = np.random.default_rng(SEED + 300)
    proofs = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(23)]
    want = O.stwo_verify_batch(proofs, O.MODE_FIXTURE)
    other = verifier.Verifier(0)
    before = torch.cuda.current_device()
    got = {}

    def work(name, v, reps):
        for r in range(reps):
            got[(name, r)] = v.verify_stwo(proofs, verifier.MODE_FIXTURE, cfg=stwo_prod.cfg)
    work("a", ver, 1)
    work("b", other, 1)
    ts = [threading.Thread(target=work, args=(n, v, 4)) for n, v in (("ta", ver), ("tb", other))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(got) == 10 and all(np.array_equal(g, want) for g in got.values())
    assert torch.cuda.current_device() == before
    other.close()


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs")
def test_context_on_a_device_that_is_not_current(stwo_prod):
    """A context on device 1 while device 0 is current: the library launches on device 1 (records path: its own
    streams and buffers) and leaves device 0 current."""
    import torch
    torch.cuda.set_device(0)
    v1 = verifier.Verifier.__new__(verifier.Verifier)
    import ctypes as C
    ctx = C.c_void_p()
    verifier.B.check(verifier.B.lib().ss_ctx_create(1, C.byref(ctx)))
    v1.ctx, v1.index, v1.device, v1.timing, v1.stwo_flags = ctx, 1, torch.device("cuda", 1), False, 0
    rng = np.random.default_rng(SEED + 301)
    proofs = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(7)]
    want = O.stwo_verify_batch(proofs, O.MODE_FIXTURE)
    assert torch.cuda.current_device() == 0
    got = v1.verify_stwo_records(stwo_prod.cfg, [verifier.stwo_record(p) for p in proofs])
    assert np.array_equal(got, want) and torch.cuda.current_device() == 0
    v1.close()


def test_one_descriptor_entry_point_equals_the_named_ones(ver, tmp_path, s101_proof, stwo_prod):
    """ss_verify_inputs (ABI 2.4, include/ss_verify.h section 4): every (form, source) pair through the ONE descriptor call
    gives the status words of the oracle -- the same the named entry point of that pair gives (ss_verify_forms.h) --, and
    combinations that do not exist are SS_ERR_ARG, not a verdict."""
    cfg = stwo_prod.cfg
    rng = np.random.default_rng(SEED + 400)
    proofs = [stwo_prod] + [formats.stwo_corrupt(stwo_prod, rng)[0] for _ in range(6)]
    want = O.stwo_verify_batch(proofs, O.MODE_FIXTURE).tolist()
    recs = [verifier.stwo_record(p) for p in proofs]
    # per-query records: host pointers, one pinned buffer
    assert ver.verify_inputs("records", "host", recs, cfg=cfg)[0].tolist() == want == ver.verify_stwo_records(cfg, recs).tolist()
    pin = ver.pinned_buffer(sum(r.size for r in recs))
    pin[:] = np.concatenate(recs)
    assert ver.verify_inputs("records", "pinned", cfg=cfg, blob=pin)[0].tolist() == want
    # shared and minimal records of the proofs that have such a form
    qs = formats.stwo_queries(stwo_prod)
    for form, make in (("shared_records", lambda p: verifier.stwo_shared_record(p, qs)),
                       ("minimal_records", lambda p: verifier.stwo_minimise_record(cfg, verifier.stwo_record(p), qs))):
        have = []
        for i, p in enumerate(proofs):
            try:
                have.append((i, make(p)))
            except ValueError:
                pass
        assert have and have[0][0] == 0
        got, _ = ver.verify_inputs(form, "host", [r for _, r in have], cfg=cfg)
        if form == "shared_records":  # expansion is exact: the per-query verdict
            assert got.tolist() == [want[i] for i, _ in have]
            assert got.tolist() == ver.verify_stwo_shared_records(cfg, [r for _, r in have]).tolist()
        else:  # a minimal record M verifies as R(M), where an omitted value is the computed one: the oracle's walk says which assert
            assert got.tolist() == [O.stwo_verify_minimal(cfg, r, O.MODE_FIXTURE) for _, r in have]
            assert got.tolist() == ver.verify_stwo_minimal_records(cfg, [r for _, r in have]).tolist()
        offs = np.zeros(len(have) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([r.size for _, r in have])
        pin = ver.pinned_buffer(int(offs[-1]))
        pin[:] = np.concatenate([r for _, r in have])
        assert ver.verify_inputs(form, "pinned", cfg=cfg, blob=pin, offsets=offs)[0].tolist() == got.tolist()
    # texts: host, pinned, files; a text that is no witness gets the stage-0 verdict on every route
    texts = [ss.stwo_to_wit(p).encode() if i % 2 else json.dumps(ss.stwo_to_json(p), separators=(",", ":")).encode()
             for i, p in enumerate(proofs)] + [b"not a witness"]
    got, stats = ver.verify_inputs("text", "host", texts, cfg=cfg)
    assert got.tolist() == want + [2] and stats["host_parsed"] == 1 and stats["text_bytes"] == sum(len(t) for t in texts)
    blob, boffs, blens = ver.pinned_text_blob(texts)
    assert ver.verify_inputs("text", "pinned", cfg=cfg, blob=blob, offsets=boffs, lengths=blens)[0].tolist() == want + [2]
    paths = []
    for i, t in enumerate(texts):
        f = tmp_path / ("w%d.txt" % i)
        f.write_bytes(t)
        paths.append(str(f))
    assert ver.verify_inputs("text", "files", paths + [str(tmp_path / "absent")], cfg=cfg)[0].tolist() == want + [2, 2]
    # stark101: records of their shape, texts
    d101 = [s101_proof] + [formats.stark101_corrupt(s101_proof, rng)[0] for _ in range(4)]
    want101 = O.s101_verify_batch(d101).tolist()
    ml, pm = verifier.s101_shape_of(d101)
    got, _ = ver.verify_inputs("records", "host", [verifier.s101_record(p, ml, pm) for p in d101], shape=(ml, pm))
    assert got.tolist() == want101 == ver.verify_stark101(d101).tolist()
    t101 = [json.dumps(ss.stark101_to_json(p)).encode() for p in d101]
    assert ver.verify_inputs("text", "host", t101)[0].tolist() == want101
    # what does not exist is an argument error
    for bad in (lambda: ver.verify_inputs("shared_records", "host", recs),                    # stark101 has no shared form
                lambda: ver.verify_inputs("records", "files", paths, cfg=cfg)):               # files hold text
        with pytest.raises(binding.SsError) as e:
            bad()
        assert e.value.code == binding.SS_ERR_ARG


def test_verdicts_do_not_depend_on_the_hardware_queue_count():
    """GPU_MAX_HW_QUEUES decides which streams share a hardware queue, i.e. what overlaps -- never a status word: the
    multi-stream tests of this file (pipeline, independent streams, graph replay, two contexts / two threads) in a child
    process that runs with the runtime's own default of 4 queues (the binding asks for 24: binding.process_defaults,
    profiles/r06_hw_queues_sweep.txt), each three times."""
    env = {k: v for k, v in os.environ.items() if k != "SS_KEEP_ENV"}
    env["GPU_MAX_HW_QUEUES"] = "4"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu and streams", "--repeat-streams", "3",
                        "-x", "-q", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
