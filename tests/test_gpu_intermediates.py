"""Stage-level parity: the workspace the HEAD kernels leave behind (queries, OODS point, fold alphas,
DEEP answers per query) against the oracle's trace, in BOTH stwo modes.

In SS_MODE_LITERAL every honest proof is rejected at the first FRI decommitment, so
the status word alone cannot show whether the single-batch DEEP quotient of fri/answers.simf:97-130
is computed correctly on the GPU; the intermediate values can."""
import numpy as np
import pytest

from stark_symphony_amd import formats, verifier
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SEED = 0x5EED2025


@pytest.mark.parametrize("mode", [verifier.MODE_FIXTURE, verifier.MODE_LITERAL])
@pytest.mark.parametrize("which", ["small", "prod"])
def test_head_kernels_leave_the_oracles_intermediates(stwo_small, stwo_prod, mode, which):
    ver = verifier.Verifier(0)
    base = stwo_small if which == "small" else stwo_prod
    rng = np.random.default_rng(SEED + 21)
    proofs = [base]
    for _ in range(4):  # corruptions that keep the transcript well-formed but change every later value
        p = base.copy()
        arr = [p.oods_trace, p.oods_cp, p.trace_vals, p.cp_vals, p.fri_witness][int(rng.integers(5))].reshape(-1)
        arr[int(rng.integers(arr.size))] ^= np.uint32(1 << int(rng.integers(31)))
        proofs.append(p)
    cfg, n = base.cfg, len(proofs)
    Q, K = cfg.n_queries, cfg.n_layers
    b = ver.stwo_batch(proofs, mode)
    b.run()
    status = b.status()
    lay = _ws_layout(b)
    assert lay.total_words * 4 <= b.ws.numel() * 4 and lay.np % 64 == 0 and lay.nip >= n * Q
    for pi, proof in enumerate(proofs):
        st, tr = O.stwo_verify(proof, mode, trace=True)
        assert int(status[pi]) == st
        got = b.intermediates(pi)  # ss_stwo_read_intermediates
        assert got["queries"].tolist() == list(tr.queries[:Q])
        assert got["oods_point"].tolist() == list(tr.oods_point.x.t()) + list(tr.oods_point.y.t())
        assert got["fold_alphas"].tolist() == [list(tr.fold_alpha[l].t()) for l in range(K + 1)]
        assert got["deep_alpha"].tolist() == list(tr.deep_alpha.t())
        # fri_answer: the evaluation the query kernel feeds into the first fold
        assert got["fri_answers"].tolist() == [list(tr.answers[q].t()) for q in range(Q)], (pi, mode)
        # the exported layout addresses the same words (a caller that reads the workspace itself)
        ws = b.ws.cpu().numpy().view(np.uint32)
        assert [int(ws[lay.ctx + (lay.c_queries + q) * lay.np + pi]) for q in range(Q)] == list(tr.queries[:Q])


@pytest.mark.parametrize("name", ["prod", "stwo_trace16.npz", "stwo_wide256_blake2s.npz"])
def test_query_kernel_leaves_the_plan_of_the_pair_memoisation(stwo_prod, name):
    """csrc/ss_layout.h ws_plan: for every query and every depth 1..T the lowest query of the proof at the same
    position (query >> (L - d)) and the lowest one at the sibling position (0xff: none), restated here in numpy from
    the oracle's queries.  The merkle kernel's byte compares and the top kernel's hashing both run on this plan."""
    from conftest import GOLDEN
    import os
    from stark_symphony_amd import records
    base = stwo_prod if name == "prod" else records.load_stwo_npz(os.path.join(GOLDEN, name))[0]
    ver = verifier.Verifier(0)
    rng = np.random.default_rng(SEED + 22)
    proofs = [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(5)]
    cfg = base.cfg
    Q, L = cfg.n_queries, cfg.lde_log
    b = ver.stwo_batch(proofs, verifier.MODE_FIXTURE)
    b.run(phases=verifier.PHASE_HEAD)
    lay = _ws_layout(b)
    assert lay.has_plan == 1 and 1 <= lay.top_levels <= 8
    ws = b.ws.cpu().numpy().view(np.uint32)
    for pi, proof in enumerate(proofs):
        queries = b.intermediates(pi)["queries"].tolist()
        assert queries == list(O.stwo_verify(proof, verifier.MODE_FIXTURE, trace=True)[1].queries[:Q])
        for q in range(Q):
            w = ws[lay.plan + (pi * Q + q) * 4: lay.plan + (pi * Q + q) * 4 + 4]
            lead = int(w[0]) | int(w[1]) << 32
            sibl = int(w[2]) | int(w[3]) << 32
            for d in range(1, lay.top_levels + 1):
                pos = [x >> (L - d) for x in queries]
                want_lead = pos.index(pos[q])
                want_sibl = pos.index(pos[q] ^ 1) if (pos[q] ^ 1) in pos else 0xff
                assert (lead >> (8 * (d - 1))) & 0xff == want_lead, (pi, q, d)
                assert (sibl >> (8 * (d - 1))) & 0xff == want_sibl, (pi, q, d)
    # query counts that do not divide 64, and SS_FLAG_TOP_CHECKS, have no plan
    ver.stwo_flags = verifier.FLAG_TOP_CHECKS
    assert _ws_layout(ver.stwo_batch(proofs[:1], verifier.MODE_FIXTURE)).has_plan == 0


def _ws_layout(b):
    import ctypes as C
    from stark_symphony_amd import binding
    lay = binding.StwoWsLayout()
    binding.check(binding.lib().ss_stwo_ws_layout_of(C.byref(b.cs), b.n, C.byref(lay)))
    return lay


def test_last_layer_asserts_in_isolation():
    """in LITERAL mode `log_size_ex == 0` (fri/verify.simf:127, code 8) and `folded_query == 0`
    (fri/layers.simf:75, code 9.0) can never be the FIRST failing assert of any batch a prover can make (stage 7 fails
    earlier), so end-to-end status words compare them only vacuously.  ss_selftest op 6 runs the query kernel's own
    last-layer block (csrc/ss_stwo_checks.h) on chosen (mode, lde_log, n_layers, query, folded position, folded value,
    last layer); the oracle exports the same lines (so_stwo_fri_tail).  4 000 random cases, every outcome in both modes,
    including the 8-bit wrap of log_size_ex."""
    import ctypes as C
    ver = verifier.Verifier(0)
    rng = np.random.default_rng(SEED + 88)
    n = 4000
    cases = np.zeros((n, 13), dtype=np.uint32)
    cases[:, 0] = rng.integers(0, 2, n)                                  # mode
    cases[:, 1] = rng.integers(1, 32, n)                                 # lde_log
    cases[:, 2] = rng.integers(0, 31, n)                                 # n_layers
    cases[:, 3] = rng.integers(0, 64, n)                                 # query number
    cases[:, 4] = np.where(rng.integers(0, 2, n) == 0, 0, rng.integers(0, 1 << 31, n))   # folded position
    cases[:, 5:9] = rng.integers(0, (1 << 31) - 1, (n, 4))
    same = rng.integers(0, 3, n)
    cases[:, 9:13] = cases[:, 5:9]
    for i in np.nonzero(same == 0)[0]:
        cases[i, 9 + int(rng.integers(0, 4))] ^= np.uint32(1 << int(rng.integers(0, 31)))
    exact = rng.integers(0, 4, n) == 0
    cases[exact, 2] = (cases[exact, 1].astype(np.int64) - 1) % 256 % 31   # around lde_log == n_layers + 1
    cases[:40, 1], cases[:40, 2] = 256 + np.arange(40) % 3, 255 + np.arange(40) % 3   # log_size_ex wraps to 0 in 8 bits
    got = ver.selftest(6, cases).reshape(-1)
    L = O.lib()
    want = [L.so_stwo_fri_tail(int(c[0]), int(c[1]), int(c[2]), int(c[3]), int(c[4]), O.qm(c[5:9].tolist()), O.qm(c[9:13].tolist()))
            for c in cases]
    assert got.tolist() == want
    kinds = {(int(c[0]), w >> 24, w & 15) for c, w in zip(cases, want)}
    assert {(0, 8, 0), (0, 9, 0), (0, 9, 1), (0, 0, 0), (1, 9, 1), (1, 0, 0)} <= kinds
    assert not any(m == 1 and (stage, sub) in ((8, 0), (9, 0)) for m, stage, sub in kinds)   # FIXTURE mode never raises D2 / D3


def test_stark101_intermediates_against_the_reference_prover_and_the_oracle(s101_proof):
    """ss_s101_read_intermediates.  Two references: (1) tests/golden/stark101_transcript.json -- the 59 channel messages
    of the reference's OWN prover (stark101/scripts/fibsquare/prover.py, imported by make_stark101_golden.py): the value
    the device carries into FRI layer i is the prover's cp_i message, the final one its last-layer constant, and cp is
    cp_0 -- what the reference's Python verifier recomputes as `rhs` (prover_test.py:32-104); the query index is the one
    of SURVEY.md Appendix C; (2) the oracle's so_s101_trace on the proof and on corruptions that keep the transcript
    well-formed (stark101/src/verifier.simf:24-42)."""
    import json
    import os
    from conftest import GOLDEN
    ver = verifier.Verifier(0)
    msgs = json.load(open(os.path.join(GOLDEN, "stark101_transcript.json")))
    # the message list: p_mt_root, 10 layer roots, the last value, then (value, path) pairs: f(x), f(gx), f(ggx),
    # cp_i and its sibling per layer, and the last value once more
    felts = [m["felt"] for m in msgs if "felt" in m]
    n_layers = len(s101_proof.layers)
    last, evals, cps = felts[0], felts[1:4], felts[4:4 + 2 * n_layers]
    assert felts[-1] == last == s101_proof.last and len(felts) == 4 + 2 * n_layers + 1
    rng = np.random.default_rng(SEED + 31)
    proofs = [s101_proof]
    for _ in range(6):
        p = s101_proof.copy()
        k = int(rng.integers(4))
        if k == 0:
            p.evals[int(rng.integers(3))].ev ^= 1 << int(rng.integers(31))
        elif k == 1:
            p.layers[int(rng.integers(n_layers))].cpb.ev ^= 1 << int(rng.integers(31))
        elif k == 2:
            p.last ^= 1 << int(rng.integers(31))
        else:
            p.root = formats._flip_bytes(p.root, int(rng.integers(256)))
        proofs.append(p)
    b = ver.stark101_batch(proofs)
    b.run()
    status = b.status()
    for pi, proof in enumerate(proofs):
        st, tr = O.s101_verify(proof, trace=True)
        assert int(status[pi]) == st
        got = b.intermediates(pi)
        assert got["alphas"].tolist() == list(tr.alpha) and got["idx"] == tr.idx and got["x"] == tr.x
        assert bytes(np.asarray(got["state"], dtype=">u4").tobytes()) == bytes(tr.state_after_commit)
        if (st >> 8) != 3:  # (a division abort leaves the later values unspecified: code 3 is ordered before them)
            assert got["cp"] == tr.cp
            upto = n_layers + 1
            for i in range(n_layers):
                if st and (st >> 8) == 4 and (st & 0xFF) % 4 == 3 and (st & 0xFF) // 4 <= i:
                    upto = i + 1  # a fold division aborted in layer i: what follows is ordered behind that code
                    break
            assert got["folds"][:upto].tolist() == list(tr.fold[:upto]), pi
    # the honest proof against the reference prover's own messages
    got = b.intermediates(0)
    assert status[0] == 0 and got["idx"] == 6160
    assert [s101_proof.evals[k].ev for k in range(3)] == evals
    assert got["cp"] == cps[0] == got["folds"][0]
    assert got["folds"][:n_layers].tolist() == cps[0::2]
    assert int(got["folds"][n_layers]) == last
