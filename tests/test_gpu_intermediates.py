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


def _layout(cfg, n):
    """Workspace offsets of csrc/ss_layout.h (words)."""
    N, Q, K = cfg.n_cols, cfg.n_queries, cfg.n_layers
    np_ = (n + 63) // 64 * 64
    nip = (n * Q + 63) // 64 * 64
    c = {"queries": 0}
    at = Q
    for name, words in (("p", 8), ("p2", 8), ("b01", 4), ("b02", 4), ("a1", 4), ("c1", 4), ("a2", 4), ("c2", 4),
                        ("m1", 4), ("fold", 4 * (K + 1))):
        c[name] = at
        at += words
    ctx_words = at
    ws_alpha = ctx_words * np_
    ws_leaf = ws_alpha + (N + 16) * 4 * np_
    return c, np_, nip, ws_alpha, ws_leaf


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
    ws = b.ws.cpu().numpy().view(np.uint32)
    c, np_, nip, ws_alpha, ws_leaf = _layout(cfg, n)
    ctx = lambda w, p: int(ws[w * np_ + p])  # noqa: E731
    for pi, proof in enumerate(proofs):
        st, tr = O.stwo_verify(proof, mode, trace=True)
        assert int(status[pi]) == st
        assert [ctx(c["queries"] + q, pi) for q in range(Q)] == list(tr.queries[:Q])
        got_p = [ctx(c["p"] + w, pi) for w in range(8)]
        assert got_p == list(tr.oods_point.x.t()) + list(tr.oods_point.y.t())
        for l in range(K + 1):
            assert [ctx(c["fold"] + 4 * l + w, pi) for w in range(4)] == list(tr.fold_alpha[l].t())
        alpha1 = [int(ws[ws_alpha + (pi * (cfg.n_cols + 16)) * 4 + w]) for w in range(4)]
        assert alpha1 == list(tr.deep_alpha.t())
        for q in range(Q):  # fri_answer: the evaluation the query kernel feeds into the first fold
            inst = pi * Q + q
            half = 4 if tr.queries[q] & 1 else 0
            ans = [int(ws[ws_leaf + (half + w) * nip + inst]) for w in range(4)]
            assert ans == list(tr.answers[q].t()), (pi, q, mode)
