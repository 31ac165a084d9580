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


def _ws_layout(b):
    import ctypes as C
    from stark_symphony_amd import binding
    lay = binding.StwoWsLayout()
    binding.check(binding.lib().ss_stwo_ws_layout_of(C.byref(b.cs), b.n, C.byref(lay)))
    return lay
