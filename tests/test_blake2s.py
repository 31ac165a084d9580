"""The Blake2s variant (BASELINE.json "Blake2s Merkle", configs 3-5).

The reference has no Blake2s, so parity for this variant is UNPINNED: the hash
is checked against RFC 7693 / hashlib, and prover, oracle and GPU kernels must agree with each
other on the same protocol (the reference's byte strings, Blake2s-256 instead of SHA-256).
"""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import stwo_prover  # noqa: E402

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import formats  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEED = 0xB1A4E25


def test_blake2s_rfc7693_vector():
    # RFC 7693 appendix B: BLAKE2s-256("abc")
    assert O.blake2s(b"abc").hex() == ("508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982")
    assert O.blake2s(b"") == hashlib.blake2s(b"").digest()


def test_blake2s_matches_hashlib_on_many_lengths():
    rng = np.random.default_rng(SEED)
    for n in list(range(0, 200, 7)) + [64, 128, 192, 1000, 4384]:
        m = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert O.blake2s(m) == hashlib.blake2s(m).digest()


def _proofs():
    out = []
    for kw in (dict(n_cols=4, trace_log=5, log_blowup=2, n_queries=3, pow_bits=5, seed=0),
               dict(n_cols=8, trace_log=7, log_blowup=3, n_queries=16, pow_bits=4, seed=3),
               dict(n_cols=20, trace_log=4, log_blowup=1, n_queries=5, pow_bits=2, seed=1)):
        out.append(ss.stwo_from_json(stwo_prover.prove(hash="blake2s", **kw)))
    return out


def test_blake2s_prover_and_oracle_agree():
    for p in _proofs():
        assert p.cfg.hash == "blake2s"
        assert O.stwo_verify(p, O.MODE_FIXTURE) == 0
        as_sha = ss.stwo_from_json(ss.stwo_to_json(p), hash="sha256")
        assert O.stwo_verify(as_sha, O.MODE_FIXTURE) != 0  # the hash family matters
        bad = p.copy()
        bad.fri_witness[1, 0, 2] ^= 4
        assert O.stwo_verify(bad, O.MODE_FIXTURE) != 0
    # work formula: a 64-byte node is one Blake2s compression, two SHA-256 ones
    assert ss.StwoConfig(4, 20, 24, 16, 19, 5, "blake2s").compressions == 6136


@pytest.mark.gpu
def test_blake2s_gpu_parity():
    from stark_symphony_amd import verifier
    ver = verifier.Verifier(0)
    rng = np.random.default_rng(SEED)
    for base in _proofs():
        batch = [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(79)]
        for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
            got = ver.verify_stwo(batch, mode, cfg=base.cfg)
            want = O.stwo_verify_batch(batch, mode)
            assert got.tolist() == want.tolist()
        assert ver.verify_stwo([base], cfg=base.cfg).tolist() == [0]
    # the same bytes under the other hash family are rejected by the GPU too
    sha = ss.stwo_from_json(ss.stwo_to_json(_proofs()[1]), hash="sha256")
    assert ver.verify_stwo([sha], cfg=sha.cfg).tolist() == [O.stwo_verify(sha, O.MODE_FIXTURE)] != [0]
    # ... and when the verifier expects Blake2s, a proof that declares SHA-256 never reaches the GPU
    assert ver.verify_stwo([sha], cfg=_proofs()[1].cfg).tolist() == [verifier.STATUS_CONFIG_MISMATCH]
