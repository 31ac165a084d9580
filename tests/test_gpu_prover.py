"""GPU prover (stark-symphony_amd/prover.py + include/ss_prover.h) against the numpy prover and
the reference's own proofs: the output must be identical, byte for byte."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import stwo_prover  # noqa: E402

import stark_symphony_amd as ss  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gp():
    from stark_symphony_amd import prover, verifier
    return prover.GpuProver(verifier.Verifier(0))


@pytest.mark.parametrize("name,kw", [
    ("stwo_proof_test.json", dict(trace_log=3, log_blowup=1, n_queries=1)),
    ("stwo_proof.json", dict(trace_log=9, log_blowup=4, n_queries=16)),
])
def test_gpu_prover_reproduces_reference_fixtures(gp, name, kw):
    want = json.load(open(os.path.join(GOLDEN, name)))
    got = gp.prove(n_cols=4, pow_bits=5, **kw)
    for key in want:
        assert got[key] == want[key], key


@pytest.mark.parametrize("kw", [
    dict(n_cols=4, trace_log=5, log_blowup=2, n_queries=3, pow_bits=5, seed=0, hash="sha256"),
    dict(n_cols=8, trace_log=4, log_blowup=1, n_queries=9, pow_bits=3, seed=7, hash="sha256"),
    dict(n_cols=32, trace_log=6, log_blowup=3, n_queries=5, pow_bits=8, seed=1, hash="blake2s"),
    dict(n_cols=3, trace_log=2, log_blowup=1, n_queries=2, pow_bits=0, seed=2, hash="sha256"),
    dict(n_cols=3, trace_log=1, log_blowup=2, n_queries=3, pow_bits=1, seed=4, hash="sha256"),  # two rows: constant partitions
    dict(n_cols=4, trace_log=11, log_blowup=2, n_queries=4, pow_bits=2, seed=6, hash="sha256"),  # LDE 2^13: first size of the LDS passes
    dict(n_cols=5, trace_log=12, log_blowup=2, n_queries=11, pow_bits=10, seed=5, hash="blake2s"),
])
def test_gpu_prover_equals_numpy_prover(gp, kw):
    got = gp.prove(**kw)
    want = stwo_prover.prove(**kw)
    assert got == want
    assert O.stwo_verify(ss.stwo_from_json(got), O.MODE_FIXTURE) == 0


def test_gpu_prove_then_gpu_verify_roundtrip(gp):
    """prove() -> verify() entirely on the device, 2^14 rows."""
    proofs = [ss.stwo_from_json(gp.prove(n_cols=4, trace_log=14, log_blowup=4, n_queries=16, seed=s))
              for s in (0, 1, 2)]
    assert len({bytes(p.roots[1]) for p in proofs}) == 3
    status = gp.ver.verify_stwo(proofs, cfg=[p.cfg for p in proofs])
    assert status.tolist() == [0, 0, 0]
    assert gp.timings["total"] > 0


# ------------------------------------------------------------------ stark101 prover (8f row 2)
@pytest.fixture(scope="module")
def gp101(gp):
    from stark_symphony_amd import prover101
    return prover101.Stark101GpuProver(gp.ver)


def test_stark101_gpu_prover_reproduces_the_reference_proof(gp101):
    """tests/golden/stark101_proof.json was written by the reference's own prove()."""
    want = json.load(open(os.path.join(GOLDEN, "stark101_proof.json")))
    got = gp101.prove()
    assert got == want
    assert gp101.claim == 2338775057 and gp101.n_fri_layers == 10
    proof = ss.stark101_from_json(got)
    assert gp101.ver.verify_stark101([proof]).tolist() == [0]
    assert O.s101_verify(proof) == 0


def test_stark101_gpu_prover_other_seed_is_rejected_at_the_boundary_constraint(gp101):
    """The boundary value is hard-coded in the reference (prover.py:44, air.simf:63): the proof of any
    other seed is internally consistent up to the composition-polynomial check and fails there, with
    the same status word on the GPU and in the oracle."""
    from stark_symphony_amd import prover101
    tried = 0
    for seed in (7, 12345, 99, 2024, 31337):
        try:
            got = gp101.prove(seed)
        except IndexError:
            continue  # query too close to the end of the coset for prover.py:145-146
        tried += 1
        assert gp101.claim == prover101.trace_reference(seed)[1022] != 2338775057
        proof = ss.stark101_from_json(got)
        status = int(gp101.ver.verify_stark101([proof])[0])
        assert status == O.s101_verify(proof) != 0
        assert status == 0x400, hex(status)  # layer 0: cp_0(x) != the AIR's value (fri.simf:77)
    assert tried >= 3


def test_random_shapes_prove_verify_and_match_the_oracle(gp):
    """Thirty random shapes (odd column counts, query counts that do not divide 64, blow-ups 2..16,
    both hashes): GPU prover -> GPU verifier accepts; with seeded corruptions mixed in, every status
    word equals the oracle's, in both modes."""
    import numpy as np
    from stark_symphony_amd import formats, verifier
    rng = np.random.default_rng(20261003)
    for _ in range(30):
        kw = dict(n_cols=int(rng.integers(3, 41)), trace_log=int(rng.integers(2, 11)),
                  log_blowup=int(rng.integers(1, 5)), n_queries=int(rng.integers(1, 21)),
                  pow_bits=int(rng.integers(0, 9)), seed=int(rng.integers(0, 1000)),
                  hash=("sha256", "blake2s")[int(rng.integers(2))])
        proof = ss.stwo_from_json(gp.prove(**kw))
        batch = [proof] + [formats.stwo_corrupt(proof, rng)[0] for _ in range(6)]
        for mode in (verifier.MODE_FIXTURE, verifier.MODE_LITERAL):
            got = gp.ver.verify_stwo(batch, mode, cfg=batch[0].cfg)
            want = O.stwo_verify_batch(batch, mode)
            assert got.tolist() == want.tolist(), (kw, mode)
        assert gp.ver.verify_stwo(batch, cfg=batch[0].cfg)[0] == 0, kw


def test_prove_many_is_byte_identical_to_one_at_a_time(gp):
    """prove_many: several proofs in flight on their own streams give, seed for seed, the proofs of prove_proof."""
    kw = dict(n_cols=4, trace_log=12, log_blowup=4, n_queries=16, pow_bits=5, hash="sha256")
    seeds = [3, 0, 11, 5, 8, 2, 9]
    want = [ss.stwo_to_json(gp.prove_proof(seed=s, **kw)) for s in seeds]
    for workers in (1, 3):
        got = gp.prove_many(seeds, workers=workers, **kw)
        assert [ss.stwo_to_json(p) for p in got] == want
    assert gp.timings["proofs_per_s"] > 0
    status = gp.ver.verify_stwo(got, cfg=got[0].cfg)
    assert status.tolist() == [0] * len(seeds)


# ------------------------------------------------------------------ `make proof` from the command line
def test_cli_prove_writes_what_the_reference_flow_writes(tmp_path):
    """`cli prove` = stark101/Makefile:14-17 (`python -m fibsquare`, then generate_wit.py): the reference prover's
    proof.json and its .wit, byte for byte; for stwo the two proofs the reference ships; and what it writes is what
    `cli verify` accepts."""
    import subprocess
    F = os.path.join(GOLDEN, "formats")

    def cli_process(*args):
        return subprocess.run([sys.executable, "-m", "stark_symphony_amd.cli", *args], cwd=ROOT, capture_output=True,
                              text=True, timeout=600)

    def cli(*args):
        """the same command through cli.main() in this process: a new python process per case costs seconds (twenty each on
        a slow box); the first prove and the first verify below go through real processes"""
        import contextlib
        import io
        import types
        from stark_symphony_amd import cli as C_
        out, err = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
            try:
                rc = C_.main(list(args))
            except SystemExit as e:
                rc = e.code
        return types.SimpleNamespace(returncode=rc, stdout=out.getvalue(), stderr=err.getvalue())
    r = cli_process("prove", "--family", "stark101")
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == json.load(open(os.path.join(GOLDEN, "stark101_proof.json")))
    wit = tmp_path / "proof.wit"
    r = cli("prove", "--family", "stark101", "--to", "wit", "--out", str(wit))
    assert r.returncode == 0 and r.stdout == "", r.stderr
    assert wit.read_text() == open(os.path.join(F, "stark101_proof.wit")).read()
    assert cli_process("verify", "--family", "stark101", "--witness", str(wit)).returncode == 0
    r = cli("prove", "--family", "stwo")  # defaults = the sizes of tests/data/proof.json
    assert r.returncode == 0 and json.loads(r.stdout) == json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))), r.stderr
    r = cli("prove", "--family", "stwo", "--trace-log", "3", "--log-blowup", "1", "--n-queries", "1", "--to", "wit")
    assert r.returncode == 0 and r.stdout == open(os.path.join(F, "stwo_proof_test.wit")).read(), r.stderr
    pj = tmp_path / "p.json"
    r = cli("prove", "--family", "stwo", "--trace-log", "12", "--seed", "3", "--hash", "blake2s", "--out", str(pj))
    assert r.returncode == 0, r.stderr
    assert cli("verify", "--family", "stwo", "--trace-log", "12", "--lde-log", "16", "--n-layers", "11", "--hash", "blake2s",
               "--proof", str(pj)).returncode == 0
    assert cli("verify", "--family", "stwo", "--proof", str(pj)).returncode == 1  # production config expected: other shape
    r = cli("prove", "--family", "stark101", "--seed", "5", "--out", str(pj))    # F3: only the reference seed verifies
    assert r.returncode == 0 and cli("verify", "--family", "stark101", "--proof", str(pj)).returncode == 1
