"""AddressSanitizer + UBSan builds of the host code that indexes caller-owned memory (CPU only; the GPU pool
has no sanitizer): the record -> batch packers, the text writers / templates / scalar reader rule, and the
oracle itself (the checker must not be the thing that reads out of bounds).  The native text readers have
their own driver in tests/test_ingest.py."""
import os
import subprocess

import numpy as np

import stark_symphony_amd as ss
from stark_symphony_amd import formats, verifier
from oracle import oracle as O

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "stark-symphony_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")


def test_packers_writers_and_the_reader_rule_under_sanitizers(tmp_path):
    exe = str(tmp_path / "host_san")
    subprocess.run(["g++", "-std=c++17"] + SAN + ["-I" + CSRC, os.path.join(NATIVE, "host_san.cpp")]
                   + [os.path.join(CSRC, f) for f in ("ss_pack.cpp", "ss_text.cpp", "ss_pool.cpp", "ss_ingest.cpp", "ss_sharedrec.cpp", "ss_minimalrec.cpp")]
                   + ["-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe, "20261003", "1500", os.path.join(GOLDEN, "stwo_proof.json"),
                        os.path.join(GOLDEN, "formats", "stwo_proof.wit")], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0 and "host_san:" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_oracle_under_sanitizers_gives_the_same_status_words(tmp_path, stwo_prod, stwo_small):
    """Valid proofs, seeded corruptions, ragged Merkle paths and unreduced words through an ASan + UBSan build
    of oracle/ss_oracle.c: no report, and the status words of the production build."""
    exe = str(tmp_path / "oracle_san")
    subprocess.run(["gcc", "-std=c11"] + SAN + ["-I" + os.path.join(ROOT, "oracle"), os.path.join(NATIVE, "oracle_san.c"),
                    os.path.join(ROOT, "oracle", "ss_oracle.c"), "-o", exe], check=True)
    rng = np.random.default_rng(0x5EED2025 + 303)
    for base in (stwo_prod, stwo_small):
        batch = [base] + [formats.stwo_corrupt(base, rng)[0] for _ in range(60)]
        short = base.copy(); short.fri_paths[1][0] = short.fri_paths[1][0][:-1]
        cut = base.copy(); cut.trace_paths[0] = cut.trace_paths[0][:1]
        big = base.copy(); big.trace_vals[0, 0] = 0xFFFFFFFF; big.oods_cp[3, 1] = 0x80000001
        batch += [short, cut, big]
        c = base.cfg
        path = tmp_path / "records.bin"
        np.stack([verifier.stwo_record(p) for p in batch]).astype("<u4").tofile(path)
        r = subprocess.run([exe, str(c.n_cols), str(c.trace_log), str(c.lde_log), str(c.n_queries), str(c.n_layers), "0", str(path)],
                           capture_output=True, text=True, env=ENV, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        got = np.array([[int(x) for x in l.split()] for l in r.stdout.split("\n") if l], dtype=np.uint32)
        assert got[:, 0].tolist() == O.stwo_verify_batch(batch, O.MODE_FIXTURE).tolist()
        assert got[:, 1].tolist() == O.stwo_verify_batch(batch, O.MODE_LITERAL).tolist()
        assert got[0, 0] == 0 and (got[1:, 0] != 0).sum() > 30
