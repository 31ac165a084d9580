"""Binary fixtures: proofs stored as their records (include/ss_verify.h) in an .npz file.

A 2^20-row proof is 170 KB as a record but several MB as JSON, so the benchmark proofs made
by tools/stwo_prover.py are committed in this form (tests/golden/*.npz).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .formats import StwoConfig, StwoProof
from .verifier import stwo_record


def _path_bytes(words: np.ndarray) -> np.ndarray:
    """uint32[len * 8] big-endian-valued words -> uint8[len, 32]."""
    return np.ascontiguousarray(words, dtype=np.uint32).astype(">u4").view(np.uint8).reshape(-1, 32)


def stwo_from_record(cfg: StwoConfig, rec: np.ndarray) -> StwoProof:
    """Inverse of verifier.stwo_record for a uniform-shape record."""
    N, L, Q, K = cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    pos = 0

    def take(n: int) -> np.ndarray:
        nonlocal pos
        out = rec[pos:pos + n]
        pos += n
        return out
    roots = _path_bytes(take(24))
    oods_trace = take(4 * N).reshape(N, 4).copy()
    oods_cp = take(64).reshape(16, 4).copy()
    fri_roots = _path_bytes(take(8 * (K + 1)))
    last = take(4).copy()
    hi, lo = take(2)
    trace_vals = np.zeros((Q, N), dtype=np.uint32)
    cp_vals = np.zeros((Q, 16), dtype=np.uint32)
    trace_paths, cp_paths = [], []
    for q in range(Q):
        trace_vals[q] = take(N)
        cp_vals[q] = take(16)
        trace_paths.append(_path_bytes(take(8 * L)))
        cp_paths.append(_path_bytes(take(8 * L)))
    fri_witness = np.zeros((K + 1, Q, 4), dtype=np.uint32)
    fri_paths: List[List[np.ndarray]] = []
    for l in range(K + 1):
        row = []
        for q in range(Q):
            fri_witness[l, q] = take(4)
            row.append(_path_bytes(take(8 * (L - 1 - l))))
        fri_paths.append(row)
    if pos + (K + 3) * Q == rec.size:  # ABI 2.x: path_len trailer
        lens = take((K + 3) * Q).reshape(K + 3, Q)
        for q in range(Q):
            trace_paths[q] = trace_paths[q][:lens[0, q]]
            cp_paths[q] = cp_paths[q][:lens[1, q]]
            for l in range(K + 1):
                fri_paths[l][q] = fri_paths[l][q][:lens[2 + l, q]]
    if pos != rec.size:  # (fixtures written before the trailer existed end here: full-length paths)
        raise ValueError("record has %d words, config needs %d" % (rec.size, pos))
    return StwoProof(cfg, roots, oods_trace, oods_cp, trace_vals, cp_vals, trace_paths, cp_paths,
                     fri_roots, last, fri_witness, fri_paths, (int(hi) << 32) | int(lo))


def save_stwo_npz(path: str, proofs: Sequence[StwoProof]) -> None:
    cfg = proofs[0].cfg
    recs = []
    for p in proofs:
        if p.cfg != cfg:
            raise ValueError("mixed configs")
        uniform = all(len(x) == cfg.lde_log for x in p.trace_paths + p.cp_paths) and all(
            len(x) == cfg.fri_path_len(l) for l in range(cfg.n_layers + 1) for x in p.fri_paths[l])
        if not uniform:  # an over-long path does not survive the fixed slots
            raise ValueError("only proofs with full-length Merkle paths are stored as records")
        recs.append(stwo_record(p))
    np.savez_compressed(path, cfg=np.array([cfg.n_cols, cfg.trace_log, cfg.lde_log, cfg.n_queries,
                                            cfg.n_layers, cfg.pow_bits,
                                            1 if cfg.hash == "blake2s" else 0], dtype=np.uint32),
                        records=np.stack(recs))


def load_stwo_npz(path: str) -> List[StwoProof]:
    z = np.load(path)
    c = [int(x) for x in z["cfg"]]
    cfg = StwoConfig(*c[:6], "blake2s" if len(c) > 6 and c[6] == 1 else "sha256")
    return [stwo_from_record(cfg, r) for r in z["records"]]
