"""ctypes binding of the CPU oracle (oracle/libss_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/ss_oracle.h.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package never
imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libss_oracle.so")
MAX_LIST = 31
MODE_LITERAL, MODE_FIXTURE = 0, 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make).  Returns the library path."""
    srcs = [os.path.join(HERE, f) for f in ("ss_oracle.c", "ss_oracle_batch.c", "ss_oracle_shared.c", "ss_oracle.h")]
    stale = (not os.path.exists(LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return LIB_PATH


class CM31(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32)]

    def t(self):
        return (self.a, self.b)


class QM31(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32), ("d", C.c_uint32)]

    def t(self):
        return (self.a, self.b, self.c, self.d)


class M31Point(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]

    def t(self):
        return (self.x, self.y)


class QM31Point(C.Structure):
    _fields_ = [("x", QM31), ("y", QM31)]

    def t(self):
        return (self.x.t(), self.y.t())


class Channel(C.Structure):
    _fields_ = [("digest", C.c_uint8 * 32), ("counter", C.c_uint32)]


class S101Eval(C.Structure):
    _fields_ = [("ev", C.c_uint32), ("len", C.c_uint32), ("path", (C.c_uint8 * 32) * MAX_LIST)]


class S101Layer(C.Structure):
    _fields_ = [("root", C.c_uint8 * 32), ("beta", C.c_uint32), ("cpa", S101Eval), ("cpb", S101Eval)]


class S101Proof(C.Structure):
    _fields_ = [("root", C.c_uint8 * 32), ("evals", S101Eval * 3), ("n_layers", C.c_uint32),
                ("layers", S101Layer * MAX_LIST), ("last", C.c_uint32)]


class S101Trace(C.Structure):
    _fields_ = [("alpha", C.c_uint32 * 3), ("idx", C.c_uint32), ("x", C.c_uint32), ("cp", C.c_uint32),
                ("fold", C.c_uint32 * (MAX_LIST + 1)), ("state_after_commit", C.c_uint8 * 32)]


class Path(C.Structure):
    _fields_ = [("len", C.c_uint32), ("nodes", C.c_void_p)]


class StwoCfg(C.Structure):
    _fields_ = [("n_cols", C.c_uint32), ("trace_log", C.c_uint32), ("lde_log", C.c_uint32),
                ("n_queries", C.c_uint32), ("n_layers", C.c_uint32), ("pow_target", C.c_uint64),
                ("hash", C.c_uint32)]


class StwoProofC(C.Structure):
    _fields_ = [("roots", (C.c_uint8 * 32) * 3), ("oods_trace", C.c_void_p), ("oods_cp", QM31 * 16),
                ("trace_vals", C.c_void_p), ("cp_vals", C.c_void_p), ("trace_paths", C.c_void_p),
                ("cp_paths", C.c_void_p), ("fri_roots", C.c_void_p), ("last_layer", QM31),
                ("fri_witness", C.c_void_p), ("fri_paths", C.c_void_p), ("pow_nonce", C.c_uint64)]


class StwoTrace(C.Structure):
    _fields_ = [("cp_alpha", QM31), ("deep_alpha", QM31), ("oods_point", QM31Point),
                ("fold_alpha", QM31 * (MAX_LIST + 1)), ("digest_after", (C.c_uint8 * 32) * 6),
                ("queries", C.c_uint32 * 64), ("answers", QM31 * 64), ("folded", QM31 * 64),
                ("folded_query", C.c_uint32 * 64), ("final_log_size", C.c_uint32)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    u32, u8p, i = C.c_uint32, C.POINTER(C.c_uint8), C.c_int
    u32p = C.POINTER(C.c_uint32)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)
    sig("so_sha256", None, C.c_char_p, C.c_size_t, u8p)
    sig("so_blake2s", None, C.c_char_p, C.c_size_t, u8p)
    sig("so_set_hash", None, C.c_int)
    sig("so_sha256_blocks", C.c_uint64)
    sig("so_sha256_blocks_reset", None)
    for n in ("add_mod", "sub_mod", "mul_mod", "exp_mod"):
        sig("so_s101_" + n, u32, u32, u32)
    sig("so_s101_div_mod", i, u32, u32, u32p)
    sig("so_s101_reduce_256_mod_32", u32, C.c_char_p, u32)
    sig("so_s101_channel_draw_32", u32, u8p, u32)
    sig("so_s101_channel_mix_32", None, u8p, u32)
    sig("so_s101_channel_mix_256", None, u8p, C.c_char_p)
    sig("so_s101_merkle_verify", i, C.c_char_p, u32, C.c_char_p, u32, C.c_char_p)
    sig("so_s101_calc_x", u32, u32)
    sig("so_s101_eval_p0", i, u32, u32, u32p)
    sig("so_s101_eval_cp", i, u32, u32, u32, u32, u32, u32, u32, u32p)
    sig("so_s101_fri_eval_cp_next", i, u32, u32, u32, u32, u32p)
    sig("so_s101_compute_auth_path", None, u32, u32, u32p, u32p)
    sig("so_s101_verify", u32, C.POINTER(S101Proof), C.POINTER(S101Trace))
    sig("so_s101_verify_batch", None, C.POINTER(S101Proof), C.c_size_t, u32p, i)
    for n in ("add", "sub", "mul", "exp"):
        sig("so_m31_" + n, u32, u32, u32)
    sig("so_m31_neg", u32, u32)
    sig("so_m31_inv", i, u32, u32p)
    for n in ("add", "sub", "mul"):
        sig("so_cm31_" + n, CM31, CM31, CM31)
        sig("so_qm31_" + n, QM31, QM31, QM31)
    sig("so_cm31_inv", i, CM31, C.POINTER(CM31))
    sig("so_cm31_div", i, CM31, CM31, C.POINTER(CM31))
    sig("so_qm31_mul_m31", QM31, QM31, u32)
    sig("so_qm31_mul_cm31", QM31, QM31, CM31)
    sig("so_qm31_inv", i, QM31, C.POINTER(QM31))
    sig("so_m31_point_add", M31Point, M31Point, M31Point)
    sig("so_m31_point_dbl", M31Point, M31Point)
    sig("so_circle_point_index_to_m31_point", M31Point, u32)
    sig("so_qm31_point_add", QM31Point, QM31Point, QM31Point)
    sig("so_qm31_point_add_m31_point", QM31Point, QM31Point, M31Point)
    sig("so_bit_reverse_position", u32, u32, C.c_uint8)
    for n in ("add", "mul"):
        sig("so_circle_point_index_" + n, u32, u32, u32)
    sig("so_circle_point_index_neg", u32, u32)
    sig("so_circle_domain", None, C.c_uint8, u32p)
    sig("so_circle_position_to_point_index", u32, C.c_uint8, u32)
    sig("so_line_position_to_x_coord", u32, C.c_uint8, u32)
    chp = C.POINTER(Channel)
    sig("so_channel_init", None, chp)
    sig("so_channel_draw_qm31", i, chp, C.POINTER(QM31))
    sig("so_channel_draw_qm31_point", i, chp, C.POINTER(QM31Point))
    sig("so_channel_mix_u256", None, chp, C.c_char_p)
    sig("so_channel_mix_u64", None, chp, C.c_uint64)
    sig("so_channel_draw_queries_8", None, chp, u32, u32p)
    sig("so_reverse_bytes_32", u32, u32)
    sig("so_check_proof_of_work", i, chp, C.c_uint64, C.c_uint64)
    sig("so_hash_u32s", None, u32p, C.c_size_t, u8p)
    sig("so_stwo_merkle_verify", i, C.c_char_p, u32, C.c_char_p, u32, C.c_char_p)
    sig("so_evals_commit", i, chp, C.c_char_p, C.POINTER(QM31))
    sig("so_composition_poly_eval_from_partitions", QM31, C.POINTER(QM31))
    sig("so_composition_poly_eval_from_decomposed", QM31, C.POINTER(QM31), QM31Point)
    sig("so_vanishing_poly_eval", QM31, C.c_uint8, QM31Point)
    sig("so_eval_composition_poly", i, C.c_uint8, QM31Point, C.POINTER(QM31), u32, QM31,
        C.POINTER(QM31))
    sig("so_channel_mix_oods_evals", None, chp, C.POINTER(QM31), u32, C.POINTER(QM31))
    sig("so_deep_quotient_denominator_inverse", i, QM31Point, M31Point, C.POINTER(CM31))
    sig("so_deep_quotient_interpolant_coefficients", None, QM31Point, QM31, QM31, C.POINTER(QM31))
    sig("so_deep_quotient_nominator", QM31, C.POINTER(QM31), M31Point, u32)
    sig("so_circle_fold", i, u32, QM31, QM31, C.c_uint8, QM31, C.POINTER(QM31))
    sig("so_line_fold", i, u32, QM31, QM31, C.c_uint8, QM31, C.POINTER(QM31))
    sig("so_stwo_verify", u32, C.POINTER(StwoCfg), C.POINTER(StwoProofC), i, C.POINTER(StwoTrace))
    sig("so_stwo_verify_batch", None, C.POINTER(StwoCfg), C.POINTER(StwoProofC), C.c_size_t, i,
        u32p, i)
    sig("so_num_procs", i)
    sig("so_stwo_fri_tail", u32, i, u32, u32, u32, u32, QM31, QM31)
    sig("so_stwo_verify_minimal", u32, C.POINTER(StwoCfg), u32p, C.c_size_t, i)
    sig("so_stwo_minimal_expand", u32, C.POINTER(StwoCfg), u32p, C.c_size_t, i, u32p)
    sig("so_shared_walk", u32, u32, u32, u32, u32p, u32p)
    sig("so_shared_expand", i, u32, u32, u32, u32, u32p, C.c_size_t, u32p)
    _lib = L
    return L


# --------------------------------------------------------------------------- helpers
def sha256(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().so_sha256(msg, len(msg), out)
    return bytes(out)


def blake2s(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().so_blake2s(msg, len(msg), out)
    return bytes(out)


def qm(v: Sequence[int]) -> QM31:
    return QM31(int(v[0]), int(v[1]), int(v[2]), int(v[3]))


def qmp(x: Sequence[int], y: Sequence[int]) -> QM31Point:
    return QM31Point(qm(x), qm(y))


def qm_array(vals: Sequence[Sequence[int]]):
    arr = (QM31 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = qm(v)
    return arr


# ------------------------------------------------------------------------- stark101
def _fill_eval(dst: S101Eval, ev) -> None:
    dst.ev = ev.ev
    dst.len = len(ev.path)
    if len(ev.path):
        C.memmove(dst.path, np.ascontiguousarray(ev.path).ctypes.data, 32 * len(ev.path))


def s101_to_c(p, out: Optional[S101Proof] = None) -> S101Proof:
    """stark_symphony_amd.formats.Stark101Proof -> so_s101_proof."""
    c = out if out is not None else S101Proof()
    C.memmove(c.root, p.root, 32)
    for k in range(3):
        _fill_eval(c.evals[k], p.evals[k])
    c.n_layers = len(p.layers)
    for i, l in enumerate(p.layers):
        C.memmove(c.layers[i].root, l.root, 32)
        c.layers[i].beta = l.beta
        _fill_eval(c.layers[i].cpa, l.cpa)
        _fill_eval(c.layers[i].cpb, l.cpb)
    c.last = p.last
    return c


def s101_verify(p, trace: bool = False):
    c = s101_to_c(p)
    tr = S101Trace() if trace else None
    st = lib().so_s101_verify(C.byref(c), C.byref(tr) if trace else None)
    return (st, tr) if trace else st


def s101_verify_batch(proofs: Sequence, threads: int = 0) -> np.ndarray:
    """`proofs` may be a list of parsed proofs or a prepared ctypes array (see s101_array)."""
    arr = proofs if isinstance(proofs, C.Array) else s101_array(proofs)
    n = len(arr)
    st = np.zeros(n, dtype=np.uint32)
    lib().so_s101_verify_batch(arr, n, st.ctypes.data_as(C.POINTER(C.c_uint32)), threads)
    return st


def s101_array(proofs: Sequence):
    arr = (S101Proof * len(proofs))()
    for i, p in enumerate(proofs):
        s101_to_c(p, arr[i])
    return arr


# ----------------------------------------------------------------------------- stwo
class StwoHolder:
    """Keeps the numpy buffers a so_stwo_proof points into alive."""

    def __init__(self, p):
        cfg = p.cfg
        Q, K = cfg.n_queries, cfg.n_layers
        # the C side indexes every array by the config: check before handing pointers over
        shapes_ok = (np.shape(p.roots) == (3, 32) and np.shape(p.oods_trace) == (cfg.n_cols, 4)
                     and np.shape(p.oods_cp) == (16, 4) and np.shape(p.trace_vals) == (Q, cfg.n_cols)
                     and np.shape(p.cp_vals) == (Q, 16) and np.shape(p.fri_roots) == (K + 1, 32)
                     and np.shape(p.fri_witness) == (K + 1, Q, 4) and len(p.trace_paths) == Q
                     and len(p.cp_paths) == Q and len(p.fri_paths) == K + 1
                     and all(len(l) == Q for l in p.fri_paths))
        paths = list(p.trace_paths) + list(p.cp_paths) + [x for l in p.fri_paths for x in l]
        if not shapes_ok or any(len(x) > 31 or (len(x) and np.shape(x)[1:] != (32,)) for x in paths):
            raise ValueError("proof arrays do not have the shape of its StwoConfig %r" % (cfg,))
        self.cfg = StwoCfg(cfg.n_cols, cfg.trace_log, cfg.lde_log, Q, K, cfg.pow_target,
                           1 if getattr(cfg, "hash", "sha256") == "blake2s" else 0)
        self.keep: List[np.ndarray] = []
        c = StwoProofC()
        C.memmove(c.roots, np.ascontiguousarray(p.roots).ctypes.data, 96)
        c.oods_trace = self._buf(p.oods_trace.astype(np.uint32))
        for k in range(16):
            c.oods_cp[k] = qm(p.oods_cp[k])
        c.trace_vals = self._buf(p.trace_vals.astype(np.uint32))
        c.cp_vals = self._buf(p.cp_vals.astype(np.uint32))
        self.tp = self._paths(p.trace_paths)
        self.cpp = self._paths(p.cp_paths)
        self.fp = self._paths([x for l in p.fri_paths for x in l])
        c.trace_paths = C.addressof(self.tp)
        c.cp_paths = C.addressof(self.cpp)
        c.fri_paths = C.addressof(self.fp)
        c.fri_roots = self._buf(p.fri_roots.astype(np.uint8))
        c.last_layer = qm(p.last_layer)
        c.fri_witness = self._buf(p.fri_witness.astype(np.uint32))
        c.pow_nonce = p.pow_nonce
        self.c = c

    def _buf(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        self.keep.append(a)
        return a.ctypes.data

    def _paths(self, paths: Sequence[np.ndarray]):
        arr = (Path * max(1, len(paths)))()
        for i, pth in enumerate(paths):
            arr[i].len = len(pth)
            arr[i].nodes = self._buf(pth.astype(np.uint8)) if len(pth) else None
        return arr


def stwo_verify(p, mode: int = MODE_FIXTURE, trace: bool = False):
    h = StwoHolder(p)
    tr = StwoTrace() if trace else None
    st = lib().so_stwo_verify(C.byref(h.cfg), C.byref(h.c), mode, C.byref(tr) if trace else None)
    return (st, tr) if trace else st


class StwoBatch:
    """A prepared array of so_stwo_proof for the timed CPU baseline."""

    def __init__(self, proofs: Sequence):
        if not proofs or any(p.cfg != proofs[0].cfg for p in proofs):
            raise ValueError("StwoBatch needs a non-empty list of proofs of ONE StwoConfig "
                             "(stwo_verify_batch groups mixed lists)")
        self.holders = [StwoHolder(p) for p in proofs]
        self.arr = (StwoProofC * len(proofs))()
        for i, h in enumerate(self.holders):
            C.memmove(C.addressof(self.arr[i]), C.addressof(h.c), C.sizeof(StwoProofC))
        self.cfg = self.holders[0].cfg

    def verify(self, mode: int = MODE_FIXTURE, threads: int = 0, repeat: int = 1) -> np.ndarray:
        n = len(self.holders)
        st = np.zeros(n, dtype=np.uint32)
        for _ in range(repeat):
            lib().so_stwo_verify_batch(C.byref(self.cfg), self.arr, n, mode,
                                       st.ctypes.data_as(C.POINTER(C.c_uint32)), threads)
        return st


def stwo_verify_batch(proofs: Sequence, mode: int = MODE_FIXTURE, threads: int = 0, cfg=None) -> np.ndarray:
    """Status word per proof, each verified under its OWN config (grouped, input order kept).  With
    `cfg` (a StwoConfig or a list of them) proofs of any other config get status 1, the checker's
    counterpart of Verifier.verify_stwo's STATUS_CONFIG_MISMATCH."""
    allowed = None if cfg is None else ([cfg] if not isinstance(cfg, (list, tuple)) else list(cfg))
    out = np.zeros(len(proofs), dtype=np.uint32)
    groups: dict = {}
    for i, p in enumerate(proofs):
        if allowed is not None and p.cfg not in allowed:
            out[i] = 1
        else:
            groups.setdefault(p.cfg, []).append(i)
    for idx in groups.values():
        out[idx] = StwoBatch([proofs[i] for i in idx]).verify(mode, threads)
    return out


def shared_walk(lde_log: int, tree: int, queries: Sequence[int]):
    """-> (plan[q][lvl], count): the first-use walk over the queries' paths of tree `tree` (0 trace, 1 cp, 2 + l FRI
    layer l); ss_oracle_shared.c."""
    ln = lde_log if tree < 2 else lde_log + 1 - tree
    qs = np.ascontiguousarray(queries, dtype=np.uint32)
    plan = np.zeros((len(qs), ln), dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    count = lib().so_shared_walk(lde_log, tree, len(qs), qs.ctypes.data_as(u32p), plan.ctypes.data_as(u32p))
    return plan, int(count)


def shared_expand(cfg, shared: np.ndarray):
    """Shared record -> (outcome, per-query record) by the definitional walk (ss_oracle_shared.c)."""
    N, L, Q, K = cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers
    words = 24 + 4 * N + 64 + 8 * (K + 1) + 6 + Q * (N + 16 + 16 * L) + sum(Q * (4 + 8 * (L - 1 - l)) for l in range(K + 1)) \
        + (K + 3) * Q
    sh = np.ascontiguousarray(shared, dtype=np.uint32)
    rec = np.zeros(words, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    rc = lib().so_shared_expand(N, L, Q, K, sh.ctypes.data_as(u32p), sh.size, rec.ctypes.data_as(u32p))
    return int(rc), rec


def _cfg_struct(cfg) -> StwoCfg:
    return StwoCfg(cfg.n_cols, cfg.trace_log, cfg.lde_log, cfg.n_queries, cfg.n_layers, cfg.pow_target,
                   1 if getattr(cfg, "hash", "sha256") == "blake2s" else 0)


def stwo_verify_minimal(cfg, rec: np.ndarray, mode: int = MODE_FIXTURE) -> int:
    """Status of a MINIMAL record (include/ss_verify.h) by the layer-by-layer walk of upstream stwo's MerkleVerifier /
    SparseEvaluation as restated in ss_oracle.c (parity unpinned: the reference has no such format)."""
    r = np.ascontiguousarray(rec, dtype=np.uint32)
    return int(lib().so_stwo_verify_minimal(C.byref(_cfg_struct(cfg)), r.ctypes.data_as(C.POINTER(C.c_uint32)), r.size, mode))


def stwo_minimal_expand(cfg, rec: np.ndarray, mode: int = MODE_FIXTURE):
    """-> (status, R(M)): the per-query record a minimal record corresponds to (every omitted sibling / fold-pair
    evaluation replaced by the value the walk computes; a tree whose lists have the wrong length gets path_len 0)."""
    N, L, Q, K = cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers
    words = 24 + 4 * N + 64 + 8 * (K + 1) + 6 + Q * (N + 16 + 16 * L) + sum(Q * (4 + 8 * (L - 1 - l)) for l in range(K + 1)) \
        + (K + 3) * Q
    r = np.ascontiguousarray(rec, dtype=np.uint32)
    out = np.zeros(words, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    st = lib().so_stwo_minimal_expand(C.byref(_cfg_struct(cfg)), r.ctypes.data_as(u32p), r.size, mode, out.ctypes.data_as(u32p))
    return int(st), out


def num_procs() -> int:
    return int(lib().so_num_procs())


def effective_cpus() -> int:
    """Host cores this process may actually use: the scheduler affinity capped by the cgroup CPU
    quota (a container can see 256 logical CPUs and be allowed 16 cores' worth of time; OpenMP over
    all 256 then runs slower than over 16)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, min(n, num_procs()))
