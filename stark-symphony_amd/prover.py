"""GPU wide-Fibonacci circle-STARK prover.

Chains the kernels of include/ss_prover.h into the protocol of the external stwo fork that made
the reference's proofs (stwo-verifier/tests/data/proof*.json): trace -> circle iFFT/LDE ->
Merkle commit -> composition polynomial in 16 partition columns -> OODS samples at P / 2P ->
two-batch DEEP quotients -> FRI commit -> proof of work -> queries -> decommit.  The
Fiat-Shamir channel (a few dozen hashes) and the O(columns) scalar QM31 algebra stay on the
host; everything proportional to the domain size runs in HIP kernels.

Output: the reference's `proof.json` schema (format C), byte for byte equal to
tools/stwo_prover.py -- and therefore to the reference's two fixtures
(tests/test_gpu_prover.py).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import binding as B

P = (1 << 31) - 1
Q4 = Tuple[int, int, int, int]
HASHES = {"sha256": hashlib.sha256, "blake2s": lambda b=b"": hashlib.blake2s(b, digest_size=32)}


# ------------------------------------------------------------- host QM31 scalars (Python ints)
def _cmul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def qmul(x: Q4, y: Q4) -> Q4:
    ac = _cmul(x[0:2], y[0:2])
    bd = _cmul(x[2:4], y[2:4])
    r = _cmul(bd, (2, 1))
    ad = _cmul(x[0:2], y[2:4])
    bc = _cmul(x[2:4], y[0:2])
    return ((ac[0] + r[0]) % P, (ac[1] + r[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)


def qadd(x: Q4, y: Q4) -> Q4:
    return tuple((a + b) % P for a, b in zip(x, y))


def qsub(x: Q4, y: Q4) -> Q4:
    return tuple((a - b) % P for a, b in zip(x, y))


def _cinv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, P - 2, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def qinv(x: Q4) -> Q4:
    a2 = _cmul(x[0:2], x[0:2])
    b2 = _cmul(x[2:4], x[2:4])
    t = _cmul(b2, (2, 1))
    den = ((a2[0] - t[0]) % P, (a2[1] - t[1]) % P)
    di = _cinv(den)
    re = _cmul(x[0:2], di)
    im = _cmul(((-x[2]) % P, (-x[3]) % P), di)
    return (re[0], re[1], im[0], im[1])


ONE: Q4 = (1, 0, 0, 0)
ZERO: Q4 = (0, 0, 0, 0)


def _interpolant(sp_y: Q4, value: Q4, alpha_i: Q4):
    """deep/quotients.simf:25-36 on canonical values."""
    a0 = (0, 0, (-2 * value[2]) % P, (-2 * value[3]) % P)
    b0 = (0, 0, (-2 * sp_y[2]) % P, (-2 * sp_y[3]) % P)
    c0 = qsub(qmul(b0, value), qmul(a0, sp_y))
    return qmul(alpha_i, a0), qmul(alpha_i, b0), qmul(alpha_i, c0)


class _Channel:
    """stwo-verifier/src/channel.simf:31-172 (host side of the Fiat-Shamir transcript)."""

    def __init__(self, h):
        self.h = h
        self.digest = bytes(32)
        self.counter = 0

    def mix(self, b: bytes) -> None:
        self.digest = self.h(self.digest + b).digest()
        self.counter = 0

    def draw_words(self) -> List[int]:
        d = self.h(self.digest + self.counter.to_bytes(4, "big")).digest()
        self.counter += 1
        return [int.from_bytes(d[4 * i:4 * i + 4], "big") for i in range(8)]

    def draw_qm31(self) -> Q4:
        while True:
            w = self.draw_words()
            if all(x < 4294967294 for x in w[:4]):
                return tuple(x % P for x in w[:4])


def _be(words: Sequence[int]) -> bytes:
    return b"".join(int(w).to_bytes(4, "big") for w in words)


class GpuProver:
    """Proves on one MI355X through the ss_p_* kernels.  `ver` is a verifier.Verifier (it owns
    the library context and the device)."""

    def __init__(self, ver):
        import torch
        self.torch = torch
        self.ver = ver
        self.dev = ver.device
        self.lib = B.lib()
        u32p, vp, sz = C.c_void_p, C.c_void_p, C.c_size_t
        L = self.lib

        def sig(name, *args):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [vp] + list(args) + [vp]
        sig("ss_p_trace", C.c_uint32, C.c_uint32, C.c_uint32, u32p)
        sig("ss_p_twiddles", C.c_uint32, u32p, u32p, u32p)
        sig("ss_p_fft", C.c_uint32, C.c_uint32, u32p, u32p, C.c_int)
        sig("ss_p_lde", C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, u32p)
        sig("ss_p_hash_rows", C.c_uint32, sz, C.c_uint32, u32p, sz, u32p)
        sig("ss_p_hash_qm31", C.c_uint32, sz, u32p, u32p)
        sig("ss_p_merkle", C.c_uint32, sz, u32p)
        sig("ss_p_merkle_dup", C.c_uint32, sz, u32p, u32p)
        sig("ss_p_composition", C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p)
        sig("ss_p_eval_at_point", C.c_uint32, u32p, u32p, u32p, u32p)
        sig("ss_p_eval_at_point_batch", C.c_uint32, C.c_uint32, u32p, sz, u32p, u32p, u32p)
        sig("ss_p_channel_fri_layer", C.c_uint32, u32p, u32p, u32p, u32p)
        sig("ss_p_fri_fold_dev", sz, u32p, u32p, u32p, u32p)
        sig("ss_p_quotients", C.c_uint32, C.c_uint32, u32p, u32p, C.c_uint32, u32p, u32p, u32p, u32p, u32p, u32p)
        sig("ss_p_fri_fold", sz, u32p, u32p, u32p, u32p)
        sig("ss_p_pow", C.c_uint32, u32p, C.c_uint64, C.c_uint64, C.c_uint64, u32p)
        self._dom: Dict[int, tuple] = {}
        self._workers: list = []  # (prover, stream) of prove_many
        self.timings: Dict[str, float] = {}

    # -- small helpers ---------------------------------------------------------------------
    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int32, device=self.dev)

    def _zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.int32, device=self.dev)

    @staticmethod
    def _q(v: Q4):
        return (C.c_uint32 * 4)(*[int(x) for x in v])

    def _call(self, name: str, *args) -> None:
        B.check(getattr(self.lib, name)(self.ver.ctx, *args, self._stream()))

    def _host(self, t) -> np.ndarray:
        return t.detach().cpu().numpy().view(np.uint32)

    def domain(self, m: int):
        """(tw, itw, hx_hy) of the canonic coset of log size m."""
        if m not in self._dom:
            tw, itw = self._empty(1 << m), self._empty(1 << m)
            hxhy = self._empty(1 << m)
            half = 1 << (m - 1)
            self._call("ss_p_twiddles", m, tw.data_ptr(), itw.data_ptr(), hxhy.data_ptr())
            hxhy[half:] = tw[:half]  # pair y = twiddle layer 0
            self._dom[m] = (tw, itw, hxhy)
        return self._dom[m]

    def fft(self, m: int, data, inverse: bool) -> None:
        tw, itw, _ = self.domain(m)
        self._call("ss_p_fft", m, data.shape[0], data.data_ptr(), (itw if inverse else tw).data_ptr(),
                   1 if inverse else 0)

    def extend(self, coefs, m: int, x_only: bool = False):
        """Coefficients [ncols, 2^k] -> evaluations on the canonic coset of log size m.
        x_only: the polynomials have no y term -- `coefs` holds the coefficients of x^k1 pi(x)^k2 ... only (the even
        positions of the full array) -- so the two points (x, +-y) of a storage pair get the same value: returns ONE
        value per pair, [ncols, 2^(m-1)].  That is the transform of size m - 1 run with the twiddles of layers 1.. of
        size m, which lie behind its first 2^(m-1) words (include/ss_prover.h, ss_p_twiddles)."""
        k = int(coefs.shape[1]).bit_length() - 1
        assert coefs.shape[1] == 1 << k and coefs.is_contiguous()
        tw = self.domain(m)[0]
        if x_only:
            tw, m = tw[1 << (m - 1):], m - 1
        out = self._empty(coefs.shape[0], 1 << m)
        # (ss_p_lde: the forward transform of the coefficients followed by zeros, without writing or reading the zeros)
        self._call("ss_p_lde", k, m, coefs.shape[0], coefs.data_ptr(), out.data_ptr(), tw.data_ptr())
        return out

    def merkle(self, hsel: int, leaves_fn, n: int):
        """levels[(2n - 1), 8]: leaves first, root last."""
        levels = self._empty(2 * n, 8)
        leaves_fn(levels)
        self._call("ss_p_merkle", hsel, n, levels.data_ptr())
        return levels

    def _root(self, levels, n: int) -> bytes:
        return self._host(levels[2 * n - 2]).astype(">u4").tobytes()

    @staticmethod
    def _path_rows(n: int, indices: Sequence[int], skip: int = 0) -> np.ndarray:
        """Rows of a tree's `levels` array holding the sibling hashes leaf -> root of every index at
        level `skip`, concatenated in query order.  Level j starts at row 2n - (2n >> j)."""
        h = n.bit_length() - 1
        if h <= skip or not len(indices):
            return np.zeros(0, dtype=np.int64)
        j = np.arange(skip, h, dtype=np.int64)
        idx = np.asarray(indices, dtype=np.int64)[:, None] >> (j - skip)[None, :]
        return ((2 * n - ((2 * n) >> j))[None, :] + (idx ^ 1)).reshape(-1)

    class _Gather:
        """Rows of several device arrays, fetched with ONE upload of all row indices and ONE
        download: requests are queued, `fetch()` uploads the concatenated indices, gathers on the
        device, concatenates and copies back (the decommitment used to cost a synchronising index
        upload and a synchronising download per tree and per layer)."""

        def __init__(self, prover):
            self.p, self.req = prover, []

        def rows(self, array, idxs) -> int:
            self.req.append((array, np.asarray(idxs, dtype=np.int64)))
            return len(self.req) - 1

        def take(self, tensor) -> int:
            self.req.append((tensor, None))
            return len(self.req) - 1

        def fetch(self) -> List[np.ndarray]:
            t = self.p.torch
            all_idx = np.concatenate([i for _, i in self.req if i is not None] or [np.zeros(0, np.int64)])
            dev_idx = t.from_numpy(all_idx).to(self.p.dev)
            parts, shapes, pos = [], [], 0
            for array, idx in self.req:
                if idx is None:
                    sel = array
                else:
                    sel = array[dev_idx[pos:pos + idx.size]]
                    pos += idx.size
                parts.append(sel.reshape(-1))
                shapes.append(tuple(sel.shape))
            flat = self.p._host(t.cat(parts)) if parts else np.zeros(0, np.uint32)
            out, pos = [], 0
            for shp in shapes:
                size = int(np.prod(shp))
                out.append(flat[pos:pos + size].reshape(shp))
                pos += size
            return out

    @staticmethod
    def _hash_rows_to_json(rows: np.ndarray) -> List[List[int]]:
        if rows.size == 0:
            return []
        return rows.reshape(-1, 8).astype(">u4").view(np.uint8).reshape(-1, 32).tolist()

    # -- the protocol ----------------------------------------------------------------------
    def prove_proof(self, n_cols: int = 4, trace_log: int = 9, log_blowup: int = 4, n_queries: int = 16,
                    pow_bits: int = 5, seed: int = 0, hash: str = "sha256"):
        """-> formats.StwoProof (arrays; what the verifier's record is made from)."""
        torch = self.torch
        t_start = time.perf_counter()
        marks: List[Tuple[str, float]] = []

        def mark(name: str) -> None:
            torch.cuda.current_stream(self.dev).synchronize()  # this proof's stream only: others may be in flight (prove_many)
            marks.append((name, time.perf_counter()))
        n, L, N = trace_log, trace_log + log_blowup, n_cols
        K = L - 1 - log_blowup
        hsel = 1 if hash == "blake2s" else 0
        H = HASHES[hash]
        ch = _Channel(H)
        size_L = 1 << L

        # ---- trace, interpolation, LDE, commitment
        cols = self._empty(N, 1 << n)
        self._call("ss_p_trace", n, N, (seed * 0x9E3779B1) % P, cols.data_ptr())
        coefs = cols
        self.fft(n, coefs, True)
        lde = self.extend(coefs, L)
        trace_tree = self.merkle(hsel, lambda lv: self._call(
            "ss_p_hash_rows", hsel, size_L, N, lde.data_ptr(), size_L, lv.data_ptr()), size_L)
        const_root = H(b"").digest()
        trace_root = self._root(trace_tree, size_L)
        mark("trace commit")
        ch.mix(const_root)
        ch.mix(trace_root)
        cp_alpha = ch.draw_qm31()

        # ---- composition polynomial on the coset of log size n + 1, 16 partition columns
        ev = self.extend(coefs, n + 1)
        _, _, hxhy_c = self.domain(n + 1)
        F = self._empty(4, 1 << (n + 1))
        self._call("ss_p_composition", n, N, ev.data_ptr(), hxhy_c.data_ptr(), self._q(cp_alpha), F.data_ptr())
        self.fft(n + 1, F, True)
        # Partition 4c + part of coordinate c takes the coefficients whose two lowest index bits are `part` and is a
        # polynomial in x alone: in the basis of size n its coefficients sit at the even positions (no y term).  Kept
        # compact (2^(n-1) per column); its LDE has one value per storage pair, and the two leaves of a pair are equal.
        cp_coefs = self._empty(16, 1 << (n - 1))
        for c in range(4):
            for part in range(4):
                cp_coefs[4 * c + part] = F[c, part::4]
        cp_lde = self.extend(cp_coefs, L, x_only=True)  # [16, 2^(L-1)]: position i -> column value at i >> 1
        half_L = size_L >> 1

        # The two leaves of a storage pair are equal, so the 2^L leaf hashes are 2^(L-1) distinct ones and the first node
        # level is H(leaf || leaf): `cp_tree` holds the tree from that level up (its "leaves" are the level-1 nodes), the
        # duplicated leaf level exists nowhere (round 3 wrote it out: 0.8 GB of copies per proof).
        pair_leaf = self._empty(half_L, 8)
        self._call("ss_p_hash_rows", hsel, half_L, 16, cp_lde.data_ptr(), half_L, pair_leaf.data_ptr())
        cp_tree = self.merkle(hsel, lambda lv: self._call(
            "ss_p_merkle_dup", hsel, half_L, pair_leaf.data_ptr(), lv.data_ptr()), half_L)
        cp_root = self._root(cp_tree, half_L)
        mark("composition commit")
        ch.mix(cp_root)

        # ---- OODS point, samples
        t = ch.draw_qm31()
        t_sq = qmul(t, t)
        inv = qinv(qadd(ONE, t_sq))
        px, py = qmul(qsub(ONE, t_sq), inv), qmul(qadd(t, t), inv)
        p2x = qsub(qadd(qmul(px, px), qmul(px, px)), ONE)
        p2y = qadd(qmul(px, py), qmul(px, py))

        def factors(x: Q4, y: Q4, m: int) -> np.ndarray:
            f = [y, x]
            cur = x
            for _ in range(2, m):
                cur = qsub(qadd(qmul(cur, cur), qmul(cur, cur)), ONE)
                f.append(cur)
            return np.array(f[:m], dtype=np.uint32).reshape(-1)
        scratch = self._empty(max(N, 16) * (3 << n))
        samples = self._empty(N + 16, 4)
        f1, f2 = factors(px, py, n), factors(p2x, p2y, n)
        # all columns of a commitment are sampled at one point: one batched fold chain each
        self._call("ss_p_eval_at_point_batch", n, N, coefs.data_ptr(), 1 << n, f1.ctypes.data, scratch.data_ptr(),
                   samples.data_ptr())
        if n > 1:
            f2x = np.ascontiguousarray(f2[4:])  # no y term: the factors x, pi(x), ... over the compact coefficients
            self._call("ss_p_eval_at_point_batch", n - 1, 16, cp_coefs.data_ptr(), 1 << (n - 1), f2x.ctypes.data,
                       scratch.data_ptr(), samples[N:].data_ptr())
        else:  # a two-row trace: every partition is a constant
            samples[N:] = 0
            samples[N:, 0] = cp_coefs[:, 0]
        samp = self._host(samples).astype(np.int64)
        oods_trace = [tuple(int(x) for x in samp[k]) for k in range(N)]
        oods_cp = [tuple(int(x) for x in samp[N + k]) for k in range(16)]
        ch.mix(b"".join(_be(v) for v in oods_trace) + b"".join(_be(v) for v in oods_cp))
        deep_alpha = ch.draw_qm31()
        mark("oods")

        # ---- DEEP quotients (trace batch at P, composition batch at 2P)
        bco: List[Q4] = []
        sums: List[Q4] = []
        for sp_y, vals in ((py, oods_trace), (p2y, oods_cp)):
            A, Cc, alpha_i = ZERO, ZERO, deep_alpha
            for v in vals:
                a, b, c = _interpolant(sp_y, v, alpha_i)
                bco.append(b)
                A, Cc = qadd(A, a), qadd(Cc, c)
                alpha_i = qmul(alpha_i, deep_alpha)
            sums += [A, Cc]
        a16 = ONE
        for _ in range(16):
            a16 = qmul(a16, deep_alpha)
        bcoef = torch.from_numpy(np.array(bco, dtype=np.uint32).view(np.int32)).to(self.dev)
        pts = (C.c_uint32 * 8)(*px, *py)
        pts2 = (C.c_uint32 * 8)(*p2x, *p2y)
        sa = (C.c_uint32 * 20)(*[x for v in sums + [a16] for x in v])
        _, itw_L, hxhy_L = self.domain(L)
        layer = self._empty(size_L, 4)
        self._call("ss_p_quotients", L, N, lde.data_ptr(), cp_lde.data_ptr(), L - 1, hxhy_L.data_ptr(),
                   bcoef.data_ptr(), pts, pts2, sa, layer.data_ptr())
        mark("quotients")

        # ---- FRI commit (fri/commit.simf:70-85), channel included, without a host round trip: per
        # layer the Merkle tree, then one lane mixes the root and draws alpha ON THE DEVICE, then the
        # fold reads alpha from device memory.  One download at the end (roots, channel, last layer).
        state = torch.from_numpy(np.concatenate([np.frombuffer(ch.digest, dtype=">u4").astype(np.uint32),
                                                 np.array([ch.counter], dtype=np.uint32)]).view(np.int32)).to(self.dev)
        roots_dev = self._empty(K + 1, 8)
        alphas_dev = self._empty(K + 1, 4)
        layers, trees = [], []
        for l in range(K + 1):
            size = size_L >> l
            layers.append(layer)
            cur = layer
            tree = self.merkle(hsel, lambda lv: self._call(
                "ss_p_hash_qm31", hsel, size, cur.data_ptr(), lv.data_ptr()), size)
            trees.append(tree)
            self._call("ss_p_channel_fri_layer", hsel, state.data_ptr(), tree[2 * size - 2].data_ptr(),
                       alphas_dev[l].data_ptr(), roots_dev[l].data_ptr())
            nxt = self._empty(size >> 1, 4)
            off = size_L - (size_L >> l)  # inverse twiddle layer l = 1 / fold coordinate
            self._call("ss_p_fri_fold_dev", size >> 1, layer.data_ptr(), itw_L[off:].data_ptr(),
                       alphas_dev[l].data_ptr(), nxt.data_ptr())
            layer = nxt
        g = self._Gather(self)
        k_state, k_roots, k_last = g.take(state), g.take(roots_dev), g.take(layer)
        got = g.fetch()
        ch.digest = got[k_state][:8].astype(">u4").tobytes()
        ch.counter = int(got[k_state][8])
        roots = [got[k_roots][l].astype(">u4").tobytes() for l in range(K + 1)]
        last_all = got[k_last]
        if not (last_all == last_all[0]).all():
            raise AssertionError("last FRI layer is not constant: the quotient is not low degree")
        last = tuple(int(x) for x in last_all[0])
        ch.mix(_be(last))
        mark("fri commit")

        # ---- proof of work (pow.simf:22-36): smallest nonce, searched on the GPU
        target = (1 << (64 - pow_bits)) - 1
        dig = (C.c_uint32 * 8)(*np.frombuffer(ch.digest, dtype=">u4").astype(np.uint32))
        out = torch.zeros(2, dtype=torch.int32, device=self.dev)
        start, span = 0, 1 << min(22, pow_bits + 8)  # expected 2^pow_bits tries: rarely a second pass
        while True:
            self._call("ss_p_pow", hsel, dig, target, start, span, out.data_ptr())
            nonce = int(self._host(out).view(np.uint64)[0])
            if nonce != 0xFFFFFFFFFFFFFFFF:
                break
            start += span
            span = min(span * 4, 1 << 22)
        ch.mix(nonce.to_bytes(8, "big"))
        mark("pow")

        # ---- queries (fri/queries.simf:29-43) and decommitment
        mask = (1 << L) - 1
        queries: List[int] = []
        while len(queries) < n_queries:
            queries += [w & mask for w in ch.draw_words()]
        queries = queries[:n_queries]
        self.queries = list(queries)  # (of the last proof made by this call path; prove_minimal uses them)
        # every gather of the decommitment is enqueued first, then downloaded at once
        g = self._Gather(self)
        lde_t, cp_lde_t = lde.T, cp_lde.T  # views: row = LDE position
        k_tq, k_cq = g.rows(lde_t, queries), g.rows(cp_lde_t, [q >> 1 for q in queries])  # [Q, N], [Q, 16]
        k_thw = g.rows(trace_tree, self._path_rows(size_L, queries))
        # composition tree: the sibling of leaf q is the other leaf of its pair (the same hash); above it, the tree over the pairs
        k_chw0 = g.rows(pair_leaf, [q >> 1 for q in queries])
        k_chw = g.rows(cp_tree, self._path_rows(half_L, [q >> 1 for q in queries]))
        k_layers = []
        cur_q = list(queries)
        for l in range(K + 1):
            size = size_L >> l
            k_sib = g.rows(layers[l], [q ^ 1 for q in cur_q])
            k_hw = g.rows(trees[l], self._path_rows(size, [q >> 1 for q in cur_q], skip=1))
            k_layers.append((k_sib, k_hw))
            cur_q = [q >> 1 for q in cur_q]
        got = g.fetch()

        def hashes(rows: np.ndarray, *shape) -> np.ndarray:
            """stored words [.., 8] -> bytes [.., 32]"""
            return np.ascontiguousarray(rows).reshape(-1, 8).astype(">u4").view(np.uint8).reshape(*shape, 32)
        from .formats import StwoConfig, StwoProof
        Q = n_queries
        proof = StwoProof(
            StwoConfig(N, n, L, Q, K, pow_bits, hash),
            np.stack([np.frombuffer(r, dtype=np.uint8) for r in (const_root, trace_root, cp_root)]),
            np.array(oods_trace, dtype=np.uint32).reshape(N, 4), np.array(oods_cp, dtype=np.uint32).reshape(16, 4),
            got[k_tq].reshape(Q, N).copy(), got[k_cq].reshape(Q, 16).copy(),
            list(hashes(got[k_thw], Q, L)),
            list(hashes(np.concatenate([got[k_chw0].reshape(Q, 1, 8), got[k_chw].reshape(Q, L - 1, 8)], axis=1), Q, L)),
            np.stack([np.frombuffer(r, dtype=np.uint8) for r in roots]), np.array(last, dtype=np.uint32),
            np.stack([got[ks].reshape(Q, 4) for ks, _ in k_layers]),
            [list(hashes(got[kh], Q, L - 1 - l)) for l, (_, kh) in enumerate(k_layers)], nonce)
        mark("decommit")
        prev = t_start
        self.timings = {}
        for name, ts in marks:
            self.timings[name] = ts - prev
            prev = ts
        self.timings["total"] = prev - t_start
        return proof

    def prove_many(self, seeds: Sequence[int], workers: int = 4, **kw) -> list:
        """Proofs of the same configuration for every seed of `seeds`, `workers` of them in flight, each on its own
        HIP stream (and host thread).  A single proof leaves the GPU idle between its many dependent launches -- the
        small upper Merkle levels, the channel round trips -- and other proofs' kernels fill those gaps; nothing else
        changes, so every proof is byte-identical to `prove_proof(seed=...)`.  Returns the StwoProofs in seed order;
        `self.timings["proofs_per_s"]` is the rate of the call."""
        import itertools
        import threading
        torch = self.torch
        seeds = list(seeds)
        out: list = [None] * len(seeds)
        if not seeds:
            return out
        t0 = time.perf_counter()
        out[0] = self.prove_proof(seed=seeds[0], **kw)  # also fills the twiddle cache every worker shares
        nxt = itertools.count(1)
        errors: list = []

        # worker provers and their streams live as long as this prover: the caching allocator keeps a pool per
        # stream, so a fresh stream per call would start every call with gigabytes of hipMalloc
        while len(self._workers) < workers:
            self._workers.append((self if not self._workers else GpuProver(self.ver), torch.cuda.Stream(device=self.dev)))

        def work(w: int) -> None:
            gp, stream = self._workers[w]
            gp._dom = self._dom  # read-only from here on
            try:
                with torch.cuda.stream(stream):
                    while not errors:
                        i = next(nxt)
                        if i >= len(seeds):
                            return
                        out[i] = gp.prove_proof(seed=seeds[i], **kw)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=work, args=(w,)) for w in range(max(1, min(workers, len(seeds) - 1)))]
        # A worker that wakes from a stream synchronisation needs the interpreter lock back; with CPython's default
        # 5 ms switch interval it can wait that long for a thread that is busy converting arrays, which is a third
        # of a whole proof.  Ask for the lock every 50 us while the workers run.
        import sys
        interval = sys.getswitchinterval()
        sys.setswitchinterval(5e-5)
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            sys.setswitchinterval(interval)
        if errors:
            raise errors[0]
        torch.cuda.synchronize(self.dev)
        dt = time.perf_counter() - t0
        self.timings = {"total": dt, "proofs_per_s": len(seeds) / dt, "workers": len(threads)}
        return out

    def prove_minimal(self, **kw):
        """The proof with ONE decommitment per tree -- queries sorted and deduplicated, only the siblings and fold
        partners the verifier cannot compute (formats.StwoMinimalProof; what upstream stwo's prover sends, where the
        reference's format repeats a full path per query: fri/queries.simf:41, scripts/generate_wit.py:36-42).  The
        prover holds every node anyway, so this is prove_proof() minus what is not sent."""
        from .formats import stwo_minimise
        proof = self.prove_proof(**kw)
        return stwo_minimise(proof, self.queries)

    def prove(self, n_cols: int = 4, trace_log: int = 9, log_blowup: int = 4, n_queries: int = 16,
              pow_bits: int = 5, seed: int = 0, hash: str = "sha256") -> dict:
        """The same proof in the reference's `proof.json` schema (format C): `prove_proof` plus the
        conversion of 170 KB of hashes into JSON's lists of byte values (3-4 ms of pure Python at the
        2^20 shape, reported as timings["json"])."""
        from .formats import stwo_to_json
        proof = self.prove_proof(n_cols, trace_log, log_blowup, n_queries, pow_bits, seed, hash)
        t0 = time.perf_counter()
        out = stwo_to_json(proof)
        self.timings["json"] = time.perf_counter() - t0
        self.timings["total"] += self.timings["json"]
        return out
