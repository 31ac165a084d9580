#!/usr/bin/env python3
"""Honest wide-Fibonacci circle-STARK prover emitting the reference's stwo `proof.json`.

The reference repository contains no stwo prover: its two proofs
(stwo-verifier/tests/data/proof.json, proof_test.json) came from an external, forked stwo
(SHA-256 channel, composition polynomial committed as 16 partition columns).  This tool
re-derives that prover from the verifier it has to satisfy (stwo-verifier/src/**/*.simf) and
from docs/{commitments,batching_samples,quotients}.md, so that valid proofs exist for
configurations other than the two shipped ones (BASELINE.json configs 3-5).

It is pinned by reproducing BOTH reference fixtures byte for byte
(`python tools/stwo_prover.py --self-check`, also run by tests/test_prover.py):
    prove(n_cols=4, trace_log=3, log_blowup=1, n_queries=1,  pow_bits=5) == proof_test.json
    prove(n_cols=4, trace_log=9, log_blowup=4, n_queries=16, pow_bits=5) == proof.json

numpy on the CPU; every array op is over M31 = 2^31 - 1 in uint64.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

P = (1 << 31) - 1
U = np.uint64

# ------------------------------------------------------------------------------- M31


def madd(a, b):
    return (a + b) % P


def msub(a, b):
    return (a + (P - b)) % P


def mmul(a, b):
    return (a * b) % P


def mpow(a, e: int):
    r = np.ones_like(a)
    base = a.copy()
    while e:
        if e & 1:
            r = mmul(r, base)
        base = mmul(base, base)
        e >>= 1
    return r


def minv(a):
    return mpow(a, P - 2)


def ipow(a: int, e: int) -> int:
    return pow(a, e, P)


# ------------------------------------------------------------- QM31 on (..., 4) uint64 arrays
def q(a, b=0, c=0, d=0):
    return np.array([a, b, c, d], dtype=U)


def cmul(ar, ai, br, bi):
    return msub(mmul(ar, br), mmul(ai, bi)), madd(mmul(ar, bi), mmul(ai, br))


def qmul(x, y):
    xa, xb, xc, xd = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    ya, yb, yc, yd = y[..., 0], y[..., 1], y[..., 2], y[..., 3]
    # (A + B u)(C + D u), u^2 = 2 + i
    acr, aci = cmul(xa, xb, ya, yb)
    bdr, bdi = cmul(xc, xd, yc, yd)
    rr, ri = cmul(bdr, bdi, U(2), U(1))
    adr, adi = cmul(xa, xb, yc, yd)
    bcr, bci = cmul(xc, xd, ya, yb)
    return np.stack([madd(acr, rr), madd(aci, ri), madd(adr, bcr), madd(adi, bci)], axis=-1)


def qadd(x, y):
    return (x + y) % P


def qsub(x, y):
    return (x + (P - y)) % P


def qscale(x, m):
    """QM31 array times M31 array/scalar."""
    return (x * np.asarray(m, dtype=U)[..., None]) % P


def cinv(ar, ai):
    n = minv(madd(mmul(ar, ar), mmul(ai, ai)))
    return mmul(ar, n), mmul(msub(np.zeros_like(ai), ai), n)


def qinv(x):
    a, b, c, d = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    ar2r, ar2i = cmul(a, b, a, b)
    ai2r, ai2i = cmul(c, d, c, d)
    # den = A^2 - (2 + i) B^2
    tr, ti = cmul(ai2r, ai2i, U(2), U(1))
    dr, di = msub(ar2r, tr), msub(ar2i, ti)
    ir, ii = cinv(dr, di)
    rr, ri = cmul(a, b, ir, ii)
    nr, ni = cmul(msub(np.zeros_like(c), c), msub(np.zeros_like(d), d), ir, ii)
    return np.stack([rr, ri, nr, ni], axis=-1)


def q_from_m(m):
    z = np.zeros_like(m)
    return np.stack([m, z, z, z], axis=-1)


# ---------------------------------------------------------------------- circle group
GEN = (2, 1268011823)


def padd(p, q_):
    (x0, y0), (x1, y1) = p, q_
    return msub(mmul(x0, x1), mmul(y0, y1)), madd(mmul(x0, y1), mmul(y0, x1))


def point_of_index(idx: int):
    rx, ry = 1, 0
    cx, cy = GEN
    for i in range(31):
        if (idx >> i) & 1:
            rx, ry = (rx * cx - ry * cy) % P, (rx * cy + ry * cx) % P
        cx, cy = (2 * cx * cx - 1) % P, (2 * cx * cy) % P
    return rx, ry


def coset_points(offset: int, step: int, log_n: int):
    """Points (offset + j * step) * G for j < 2^log_n, natural order, as uint64 arrays."""
    x = np.array([point_of_index(offset)[0]], dtype=U)
    y = np.array([point_of_index(offset)[1]], dtype=U)
    for b in range(log_n):
        sx, sy = point_of_index((step << b) & 0x7FFFFFFF)
        nx, ny = padd((x, y), (U(sx), U(sy)))
        x, y = np.concatenate([x, nx]), np.concatenate([y, ny])
    return x, y


def bitrev_perm(log_n: int) -> np.ndarray:
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint32)
    r = np.zeros(n, dtype=np.uint32)
    for b in range(log_n):
        r |= ((idx >> b) & 1) << (log_n - 1 - b)
    return r


class Domain:
    """Canonic coset of log size m (groups/circle_domain.simf:17-25) with FFT twiddles.
    Storage order is bit-reversed: value[i] = f(at(bitrev(i)))."""

    def __init__(self, m: int):
        self.m = m
        half = m - 1
        # half coset: offset 2^(30-m), step 2^(32-m), natural order then bit-reversed
        hx, hy = coset_points(1 << (30 - m), (1 << (32 - m)) & 0x7FFFFFFF, half)
        br = bitrev_perm(half)
        self.hx, self.hy = hx[br], hy[br]  # point of storage pair h = (hx[h], +-hy[h])
        # twiddles: layer 0 -> y_h ; layer i >= 1 -> pi^(i-1)(x_{h * 2^i})
        self.tw = [self.hy]
        cur = self.hx
        for _ in range(1, m):
            cur = cur[::2]
            self.tw.append(cur.copy())
            cur = msub(madd(mmul(cur, cur), mmul(cur, cur)), U(1))
        self.itw = None

    def points(self):
        """(x, y) of every storage position."""
        x = np.repeat(self.hx, 2)
        y = np.empty_like(x)
        y[0::2] = self.hy
        y[1::2] = msub(np.zeros_like(self.hy), self.hy)
        return x, y

    def _itw(self):
        if self.itw is None:
            self.itw = [minv(t) for t in self.tw]
        return self.itw


def ifft(vals: np.ndarray, dom: Domain) -> np.ndarray:
    """Evaluations (storage order) -> coefficients in the basis y^k0 x^k1 pi(x)^k2 ..."""
    m = dom.m
    v = vals.astype(U).copy()
    itw = dom._itw()
    for i in range(m):
        a = v.reshape(-1, 2, 1 << i)
        t = itw[i][:, None]
        v0, v1 = a[:, 0, :].copy(), a[:, 1, :].copy()
        a[:, 0, :] = madd(v0, v1)
        a[:, 1, :] = mmul(msub(v0, v1), t)
    return mmul(v, U(ipow(1 << m, P - 2)))


def fft(coeffs: np.ndarray, dom: Domain) -> np.ndarray:
    """Coefficients (length <= 2^m, zero extended) -> evaluations in storage order."""
    m = dom.m
    v = np.zeros(1 << m, dtype=U)
    v[:len(coeffs)] = coeffs
    # a polynomial of smaller log size only uses the low basis elements; its coefficient of
    # index k keeps its meaning because the basis is ordered by bit significance
    for i in range(m - 1, -1, -1):
        a = v.reshape(-1, 2, 1 << i)
        t = dom.tw[i][:, None]
        v0, v1t = a[:, 0, :].copy(), mmul(a[:, 1, :], t)
        a[:, 0, :] = madd(v0, v1t)
        a[:, 1, :] = msub(v0, v1t)
    return v


def eval_at_qpoint(coeffs: np.ndarray, px: np.ndarray, py: np.ndarray) -> np.ndarray:
    """sum_k c_k y^k0 x^k1 pi(x)^k2 ...  at a QM31 point."""
    m = int(np.log2(len(coeffs)))
    vals = q_from_m(coeffs.astype(U))
    factors = [py, px]
    cur = px
    for _ in range(2, m):
        cur = qsub(qadd(qmul(cur, cur), qmul(cur, cur)), q(1))
        factors.append(cur)
    for b in range(m):
        vals = qadd(vals[0::2], qmul(vals[1::2], factors[b][None, :]))
    return vals[0]


# ------------------------------------------------------------------------- hashing
def be_words(arr: np.ndarray) -> bytes:
    return np.ascontiguousarray(arr, dtype=">u4").tobytes()


HASHES = {"sha256": hashlib.sha256, "blake2s": lambda b=b"": hashlib.blake2s(b, digest_size=32)}
_H = hashlib.sha256  # hash family of the running proof (set by prove())


def hash_rows(rows: np.ndarray) -> np.ndarray:
    """rows uint[n, w] -> hash of each row's big-endian words, uint8[n, 32]."""
    n, w = rows.shape
    buf = be_words(rows)
    sha = _H
    step = 4 * w
    out = b"".join([sha(buf[i:i + step]).digest() for i in range(0, n * step, step)])
    return np.frombuffer(out, dtype=np.uint8).reshape(n, 32)


def merkle_levels(leaves: np.ndarray):
    """[leaf level, ..., root level]; node = sha256(left || right)."""
    levels = [leaves]
    sha = _H
    cur = leaves
    while len(cur) > 1:
        buf = cur.tobytes()
        out = b"".join([sha(buf[i:i + 64]).digest() for i in range(0, len(buf), 64)])
        cur = np.frombuffer(out, dtype=np.uint8).reshape(-1, 32)
        levels.append(cur)
    return levels


def merkle_path(levels, index: int, skip: int = 0):
    """Sibling hashes leaf -> root for `index` at level `skip`."""
    out = []
    idx = index
    for lv in levels[skip:-1]:
        out.append(lv[idx ^ 1])
        idx >>= 1
    return out


class Channel:
    """stwo-verifier/src/channel.simf:31-172."""

    def __init__(self):
        self.digest = bytes(32)
        self.counter = 0

    def mix_u256(self, b: bytes):
        self.digest = _H(self.digest + b).digest()
        self.counter = 0

    def mix_bytes(self, b: bytes):
        self.mix_u256(b)

    def draw_words(self):
        d = _H(self.digest + self.counter.to_bytes(4, "big")).digest()
        self.counter += 1
        return [int.from_bytes(d[4 * i:4 * i + 4], "big") for i in range(8)]

    def draw_qm31(self):
        while True:
            w = self.draw_words()
            if all(x < 4294967294 for x in w[:4]):
                return q(*[x % P for x in w[:4]])


def prove(n_cols: int = 4, trace_log: int = 9, log_blowup: int = 4, n_queries: int = 16,
          pow_bits: int = 5, verbose: bool = False, seed: int = 0, hash: str = "sha256") -> dict:
    """seed = 0 is the external prover's trace (row r starts 1, r); other seeds start row r at
    (1, r + seed * 0x9E3779B1) -- any start satisfies the wide-Fibonacci transition constraints
    (constraints/wide_fibonacci.simf:24-38), so every seed gives a distinct valid proof."""
    global _H
    _H = HASHES[hash]
    t_start = time.time()

    def log(msg):
        if verbose:
            print("[%.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)
    n, L = trace_log, trace_log + log_blowup
    N = n_cols
    K = L - 1 - log_blowup
    ch = Channel()

    # ---- trace: row r = [1, r, 1 + r^2, ...]; stored vector = [row0, row1, ...]
    r = np.arange(1 << n, dtype=U)
    cols = [np.ones(1 << n, dtype=U), (r + U((seed * 0x9E3779B1) % P)) % P]
    for k in range(2, N):
        cols.append(madd(mmul(cols[k - 1], cols[k - 1]), mmul(cols[k - 2], cols[k - 2])))
    dom_n, dom_L, dom_c = Domain(n), Domain(L), Domain(n + 1)
    log("domains")
    coefs = [ifft(c, dom_n) for c in cols]
    lde = np.stack([fft(c, dom_L) for c in coefs], axis=1)  # [2^L, N]
    log("trace LDE")
    trace_tree = merkle_levels(hash_rows(lde))
    const_root = _H(b"").digest()
    trace_root = trace_tree[-1][0].tobytes()
    log("trace tree")
    ch.mix_u256(const_root)
    ch.mix_u256(trace_root)
    cp_alpha = ch.draw_qm31()

    # ---- composition polynomial on the canonic coset of log size n + 1
    ev = [fft(c, dom_c) for c in coefs]
    cx, _ = dom_c.points()
    van = cx.copy()
    for _ in range(n - 1):
        van = msub(madd(mmul(van, van), mmul(van, van)), U(1))
    van_inv = minv(van)
    acc = np.zeros((1 << (n + 1), 4), dtype=U)
    for k in range(2, N):
        cons = msub(ev[k], madd(mmul(ev[k - 1], ev[k - 1]), mmul(ev[k - 2], ev[k - 2])))
        acc = qadd(qmul(acc, cp_alpha[None, :]), q_from_m(cons))
    F = qscale(acc, van_inv)
    cp_coefs = []  # 16 circle polynomials of log size n, index 4 * coord + part
    for c in range(4):
        fc = ifft(F[:, c], dom_c)
        for part in range(4):
            a = fc[part::4]  # line-basis coefficients of the part polynomial
            cc = np.zeros(1 << n, dtype=U)
            cc[0::2] = a       # circle basis with k0 = 0: a polynomial in x only
            cp_coefs.append(cc)
    cp_lde = np.stack([fft(c, dom_L) for c in cp_coefs], axis=1)  # [2^L, 16]
    log("composition LDE")
    cp_tree = merkle_levels(hash_rows(cp_lde))
    cp_root = cp_tree[-1][0].tobytes()
    ch.mix_u256(cp_root)
    log("composition tree")

    # ---- OODS
    t = ch.draw_qm31()
    t_sq = qmul(t, t)
    inv = qinv(qadd(q(1), t_sq))
    px = qmul(qsub(q(1), t_sq), inv)
    py = qmul(qadd(t, t), inv)
    p2x = qsub(qadd(qmul(px, px), qmul(px, px)), q(1))
    p2y = qadd(qmul(px, py), qmul(px, py))
    oods_trace = np.stack([eval_at_qpoint(c, px, py) for c in coefs])
    oods_cp = np.stack([eval_at_qpoint(c, p2x, p2y) for c in cp_coefs])
    msg = b"".join(be_words(v) for v in oods_trace) + b"".join(be_words(v) for v in oods_cp)
    ch.mix_bytes(msg)
    deep_alpha = ch.draw_qm31()
    log("oods")

    # ---- DEEP quotients over the LDE domain (fixture semantics)
    lx, ly = dom_L.points()

    def batch_row(sx, sy, samples, values):
        # deep/quotients.simf:15-44 with the line through the sample point and its conjugate
        prx, pix = sx[0:2], sx[2:4]
        pry, piy = sy[0:2], sy[2:4]
        dxr, dxi = msub(prx[0], lx), np.full_like(lx, prx[1])
        dyr, dyi = msub(pry[0], ly), np.full_like(ly, pry[1])
        d1r, d1i = cmul(dxr, dxi, piy[0], piy[1])
        d2r, d2i = cmul(dyr, dyi, pix[0], pix[1])
        ir, ii = cinv(msub(d1r, d2r), msub(d1i, d2i))
        num = np.zeros((len(lx), 4), dtype=U)
        alpha_i = deep_alpha.copy()
        for k in range(len(samples)):
            val = samples[k]
            zero2 = np.zeros(2, dtype=U)
            a = np.concatenate([zero2, msub(zero2, madd(val[2:4], val[2:4]))])
            b = np.concatenate([zero2, msub(zero2, madd(sy[2:4], sy[2:4]))])
            c = qsub(qmul(b, val), qmul(a, sy))
            a, b, c = qmul(alpha_i, a), qmul(alpha_i, b), qmul(alpha_i, c)
            term = qsub(qscale(b[None, :], values[:, k]), qadd(qscale(a[None, :], ly), c[None, :]))
            num = qadd(num, term)
            alpha_i = qmul(alpha_i, deep_alpha)
        # multiply by the CM31 denominator inverse
        n0r, n0i = cmul(num[:, 0], num[:, 1], ir, ii)
        n1r, n1i = cmul(num[:, 2], num[:, 3], ir, ii)
        return np.stack([n0r, n0i, n1r, n1i], axis=-1)

    b1 = batch_row(px, py, oods_trace, lde)
    b2 = batch_row(p2x, p2y, oods_cp, cp_lde)
    a16 = q(1)
    for _ in range(16):
        a16 = qmul(a16, deep_alpha)
    layer = qadd(qmul(b1, a16[None, :]), b2)  # [2^L, 4]
    log("quotients")

    # ---- FRI commit
    fri_layers, fri_trees, fri_roots, fold_alphas = [], [], [], []
    coord = dom_L.hy  # y of the even member of each storage pair
    xs = dom_L.hx
    for l in range(K + 1):
        fri_layers.append(layer)
        tree = merkle_levels(hash_rows(layer))
        fri_trees.append(tree)
        root = tree[-1][0].tobytes()
        fri_roots.append(root)
        ch.mix_u256(root)
        alpha = ch.draw_qm31()
        fold_alphas.append(alpha)
        v0, v1 = layer[0::2], layer[1::2]
        cinv_ = minv(coord)
        f0 = qadd(v0, v1)
        f1 = qscale(qsub(v0, v1), cinv_)
        layer = qadd(f0, qmul(f1, alpha[None, :]))
        # next layer's coordinates: x of the even member of each pair of the folded domain
        if l == 0:
            coord = xs[0::2]
        else:
            nxt = msub(madd(mmul(coord, coord), mmul(coord, coord)), U(1))
            coord = nxt[0::2]
        log("fri layer %d" % l)
    if not (layer == layer[0]).all():
        raise AssertionError("last FRI layer is not constant: the quotient is not low degree")
    last = layer[0]
    ch.mix_bytes(be_words(last))

    # ---- proof of work (pow.simf:22-36): smallest nonce with LE64(last 8 digest bytes) < target
    target = (1 << (64 - pow_bits)) - 1
    nonce = 0
    while True:
        d = _H(ch.digest + nonce.to_bytes(8, "big")).digest()
        if int.from_bytes(d[24:32], "little") < target:
            break
        nonce += 1
    ch.digest, ch.counter = d, 0

    # ---- queries (fri/queries.simf:29-43), no sort / dedup
    mask = (1 << L) - 1
    queries = []
    while len(queries) < n_queries:
        queries += [w & mask for w in ch.draw_words()]
    queries = queries[:n_queries]
    log("queries %s" % queries[:4])

    # ---- decommit
    def hw(nodes):
        return [[int(b) for b in node] for node in nodes]
    trace_hw, cp_hw, trace_q, cp_q = [], [], [], []
    for qi in queries:
        trace_q += [int(v) for v in lde[qi]]
        cp_q += [int(v) for v in cp_lde[qi]]
        trace_hw += hw(merkle_path(trace_tree, qi))
        cp_hw += hw(merkle_path(cp_tree, qi))

    def qj(v):
        return [[int(v[0]), int(v[1])], [int(v[2]), int(v[3])]]
    fri_json = []
    cur = list(queries)
    for l in range(K + 1):
        wit, hwl = [], []
        for j, qi in enumerate(cur):
            wit.append(qj(fri_layers[l][qi ^ 1]))
            hwl += hw(merkle_path(fri_trees[l], qi >> 1, skip=1))
            cur[j] = qi >> 1
        fri_json.append({"fri_witness": wit,
                         "decommitment": {"hash_witness": hwl, "column_witness": []},
                         "commitment": [int(b) for b in fri_roots[l]]})
    log("decommit")
    conf = {"pow_bits": pow_bits,
            "fri_config": {"log_blowup_factor": log_blowup, "log_last_layer_degree_bound": 0,
                           "n_queries": n_queries}}
    if hash != "sha256":
        conf["hash"] = hash  # extension: the reference's proofs are always SHA-256
    return {
        "config": conf,
        "commitments": [[int(b) for b in const_root], [int(b) for b in trace_root],
                        [int(b) for b in cp_root]],
        "sampled_values": [[], [[qj(v)] for v in oods_trace], [[qj(v)] for v in oods_cp]],
        "decommitments": [{"hash_witness": [], "column_witness": []},
                          {"hash_witness": trace_hw, "column_witness": []},
                          {"hash_witness": cp_hw, "column_witness": []}],
        "queried_values": [[], trace_q, cp_q],
        "proof_of_work": nonce,
        "fri_proof": {"first_layer": fri_json[0], "inner_layers": fri_json[1:],
                      "last_layer_poly": {"coeffs": [qj(last)], "log_size": 0}},
    }


def self_check(golden_dir: str) -> None:
    for name, kw in (("stwo_proof_test.json", dict(trace_log=3, log_blowup=1, n_queries=1)),
                     ("stwo_proof.json", dict(trace_log=9, log_blowup=4, n_queries=16))):
        want = json.load(open(os.path.join(golden_dir, name)))
        got = prove(n_cols=4, pow_bits=5, **kw)
        for key in want:
            if got[key] != want[key]:
                raise SystemExit("%s: field %r differs" % (name, key))
        print("%s reproduced byte for byte" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--self-check", action="store_true")
    ap.add_argument("--n-cols", type=int, default=4)
    ap.add_argument("--trace-log", type=int, default=9)
    ap.add_argument("--log-blowup", type=int, default=4)
    ap.add_argument("--n-queries", type=int, default=16)
    ap.add_argument("--pow-bits", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hash", choices=["sha256", "blake2s"], default="sha256")
    ap.add_argument("--minimal", action="store_true",
                    help="print the minimal proof.json: one sorted, deduplicated decommitment per tree, as upstream stwo's "
                         "prover sends it (formats.stwo_minimal_to_json), instead of one path per query")
    ap.add_argument("-o", "--out", default="-")
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    if args.self_check:
        self_check(os.path.join(here, "..", "tests", "golden"))
        return
    proof = prove(args.n_cols, args.trace_log, args.log_blowup, args.n_queries, args.pow_bits,
                  verbose=True, seed=args.seed, hash=args.hash)
    if args.minimal:  # a selection from the per-query proof: nothing new is computed
        sys.path.insert(0, os.path.join(here, ".."))
        from stark_symphony_amd import formats
        proof = formats.stwo_minimal_to_json(formats.stwo_minimise(formats.stwo_from_json(proof, trace_log=args.trace_log)))
    text = json.dumps(proof)
    if args.out == "-":
        print(text)
    else:
        with open(args.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
