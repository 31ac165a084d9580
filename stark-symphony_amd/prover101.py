"""GPU stark101 (FibonacciSq) prover.

Mirrors ``prove()`` of the reference's Python prover (stark101/scripts/fibsquare/prover.py:94-171)
and returns its ``res`` dict (format A, ``proof.json``) -- for the reference's seed the same bytes as
tests/golden/stark101_proof.json, which the reference's own prover wrote.  Everything proportional
to the domain runs in the ss_p101_* / ss_p_hash_rows / ss_p_merkle kernels (include/ss_prover.h);
the Fiat-Shamir channel (channel.py:41-96, three dozen hashes) stays on the host.

The reference proves in ~16 s of Python big-int polynomial arithmetic; the GPU path needs no
polynomial division at all (see csrc/ss_s101_prover.hip for why the values are the same).

Only the reference's seed gives a proof its verifier accepts: the boundary value 2338775057 is
hard-coded in both prover.py:44 and air.simf:63.  ``seed`` is still a parameter -- the proof of
another seed carries its own claim a_1022 in ``Stark101GpuProver.claim`` and the verifiers reject it
at the composition-polynomial check, which tests/test_gpu_prover.py uses as a negative case.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import List, Sequence

import numpy as np

from . import binding as B

P101 = 3 * 2 ** 30 + 1
REFERENCE_SEED = 3141592       # prover.py:27
REFERENCE_CLAIM = 2338775057   # prover.py:44
LDE = 8192
BLOWUP = 8


class _Channel:
    """channel.py:41-96."""

    def __init__(self) -> None:
        self.state = b""

    def mix(self, data: bytes) -> None:
        self.state = hashlib.sha256(self.state + data).digest()

    def random_int(self, lo: int, hi: int) -> int:
        num = lo + int.from_bytes(self.state, "big") % (hi - lo + 1)
        self.state = hashlib.sha256(self.state).digest()
        return num

    def field_element(self) -> int:
        return self.random_int(0, P101 - 1)


class Stark101GpuProver:
    """Proves on one MI355X; ``ver`` is a verifier.Verifier (owns the library context and device)."""

    def __init__(self, ver):
        import torch
        self.torch = torch
        self.ver = ver
        self.dev = ver.device
        self.lib = B.lib()
        vp = C.c_void_p
        for name, args in (("ss_p101_trace_poly", (C.c_uint32, vp, vp)),
                           ("ss_p101_lde", (vp, vp)),
                           ("ss_p101_composition", (vp, vp, C.c_uint32, vp)),
                           ("ss_p101_fold", (C.c_uint32, C.c_uint32, C.c_uint32, vp, vp)),
                           ("ss_p_hash_rows", (C.c_uint32, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp)),
                           ("ss_p_merkle", (C.c_uint32, C.c_size_t, vp))):
            f = getattr(self.lib, name)
            f.restype = C.c_int
            f.argtypes = [vp] + list(args) + [vp]
        self.claim = REFERENCE_CLAIM
        self.n_fri_layers = 0

    def _call(self, name: str, *args) -> None:
        stream = int(self.torch.cuda.current_stream(self.dev).cuda_stream)
        B.check(getattr(self.lib, name)(self.ver.ctx, *args, stream))

    def _empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int32, device=self.dev)

    def _commit(self, values, n: int) -> np.ndarray:
        """merkle.py:29-66 over n = 2^k field elements -> host levels uint8[2n - 1, 32] (leaves first)."""
        levels = self._empty(2 * n, 8)
        self._call("ss_p_hash_rows", 0, n, 1, values.data_ptr(), n, levels.data_ptr())
        if n > 1:
            self._call("ss_p_merkle", 0, n, levels.data_ptr())
        host = levels[:2 * n - 1].cpu().numpy().view(np.uint32)
        return host.astype(">u4").view(np.uint8).reshape(2 * n - 1, 32)

    @staticmethod
    def _path(levels: np.ndarray, n: int, index: int) -> List[int]:
        """Siblings leaf -> root (merkle.py:38-52 returns root -> leaf; prover.py:143 reverses)."""
        out, off, size, idx = [], 0, n, index
        while size > 1:
            out.append(int.from_bytes(bytes(levels[off + (idx ^ 1)]), "big"))
            off += size
            size >>= 1
            idx >>= 1
        return out

    def prove(self, seed: int = REFERENCE_SEED) -> dict:
        torch = self.torch
        ch = _Channel()
        trace, coef = self._empty(1024), self._empty(1024)
        self._call("ss_p101_trace_poly", seed % P101, trace.data_ptr(), coef.data_ptr())
        p_ev = self._empty(LDE)
        self._call("ss_p101_lde", coef.data_ptr(), p_ev.data_ptr())
        p_tree = self._commit(p_ev, LDE)
        self.claim = int(trace[1022].item()) & 0xFFFFFFFF
        ch.mix(bytes(p_tree[-1]))

        alphas = (C.c_uint32 * 3)(ch.field_element(), ch.field_element(), ch.field_element())
        cp_ev = self._empty(LDE)
        self._call("ss_p101_composition", p_ev.data_ptr(), alphas, self.claim, cp_ev.data_ptr())
        layers = [cp_ev]
        trees = [self._commit(cp_ev, LDE)]
        ch.mix(bytes(trees[0][-1]))

        # prover.py:128-137: fold until the polynomial is constant; that layer's root is not sent.
        betas: List[int] = []
        while True:
            beta = ch.field_element()
            cur = layers[-1]
            nxt = self._empty(cur.numel() // 2)
            self._call("ss_p101_fold", len(layers) - 1, cur.numel(), beta, cur.data_ptr(), nxt.data_ptr())
            betas.append(beta)
            layers.append(nxt)
            constant = bool((nxt == nxt[0]).all().item())
            if constant or nxt.numel() == 1:
                if not constant:
                    raise ValueError("composition polynomial has degree >= the coset size")
                break
            trees.append(self._commit(nxt, nxt.numel()))
            ch.mix(bytes(trees[-1][-1]))
        last = int(layers[-1][0].item()) & 0xFFFFFFFF
        ch.mix(last.to_bytes(4, "big"))
        self.n_fri_layers = len(layers) - 1

        idx = ch.random_int(0, LDE - 1)
        if idx + 2 * BLOWUP >= LDE:
            raise IndexError("query %d: prover.py:145-146 index f(gx), f(g^2 x) without wrapping" % idx)
        p_host = p_ev.cpu().numpy().view(np.uint32)
        res = {"p_mt_root": int.from_bytes(bytes(p_tree[-1]), "big"),
               "evals": [[int(p_host[i]), self._path(p_tree, LDE, i)]
                         for i in (idx, idx + BLOWUP, idx + 2 * BLOWUP)],
               "fri_layers": []}
        for i in range(len(layers) - 1):
            length = layers[i].numel()
            host = layers[i].cpu().numpy().view(np.uint32)
            a, b = idx % length, (idx + length // 2) % length
            res["fri_layers"].append([int.from_bytes(bytes(trees[i][-1]), "big"), betas[i],
                                      int(host[a]), self._path(trees[i], length, a),
                                      int(host[b]), self._path(trees[i], length, b)])
        res["fri_last_layer"] = last
        return res


def trace_reference(seed: int = REFERENCE_SEED) -> Sequence[int]:
    """prover.py:25-30 on Python ints (tests)."""
    t = [1, seed % P101]
    while len(t) < 1023:
        t.append((t[-2] * t[-2] + t[-1] * t[-1]) % P101)
    return t
