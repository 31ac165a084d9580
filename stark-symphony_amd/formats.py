"""Proof / witness formats of the reference, parsed into plain host-side records.

This is the drop-in boundary on the *input* side: the reference's
callers hand a verifier one of four files, and this module reads (and writes) all
of them without touching the reference's code:

  A  stark101 ``proof.json``   written by ``python -m fibsquare``
        (stark101/scripts/fibsquare/prover.py:108,143-147,150-167, __main__.py:8-11)
  B  stark101 ``proof.wit``    (stark101/scripts/generate_wit.py:13-30; read by
        stark101/src/main.simf:13-16 as witness::P_MT_ROOT / P_EVALS / FRI_LAYERS /
        FRI_LAST_LAYER)
  C  stwo ``proof.json``       (schema consumed by stwo-verifier/scripts/generate_wit.py:106-245)
  D  stwo ``proof.wit``        (stwo-verifier/scripts/generate_wit.py:218-243; read by
        stwo-verifier/src/main.simf:10-15)

A ``.wit`` file is a JSON object ``{NAME: {"value": <SimplicityHL literal>, "type": ...}}``;
the literal grammar needed here is integers (decimal or 0x-hex), tuples ``( .. )``,
arrays ``[ .. ]`` and ``list![ .. ]``.

Hashes are kept as 32 raw bytes in SHA-256 output order (== the big-endian u256 the
reference prints).  Field elements are raw ``u32`` words exactly as given -- the
reference never range-checks them (SURVEY.md 7 "Unreduced / adversarial witness
values"), so neither does the parser.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Any, List, Sequence, Tuple

import numpy as np

MAX_LIST = 31  # SimplicityHL List<T, 32> holds strictly fewer than 32 elements


class MalformedProof(ValueError):
    """The input cannot be typed as the reference's witness (``simfony run`` exits 1 with a
    type error before executing anything: simfony-cli/src/main.rs:77-81,187-190)."""


# --------------------------------------------------------------------------- helpers
def u256_to_bytes(v: int) -> bytes:
    if not (0 <= v < 1 << 256):
        raise MalformedProof("u256 out of range")
    return int(v).to_bytes(32, "big")


def bytes_to_u256(b: bytes) -> int:
    return int.from_bytes(b, "big")


def _u32(v: Any) -> int:
    # integers only: int() would also take 1.5 or "7", which no witness of the reference can hold
    # (generate_wit.py prints the value into a u32 literal)
    if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
        raise MalformedProof("u32 expected, got %r" % (v,))
    v = int(v)
    if not (0 <= v < 1 << 32):
        raise MalformedProof("u32 out of range: %r" % (v,))
    return v


def _uint(v: Any, bits: int) -> int:
    if isinstance(v, bool) or not isinstance(v, (int, np.integer)) or not (0 <= int(v) < 1 << bits):
        raise MalformedProof("u%d expected, got %r" % (bits, v))
    return int(v)


def _path(nodes: Sequence[Any]) -> np.ndarray:
    """List of u256 ints / 32-byte sequences -> uint8[len, 32] (leaf -> root order)."""
    if len(nodes) > MAX_LIST:
        raise MalformedProof("List<u256, 32> holds at most 31 elements")
    out = np.zeros((len(nodes), 32), dtype=np.uint8)
    for i, n in enumerate(nodes):
        if isinstance(n, int):
            out[i] = np.frombuffer(u256_to_bytes(n), dtype=np.uint8)
        else:
            b = bytes(n)
            if len(b) != 32:
                raise MalformedProof("expected 32 bytes")
            out[i] = np.frombuffer(b, dtype=np.uint8)
    return out


# ----------------------------------------------------------- SimplicityHL literal parser
def parse_literal(text: str) -> Any:
    """Parse a SimplicityHL value literal into nested Python lists / ints."""
    if not isinstance(text, str):
        raise MalformedProof("a witness value is a string holding a literal")
    pos = 0
    n = len(text)

    def skip() -> None:
        nonlocal pos
        while pos < n and text[pos] in " \t\r\n":
            pos += 1

    def value() -> Any:
        nonlocal pos
        skip()
        if pos >= n:
            raise MalformedProof("unexpected end of literal")
        ch = text[pos]
        if text.startswith("list!", pos):
            pos += 5
            skip()
            if pos >= n or text[pos] != "[":
                raise MalformedProof("expected '[' after list!")
            return seq("]")
        if text.startswith("qm31", pos):  # qm31(a, b, c, d) constructor of the .simf snippets
            pos += 4
            skip()
            if pos >= n or text[pos] != "(":
                raise MalformedProof("expected '(' after qm31")
            w = seq(")")
            if not isinstance(w, list) or len(w) != 4:
                raise MalformedProof("qm31 takes four words")
            return [[w[0], w[1]], [w[2], w[3]]]
        if ch == "(":
            return seq(")")
        if ch == "[":
            return seq("]")
        start = pos
        if text.startswith("0x", pos) or text.startswith("0X", pos):
            pos += 2
            while pos < n and text[pos] in "0123456789abcdefABCDEF_":
                pos += 1
            digits = text[start + 2:pos].replace("_", "")
            if not digits:
                raise MalformedProof("hex literal without digits at %d" % start)
            return int(digits, 16)
        while pos < n and (text[pos] in "0123456789_"):
            pos += 1
        digits = text[start:pos].replace("_", "")
        if not digits:
            raise MalformedProof("unexpected character %r at %d" % (ch, start))
        return int(digits)

    def seq(close: str) -> Any:
        nonlocal pos
        pos += 1  # opening bracket
        items: List[Any] = []
        commas = 0
        while True:
            skip()
            if pos >= n:
                raise MalformedProof("unterminated sequence")
            if text[pos] == close:
                pos += 1
                # "(x)" is a parenthesised value, not a 1-tuple (generate_wit.py:8 wraps
                # every FriLayer in a redundant pair of parentheses)
                if close == ")" and len(items) == 1 and commas == 0:
                    return items[0]
                return items
            items.append(value())
            skip()
            if pos < n and text[pos] == ",":
                pos += 1
                commas += 1

    v = value()
    skip()
    if pos != n:
        raise MalformedProof("trailing characters in literal")
    return v


# ------------------------------------------------------------------------- stark101
@dataclass
class Stark101Eval:
    """``Eval = (u32, MerkleProof32)`` -- stark101/src/air.simf:24."""
    ev: int
    path: np.ndarray  # uint8[len, 32]


@dataclass
class Stark101Layer:
    """``FriLayer`` -- stark101/src/fri.simf:30."""
    root: bytes
    beta: int
    cpa: Stark101Eval
    cpb: Stark101Eval


@dataclass
class Stark101Proof:
    """``FibSquareProof`` -- stark101/src/verifier.simf:17."""
    root: bytes
    evals: List[Stark101Eval]
    layers: List[Stark101Layer]
    last: int

    def copy(self) -> "Stark101Proof":
        return Stark101Proof(
            self.root,
            [Stark101Eval(e.ev, e.path.copy()) for e in self.evals],
            [Stark101Layer(l.root, l.beta, Stark101Eval(l.cpa.ev, l.cpa.path.copy()),
                           Stark101Eval(l.cpb.ev, l.cpb.path.copy())) for l in self.layers],
            self.last)


def _s101_from_parts(root: Any, evals: Any, layers: Any, last: Any) -> Stark101Proof:
    try:
        if len(evals) != 3:
            raise MalformedProof("P_EVALS must hold three evaluations")
        if len(layers) > MAX_LIST:
            raise MalformedProof("List<FriLayer, 32> holds at most 31 layers")
        ev = [Stark101Eval(_u32(e[0]), _path(e[1])) for e in evals]
        ls = []
        for l in layers:
            if len(l) != 6:
                raise MalformedProof("FriLayer has six fields")
            ls.append(Stark101Layer(u256_to_bytes(_uint(l[0], 256)), _u32(l[1]),
                                    Stark101Eval(_u32(l[2]), _path(l[3])),
                                    Stark101Eval(_u32(l[4]), _path(l[5]))))
        return Stark101Proof(u256_to_bytes(_uint(root, 256)), ev, ls, _u32(last))
    except (TypeError, IndexError, KeyError) as e:  # wrong nesting
        raise MalformedProof(str(e)) from e


def stark101_from_json(obj: Any) -> Stark101Proof:
    """Format A: the prover's ``res`` dict / ``target/proof.json``."""
    if isinstance(obj, (str, bytes)):
        obj = json.loads(obj)
    try:
        return _s101_from_parts(obj["p_mt_root"], obj["evals"], obj["fri_layers"],
                                obj["fri_last_layer"])
    except KeyError as e:
        raise MalformedProof("missing key %s" % e) from e


S101_P = 3 * 2 ** 30 + 1


def stark101_from_transcript(messages: Sequence[Any]) -> Stark101Proof:
    """The in-Python caller format: `channel.proof`, the message list that
    fibsquare.prover.prove() returns beside `res` and that the reference's own Python verifier replays
    (stark101/scripts/fibsquare/prover_test.py:32-104): p_mt_root, the FRI layer roots, the last
    layer's constant, then (value, authentication path) per decommitment -- the three trace
    evaluations and cpa / cpb of every layer -- and the constant once more.  A message is `bytes`, a
    FieldElement (anything with `.val`) or int, or a list of `bytes` (paths, ROOT -> LEAF as the
    prover's channel records them; `res` and the records hold them leaf -> root, prover.py:144-146).

    The list carries no betas (the verifier draws them); `res` does (prover.py:150-167), so they are
    re-drawn here exactly as prove() packs them: state = sha256(state || root), beta = state mod p,
    state = sha256(state) (channel.py:48-83).  The GPU verifier still draws its own and compares."""
    import hashlib
    msgs = list(messages)
    pos = 0

    def kind(m: Any) -> str:
        if isinstance(m, (bytes, bytearray)):
            return "bytes"
        if isinstance(m, (list, tuple)):
            return "path"
        return "felt"

    def take(want: str) -> Any:
        nonlocal pos
        if pos >= len(msgs) or kind(msgs[pos]) != want:
            raise MalformedProof("transcript message %d: expected %s" % (pos, want))
        m = msgs[pos]
        pos += 1
        if want == "felt":
            return _u32(getattr(m, "val", m))
        if want == "bytes":
            if len(m) != 32:
                raise MalformedProof("transcript message %d: a commitment is 32 bytes" % (pos - 1))
            return bytes(m)
        return _path([bytes(x) for x in m][::-1])
    root = take("bytes")
    roots: List[bytes] = []
    while pos < len(msgs) and kind(msgs[pos]) == "bytes":
        roots.append(take("bytes"))
    if len(roots) > MAX_LIST:
        raise MalformedProof("List<FriLayer, 32> holds at most 31 layers")
    last = take("felt")
    evals = [Stark101Eval(take("felt"), take("path")) for _ in range(3)]
    state = hashlib.sha256(root).digest()
    for _ in range(3):  # the three composition coefficients (air.simf:30-35)
        state = hashlib.sha256(state).digest()
    layers = []
    for r in roots:
        state = hashlib.sha256(state + r).digest()
        beta = int.from_bytes(state, "big") % S101_P
        state = hashlib.sha256(state).digest()
        cpa = Stark101Eval(take("felt"), take("path"))
        cpb = Stark101Eval(take("felt"), take("path"))
        layers.append(Stark101Layer(r, beta, cpa, cpb))
    if take("felt") != last or pos != len(msgs):
        raise MalformedProof("the transcript must end with the last layer's constant, sent twice with one value")
    return Stark101Proof(root, evals, layers, last)


def stark101_to_json(p: Stark101Proof) -> dict:
    def pth(a: np.ndarray) -> List[int]:
        return [bytes_to_u256(bytes(r)) for r in a]
    return {
        "p_mt_root": bytes_to_u256(p.root),
        "evals": [[e.ev, pth(e.path)] for e in p.evals],
        "fri_layers": [[bytes_to_u256(l.root), l.beta, l.cpa.ev, pth(l.cpa.path), l.cpb.ev,
                        pth(l.cpb.path)] for l in p.layers],
        "fri_last_layer": p.last,
    }


def stark101_from_wit(text: Any) -> Stark101Proof:
    """Format B: ``proof.wit`` (stark101/scripts/generate_wit.py:13-30)."""
    obj = json.loads(text) if isinstance(text, (str, bytes)) else text
    try:
        root = parse_literal(obj["P_MT_ROOT"]["value"])
        evals = parse_literal(obj["P_EVALS"]["value"])
        layers = parse_literal(obj["FRI_LAYERS"]["value"])
        last = parse_literal(obj["FRI_LAST_LAYER"]["value"])
    except (KeyError, TypeError) as e:
        raise MalformedProof("missing witness %s" % e) from e
    return _s101_from_parts(root, evals, layers, last)


def stark101_to_wit(p: Stark101Proof) -> str:
    """Writer for format B, same text the reference's generate_wit.py prints."""
    j = stark101_to_json(p)

    def layer(l: List[Any]) -> str:
        return "((%s, %s, %s, list!%s, %s, list!%s))" % (l[0], l[1], l[2], l[3], l[4], l[5])

    p_evals = ", ".join("(%s, list!%s)" % (x[0], x[1]) for x in j["evals"])
    fri_layers = ", ".join(layer(l) for l in j["fri_layers"])
    res = {
        "P_MT_ROOT": {"value": str(j["p_mt_root"]), "type": "u256"},
        "P_EVALS": {"value": "(%s)" % p_evals,
                    "type": "((u32, List<u256, 32>), (u32, List<u256, 32>), (u32, List<u256, 32>))"},
        "FRI_LAYERS": {"value": "list![%s]" % fri_layers,
                       "type": "List<((u256, u32, u32, List<u256, 32>, u32, List<u256, 32>), 32)"},
        "FRI_LAST_LAYER": {"value": str(j["fri_last_layer"]), "type": "u32"},
    }
    return json.dumps(res, indent=4)


def stark101_to_simf(p: Stark101Proof) -> str:
    """The ``let proof: FibSquareProof = (...);`` snippet stark101/scripts/generate_simf.py prints
    (the literal embedded in stark101/src/verifier.simf:44-388), same text."""
    j = stark101_to_json(p)

    def nodes(path: List[int]) -> str:
        return "list![\n                %s\n            ]" % ",\n                ".join(str(x) for x in path)

    def ev(e: List[Any]) -> str:
        return "        (\n            %d,\n            %s,\n        )," % (e[0], nodes(e[1]))

    def layer(l: List[Any]) -> str:
        return ("\n        (\n            %d,\n            %d,\n            %d,\n            %s,\n            %d,\n"
                "            %s\n        )" % (l[0], l[1], l[2], nodes(l[3]), l[4], nodes(l[5])))
    return ("\nlet proof: FibSquareProof = (\n    %d,\n    (\n%s\n    ),\n    list![\n        %s\n    ],\n    %d\n);\n"
            % (j["p_mt_root"], "\n".join(ev(e) for e in j["evals"]),
               ",\n             ".join(layer(l) for l in j["fri_layers"]), j["fri_last_layer"]))


def stark101_from_simf(text: str) -> Stark101Proof:
    v = _simf_body(text, "FibSquareProof")
    if not isinstance(v, list) or len(v) != 4:
        raise MalformedProof("FibSquareProof has four fields")
    return _s101_from_parts(v[0], v[1], v[2], v[3])


# ----------------------------------------------------------------------------- stwo
N_CP_PARTITIONS = 16  # evals/composition_poly.simf:12


@dataclass(frozen=True)
class StwoConfig:
    """Runtime form of the compile-time macros in stwo-verifier/src/config.simf:10-51."""
    n_cols: int = 4        # NUM_COLUMNS
    trace_log: int = 9     # TRACE_LOG_SIZE
    lde_log: int = 13      # LDE_LOG_SIZE
    n_queries: int = 16    # NUM_FRI_QUERIES
    n_layers: int = 8      # NUM_FRI_LAYERS (inner layers)
    pow_bits: int = 5      # POW_TARGET_64 = 2^(64-pow_bits) - 1, strict '<'
    hash: str = "sha256"   # "sha256" = the reference; "blake2s" = BASELINE.json variant (unpinned)

    @property
    def pow_target(self) -> int:
        return (1 << (64 - self.pow_bits)) - 1

    @property
    def log_blowup(self) -> int:
        return self.lde_log - self.trace_log

    def fri_path_len(self, layer: int) -> int:
        """Merkle path length of FRI layer `layer` (0 = first): fri/layers.simf:40-48."""
        return self.lde_log - 1 - layer

    @property
    def packed_bytes(self) -> int:
        """Algorithmic bytes per proof."""
        N, L, K, Q = self.n_cols, self.lde_log, self.n_layers, self.n_queries
        return (96 + Q * (4 * N + 64 + 64 * L) + 16 * (N + 16) + 32 * (K + 1) + 16
                + Q * sum(16 + 32 * (L - 1 - i) for i in range(K + 1)) + 8)

    @property
    def compressions(self) -> int:
        """Hash compression calls per proof of the reference algorithm."""
        N, L, K, Q = self.n_cols, self.lde_log, self.n_layers, self.n_queries
        b2s = self.hash == "blake2s"

        def blk(n: int) -> int:
            if b2s:  # no padding block: ceil(n / 64), at least one
                return max(1, (n + 63) // 64)
            return (n + 9 + 63) // 64
        pair = blk(64)  # 2 for SHA-256 (padding block), 1 for Blake2s
        per_query = blk(4 * N) + blk(64) + 2 * L * pair
        fri = sum(2 * blk(16) + (L - i) * pair for i in range(K + 1))
        channel = (3 * pair + (3 + K + 1 + (Q + 7) // 8) + blk(32 + 16 * (N + 16)) + (K + 1) * pair
                   + blk(48) + blk(40))
        return Q * (per_query + fri) + channel


TESTING_CONFIG = StwoConfig(4, 3, 4, 1, 2, 5)      # config.simf:16-33
PRODUCTION_CONFIG = StwoConfig(4, 9, 13, 16, 8, 5)  # config.simf:34-52


@dataclass
class StwoProof:
    """``StarkProof`` -- stwo-verifier/src/verifier.simf:22-29, arrays in query-major order."""
    cfg: StwoConfig
    roots: np.ndarray        # uint8[3, 32]          Commitments (const, trace, cp)
    oods_trace: np.ndarray   # uint32[n_cols, 4]     OodsEvals.0
    oods_cp: np.ndarray      # uint32[16, 4]         OodsEvals.1
    trace_vals: np.ndarray   # uint32[Q, n_cols]
    cp_vals: np.ndarray      # uint32[Q, 16]
    trace_paths: List[np.ndarray]       # Q x uint8[len, 32]
    cp_paths: List[np.ndarray]          # Q x uint8[len, 32]
    fri_roots: np.ndarray    # uint8[K+1, 32]        first + inner commitments
    last_layer: np.ndarray   # uint32[4]
    fri_witness: np.ndarray  # uint32[K+1, Q, 4]
    fri_paths: List[List[np.ndarray]]   # (K+1) x Q x uint8[len, 32]
    pow_nonce: int

    def copy(self) -> "StwoProof":
        return StwoProof(self.cfg, self.roots.copy(), self.oods_trace.copy(), self.oods_cp.copy(),
                         self.trace_vals.copy(), self.cp_vals.copy(),
                         [p.copy() for p in self.trace_paths], [p.copy() for p in self.cp_paths],
                         self.fri_roots.copy(), self.last_layer.copy(), self.fri_witness.copy(),
                         [[p.copy() for p in l] for l in self.fri_paths], self.pow_nonce)


def _qm31(node: Any) -> Tuple[int, int, int, int]:
    x = node
    while isinstance(x, list) and len(x) == 1 and isinstance(x[0], list):
        x = x[0]
    try:
        (a, b), (c, d) = x
    except (TypeError, ValueError) as e:
        raise MalformedProof("bad QM31 literal") from e
    return _u32(a), _u32(b), _u32(c), _u32(d)


def _split(lst: Sequence[Any], n: int) -> List[Sequence[Any]]:
    if n <= 0 or len(lst) % n:
        raise MalformedProof("list length must be divisible by the number of queries")
    k = len(lst) // n
    return [lst[i * k:(i + 1) * k] for i in range(n)]


MAX_SHARED_QUERIES = 64


def shared_path_order(lde_log: int, n_layers: int, queries: Sequence[int]):
    """The "shared paths" variant of proof.json (what the reference notes it does not do,
    stwo-verifier/src/fri/queries.simf:41).  The per-query format repeats a sibling node for every query whose
    authentication path passes it; the shared variant stores every DISTINCT sibling of a tree once, in the order a
    verifier that walks query 0, 1, .. leaf -> root first needs it, and names the query positions in a top-level
    "queries" member so that a reader can undo the sharing WITHOUT hashing (the positions are a hint, not trusted:
    the verifier draws its own and checks the expanded paths in full, so a wrong hint can only make a proof fail).
    Upstream stwo's MerkleDecommitment goes further and also omits siblings the verifier can compute from other
    queries' leaves; undoing that needs hashing, i.e. a different verifier, and no bytes of either variant exist
    in the reference to be equal to.

    -> for each tree (trace, composition, FRI layer 0..K): (len, plan) with plan[q][lvl] = index into the tree's
    shared node list of query q's sibling at level lvl."""
    out = []
    lens = [lde_log, lde_log] + [lde_log - 1 - l for l in range(n_layers + 1)]
    for t, ln in enumerate(lens):
        shift = 0 if t < 2 else t - 1          # FRI layer l (tree 2 + l) is indexed by query >> (l + 1)
        seen: dict = {}
        plan = []
        for q in queries:
            idx = int(q) >> shift
            row = []
            for lvl in range(ln):
                key = (lvl, (idx >> lvl) ^ 1)
                if key not in seen:
                    seen[key] = len(seen)
                row.append(seen[key])
            plan.append(row)
        out.append((ln, plan, len(seen)))
    return out


def _expand_shared(data: dict, lde_log: int, Q: int) -> dict:
    """proof.json with "queries" and shared hash_witness lists -> the same object in the per-query form."""
    import copy
    queries = data["queries"]
    if Q > MAX_SHARED_QUERIES:  # the format's own bound (the C ABI's n_queries limit): the plan stays small whatever a text claims
        raise MalformedProof("shared-path proof: more than %d queries" % MAX_SHARED_QUERIES)
    if not isinstance(queries, list) or len(queries) != Q:
        raise MalformedProof("shared-path proof: one query position per query expected")
    qs = [_uint(q, 32) for q in queries]
    if lde_log < 1 or lde_log > MAX_LIST or any(q >> lde_log for q in qs):
        raise MalformedProof("shared-path proof: query position outside the LDE domain")
    fri = data["fri_proof"]
    layers = [fri["first_layer"]] + list(fri.get("inner_layers", []))
    K = len(layers) - 1
    if K + 1 >= lde_log or K > MAX_LIST:
        raise MalformedProof("shared-path proof: too many FRI layers for the LDE size")
    out = copy.copy(data)
    out.pop("queries")
    plans = shared_path_order(lde_log, K, qs)

    def expand(nodes, plan):
        ln, rows, count = plan
        if not isinstance(nodes, list) or len(nodes) != count:
            raise MalformedProof("shared-path proof: %d distinct siblings expected" % count)
        return [nodes[i] for row in rows for i in row]
    dec = [dict(d) if isinstance(d, dict) else d for d in data["decommitments"]]
    dec[1]["hash_witness"] = expand(dec[1]["hash_witness"], plans[0])
    dec[2]["hash_witness"] = expand(dec[2]["hash_witness"], plans[1])
    out["decommitments"] = dec
    new_layers = []
    for l, layer in enumerate(layers):
        layer = dict(layer)
        layer["decommitment"] = dict(layer["decommitment"])
        layer["decommitment"]["hash_witness"] = expand(layer["decommitment"]["hash_witness"], plans[2 + l])
        new_layers.append(layer)
    out["fri_proof"] = dict(fri, first_layer=new_layers[0], inner_layers=new_layers[1:])
    return out


# ------------------------------------------------------------------ minimal decommitment
def minimal_order(lde_log: int, queries: Sequence[int]):
    """The structure of upstream stwo's one-decommitment-per-tree form ("sorted multi-proof Merkle
    (real stwo format)"; the reference keeps one full path per query instead, stwo-verifier/src/fri/queries.simf:41,
    scripts/generate_wit.py:36-42, merkle.simf:22-44).  With Nodes(a) = the distinct positions `query >> a` of the
    queries at absolute level a (0 = the leaves of the LDE-sized trees), ascending, and Lone(a) = the nodes of
    Nodes(a) whose sibling `x ^ 1` is NOT in Nodes(a):
      * a tree's queried values are listed once per node of Nodes(0);
      * its hash witness holds, level by level from the leaves, the siblings of Lone(a) -- every other sibling is a
        node the verifier computes itself (stwo MerkleVerifier::verify reads a child from the witness only when the
        layer below did not produce it, left before right);
      * FRI layer l lives at levels >= l: its fold pairs are Nodes(l + 1), the evaluations of a pair's member that is
        not queried -- the partners of Lone(l) -- are its fri_witness (stwo FriLayerVerifier::extract_evaluation), and
        its tree's witness starts above the pairs: Lone(l + 1), Lone(l + 2), ...
    -> (nodes, lone): two lists indexed by level 0..lde_log - 1.  Definition by sets; the closed form the library uses
    (csrc/ss_minimal.h) and the test checker's sorted walk are compared with it in tests/test_minimal.py."""
    nodes, lone = [], []
    for a in range(lde_log):
        ns = sorted({int(q) >> a for q in queries})
        present = set(ns)
        nodes.append(ns)
        lone.append([x for x in ns if (x ^ 1) not in present])
    return nodes, lone


@dataclass
class StwoMinimalProof:
    """An stwo proof with one decommitment per tree (minimal_order): what upstream stwo's prover emits before the
    reference's adapter-side convention of one path per query.  Parity unpinned (no bytes of it in the reference)."""
    cfg: StwoConfig
    roots: np.ndarray            # uint8[3, 32]
    oods_trace: np.ndarray       # uint32[n_cols, 4]
    oods_cp: np.ndarray          # uint32[16, 4]
    fri_roots: np.ndarray        # uint8[K+1, 32]
    last_layer: np.ndarray       # uint32[4]
    pow_nonce: int
    trace_vals: np.ndarray       # uint32[distinct positions, n_cols], ascending position
    cp_vals: np.ndarray          # uint32[distinct positions, 16]
    fri_witness: List[np.ndarray]   # (K+1) x uint32[n_l, 4]
    hash_witness: List[np.ndarray]  # (K+3) x uint8[n_t, 32]: trace, composition, FRI layer 0..K

    def copy(self) -> "StwoMinimalProof":
        return StwoMinimalProof(self.cfg, self.roots.copy(), self.oods_trace.copy(), self.oods_cp.copy(),
                                self.fri_roots.copy(), self.last_layer.copy(), self.pow_nonce, self.trace_vals.copy(),
                                self.cp_vals.copy(), [w.copy() for w in self.fri_witness],
                                [h.copy() for h in self.hash_witness])


def stwo_minimise(p: "StwoProof", queries: "Sequence[int] | None" = None) -> StwoMinimalProof:
    """Per-query proof -> minimal proof: a selection, no hashing (a prover holds every node anyway; this is what it
    would not send).  `queries`: the positions the prover drew, else the public transcript is replayed.  Only proofs
    whose paths all have the config's lengths and whose queries agree wherever they present the same thing have a
    minimal form (MalformedProof otherwise).  Siblings that the verifier recomputes are dropped WITHOUT being
    checked -- checking them is verifying."""
    c = p.cfg
    L, Q, K = c.lde_log, c.n_queries, c.n_layers
    qs = [int(x) for x in (queries if queries is not None else stwo_queries(p))]
    if len(qs) != Q or any(q >> L for q in qs):
        raise MalformedProof("one position inside the LDE domain per query expected")
    nodes, lone = minimal_order(L, qs)

    def pick(items, what):  # the chains at one node must present the same thing
        first = items[0]
        for x in items[1:]:
            if not np.array_equal(first, x):
                raise MalformedProof("two queries present different %s for one position: no minimal form" % what)
        return first

    def at(a, x):
        return [q for q in range(Q) if (qs[q] >> a) == x]
    tv = np.array([pick([p.trace_vals[q] for q in at(0, x)], "values") for x in nodes[0]], dtype=np.uint32).reshape(-1, c.n_cols)
    cv = np.array([pick([p.cp_vals[q] for q in at(0, x)], "values") for x in nodes[0]], dtype=np.uint32).reshape(-1, 16)

    def tree(paths, shift):
        ln = L - shift
        if any(len(pth) != ln for pth in paths):
            raise MalformedProof("only full-length Merkle paths have a minimal form")
        out = [pick([paths[q][a - shift] for q in at(a, x)], "siblings") for a in range(shift, L) for x in lone[a]]
        return np.array(out, dtype=np.uint8).reshape(-1, 32)
    hw = [tree(p.trace_paths, 0), tree(p.cp_paths, 0)]
    fw = []
    for l in range(K + 1):
        fw.append(np.array([pick([p.fri_witness[l, q] for q in at(l, x)], "evaluations") for x in lone[l]],
                           dtype=np.uint32).reshape(-1, 4))
        hw.append(tree(p.fri_paths[l], l + 1))
    return StwoMinimalProof(c, p.roots.copy(), p.oods_trace.copy(), p.oods_cp.copy(), p.fri_roots.copy(),
                            p.last_layer.copy(), p.pow_nonce, tv, cv, fw, hw)


def stwo_minimal_to_json(m: StwoMinimalProof) -> dict:
    """The proof.json of a minimal proof: the schema of format C (stwo-verifier/scripts/generate_wit.py:106-245 reads
    it) with the lists as upstream stwo fills them -- `queried_values` once per distinct position, `hash_witness` and
    `fri_witness` without what the verifier computes.  Nothing in the text says which form it is: a reader tells by
    the list lengths (stwo_from_json_any) or is told (SS_TEXT_JSON_MINIMAL)."""
    def q(v: Sequence[int]) -> List[List[int]]:
        return [[int(v[0]), int(v[1])], [int(v[2]), int(v[3])]]

    def hw(a: np.ndarray) -> List[List[int]]:
        return [[int(b) for b in node] for node in a]
    c = m.cfg
    conf = {"pow_bits": c.pow_bits,
            "fri_config": {"log_blowup_factor": c.log_blowup, "log_last_layer_degree_bound": 0, "n_queries": c.n_queries}}
    if c.hash != "sha256":
        conf["hash"] = c.hash
    layers = [{"fri_witness": [q(w) for w in m.fri_witness[l]],
               "decommitment": {"hash_witness": hw(m.hash_witness[2 + l]), "column_witness": []},
               "commitment": [int(b) for b in m.fri_roots[l]]} for l in range(c.n_layers + 1)]
    return {
        "config": conf,
        "commitments": [[int(b) for b in r] for r in m.roots],
        "sampled_values": [[], [[q(v)] for v in m.oods_trace], [[q(v)] for v in m.oods_cp]],
        "decommitments": [{"hash_witness": [], "column_witness": []},
                          {"hash_witness": hw(m.hash_witness[0]), "column_witness": []},
                          {"hash_witness": hw(m.hash_witness[1]), "column_witness": []}],
        "queried_values": [[], [int(x) for x in m.trace_vals.reshape(-1)], [int(x) for x in m.cp_vals.reshape(-1)]],
        "proof_of_work": int(m.pow_nonce),
        "fri_proof": {"first_layer": layers[0], "inner_layers": layers[1:],
                      "last_layer_poly": {"coeffs": [q(m.last_layer)], "log_size": 0}},
    }


def stwo_minimal_from_json(data: Any, expect: StwoConfig) -> StwoMinimalProof:
    """Reads a minimal proof.json against the config the caller expects (a minimal text has no path whose length
    would give the LDE size away, so there is no reading it without one).  The returned cfg is what the text DECLARES
    where it declares (pow_bits, blow-up, n_queries, hash, the column and layer counts); list lengths are data."""
    if isinstance(data, (str, bytes)):
        data = json.loads(data)
    try:
        conf = data.get("config", {})
        fri_conf = conf.get("fri_config", {})
        Q = _uint(fri_conf["n_queries"], 32) if "n_queries" in fri_conf else expect.n_queries
        pow_bits = _uint(conf["pow_bits"], 32) if "pow_bits" in conf else expect.pow_bits
        blow = _uint(fri_conf["log_blowup_factor"], 32) if "log_blowup_factor" in fri_conf else expect.log_blowup
        if "hash" in conf and (not isinstance(conf["hash"], str) or conf["hash"] not in ("sha256", "blake2s")):
            raise MalformedProof("unknown hash %r" % (conf["hash"],))
        hname = conf.get("hash") or expect.hash
        roots = np.stack([np.frombuffer(bytes(c), dtype=np.uint8) for c in data["commitments"]])
        if roots.shape != (3, 32):
            raise MalformedProof("expected three 32-byte commitments")
        sv = data["sampled_values"]
        oods_trace = np.array([_qm31(c) for c in sv[1]], dtype=np.uint32).reshape(-1, 4)
        oods_cp = np.array([_qm31(c) for c in sv[2]], dtype=np.uint32).reshape(-1, 4)
        if oods_cp.shape[0] != N_CP_PARTITIONS:
            raise MalformedProof("expected 16 composition-polynomial partitions")
        N = oods_trace.shape[0]
        qv, dec = data["queried_values"], data["decommitments"]
        if N == 0 or len(qv[1]) % N or len(qv[2]) % N_CP_PARTITIONS:
            raise MalformedProof("queried values are listed column-count at a time")
        tv = np.array([_u32(x) for x in qv[1]], dtype=np.uint32).reshape(-1, N)
        cv = np.array([_u32(x) for x in qv[2]], dtype=np.uint32).reshape(-1, N_CP_PARTITIONS)

        def hashes(lst) -> np.ndarray:
            out = np.zeros((len(lst), 32), dtype=np.uint8)
            for i, n in enumerate(lst):
                b = bytes(n)
                if len(b) != 32:
                    raise MalformedProof("expected 32 bytes")
                out[i] = np.frombuffer(b, dtype=np.uint8)
            return out
        fri = data["fri_proof"]
        layers = [fri["first_layer"]] + list(fri.get("inner_layers", []))
        K = len(layers) - 1
        if K > MAX_LIST:
            raise MalformedProof("too many FRI layers")
        hw = [hashes(dec[1]["hash_witness"]), hashes(dec[2]["hash_witness"])]
        fw = []
        for l in layers:
            fw.append(np.array([_qm31(x) for x in l["fri_witness"]], dtype=np.uint32).reshape(-1, 4))
            hw.append(hashes(l["decommitment"]["hash_witness"]))
        fri_roots = np.stack([np.frombuffer(bytes(l["commitment"]), dtype=np.uint8) for l in layers])
        coeffs = fri["last_layer_poly"]["coeffs"]
        if len(coeffs) != 1:
            raise MalformedProof("expected a degree-0 last layer")
        cfg = StwoConfig(N, expect.lde_log - blow, expect.lde_log, Q, K, pow_bits, hname)
        return StwoMinimalProof(cfg, roots.copy(), oods_trace, oods_cp, fri_roots.copy(),
                                np.array(_qm31(coeffs[0]), dtype=np.uint32), _uint(data.get("proof_of_work", 0), 64),
                                tv, cv, fw, hw)
    except (KeyError, IndexError, TypeError, AttributeError, ValueError) as e:
        if isinstance(e, MalformedProof):
            raise
        raise MalformedProof(str(e)) from e


def stwo_json_is_minimal(data: dict, expect: StwoConfig) -> bool:
    """Which of the two forms a proof.json of the expected config is: per-query lists hold n_queries equal shares --
    n_queries x n_cols queried values and n_queries x lde_log trace siblings -- and a minimal text with two or more
    queries always has fewer siblings (the paths meet below the root); with ONE query the two forms are the same bytes."""
    try:
        return not (len(data["queried_values"][1]) == expect.n_queries * expect.n_cols
                    and len(data["decommitments"][1]["hash_witness"]) == expect.n_queries * expect.lde_log)
    except (KeyError, IndexError, TypeError) as e:
        raise MalformedProof(str(e)) from e


def stwo_queries(p: "StwoProof") -> List[int]:
    """The query positions of a proof: the public part of the Fiat-Shamir transcript replayed with hashlib
    (stwo-verifier/src/channel.simf:31-172 in the order of verifier.simf:32-58).  No check of the proof is made;
    `convert --to json-shared` uses this to know which siblings coincide."""
    import hashlib
    H = hashlib.sha256 if p.cfg.hash == "sha256" else (lambda b: hashlib.blake2s(b, digest_size=32))
    state = {"d": bytes(32), "c": 0}

    def mix(b: bytes) -> None:
        state["d"], state["c"] = H(state["d"] + b).digest(), 0

    def draw_words() -> List[int]:
        d = H(state["d"] + state["c"].to_bytes(4, "big")).digest()
        state["c"] += 1
        return [int.from_bytes(d[4 * i:4 * i + 4], "big") for i in range(8)]

    def draw_qm31() -> None:
        for _ in range(256):  # channel.simf:127-135: retry while a word is >= 2^32 - 2
            if all(x < 4294967294 for x in draw_words()[:4]):
                return
        raise MalformedProof("channel draw exhausted")

    def be(words) -> bytes:
        return b"".join(int(w).to_bytes(4, "big") for w in np.asarray(words).reshape(-1))
    mix(bytes(p.roots[0])); mix(bytes(p.roots[1])); draw_qm31(); mix(bytes(p.roots[2]))
    draw_qm31()
    mix(be(p.oods_trace) + be(p.oods_cp))
    draw_qm31()
    for r in p.fri_roots:
        mix(bytes(r))
        draw_qm31()
    mix(be(p.last_layer))
    mix(int(p.pow_nonce).to_bytes(8, "big"))
    mask, out = (1 << p.cfg.lde_log) - 1, []
    while len(out) < p.cfg.n_queries:
        out += [w & mask for w in draw_words()]
    return out[:p.cfg.n_queries]


def stwo_from_json(data: Any, trace_log: int | None = None, hash: str | None = None,
                   expect: "StwoConfig | None" = None) -> StwoProof:
    """Format C.  The JSON carries pow_bits / log_blowup / n_queries; the LDE size is
    implied by the Merkle path length (the reference hard-codes it: config.simf:21,39).

    The returned proof's `cfg` is what the proof DECLARES (its `config` object and its array
    shapes) -- untrusted input.  The verifier compares it with the config its caller expects
    (Verifier.verify_stwo(..., cfg=...)) and never verifies against a self-declared one.  A
    parameter the JSON does not declare is taken from `expect`; without `expect` it is an error
    (in particular a missing `pow_bits` never means "no proof of work")."""
    if isinstance(data, (str, bytes)):
        data = json.loads(data)
    try:
        conf = data.get("config", {})
        fri_conf = conf.get("fri_config", {})

        def declared(d: dict, key: str, fallback: Any) -> int:
            if key in d:
                return _uint(d[key], 32)
            if fallback is None:
                raise MalformedProof("config has no %r and no expected config was given" % key)
            return int(fallback)
        Q = declared(fri_conf, "n_queries", expect and expect.n_queries)
        if "queries" in data:  # the shared-path variant: undo the sharing first (needs the LDE size: paths have no ends)
            if expect is not None:
                lde_hint = expect.lde_log
            elif trace_log is not None and "log_blowup_factor" in fri_conf:
                lde_hint = trace_log + _uint(fri_conf["log_blowup_factor"], 32)
            else:
                raise MalformedProof("a shared-path proof.json can only be read against an expected config")
            data = _expand_shared(data, lde_hint, Q)
        roots = np.stack([np.frombuffer(bytes(c), dtype=np.uint8) for c in data["commitments"]])
        if roots.shape != (3, 32):
            raise MalformedProof("expected three 32-byte commitments")
        sv = data["sampled_values"]
        oods_trace = np.array([_qm31(c) for c in sv[1]], dtype=np.uint32).reshape(-1, 4)
        oods_cp = np.array([_qm31(c) for c in sv[2]], dtype=np.uint32).reshape(-1, 4)
        if oods_cp.shape[0] != N_CP_PARTITIONS:
            raise MalformedProof("expected 16 composition-polynomial partitions")
        N = oods_trace.shape[0]
        dec = data["decommitments"]
        trace_paths = [_path(c) for c in _split(dec[1]["hash_witness"], Q)]
        cp_paths = [_path(c) for c in _split(dec[2]["hash_witness"], Q)]
        qv = data["queried_values"]
        trace_vals = np.array([[_u32(x) for x in c] for c in _split(qv[1], Q)], dtype=np.uint32)
        cp_vals = np.array([[_u32(x) for x in c] for c in _split(qv[2], Q)], dtype=np.uint32)
        if trace_vals.shape != (Q, N) or cp_vals.shape != (Q, N_CP_PARTITIONS):
            raise MalformedProof("queried value count mismatch")
        fri = data["fri_proof"]
        layers = [fri["first_layer"]] + list(fri.get("inner_layers", []))
        K = len(layers) - 1
        if K > MAX_LIST:
            raise MalformedProof("too many FRI layers")
        fri_roots = np.stack([np.frombuffer(bytes(l["commitment"]), dtype=np.uint8) for l in layers])
        fri_witness = np.zeros((K + 1, Q, 4), dtype=np.uint32)
        fri_paths: List[List[np.ndarray]] = []
        for i, l in enumerate(layers):
            w = l["fri_witness"]
            if len(w) != Q:
                raise MalformedProof("one FRI witness per query expected")
            fri_witness[i] = np.array([_qm31(x) for x in w], dtype=np.uint32)
            fri_paths.append([_path(c) for c in _split(l["decommitment"]["hash_witness"], Q)])
        coeffs = fri["last_layer_poly"]["coeffs"]
        if len(coeffs) != 1:
            raise MalformedProof("expected a degree-0 last layer")
        last = np.array(_qm31(coeffs[0]), dtype=np.uint32)
        lde_log = len(trace_paths[0]) if Q else 0
        if trace_log is not None:
            tl = trace_log
        else:
            tl = lde_log - declared(fri_conf, "log_blowup_factor", expect and expect.log_blowup)
        if "hash" in conf:  # declared: must be one of the two names (a present null / 0 / "" is not "absent")
            if not isinstance(conf["hash"], str) or conf["hash"] not in ("sha256", "blake2s"):
                raise MalformedProof("unknown hash %r" % (conf["hash"],))
        hname = hash or conf.get("hash") or (expect.hash if expect else "sha256")  # the reference: SHA-256 only
        if hname not in ("sha256", "blake2s"):
            raise MalformedProof("unknown hash %r" % (hname,))
        cfg = StwoConfig(N, tl, lde_log, Q, K, declared(conf, "pow_bits", expect and expect.pow_bits), hname)
        nonce = _uint(data.get("proof_of_work", 0), 64)
        return StwoProof(cfg, roots.copy(), oods_trace, oods_cp, trace_vals, cp_vals, trace_paths,
                         cp_paths, fri_roots.copy(), last, fri_witness, fri_paths, nonce)
    except (KeyError, IndexError, TypeError, AttributeError) as e:  # wrong nesting, a list where an object belongs
        raise MalformedProof(str(e)) from e


def stwo_to_json(p: StwoProof, shared: bool = False, queries: "Sequence[int] | None" = None) -> dict:
    """Writer for format C (what an stwo-style prover would emit).  shared=True writes the shared-path variant
    (shared_path_order): every distinct sibling once + a "queries" member; the positions come from `queries` (a
    prover knows them) or from replaying the public transcript (stwo_queries).  Only proofs whose paths all have
    the config's lengths and agree wherever they meet can be written that way."""
    def q(v: Sequence[int]) -> List[List[int]]:
        return [[int(v[0]), int(v[1])], [int(v[2]), int(v[3])]]

    plans = None
    if shared:
        qs = [int(x) for x in (queries if queries is not None else stwo_queries(p))]
        plans = shared_path_order(p.cfg.lde_log, p.cfg.n_layers, qs)
        tree_no = {"n": 0}

    def hw(paths: List[np.ndarray]) -> List[List[int]]:
        if plans is None:
            return [[int(b) for b in node] for pth in paths for node in pth]
        ln, rows, count = plans[tree_no["n"]]
        tree_no["n"] += 1
        nodes: List[Any] = [None] * count
        for pth, row in zip(paths, rows):
            if len(pth) != ln:
                raise MalformedProof("only full-length Merkle paths can be shared")
            for node, i in zip(pth, row):
                val = [int(b) for b in node]
                if nodes[i] is not None and nodes[i] != val:
                    raise MalformedProof("two queries present different bytes for one node: no shared form")
                nodes[i] = val
        return nodes

    def layer(i: int) -> dict:
        return {"fri_witness": [q(w) for w in p.fri_witness[i]],
                "decommitment": {"hash_witness": hw(p.fri_paths[i]), "column_witness": []},
                "commitment": [int(b) for b in p.fri_roots[i]]}
    c = p.cfg
    conf = {"pow_bits": c.pow_bits,
            "fri_config": {"log_blowup_factor": c.log_blowup,
                           "log_last_layer_degree_bound": 0, "n_queries": c.n_queries}}
    if c.hash != "sha256":
        conf["hash"] = c.hash  # extension key; the reference's JSON has none (always SHA-256)
    # (hw() consumes the trees in the order trace, composition, FRI layer 0.. : keep the evaluation order below)
    dec = [{"hash_witness": [], "column_witness": []},
           {"hash_witness": hw(p.trace_paths), "column_witness": []},
           {"hash_witness": hw(p.cp_paths), "column_witness": []}]
    fri_layers = [layer(i) for i in range(c.n_layers + 1)]
    out = {
        "config": conf,
        "commitments": [[int(b) for b in r] for r in p.roots],
        "sampled_values": [[], [[q(v)] for v in p.oods_trace], [[q(v)] for v in p.oods_cp]],
        "decommitments": dec,
        "queried_values": [[], [int(x) for x in p.trace_vals.reshape(-1)],
                           [int(x) for x in p.cp_vals.reshape(-1)]],
        "proof_of_work": int(p.pow_nonce),
        "fri_proof": {"first_layer": fri_layers[0],
                      "inner_layers": fri_layers[1:],
                      "last_layer_poly": {"coeffs": [q(p.last_layer)], "log_size": 0}},
    }
    if shared:
        out["queries"] = qs
    return out


def stwo_from_wit(text: Any, trace_log: int, pow_bits: int = 5, hash: str = "sha256") -> StwoProof:
    """Format D (stwo-verifier/scripts/generate_wit.py:218-243).  A ``.wit`` carries no
    config, so TRACE_LOG_SIZE and the PoW target come from the caller, as they come from
    config.simf for the reference."""
    obj = json.loads(text) if isinstance(text, (str, bytes)) else text
    try:
        com = parse_literal(obj["COMMITMENTS"]["value"])
        dec = parse_literal(obj["DECOMMITMENTS"]["value"])
        oods = parse_literal(obj["OODS_EVALS"]["value"])
        fric = parse_literal(obj["FRI_COMMITMENTS"]["value"])
        frid = parse_literal(obj["FRI_DECOMMITMENTS"]["value"])
        nonce = parse_literal(obj["POW_NONCE"]["value"])
    except (KeyError, TypeError) as e:
        raise MalformedProof("missing witness %s" % e) from e
    return _stwo_from_parts(com, dec, oods, fric, frid, nonce, trace_log, pow_bits, hash)


def _stwo_from_parts(com: Any, dec: Any, oods: Any, fric: Any, frid: Any, nonce: Any, trace_log: int,
                     pow_bits: int, hash: str) -> StwoProof:
    """The six fields of `StarkProof` (stwo-verifier/src/verifier.simf:22-33) as parsed literals."""
    try:
        roots = np.stack([np.frombuffer(u256_to_bytes(c), dtype=np.uint8) for c in com])
        Q = len(dec)
        oods_trace = np.array([_qm31(c[0]) for c in oods[0]], dtype=np.uint32).reshape(-1, 4)
        oods_cp = np.array([_qm31(c) for c in oods[1]], dtype=np.uint32).reshape(-1, 4)
        N = oods_trace.shape[0]
        trace_vals = np.array([[_u32(c[0]) for c in d[0][0]] for d in dec], dtype=np.uint32)
        cp_vals = np.array([[_u32(x) for x in d[1][0]] for d in dec], dtype=np.uint32)
        trace_paths = [_path(d[0][1]) for d in dec]
        cp_paths = [_path(d[1][1]) for d in dec]
        layers_d = [frid[0]] + list(frid[1])
        K = len(layers_d) - 1
        fri_roots = np.stack([np.frombuffer(u256_to_bytes(c), dtype=np.uint8)
                              for c in [fric[0]] + list(fric[1])])
        last = np.array(_qm31(fric[2]), dtype=np.uint32)
        fri_witness = np.array([[_qm31(x[0]) for x in l] for l in layers_d], dtype=np.uint32)
        fri_paths = [[_path(x[1]) for x in l] for l in layers_d]
        if (roots.shape != (3, 32) or trace_vals.shape != (Q, N)
                or cp_vals.shape != (Q, N_CP_PARTITIONS) or fri_roots.shape != (K + 1, 32)
                or fri_witness.shape != (K + 1, Q, 4) or oods_cp.shape[0] != N_CP_PARTITIONS):
            raise MalformedProof("witness shape mismatch")
        lde_log = len(trace_paths[0]) if Q else 0
        cfg = StwoConfig(N, trace_log, lde_log, Q, K, pow_bits, hash)
        return StwoProof(cfg, roots.copy(), oods_trace, oods_cp, trace_vals, cp_vals, trace_paths,
                         cp_paths, fri_roots.copy(), last, fri_witness, fri_paths, _uint(nonce, 64))
    except (IndexError, TypeError, ValueError) as e:
        if isinstance(e, MalformedProof):
            raise
        raise MalformedProof(str(e)) from e


def stwo_to_wit(p: StwoProof) -> str:
    """Writer for format D, same layout as the reference's generate_wit.py."""
    def hx(b: Any) -> str:
        return "0x" + bytes(b).hex()

    def q(v: Sequence[int]) -> str:
        return "((%d, %d), (%d, %d))" % (int(v[0]), int(v[1]), int(v[2]), int(v[3]))

    def lst(pth: np.ndarray) -> str:
        return "list![" + ", ".join(hx(n) for n in pth) + "]"
    c = p.cfg
    Q, K, N = c.n_queries, c.n_layers, c.n_cols
    M31, MP = "u32", "List<u256, 32>"
    CM31 = "(%s, %s)" % (M31, M31)
    QM31 = "(%s, %s)" % (CM31, CM31)
    dec_t = "(([[%s; 1]; %d], %s), ([%s; 16], %s))" % (M31, N, MP, M31, MP)
    fqd = "(%s, %s)" % (QM31, MP)
    fld = "[%s; %d]" % (fqd, Q)
    dec_items = []
    for i in range(Q):
        tv = "[" + ", ".join("[%d]" % int(x) for x in p.trace_vals[i]) + "]"
        cv = "[" + ", ".join(str(int(x)) for x in p.cp_vals[i]) + "]"
        dec_items.append("((%s, %s), (%s, %s))" % (tv, lst(p.trace_paths[i]), cv, lst(p.cp_paths[i])))

    def fl(i: int) -> str:
        return "[" + ", ".join("(%s, %s)" % (q(p.fri_witness[i, j]), lst(p.fri_paths[i][j]))
                               for j in range(Q)) + "]"
    wit = {
        "COMMITMENTS": {"value": "(%s, %s, %s)" % tuple(hx(r) for r in p.roots),
                        "type": "(u256, u256, u256)"},
        "DECOMMITMENTS": {"value": "[" + ", ".join(dec_items) + "]",
                          "type": "[%s; %d]" % (dec_t, Q)},
        "OODS_EVALS": {"value": "(%s, %s)" % (
            "[" + ", ".join("[" + q(v) + "]" for v in p.oods_trace) + "]",
            "[" + ", ".join(q(v) for v in p.oods_cp) + "]"),
            "type": "([[%s; 1]; %d], [%s; 16])" % (QM31, N, QM31)},
        "FRI_COMMITMENTS": {"value": "(%s, [%s], %s)" % (
            hx(p.fri_roots[0]), ", ".join(hx(r) for r in p.fri_roots[1:]), q(p.last_layer)),
            "type": "(u256, [u256; %d], %s)" % (K, QM31)},
        "FRI_DECOMMITMENTS": {"value": "(%s, [%s])" % (
            fl(0), ", ".join(fl(i) for i in range(1, K + 1))),
            "type": "(%s, [%s; %d])" % (fld, fld, K)},
        "POW_NONCE": {"value": str(int(p.pow_nonce)), "type": "u64"},
    }
    return json.dumps(wit, indent=4)


def _simf_body(text: str, what: str) -> Any:
    """``let proof: T = <literal>;`` -> the parsed literal."""
    s = text.strip()
    if not s.startswith("let ") or "=" not in s:
        raise MalformedProof("expected `let proof: %s = (...);`" % what)
    body = s[s.index("=") + 1:].strip()
    if body.endswith(";"):
        body = body[:-1]
    return parse_literal(body)


def stwo_to_simf(p: StwoProof) -> str:
    """The ``let proof: Proof = (...);`` snippet stwo-verifier/scripts/generate_simf.py:160-228
    prints for pasting into a ``.simf`` test (e.g. verifier.simf:63), same text."""
    def hx(b: Any) -> str:
        return "0x" + bytes(b).hex()

    def q(v: Sequence[int]) -> str:
        return "qm31(%d, %d, %d, %d)" % (int(v[0]), int(v[1]), int(v[2]), int(v[3]))

    def lst(pth: np.ndarray) -> str:
        return "list![" + ", ".join(hx(n) for n in pth) + "]"

    def indent(text: str, spaces: int) -> str:
        pad = " " * spaces
        return "\n".join(pad + line if line else line for line in text.splitlines())

    def layer(i: int) -> str:
        items = ["(\n            %s,\n            %s\n        )" % (q(p.fri_witness[i, j]), lst(p.fri_paths[i][j]))
                 for j in range(p.cfg.n_queries)]
        return "[\n        " + ",\n        ".join(items) + "\n    ]"
    Q, K = p.cfg.n_queries, p.cfg.n_layers
    commitments = "(\n        %s,\n        %s,\n        %s,\n    )" % tuple(hx(r) for r in p.roots)
    dec_items = []
    for i in range(Q):
        tv = "[" + ", ".join("[%d]" % int(x) for x in p.trace_vals[i]) + "]"
        cv = "[" + ", ".join(str(int(x)) for x in p.cp_vals[i]) + "]"
        dec_items.append("(\n            (%s, %s),\n            (%s, %s),\n        )"
                         % (tv, lst(p.trace_paths[i]), cv, lst(p.cp_paths[i])))
    decommitments = "[\n        " + ",\n        ".join(dec_items) + "\n    ]"
    oods = "(\n        %s,\n        %s,\n    )" % (
        "[" + ", ".join("[" + q(v) + "]" for v in p.oods_trace) + "]",
        "[" + ", ".join(q(v) for v in p.oods_cp) + "]")
    fri_commitments = "(\n        %s,\n        %s,\n        %s,\n    )" % (
        hx(p.fri_roots[0]), "[" + ", ".join(hx(r) for r in p.fri_roots[1:]) + "]", q(p.last_layer))
    inner = "[\n" + ",\n".join(indent(layer(i), 8) for i in range(1, K + 1)) + "\n    ]"
    fri_decommitments = "(\n" + indent(layer(0), 8) + ",\n" + indent(inner, 4) + "\n    )"
    return ("let proof: Proof = (\n    %s,\n    %s,\n    %s,\n    %s,\n    %s,\n    %d\n);"
            % (commitments, decommitments, oods, fri_commitments, fri_decommitments, int(p.pow_nonce)))


def stwo_from_simf(text: str, trace_log: int, pow_bits: int = 5, hash: str = "sha256") -> StwoProof:
    """Reads the snippet back (it carries no config, like the ``.wit``)."""
    v = _simf_body(text, "Proof")
    if not isinstance(v, list) or len(v) != 6:
        raise MalformedProof("Proof has six fields")
    return _stwo_from_parts(v[0], v[1], v[2], v[3], v[4], v[5], trace_log, pow_bits, hash)


# ------------------------------------------------------------- seeded corruption (tests / bench)
def stark101_fields(p: Stark101Proof) -> List[Tuple[str, Any]]:
    """Every mutable location of a stark101 proof, for bit-flip negative tests."""
    locs: List[Tuple[str, Any]] = [("root", None), ("last", None)]
    for k in range(3):
        locs.append(("eval_ev", k))
        locs += [("eval_path", (k, i)) for i in range(len(p.evals[k].path))]
    for li, l in enumerate(p.layers):
        locs += [("layer_root", li), ("layer_beta", li), ("cpa_ev", li), ("cpb_ev", li)]
        locs += [("cpa_path", (li, i)) for i in range(len(l.cpa.path))]
        locs += [("cpb_path", (li, i)) for i in range(len(l.cpb.path))]
    return locs


def _flip_bytes(b: bytes, bit: int) -> bytes:
    a = bytearray(b)
    a[(bit // 8) % len(a)] ^= 1 << (bit % 8)
    return bytes(a)


def stark101_corrupt(p: Stark101Proof, rng: np.random.Generator) -> Tuple[Stark101Proof, str]:
    """Flip one seeded bit somewhere in the proof.  Returns (new proof, description)."""
    q = p.copy()
    locs = stark101_fields(q)
    kind, arg = locs[int(rng.integers(len(locs)))]
    bit = int(rng.integers(256))
    if kind == "root":
        q.root = _flip_bytes(q.root, bit)
    elif kind == "last":
        q.last ^= 1 << (bit % 32)
    elif kind == "eval_ev":
        q.evals[arg].ev ^= 1 << (bit % 32)
    elif kind == "eval_path":
        k, i = arg
        q.evals[k].path[i, (bit // 8) % 32] ^= 1 << (bit % 8)
    elif kind == "layer_root":
        q.layers[arg].root = _flip_bytes(q.layers[arg].root, bit)
    elif kind == "layer_beta":
        q.layers[arg].beta ^= 1 << (bit % 32)
    elif kind == "cpa_ev":
        q.layers[arg].cpa.ev ^= 1 << (bit % 32)
    elif kind == "cpb_ev":
        q.layers[arg].cpb.ev ^= 1 << (bit % 32)
    elif kind == "cpa_path":
        li, i = arg
        q.layers[li].cpa.path[i, (bit // 8) % 32] ^= 1 << (bit % 8)
    elif kind == "cpb_path":
        li, i = arg
        q.layers[li].cpb.path[i, (bit // 8) % 32] ^= 1 << (bit % 8)
    return q, "%s%s bit %d" % (kind, "" if arg is None else str(arg), bit)


def stwo_corrupt(p: StwoProof, rng: np.random.Generator) -> Tuple[StwoProof, str]:
    """Flip one seeded bit in one seeded section of an stwo proof."""
    q = p.copy()
    Q, K = p.cfg.n_queries, p.cfg.n_layers
    sections = ["roots", "oods_trace", "oods_cp", "trace_vals", "cp_vals", "trace_path", "cp_path",
                "fri_roots", "last_layer", "fri_witness", "fri_path", "pow_nonce"]
    s = sections[int(rng.integers(len(sections)))]
    bit = int(rng.integers(1 << 30))

    def flip_u8(a: np.ndarray) -> None:
        flat = a.reshape(-1)
        flat[(bit // 8) % flat.size] ^= np.uint8(1 << (bit % 8))

    def flip_u32(a: np.ndarray) -> None:
        flat = a.reshape(-1)
        flat[(bit // 32) % flat.size] ^= np.uint32(1 << (bit % 32))
    if s == "roots":
        flip_u8(q.roots)
    elif s == "oods_trace":
        flip_u32(q.oods_trace)
    elif s == "oods_cp":
        flip_u32(q.oods_cp)
    elif s == "trace_vals":
        flip_u32(q.trace_vals)
    elif s == "cp_vals":
        flip_u32(q.cp_vals)
    elif s == "trace_path":
        flip_u8(q.trace_paths[int(rng.integers(Q))])
    elif s == "cp_path":
        flip_u8(q.cp_paths[int(rng.integers(Q))])
    elif s == "fri_roots":
        flip_u8(q.fri_roots)
    elif s == "last_layer":
        flip_u32(q.last_layer)
    elif s == "fri_witness":
        flip_u32(q.fri_witness)
    elif s == "fri_path":
        cand = [(l, j) for l in range(K + 1) for j in range(Q) if q.fri_paths[l][j].size]
        l, j = cand[int(rng.integers(len(cand)))]
        flip_u8(q.fri_paths[l][j])
    elif s == "pow_nonce":
        q.pow_nonce ^= 1 << (bit % 64)
    return q, "%s bit %d" % (s, bit)
