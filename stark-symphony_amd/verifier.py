"""Host side of the batch verifier: records, device batches, verify().

Mirrors the reference's caller-side contract: a proof file goes in, ACCEPT /
REJECT comes out (`simfony run` exit status, simfony-cli/src/main.rs:205-206,254-257) -- for
N proofs at once.  All verification work happens in libss_verify.so's HIP kernels; this
module only re-orders bytes (records -> batch, via ss_*_pack) and moves them with PyTorch,
which is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import binding as B
from .formats import Stark101Proof, StwoConfig, StwoProof

MODE_LITERAL, MODE_FIXTURE = B.MODE_LITERAL, B.MODE_FIXTURE
PHASE_HEAD, PHASE_TAIL, PHASE_ALL = 1, 2, 3


def _after_current(streams, dev) -> None:
    """Order `streams` after everything already enqueued on the caller's current stream of `dev`.

    torch's side streams are non-blocking: they are NOT ordered after the legacy null stream, so a `fill_` / `zeros` /
    `copy_` a caller issued on its current stream can still be pending when a pass on a side stream starts -- or land
    after that pass has finished, in which case the caller's write replaces the verifier's status words (round 5's red
    test: profiles/r06_null_stream_order.txt).  The status word is the verifier's and only the verifier's
    (stark101/src/verifier.simf:24-42, simfony-cli/src/main.rs:254-257), so every helper below joins the current
    stream when it is made, and again on request (`wait_current`)."""
    torch = _torch()
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        if s != cur:
            s.wait_stream(cur)


class Pipeline:
    """Keeps `depth` passes over resident batches in flight on two HIP streams: the HEAD half
    (transcript + query kernels, latency bound) of pass i+1 runs on the head stream while the
    TAIL half (Merkle kernel, ALU bound) of pass i runs on the tail stream.  With one tail stream
    (the default) Merkle kernels never overlap each other, so their event-measured durations stay
    meaningful.  `tail_streams=2` alternates the TAIL halves over two streams: the drain of one Merkle
    launch (its last, half-empty round of workgroups) overlaps the start of the next, which is worth a
    few per cent when a launch is only a few milliseconds (one GPU's 8 192-proof share of a batch);
    per-kernel durations measured under that overlap are inflated."""

    def __init__(self, slots: Sequence["_DeviceBatch"], tail_streams: int = 1, head_streams: int = 0):
        torch = _torch()
        self.slots = list(slots)
        dev = self.slots[0].ver.device
        # one head stream per slot (head_streams = 0): HEAD halves are latency bound (a few waves per CU), so
        # several of them also overlap each other; TAIL halves share one stream.  head_streams = 1 measures the same
        # at 8 192 and 65 536 proofs per pass (profiles/r04_accept_reduce.txt) and leaves a process that keeps the
        # runtime's 4 hardware queues with a stream per queue.
        made = [torch.cuda.Stream(device=dev) for _ in range(head_streams or len(self.slots))]
        self.head_streams = [made[i % len(made)] for i in range(len(self.slots))]
        # (a high-priority tail stream was tried: no measurable effect on MI355X)
        self.tail_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, tail_streams))]
        self.tail_stream = self.tail_streams[0]
        self.comm_stream = torch.cuda.Stream(device=dev)  # accept-reduce, off the kernels' path
        self.head_done = [torch.cuda.Event() for _ in self.slots]
        self.tail_done = [None for _ in self.slots]
        self.i = 0
        self.wait_current()

    def wait_current(self) -> None:
        """Order every later pass after what the caller has enqueued on its current stream so far (done once by the
        constructor; call it again after writing to a slot's buffers from the current stream)."""
        _after_current(set(self.head_streams + self.tail_streams + [self.comm_stream]), self.slots[0].ver.device)

    def submit(self, after_tail=None, on_reuse=None) -> int:
        """Enqueue one pass; returns the slot it used.
        `after_tail(slot)` is called with the communication stream current, ordered after the pass (e.g. to all-reduce
        its accept count at once); the slot is not reused before that has completed.
        `on_reuse(slot)` is called with the slot's HEAD stream current when the slot comes round again, after its
        previous pass has completed and before the new one touches it (and by `flush` for the passes still pending at
        the end): the same exchange, one pipeline depth later, without a stream of its own -- every stream more shifts
        the hardware queue the others land on (bench.py, GPU_MAX_HW_QUEUES), and the head stream has the slack.  Hooks of
        different slots run on different head streams, possibly at the same time: keep their state per slot."""
        torch = _torch()
        k = self.i % len(self.slots)
        self.i += 1
        slot, hs = self.slots[k], self.head_streams[k]
        if self.tail_done[k] is not None:  # the slot's workspace is free again
            hs.wait_event(self.tail_done[k])
            if on_reuse is not None:
                with torch.cuda.stream(hs):
                    on_reuse(k)
        slot.run(hs, PHASE_HEAD)
        # a fresh event per pass: re-recording one event inside a stream capture crashes
        # hipStreamEndCapture on ROCm 7.0 (tools/probes/graph_capture_probe.py)
        self.head_done[k] = torch.cuda.Event()
        self.head_done[k].record(hs)
        ts = self.tail_streams[(self.i - 1) % len(self.tail_streams)]
        ts.wait_event(self.head_done[k])
        slot.run(ts, PHASE_TAIL)
        ev = torch.cuda.Event()
        ev.record(ts)
        if after_tail is not None:
            # e.g. the RCCL accept-reduce: on its own stream, so the next Merkle kernel on the
            # tail stream never waits for a collective (or for a slower rank)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                after_tail(k)
            ev = torch.cuda.Event()
            ev.record(self.comm_stream)
        self.tail_done[k] = ev
        return k

    def flush(self, on_reuse) -> None:
        """on_reuse for every slot whose last pass has not had it yet (call once after the last submit)."""
        torch = _torch()
        for k, hs in enumerate(self.head_streams):
            if self.tail_done[k] is not None:
                hs.wait_event(self.tail_done[k])
                with torch.cuda.stream(hs):
                    on_reuse(k)
                self.tail_done[k] = None

    def synchronize(self) -> None:
        for s in self.head_streams:
            s.synchronize()
        for s in self.tail_streams:
            s.synchronize()
        self.comm_stream.synchronize()


class IndependentStreams:
    """Every slot on its own stream, HEAD and TAIL back to back, no events between slots: for batches
    too small to fill the chip (a stark101 x 4096 pass is 64 transcript waves, then 1 472 short Merkle
    waves, for 1 024 SIMDs) many whole passes must overlap, and one library call per pass (two memsets
    and three launches) is cheap enough to enqueue eagerly.  Kernel durations measured under this
    overlap are not meaningful; take them from a Pipeline pass."""

    def __init__(self, slots: Sequence["_DeviceBatch"]):
        torch = _torch()
        self.slots = list(slots)
        dev = self.slots[0].ver.device
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.slots]
        self.i = 0
        self.wait_current()

    def wait_current(self) -> None:
        """Order every later pass after what the caller has enqueued on its current stream so far (see Pipeline)."""
        _after_current(self.streams, self.slots[0].ver.device)

    def submit(self) -> int:
        k = self.i % len(self.slots)
        self.i += 1
        self.slots[k].run(self.streams[k], PHASE_ALL)  # same slot, same stream: ordered by the stream
        return k

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()


class GraphedPipeline:
    """One pipelined pass over every slot captured ONCE into a hipGraph (the head streams and the
    tail stream fork from and join the capture stream), then replayed with a single launch.

    For small batches (stark101, a few thousand proofs) a pass costs less GPU time than the ~10
    host-side launches and event operations that enqueue it; replaying a graph removes that bound.
    Within a replay all HEAD halves run concurrently and the TAIL halves chain; replays on one
    stream serialize, so keep two GraphedPipelines (own slots, own streams) in flight to overlap
    the HEAD latency of one with the Merkle kernels of the other (tools/graph_bench.py).

    Each slot is used once per graph: a head stream that waits again on a tail-stream event inside
    the capture crashes hipStreamEndCapture on ROCm 7.0 (tools/probes/graph_capture_probe.py).
    Per-kernel timing (`Verifier.set_timing`) must be off: timing events cannot be captured."""

    def __init__(self, slots: Sequence["_DeviceBatch"], concurrent_tails: bool = False):
        """concurrent_tails: every slot is its own branch of the graph (HEAD then TAIL on its own
        stream), so the Merkle kernels of different slots overlap too.  For batches too small to
        fill the chip on their own (stark101 x 4096 is 1 472 short waves for 1 024 SIMDs)."""
        torch = _torch()
        ver = slots[0].ver
        if ver.timing:
            raise RuntimeError("GraphedPipeline: switch Verifier.set_timing off before capturing")
        self.slots = list(slots)
        self.steps_per_replay = len(self.slots)
        self.stream = torch.cuda.Stream(device=ver.device)
        if concurrent_tails:
            pipe = None
            branches = [torch.cuda.Stream(device=ver.device) for _ in self.slots]
        else:
            pipe = Pipeline(self.slots)
            branches = pipe.head_streams + [pipe.tail_stream]
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(ver.device)
        with torch.cuda.graph(self.graph, stream=self.stream):
            fork = torch.cuda.Event()
            fork.record(self.stream)
            for s in branches:
                s.wait_event(fork)
            if concurrent_tails:
                for slot, s in zip(self.slots, branches):
                    slot.run(s, PHASE_ALL)
            else:
                for _ in self.slots:
                    pipe.submit()
            for s in branches:
                join = torch.cuda.Event()
                join.record(s)
                self.stream.wait_event(join)
        self._pipe, self._branches = pipe, branches  # keeps the captured streams and events alive

    def replay(self) -> int:
        """Enqueue one pass over every slot, ordered after what the caller's current stream holds so far (torch's own
        CUDAGraph.replay runs ON the current stream; this one runs on the capture stream, so it joins it instead);
        returns the number of passes."""
        _after_current([self.stream], self.slots[0].ver.device)
        with _torch().cuda.stream(self.stream):
            self.graph.replay()
        return self.steps_per_replay

    def synchronize(self) -> None:
        self.stream.synchronize()


# ----------------------------------------------------------------------------- records
def _hash_words(b) -> np.ndarray:
    """32*k bytes -> big-endian u32 words (SHA-256 state words)."""
    return np.frombuffer(bytes(b), dtype=">u4").astype(np.uint32)


def _fixed_path(path: np.ndarray, n: int) -> np.ndarray:
    """uint8[len, 32] -> uint32[n * 8], truncated / zero padded to n levels."""
    out = np.zeros(n * 8, dtype=np.uint32)
    k = min(len(path), n)
    if k:
        out[:k * 8] = np.ascontiguousarray(path[:k]).view(">u4").astype(np.uint32).reshape(-1)
    return out


def stwo_code(stage: int, layer: int, query: int, sub: int) -> int:
    return (stage << 24) | (layer << 16) | (query << 4) | sub


def stwo_record(p: StwoProof) -> np.ndarray:
    """Proof -> record (layout in include/ss_verify.h).  Merkle paths go into fixed slots (first
    min(len, slot) siblings, zero padded) and their real lengths into the path_len trailer; what a
    wrong length means for the verdict is decided by the library (merkle.simf:42), not here."""
    c = p.cfg
    N, L, Q, K = c.n_cols, c.lde_log, c.n_queries, c.n_layers
    parts: List[np.ndarray] = [
        _hash_words(p.roots.tobytes()), p.oods_trace.astype(np.uint32).reshape(-1),
        p.oods_cp.astype(np.uint32).reshape(-1), _hash_words(p.fri_roots.tobytes()),
        p.last_layer.astype(np.uint32).reshape(-1),
        np.array([p.pow_nonce >> 32, p.pow_nonce & 0xFFFFFFFF], dtype=np.uint32)]
    for q in range(Q):
        parts += [p.trace_vals[q].astype(np.uint32), p.cp_vals[q].astype(np.uint32),
                  _fixed_path(p.trace_paths[q], L), _fixed_path(p.cp_paths[q], L)]
    for l in range(K + 1):
        n = L - 1 - l
        for q in range(Q):
            parts += [p.fri_witness[l, q].astype(np.uint32), _fixed_path(p.fri_paths[l][q], n)]
    lens = [[len(x) for x in p.trace_paths], [len(x) for x in p.cp_paths]]
    lens += [[len(x) for x in p.fri_paths[l]] for l in range(K + 1)]
    parts.append(np.array(lens, dtype=np.uint32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)


def stwo_shared_record(p: StwoProof, queries: "Sequence[int] | None" = None, mode: int = MODE_FIXTURE) -> np.ndarray:
    """Proof -> SHARED record (include/ss_verify.h: every distinct Merkle sibling of a tree once + the query
    positions as a hint; the reference repeats them per query, fri/queries.simf:41).  `queries` are the
    positions the prover drew; without them the public transcript is replayed (formats.stwo_queries).
    Raises ValueError for a proof that has no shared form (a path of another length, or two queries that
    present different bytes for one node) -- such a proof travels as a per-query record."""
    from .formats import stwo_queries
    cs = stwo_cfg_struct(p.cfg, mode)
    L = B.lib()
    rec = stwo_record(p)
    qs = np.ascontiguousarray(queries if queries is not None else stwo_queries(p), dtype=np.uint32)
    if qs.size != p.cfg.n_queries:
        raise ValueError("one position per query expected")
    out = np.zeros(L.ss_stwo_shared_max_words(C.byref(cs)), dtype=np.uint32)
    words = C.c_size_t(0)
    rc = B.check(L.ss_stwo_share_record(C.byref(cs), rec.ctypes.data, qs.ctypes.data, out.ctypes.data, out.size,
                                        C.byref(words)))
    if rc:
        raise ValueError("this proof has no shared form")
    return out[:words.value].copy()


def stwo_unshare_record(cfg: StwoConfig, shared: np.ndarray, mode: int = MODE_FIXTURE):
    """Shared record -> (outcome, per-query record) on the host (ss_stwo_unshare_record): outcome 0 or
    STATUS_MALFORMED.  The GPU does the same in ss_stwo_expand_shared_dev."""
    cs = stwo_cfg_struct(cfg, mode)
    L = B.lib()
    sh = np.ascontiguousarray(shared, dtype=np.uint32)
    rec = np.zeros(L.ss_stwo_record_words(C.byref(cs)), dtype=np.uint32)
    return B.check(L.ss_stwo_unshare_record(C.byref(cs), sh.ctypes.data, sh.size, rec.ctypes.data)), rec


def stwo_shared_counts(cfg: StwoConfig, queries: Sequence[int], mode: int = MODE_FIXTURE) -> np.ndarray:
    """Distinct siblings per tree (trace, composition, FRI layer 0..) for these positions (ss_stwo_shared_counts)."""
    cs = stwo_cfg_struct(cfg, mode)
    qs = np.ascontiguousarray(queries, dtype=np.uint32)
    out = np.zeros(cfg.n_layers + 3, dtype=np.uint32)
    B.check(B.lib().ss_stwo_shared_counts(C.byref(cs), qs.ctypes.data, out.ctypes.data))
    return out


def stwo_minimal_record(m) -> np.ndarray:
    """StwoMinimalProof -> minimal record (include/ss_verify.h: head, the list lengths, then the lists)."""
    c = m.cfg
    K = c.n_layers
    if len(m.fri_witness) != K + 1 or len(m.hash_witness) != K + 3:
        raise ValueError("a minimal proof has one fri_witness list per layer and one hash_witness list per tree")
    parts: List[np.ndarray] = [
        _hash_words(m.roots.tobytes()), m.oods_trace.astype(np.uint32).reshape(-1),
        m.oods_cp.astype(np.uint32).reshape(-1), _hash_words(m.fri_roots.tobytes()),
        m.last_layer.astype(np.uint32).reshape(-1),
        np.array([m.pow_nonce >> 32, m.pow_nonce & 0xFFFFFFFF], dtype=np.uint32),
        np.array([len(m.trace_vals), len(m.cp_vals)] + [len(w) for w in m.fri_witness] + [len(h) for h in m.hash_witness],
                 dtype=np.uint32),
        np.asarray(m.trace_vals, dtype=np.uint32).reshape(-1), np.asarray(m.cp_vals, dtype=np.uint32).reshape(-1)]
    parts += [np.asarray(w, dtype=np.uint32).reshape(-1) for w in m.fri_witness]
    parts += [_hash_words(np.ascontiguousarray(h, dtype=np.uint8).tobytes()) for h in m.hash_witness]
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)


def stwo_minimal_from_record(cfg: StwoConfig, rec: np.ndarray):
    """Inverse of stwo_minimal_record (ValueError when the size is not what the record's own counts give)."""
    from .formats import StwoMinimalProof
    N, K = cfg.n_cols, cfg.n_layers
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    head = 24 + 4 * N + 64 + 8 * (K + 1) + 6
    if rec.size < head + 2 + (K + 1) + (K + 3):
        raise ValueError("shorter than the fixed words of a minimal record")
    pos = 0

    def take(n: int) -> np.ndarray:
        nonlocal pos
        if pos + n > rec.size:
            raise ValueError("the record ends inside a list")
        out = rec[pos:pos + n]
        pos += n
        return out

    def hashes(words: np.ndarray) -> np.ndarray:
        return words.astype(">u4").view(np.uint8).reshape(-1, 32).copy()
    roots = hashes(take(24))
    oods_trace = take(4 * N).reshape(N, 4).copy()
    oods_cp = take(64).reshape(16, 4).copy()
    fri_roots = hashes(take(8 * (K + 1)))
    last = take(4).copy()
    hi, lo = take(2)
    nv = [int(x) for x in take(2)]
    nfw = [int(x) for x in take(K + 1)]
    nhw = [int(x) for x in take(K + 3)]
    tv = take(nv[0] * N).reshape(-1, N).copy()
    cv = take(nv[1] * 16).reshape(-1, 16).copy()
    fw = [take(4 * n).reshape(-1, 4).copy() for n in nfw]
    hw = [hashes(take(8 * n)) for n in nhw]
    if pos != rec.size:
        raise ValueError("the record is longer than its lists")
    return StwoMinimalProof(cfg, roots, oods_trace, oods_cp, fri_roots, last, (int(hi) << 32) | int(lo), tv, cv, fw, hw)


def stwo_minimise_record(cfg: StwoConfig, record: np.ndarray, queries: Sequence[int], mode: int = MODE_FIXTURE) -> np.ndarray:
    """Per-query record + the positions its prover drew -> MINIMAL record (ss_stwo_minimise_record: a selection, no
    hashing).  ValueError for a proof that has no minimal form."""
    cs = stwo_cfg_struct(cfg, mode)
    L = B.lib()
    rec = np.ascontiguousarray(record, dtype=np.uint32)
    qs = np.ascontiguousarray(queries, dtype=np.uint32)
    if qs.size != cfg.n_queries or rec.size != L.ss_stwo_record_words(C.byref(cs)):
        raise ValueError("a record of the config and one position per query expected")
    out = np.zeros(L.ss_stwo_minimal_max_words(C.byref(cs)), dtype=np.uint32)
    words = C.c_size_t(0)
    rc = B.check(L.ss_stwo_minimise_record(C.byref(cs), rec.ctypes.data, qs.ctypes.data, out.ctypes.data, out.size, C.byref(words)))
    if rc:
        raise ValueError("this proof has no minimal form")
    return out[:words.value].copy()


READER_AUTO, READER_GENERAL, READER_STREAM, READER_DECLINED = 0, 1, 2, 3


def parse_stwo_minimal_text(cfg: StwoConfig, text: bytes, mode: int = MODE_FIXTURE, reader: int = READER_AUTO):
    """One minimal proof.json -> (outcome, minimal record or None) with the library's native readers
    (ss_stwo_parse_minimal_route; no GPU): outcome 0 = parsed, STATUS_CONFIG_MISMATCH (1), STATUS_MALFORMED (2);
    reader=READER_STREAM asks the streaming reader alone, which answers READER_DECLINED (3) for a text it leaves to the
    general one."""
    cs = stwo_cfg_struct(cfg, mode)
    L = B.lib()
    text = bytes(text)
    out = np.zeros(L.ss_stwo_minimal_max_words(C.byref(cs)), dtype=np.uint32)
    words = C.c_size_t(0)
    rc = B.check(L.ss_stwo_parse_minimal_route(C.byref(cs), text, len(text), reader, out.ctypes.data, out.size, C.byref(words)))
    return rc, (out[:words.value].copy() if rc == 0 else None)


def stwo_minimal_from_capacity(cfg: StwoConfig, capacity: np.ndarray, mode: int = MODE_FIXTURE) -> np.ndarray:
    """Capacity-form minimal record (every list at a fixed base: what the GPU reader of the minimal proof.json writes and
    ss_stwo_text_is_canonical returns for SS_TEXT_JSON_MINIMAL) -> the minimal record of include/ss_verify.h."""
    cs = stwo_cfg_struct(cfg, mode)
    cap = np.ascontiguousarray(capacity, dtype=np.uint32)
    out = np.zeros(cap.size, dtype=np.uint32)
    words = C.c_size_t(0)
    B.check(B.lib().ss_stwo_minimal_from_capacity(C.byref(cs), cap.ctypes.data, out.ctypes.data, out.size, C.byref(words)))
    return out[:words.value].copy()


def stwo_minimal_text_is_canonical(cfg: StwoConfig, text: bytes, mode: int = MODE_FIXTURE):
    """Would the GPU reader take this minimal proof.json?  -> (taken, minimal record or None): the scalar statement of its rule
    (ss_stwo_text_is_canonical with SS_TEXT_JSON_MINIMAL; no GPU)."""
    cs = stwo_cfg_struct(cfg, mode)
    L = B.lib()
    text = bytes(text)
    cap = np.zeros(L.ss_stwo_minimal_max_words(C.byref(cs)), dtype=np.uint32)
    r = B.check(L.ss_stwo_text_is_canonical(C.byref(cs), text, len(text), B.TEXT_JSON_MINIMAL, cap.ctypes.data))
    return bool(r), (stwo_minimal_from_capacity(cfg, cap, mode) if r else None)


def write_stwo_minimal_text(cfg: StwoConfig, minimal: np.ndarray, python_separators: bool = True, mode: int = MODE_FIXTURE) -> bytes:
    """Minimal record -> the minimal proof.json, byte for byte what json.dumps prints for formats.stwo_minimal_to_json
    (ss_stwo_write_minimal_text).  ValueError when `minimal` is no minimal record of the config."""
    cs = stwo_cfg_struct(cfg, mode)
    L = B.lib()
    rec = np.ascontiguousarray(minimal, dtype=np.uint32)
    n = L.ss_stwo_write_minimal_text(C.byref(cs), rec.ctypes.data, rec.size, 1 if python_separators else 0, None, 0)
    if n == 0:
        raise ValueError("not a minimal record of this config")
    buf = C.create_string_buffer(n)
    L.ss_stwo_write_minimal_text(C.byref(cs), rec.ctypes.data, rec.size, 1 if python_separators else 0, buf, n)
    return buf.raw[:n]


def stwo_minimal_counts(cfg: StwoConfig, queries: Sequence[int], mode: int = MODE_FIXTURE) -> np.ndarray:
    """List lengths of a minimal record for these positions: [n_vals x 2, n_fw per layer, n_hw per tree]."""
    cs = stwo_cfg_struct(cfg, mode)
    qs = np.ascontiguousarray(queries, dtype=np.uint32)
    out = np.zeros(2 + (cfg.n_layers + 1) + (cfg.n_layers + 3), dtype=np.uint32)
    B.check(B.lib().ss_stwo_minimal_counts(C.byref(cs), qs.ctypes.data, out.ctypes.data))
    return out


STATUS_CONFIG_MISMATCH = 1
"""Status of a proof whose shape / declared parameters are not the config the caller expects.  The
reference fixes NUM_COLUMNS, LDE_LOG_SIZE, NUM_FRI_QUERIES, NUM_FRI_LAYERS and POW_TARGET_64 at
compile time (stwo-verifier/src/config.simf:10-51); a witness of any other shape fails typing and
`simfony run` exits 1 before executing anything (simfony-cli/src/main.rs:77-81,187-190).  Stage 0
precedes every assert of verify_proof, so this is the smallest status code."""


def apply_config_policy(proofs: Sequence[StwoProof], cfg) -> Tuple[np.ndarray, List[List[int]]]:
    """-> (status, groups): status[i] = STATUS_CONFIG_MISMATCH for every proof whose config is not
    the expected one (or on the allow-list) and 0xFFFFFFFF (not yet verified = REJECT) for the
    others; groups = indices of the admissible proofs, one list per config."""
    allowed = _allowed_configs(cfg)
    status = np.full(len(proofs), 0xFFFFFFFF, dtype=np.uint32)
    ok = [i for i, p in enumerate(proofs) if p.cfg in allowed]
    for i, p in enumerate(proofs):
        if p.cfg not in allowed:
            status[i] = STATUS_CONFIG_MISMATCH
    return status, [[ok[j] for j in g] for g in group_by_config([proofs[i] for i in ok])]


def _allowed_configs(cfg) -> List[StwoConfig]:
    if isinstance(cfg, StwoConfig):
        return [cfg]
    allowed = list(cfg)
    if not allowed or not all(isinstance(c, StwoConfig) for c in allowed):
        raise TypeError("cfg must be the expected StwoConfig or a non-empty allow-list of them")
    return allowed


def merkle_node_hashes(cfg: StwoConfig, queries: Sequence[int], levels: Optional[int] = None) -> Tuple[int, int]:
    """-> (node hashes of the reference algorithm, node hashes the kernels execute) for one proof
    with these query indices.  The reference hashes every sibling of every query's path
    (fri/queries.simf:41: no deduplication); stwo_top_kernel hashes each distinct pair of the top
    `levels` levels once (default: what csrc/ss_layout.h picks).  Depth d of every tree of a proof
    has position query >> (lde_log - d)."""
    L, Q, K = cfg.lde_log, cfg.n_queries, cfg.n_layers
    if levels is None:
        levels = min((Q - 1).bit_length() + (1 if cfg.hash == "blake2s" else 2), L) if Q > 1 else 0
    lens = [L, L] + [L - 1 - l for l in range(K + 1)]
    full = Q * sum(lens)
    done = 0
    for ln in lens:
        top = min(levels, ln)
        done += Q * (ln - top) + sum(len({int(q) >> (L - d) for q in queries}) for d in range(top))
    return full, done


def parse_stwo_text(cfg: StwoConfig, text: bytes, mode: int = MODE_FIXTURE, fmt: int = B.TEXT_AUTO):
    """One proof.json / proof.wit text -> (outcome, record) with the library's native reader
    (ss_stwo_parse; no GPU): outcome 0 = parsed, STATUS_CONFIG_MISMATCH (1), STATUS_MALFORMED (2)."""
    cs = stwo_cfg_struct(cfg, mode)
    rec = np.zeros(B.lib().ss_stwo_record_words(C.byref(cs)), dtype=np.uint32)
    text = bytes(text)
    return B.check(B.lib().ss_stwo_parse(C.byref(cs), text, len(text), fmt, rec.ctypes.data)), rec


def parse_s101_text(text: bytes, fmt: int = B.TEXT_AUTO):
    """-> (outcome, (n_layers, max_path), record or None) with the native reader (ss_s101_parse)."""
    text = bytes(text)
    sh = B.S101Shape(0, 0)
    rc = B.check(B.lib().ss_s101_parse(text, len(text), fmt, C.byref(sh), None))
    if rc:
        return rc, None, None
    rec = np.zeros(B.lib().ss_s101_record_words(C.byref(sh)), dtype=np.uint32)
    B.check(B.lib().ss_s101_parse(text, len(text), fmt, C.byref(sh), rec.ctypes.data))
    return 0, (sh.max_layers, sh.max_path), rec


def group_by_config(proofs: Sequence[StwoProof]) -> List[List[int]]:
    """Indices of `proofs` grouped by StwoConfig, groups in order of first appearance."""
    groups: dict = {}
    for i, p in enumerate(proofs):
        groups.setdefault(p.cfg, []).append(i)
    return list(groups.values())


def s101_shape_of(proofs: Sequence[Stark101Proof]) -> Tuple[int, int]:
    ml = max([len(p.layers) for p in proofs] + [0])
    pm = 0
    for p in proofs:
        pm = max([pm] + [len(e.path) for e in p.evals]
                 + [len(x.path) for l in p.layers for x in (l.cpa, l.cpb)])
    return ml, pm


def s101_record(p: Stark101Proof, max_layers: int, max_path: int) -> np.ndarray:
    def chain(e) -> List[np.ndarray]:
        return [np.array([e.ev, len(e.path)], dtype=np.uint32), _fixed_path(e.path, max_path)]
    parts: List[np.ndarray] = [_hash_words(p.root),
                               np.array([len(p.layers), p.last], dtype=np.uint32)]
    for e in p.evals:
        parts += chain(e)
    for i in range(max_layers):
        if i < len(p.layers):
            l = p.layers[i]
            parts += [_hash_words(l.root), np.array([l.beta], dtype=np.uint32)]
            parts += chain(l.cpa) + chain(l.cpb)
        else:
            parts.append(np.zeros(9 + 2 * (2 + 8 * max_path), dtype=np.uint32))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)


def _ptr_array(records):
    """`const uint32_t *const *` for the C ABI.  A 2-d C-contiguous array (one record per row) costs nothing per record:
    the row addresses are computed by numpy; a list of arrays costs about a microsecond each."""
    if isinstance(records, np.ndarray) and records.ndim == 2 and records.flags["C_CONTIGUOUS"]:
        ptrs = records.ctypes.data + np.arange(records.shape[0], dtype=np.uint64) * np.uint64(records.strides[0])
        arr = (C.c_void_p * records.shape[0]).from_buffer(ptrs)
        arr._keepalive = (ptrs, records)
        return arr
    arr = (C.c_void_p * len(records))()
    for i, r in enumerate(records):
        arr[i] = r.ctypes.data
    return arr


FLAG_NO_DEDUP = 1  # SS_FLAG_NO_DEDUP: hash every query's Merkle path in full (A/B and tests)
FLAG_TOP_CHECKS = 2  # SS_FLAG_TOP_CHECKS: the memoisation's byte compares in the top kernel whatever the query count


def stwo_cfg_struct(cfg: StwoConfig, mode: int, flags: int = 0) -> B.StwoCfg:
    return B.StwoCfg(cfg.n_cols, cfg.trace_log, cfg.lde_log, cfg.n_queries, cfg.n_layers, mode,
                     cfg.pow_target, 1 if cfg.hash == "blake2s" else 0, flags)


def pack_stwo(cfg: StwoConfig, mode: int, records: Sequence[np.ndarray], flags: int = 0) -> np.ndarray:
    """records (one per proof; entries may repeat) -> host batch buffer (uint32).  A batch is laid out for
    the cfg it will be verified with, hash family and SS_FLAG_* included (csrc/ss_layout.h: with pair
    memoisation the top levels of every tree are stored per proof, not in the 64-chain tiles)."""
    L = B.lib()
    cs = stwo_cfg_struct(cfg, mode, flags)
    want = L.ss_stwo_record_words(C.byref(cs))
    if want == 0:
        raise B.SsError(B.SS_ERR_ARG, "unsupported stwo config %r" % (cfg,))
    for r in records:
        if r.dtype != np.uint32 or r.size != want or not r.flags["C_CONTIGUOUS"]:
            raise ValueError("record must be %d contiguous uint32 words" % want)
    out = np.empty(L.ss_stwo_batch_words(C.byref(cs), len(records)), dtype=np.uint32)
    B.check(L.ss_stwo_pack(C.byref(cs), len(records), _ptr_array(records), out.ctypes.data))
    return out


def pack_s101(max_layers: int, max_path: int, records: Sequence[np.ndarray]) -> np.ndarray:
    L = B.lib()
    sh = B.S101Shape(max_layers, max_path)
    want = L.ss_s101_record_words(C.byref(sh))
    for r in records:
        if r.dtype != np.uint32 or r.size != want or not r.flags["C_CONTIGUOUS"]:
            raise ValueError("record must be %d contiguous uint32 words" % want)
    out = np.empty(L.ss_s101_batch_words(C.byref(sh), len(records)), dtype=np.uint32)
    B.check(L.ss_s101_pack(C.byref(sh), len(records), _ptr_array(records), out.ctypes.data))
    return out


# ---------------------------------------------------------------------- device batches
def _torch():
    import torch
    return torch


def _to_dev(a: np.ndarray, device):
    torch = _torch()
    return torch.from_numpy(a.view(np.int32)).to(device, non_blocking=False)


class _DeviceBatch:
    """A batch resident in HBM plus its workspace / status buffers."""

    def __init__(self, ver: "Verifier", n: int, batch_host, ws_bytes: int):
        """`batch_host`: the packed batch as a numpy array (uploaded here) or as an int32 device
        tensor that is already resident."""
        torch = _torch()
        self.ver, self.n = ver, n
        self.batch = batch_host if torch.is_tensor(batch_host) else _to_dev(batch_host, ver.device)
        self.ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.int32, device=ver.device)
        self.status_dev = torch.empty(n, dtype=torch.int32, device=ver.device)
        self.accept_dev = torch.zeros(1, dtype=torch.int32, device=ver.device)
        self.batch_bytes = self.batch.numel() * 4
        # nothing of the construction (the fill above, a subclass's device gather) may still be pending on the caller's
        # current stream when a pass starts on another stream: passes run on non-blocking side streams (_after_current)
        torch.cuda.current_stream(ver.device).synchronize()

    def _stream(self, stream) -> int:
        torch = _torch()
        s = stream if stream is not None else torch.cuda.current_stream(self.ver.device)
        return int(s.cuda_stream)

    def sibling(self):
        """Another run slot over the SAME resident batch (own workspace / status / accept
        count), so two passes can be in flight on two streams: the sequential transcript
        kernel of one overlaps the Merkle kernel of the other."""
        import copy
        torch = _torch()
        other = copy.copy(self)
        other.ws = torch.empty_like(self.ws)
        other.status_dev = torch.empty_like(self.status_dev)
        other.accept_dev = torch.zeros_like(self.accept_dev)
        torch.cuda.current_stream(self.ver.device).synchronize()  # as in __init__: no fill left pending
        return other

    def status(self) -> np.ndarray:
        """Waits for every stream of the device (the passes may have run on any of them), returns the per-proof
        status words (0 = ACCEPT)."""
        _torch().cuda.synchronize(self.ver.device)
        return self.status_dev.cpu().numpy().view(np.uint32).copy()

    def accepted(self) -> int:
        """Waits like status(); the number of accepted proofs of the last pass."""
        _torch().cuda.synchronize(self.ver.device)
        return int(self.accept_dev.item())


class StwoDeviceBatch(_DeviceBatch):
    def __init__(self, ver: "Verifier", cfg: StwoConfig, mode: int, records: Sequence[np.ndarray],
                 index: Optional[Sequence[int]] = None):
        """`index` (optional): proof i of the batch is records[index[i]].  The distinct records are
        then uploaded once, replicated by a device gather and re-tiled by ss_stwo_pack_dev, so a
        65 536-proof batch made of a few distinct proofs costs no 11 GB host pack and upload."""
        L = B.lib()
        self.cfg, self.mode = cfg, mode
        self.cs = stwo_cfg_struct(cfg, mode, ver.stwo_flags)
        if index is None:
            n = len(records)
            packed = pack_stwo(cfg, mode, records, ver.stwo_flags)
        else:
            torch = _torch()
            n = len(index)
            distinct = _to_dev(np.ascontiguousarray(np.stack(records), dtype=np.uint32), ver.device)
            idx = torch.as_tensor(np.asarray(index, dtype=np.int64), device=ver.device)
            full = distinct[idx].contiguous()
            packed = torch.empty(L.ss_stwo_batch_words(C.byref(self.cs), n), dtype=torch.int32, device=ver.device)
            B.check(L.ss_stwo_pack_dev(ver.ctx, C.byref(self.cs), n, full.data_ptr(), packed.data_ptr(),
                                       int(torch.cuda.current_stream(ver.device).cuda_stream)))
            torch.cuda.synchronize(ver.device)
            del full, distinct
        super().__init__(ver, n, packed, L.ss_stwo_workspace_bytes(C.byref(self.cs), n))

    def run(self, stream=None, phases: int = PHASE_ALL) -> None:
        """Asynchronous: enqueue the verification of the whole batch (or one half of it, see
        SS_PHASE_* in include/ss_verify.h) on `stream`."""
        B.check(B.lib().ss_stwo_verify_phase_dev(
            self.ver.ctx, C.byref(self.cs), self.n, self.batch.data_ptr(),
            self.ws.data_ptr(), self.ws.numel() * 4, self.status_dev.data_ptr(),
            self.accept_dev.data_ptr(), phases, self._stream(stream)))

    def intermediates(self, proof: int, stream=None) -> dict:
        """Per-stage values of proof `proof` left by the last run (ss_stwo_read_intermediates; the
        reference's counterpart is the dbg! tracker, simfony-cli/src/tracker.rs:48-80)."""
        Q, K = self.cfg.n_queries, self.cfg.n_layers
        out = {"queries": np.zeros(Q, np.uint32), "oods_point": np.zeros(8, np.uint32),
               "deep_alpha": np.zeros(4, np.uint32), "fold_alphas": np.zeros((K + 1, 4), np.uint32),
               "fri_answers": np.zeros((Q, 4), np.uint32)}
        B.check(B.lib().ss_stwo_read_intermediates(
            self.ver.ctx, C.byref(self.cs), self.n, self.ws.data_ptr(), proof, self._stream(stream),
            out["queries"].ctypes.data, out["oods_point"].ctypes.data, out["deep_alpha"].ctypes.data,
            out["fold_alphas"].ctypes.data, out["fri_answers"].ctypes.data))
        return out


class StwoMinimalDeviceBatch(_DeviceBatch):
    """Minimal records resident in HBM (back to back + their offset table) with the scratch batch / workspace the
    verification fills: ss_stwo_verify_minimal_dev.  `index`: proof i is records[index[i]] (replicated on the device)."""

    def __init__(self, ver: "Verifier", cfg: StwoConfig, mode: int, records: Sequence[np.ndarray],
                 index: Optional[Sequence[int]] = None):
        torch = _torch()
        L = B.lib()
        self.cfg, self.mode = cfg, mode
        self.cs = stwo_cfg_struct(cfg, mode, ver.stwo_flags)
        order = list(range(len(records))) if index is None else [int(i) for i in index]
        n = len(order)
        sizes = np.array([records[i].size for i in order], dtype=np.uint64)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(sizes)
        distinct = _to_dev(np.concatenate([np.ascontiguousarray(r, dtype=np.uint32) for r in records]), ver.device)
        starts = np.zeros(len(records) + 1, dtype=np.int64)
        starts[1:] = np.cumsum([r.size for r in records])
        if index is None:
            flat = distinct
        else:  # gather on the device: element j of proof i comes from starts[order[i]] + j
            src0 = torch.as_tensor(starts[:-1][order], device=ver.device)
            rep = torch.as_tensor(sizes.astype(np.int64), device=ver.device)
            base = torch.repeat_interleave(src0 - torch.as_tensor(offs[:-1].astype(np.int64), device=ver.device), rep)
            flat = distinct[base + torch.arange(int(offs[-1]), device=ver.device)]
        self.records = flat
        self.offs = torch.from_numpy(offs.view(np.int64)).to(ver.device)
        self.record_bytes = int(offs[-1]) * 4
        packed = torch.empty(L.ss_stwo_minimal_batch_words(C.byref(self.cs), n), dtype=torch.int32, device=ver.device)
        super().__init__(ver, n, packed, L.ss_stwo_minimal_workspace_bytes(C.byref(self.cs), n))

    def run(self, stream=None, phases: int = PHASE_ALL) -> None:
        B.check(B.lib().ss_stwo_verify_minimal_dev(
            self.ver.ctx, C.byref(self.cs), self.n, self.records.data_ptr(), self.offs.data_ptr(), self.batch.data_ptr(),
            self.ws.data_ptr(), self.ws.numel() * 4, self.status_dev.data_ptr(), self.accept_dev.data_ptr(), phases,
            self._stream(stream)))


class S101DeviceBatch(_DeviceBatch):
    def __init__(self, ver: "Verifier", max_layers: int, max_path: int, records: Sequence[np.ndarray]):
        L = B.lib()
        self.sh = B.S101Shape(max_layers, max_path)
        host = pack_s101(max_layers, max_path, records)
        super().__init__(ver, len(records), host, L.ss_s101_workspace_bytes(C.byref(self.sh), len(records)))

    def run(self, stream=None, phases: int = PHASE_ALL) -> None:
        B.check(B.lib().ss_s101_verify_phase_dev(
            self.ver.ctx, C.byref(self.sh), self.n, self.batch.data_ptr(), self.ws.data_ptr(),
            self.ws.numel() * 4, self.status_dev.data_ptr(), self.accept_dev.data_ptr(),
            phases, self._stream(stream)))

    def intermediates(self, proof: int, stream=None) -> dict:
        """Per-stage values of proof `proof` left by the last run (ss_s101_read_intermediates): the composition
        coefficients, query, x, cp, the value entering every FRI layer (+ the final one), the channel state before the
        query draw -- what stark101/scripts/fibsquare/prover_test.py:32-104 recomputes."""
        ml = self.sh.max_layers
        out = {"alphas": np.zeros(3, np.uint32), "idx": np.zeros(1, np.uint32), "x": np.zeros(1, np.uint32),
               "cp": np.zeros(1, np.uint32), "folds": np.zeros(ml + 1, np.uint32), "state": np.zeros(8, np.uint32)}
        B.check(B.lib().ss_s101_read_intermediates(
            self.ver.ctx, C.byref(self.sh), self.n, self.ws.data_ptr(), proof, self._stream(stream),
            out["alphas"].ctypes.data, out["idx"].ctypes.data, out["x"].ctypes.data, out["cp"].ctypes.data,
            out["folds"].ctypes.data, out["state"].ctypes.data))
        return {k: (int(v[0]) if k in ("idx", "x", "cp") else v) for k, v in out.items()}


class Verifier:
    """One context on one MI355X.  Raises SsError if the GPU or the library is unusable."""

    def __init__(self, device: int = 0):
        torch = _torch()
        self.index = device
        self.device = torch.device("cuda", device)
        ctx = C.c_void_p()
        B.check(B.lib().ss_ctx_create(device, C.byref(ctx)))
        self.ctx = ctx
        self.timing = False
        self.stwo_flags = 0  # SS_FLAG_* for the batches this verifier builds
        torch.cuda.set_device(self.device)

    def close(self) -> None:
        if self.ctx:
            B.lib().ss_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- timing -----------------------------------------------------------------------
    def set_timing(self, on: bool) -> None:
        B.check(B.lib().ss_ctx_set_timing(self.ctx, 1 if on else 0))
        self.timing = bool(on)

    def collect_timing(self) -> dict:
        """{kernel name: (total_ms, launches)} since the last collect (waits for the events)."""
        cap = 64  # more kernel names than the library has
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        cnt = (C.c_uint32 * cap)()
        k = B.check(B.lib().ss_ctx_collect_timing(self.ctx, cap, names, ms, cnt))
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    # -- stark101 ---------------------------------------------------------------------
    def stark101_batch(self, proofs: Sequence[Stark101Proof], replicate: int = 1) -> S101DeviceBatch:
        ml, pm = s101_shape_of(proofs)
        recs = [s101_record(p, ml, pm) for p in proofs]
        return S101DeviceBatch(self, ml, pm, recs * replicate)

    def verify_stark101(self, proofs: Sequence[Stark101Proof]) -> np.ndarray:
        if len(proofs) == 0:  # the C ABI rejects empty batches; an empty list has an empty answer
            return np.empty(0, dtype=np.uint32)
        b = self.stark101_batch(proofs)
        b.run()
        return b.status()

    # -- stwo -------------------------------------------------------------------------
    def stwo_batch(self, proofs: Sequence[StwoProof], mode: int = MODE_FIXTURE,
                   replicate: int = 1) -> StwoDeviceBatch:
        """A resident batch of proofs that all have the config of proofs[0] (no policy here: the
        caller has already decided that this config is the one it accepts, see verify_stwo)."""
        cfg = proofs[0].cfg
        if any(p.cfg != cfg for p in proofs):
            raise ValueError("all proofs of a batch must share one StwoConfig")
        recs = [stwo_record(p) for p in proofs]
        if replicate > 1:  # the proofs repeated `replicate` times, replicated on the device
            return StwoDeviceBatch(self, cfg, mode, recs, index=list(range(len(recs))) * replicate)
        return StwoDeviceBatch(self, cfg, mode, recs)

    def verify_stwo(self, proofs: Sequence[StwoProof], mode: int = MODE_FIXTURE, *, cfg) -> np.ndarray:
        """Status word per proof.  `cfg` is the StwoConfig the CALLER expects (or an explicit
        allow-list of them): security parameters -- queries, blow-up, PoW bits, hash family, column
        count -- are the verifier's, never the proof's.  A proof whose parsed shape or declared
        config is not in `cfg` gets STATUS_CONFIG_MISMATCH without reaching the GPU; the others are
        verified per config as their own device batch and the statuses return in input order."""
        out, groups = apply_config_policy(proofs, cfg)
        for idx in groups:
            b = self.stwo_batch([proofs[i] for i in idx], mode)
            b.run()
            out[idx] = b.status()
        return out

    def verify_stwo_records(self, cfg: StwoConfig, records: Sequence[np.ndarray],
                            mode: int = MODE_FIXTURE) -> np.ndarray:
        """Host-buffer path (ss_stwo_verify_records): raw records are uploaded in pinned
        chunks, re-tiled on the GPU and verified.  PCIe-bound; synchronous."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        want = B.lib().ss_stwo_record_words(C.byref(cs))
        if want == 0:
            raise B.SsError(B.SS_ERR_ARG, "unsupported stwo config %r" % (cfg,))
        if isinstance(records, np.ndarray) and records.ndim == 2:  # one record per row: checked once, no per-record work
            if records.dtype != np.uint32 or records.shape[1] != want or not records.flags["C_CONTIGUOUS"]:
                raise ValueError("records must be a C-contiguous uint32 array of shape (n, %d)" % want)
        else:
            for r in records:  # the library memcpy's `want` words from every pointer
                if r.dtype != np.uint32 or r.size != want or not r.flags["C_CONTIGUOUS"]:
                    raise ValueError("record must be %d contiguous uint32 words" % want)
        if len(records) == 0:
            return np.empty(0, dtype=np.uint32)
        status = np.full(len(records), 0xFFFFFFFF, dtype=np.uint32)  # unwritten = REJECT
        B.check(B.lib().ss_stwo_verify_records(self.ctx, C.byref(cs), len(records), _ptr_array(records),
                                               status.ctypes.data))
        return status

    def verify_stwo_shared_records(self, cfg: StwoConfig, shared, mode: int = MODE_FIXTURE) -> np.ndarray:
        """Host-buffer path for SHARED records (ss_stwo_verify_shared_records): 9-21 % fewer bytes on the
        link; each chunk is expanded to per-query records on the GPU (csrc/ss_shared.hip), re-tiled and
        verified behind the next upload.  A record that is no shared record of `cfg` gets STATUS_MALFORMED.
        `shared`: a list of 1-d uint32 arrays, or -- no per-record Python work -- a pair (flat, offsets): the
        records back to back in one uint32 array and the n + 1 word offsets of their starts."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        if B.lib().ss_stwo_record_words(C.byref(cs)) == 0:
            raise B.SsError(B.SS_ERR_ARG, "unsupported stwo config %r" % (cfg,))
        if isinstance(shared, tuple):
            flat, offs = shared
            offs = np.ascontiguousarray(offs, dtype=np.uint64)
            if flat.dtype != np.uint32 or flat.ndim != 1 or not flat.flags["C_CONTIGUOUS"] or offs.ndim != 1 or offs.size < 1 \
                    or (np.diff(offs.astype(np.int64)) < 0).any() or int(offs[-1]) > flat.size:
                raise ValueError("(flat, offsets): a contiguous uint32 array and ascending word offsets inside it")
            n = offs.size - 1
            ptr_vals = np.uint64(flat.ctypes.data) + offs[:-1] * np.uint64(4)
            ptrs = (C.c_void_p * n).from_buffer(ptr_vals) if n else None
            lens = np.ascontiguousarray(np.diff(offs), dtype=np.uint64)
            words = (C.c_size_t * n).from_buffer(lens) if n else None
        else:
            for r in shared:
                if r.dtype != np.uint32 or r.ndim != 1 or not r.flags["C_CONTIGUOUS"]:
                    raise ValueError("a shared record is a contiguous 1-d uint32 array")
            n = len(shared)
            ptrs = _ptr_array(shared) if n else None
            words = (C.c_size_t * n)(*[int(r.size) for r in shared]) if n else None
        if n == 0:
            return np.empty(0, dtype=np.uint32)
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)  # unwritten = REJECT
        B.check(B.lib().ss_stwo_verify_shared_records(self.ctx, C.byref(cs), n, ptrs, words, status.ctypes.data))
        return status

    def verify_stwo_minimal_records(self, cfg: StwoConfig, minimal, mode: int = MODE_FIXTURE) -> np.ndarray:
        """Host-buffer path for MINIMAL records (ss_stwo_verify_minimal_records): one sorted, deduplicated decommitment
        per tree, verified without an expansion pass (csrc/ss_minimal.hip).  A record that is no minimal record of
        `cfg` gets STATUS_MALFORMED.  `minimal`: a list of 1-d uint32 arrays, or a pair (flat, offsets) as for
        verify_stwo_shared_records."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        if B.lib().ss_stwo_record_words(C.byref(cs)) == 0:
            raise B.SsError(B.SS_ERR_ARG, "unsupported stwo config %r" % (cfg,))
        if isinstance(minimal, tuple):
            flat, offs = minimal
            offs = np.ascontiguousarray(offs, dtype=np.uint64)
            if flat.dtype != np.uint32 or flat.ndim != 1 or not flat.flags["C_CONTIGUOUS"] or offs.ndim != 1 or offs.size < 1 \
                    or (np.diff(offs.astype(np.int64)) < 0).any() or int(offs[-1]) > flat.size:
                raise ValueError("(flat, offsets): a contiguous uint32 array and ascending word offsets inside it")
            n = offs.size - 1
            ptr_vals = np.uint64(flat.ctypes.data) + offs[:-1] * np.uint64(4)
            ptrs = (C.c_void_p * n).from_buffer(ptr_vals) if n else None
            lens = np.ascontiguousarray(np.diff(offs), dtype=np.uint64)
            words = (C.c_size_t * n).from_buffer(lens) if n else None
        else:
            for r in minimal:
                if r.dtype != np.uint32 or r.ndim != 1 or not r.flags["C_CONTIGUOUS"]:
                    raise ValueError("a minimal record is a contiguous 1-d uint32 array")
            n = len(minimal)
            ptrs = _ptr_array(minimal) if n else None
            words = (C.c_size_t * n)(*[int(r.size) for r in minimal]) if n else None
        if n == 0:
            return np.empty(0, dtype=np.uint32)
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)  # unwritten = REJECT
        B.check(B.lib().ss_stwo_verify_minimal_records(self.ctx, C.byref(cs), n, ptrs, words, status.ctypes.data))
        return status

    # -- the same three paths from a caller-pinned buffer: no staging copy (csrc/ss_pinned.hip) ----------------------
    def pinned_buffer(self, words: int) -> np.ndarray:
        """A uint32 array of `words` words in page-locked host memory (torch's pinned allocator = hipHostMalloc)."""
        t = _torch().empty(max(int(words), 1), dtype=_torch().int32, pin_memory=True)
        return t.numpy().view(np.uint32)[:int(words)]  # (the array keeps the tensor, i.e. the pinned allocation, alive)

    def register_host(self, a: np.ndarray) -> None:
        """Page-lock an existing array in place (ss_host_register = hipHostRegister); undo with unregister_host."""
        B.check(B.lib().ss_host_register(self.ctx, a.ctypes.data, a.nbytes))

    def unregister_host(self, a: np.ndarray) -> None:
        B.check(B.lib().ss_host_unregister(self.ctx, a.ctypes.data))

    def verify_stwo_pinned(self, cfg: StwoConfig, flat: np.ndarray, offsets=None, kind: str = "records",
                           mode: int = MODE_FIXTURE) -> np.ndarray:
        """records / shared / minimal records lying back to back in ONE page-locked uint32 array (pinned_buffer, or any
        array passed to register_host) -> verdicts, DMA straight from that array (ss_stwo_verify_*_pinned).  `offsets`:
        n + 1 ascending word offsets (shared, minimal); per-query records are n x record_words words."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        if flat.dtype != np.uint32 or flat.ndim != 1 or not flat.flags["C_CONTIGUOUS"]:
            raise ValueError("a contiguous 1-d uint32 array expected")
        L = B.lib()
        if kind == "records":
            W = L.ss_stwo_record_words(C.byref(cs))
            if W == 0 or flat.size % W:
                raise ValueError("n x %d words expected" % W)
            n = flat.size // W
            status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
            if n:
                B.check(L.ss_stwo_verify_records_pinned(self.ctx, C.byref(cs), n, flat.ctypes.data, status.ctypes.data))
            return status
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        if offs.ndim != 1 or offs.size < 1 or (np.diff(offs.astype(np.int64)) < 0).any() or int(offs[-1]) > flat.size:
            raise ValueError("n + 1 ascending word offsets inside the array expected")
        n = offs.size - 1
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        fn = {"shared": L.ss_stwo_verify_shared_records_pinned, "minimal": L.ss_stwo_verify_minimal_records_pinned}[kind]
        if n:
            B.check(fn(self.ctx, C.byref(cs), n, flat.ctypes.data, offs.ctypes.data, status.ctypes.data))
        return status

    def pinned_text_blob(self, texts: Sequence[bytes]):
        """texts -> (blob, offsets, lengths): the texts back to back, each at a multiple of 16, in ONE page-locked uint8
        array -- what verify_stwo_texts_pinned takes (a caller that reads files would read them straight into such a buffer)."""
        offs = np.zeros(len(texts) + 1, dtype=np.uint64)
        lens = np.array([len(t) for t in texts], dtype=np.uint64)
        offs[1:] = np.cumsum((lens + np.uint64(15)) & ~np.uint64(15))
        blob = self.pinned_buffer((int(offs[-1]) + 3) // 4 + 4).view(np.uint8)
        for t, o in zip(texts, offs[:-1]):
            blob[int(o):int(o) + len(t)] = np.frombuffer(t, dtype=np.uint8)
        return blob, offs, lens

    def verify_stwo_texts_pinned(self, cfg: StwoConfig, blob: np.ndarray, offsets, lengths, mode: int = MODE_FIXTURE,
                                 fmt: int = B.TEXT_AUTO):
        """ss_stwo_verify_texts_pinned: proof.json / proof.wit texts lying in one page-locked uint8 array (pinned_text_blob, or
        any array passed to register_host) -> (status, stats); the DMA engine reads them where they are."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        n = lens.size
        if blob.dtype != np.uint8 or not blob.flags["C_CONTIGUOUS"] or offs.size != n + 1 or (n and int(offs[-1]) > blob.size):
            raise ValueError("a contiguous uint8 array, n + 1 byte offsets inside it and n lengths expected")
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n == 0:
            return status, {k: 0 for k, _ in B.IngestStats._fields_}
        clens = (C.c_size_t * n).from_buffer(lens)
        B.check(B.lib().ss_stwo_verify_texts_pinned(self.ctx, C.byref(cs), n, blob.ctypes.data, offs.ctypes.data, clens, fmt,
                                                    status.ctypes.data, C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def verify_stark101_texts_pinned(self, blob: np.ndarray, offsets, lengths, fmt: int = B.TEXT_AUTO):
        """ss_s101_verify_texts_pinned: stark101 proof.json / proof.wit texts lying in one page-locked uint8 array."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        n = lens.size
        if blob.dtype != np.uint8 or not blob.flags["C_CONTIGUOUS"] or offs.size != n + 1 or (n and int(offs[-1]) > blob.size):
            raise ValueError("a contiguous uint8 array, n + 1 byte offsets inside it and n lengths expected")
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n == 0:
            return status, {k: 0 for k, _ in B.IngestStats._fields_}
        clens = (C.c_size_t * n).from_buffer(lens)
        B.check(B.lib().ss_s101_verify_texts_pinned(self.ctx, n, blob.ctypes.data, offs.ctypes.data, clens, fmt, status.ctypes.data,
                                                    C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def verify_stwo_minimal_texts_pinned(self, cfg: StwoConfig, blob: np.ndarray, offsets, lengths, mode: int = MODE_FIXTURE):
        """ss_stwo_verify_minimal_texts_pinned: minimal proof.json texts lying in one page-locked uint8 array (pinned_text_blob)
        -> (status, stats)."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        n = lens.size
        if blob.dtype != np.uint8 or not blob.flags["C_CONTIGUOUS"] or offs.size != n + 1 or (n and int(offs[-1]) > blob.size):
            raise ValueError("a contiguous uint8 array, n + 1 byte offsets inside it and n lengths expected")
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n == 0:
            return status, {k: 0 for k, _ in B.IngestStats._fields_}
        clens = (C.c_size_t * n).from_buffer(lens)
        B.check(B.lib().ss_stwo_verify_minimal_texts_pinned(self.ctx, C.byref(cs), n, blob.ctypes.data, offs.ctypes.data, clens,
                                                            status.ctypes.data, C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def verify_stwo_minimal_texts(self, cfg: StwoConfig, texts: Sequence[bytes], mode: int = MODE_FIXTURE):
        """Minimal proof.json texts -> (status, stats) (ss_stwo_verify_minimal_texts: read into capacity-form minimal records
        by the GPU reader -- texts it does not take by the library's host readers --, then the minimal-record path)."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        n = len(texts)
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n == 0:
            return status, {k: 0 for k, _ in B.IngestStats._fields_}
        bufs = [bytes(t) for t in texts]
        arr = (C.c_char_p * n)(*bufs)
        lens = (C.c_size_t * n)(*[len(b) for b in bufs])
        B.check(B.lib().ss_stwo_verify_minimal_texts(self.ctx, C.byref(cs), n, arr, lens, status.ctypes.data, C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def verify_stwo_minimal(self, proofs, mode: int = MODE_FIXTURE, *, cfg: StwoConfig) -> np.ndarray:
        """Status word per StwoMinimalProof against the config the caller expects (others: STATUS_CONFIG_MISMATCH)."""
        out = np.full(len(proofs), 0xFFFFFFFF, dtype=np.uint32)
        idx = [i for i, p in enumerate(proofs) if p.cfg == cfg]
        for i, p in enumerate(proofs):
            if p.cfg != cfg:
                out[i] = STATUS_CONFIG_MISMATCH
        if idx:
            out[idx] = self.verify_stwo_minimal_records(cfg, [stwo_minimal_record(proofs[i]) for i in idx], mode)
        return out

    def stwo_minimal_batch(self, cfg: StwoConfig, records: Sequence[np.ndarray], mode: int = MODE_FIXTURE,
                           index: Optional[Sequence[int]] = None) -> StwoMinimalDeviceBatch:
        return StwoMinimalDeviceBatch(self, cfg, mode, records, index)

    def expand_shared_on_device(self, cfg: StwoConfig, shared: Sequence[np.ndarray], mode: int = MODE_FIXTURE):
        """The expansion kernel alone (ss_stwo_expand_shared_dev): -> (records uint32[n, W], outcome uint32[n])."""
        torch = _torch()
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        n = len(shared)
        W = B.lib().ss_stwo_record_words(C.byref(cs))
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([r.size for r in shared])
        flat = np.concatenate([np.ascontiguousarray(r, dtype=np.uint32) for r in shared] + [np.zeros(1, np.uint32)])
        sh_dev = _to_dev(flat, self.device)
        offs_dev = torch.from_numpy(offs.view(np.int64)).to(self.device)
        rec_dev = torch.empty(n * W, dtype=torch.int32, device=self.device)
        out_dev = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        B.check(B.lib().ss_stwo_expand_shared_dev(self.ctx, C.byref(cs), n, sh_dev.data_ptr(), offs_dev.data_ptr(),
                                                  rec_dev.data_ptr(), out_dev.data_ptr(),
                                                  int(torch.cuda.current_stream(self.device).cuda_stream)))
        torch.cuda.synchronize(self.device)
        return (rec_dev.cpu().numpy().view(np.uint32).reshape(n, W).copy(),
                out_dev.cpu().numpy().view(np.uint32).copy())

    # -- text in, verdicts out (native readers, csrc/ss_ingest.cpp) --------------------------
    def verify_inputs(self, form: str, source: str, inputs=None, *, cfg: Optional[StwoConfig] = None, mode: int = MODE_FIXTURE,
                      shape=None, blob=None, offsets=None, lengths=None, fmt: int = B.TEXT_AUTO):
        """ss_verify_inputs, the library's ONE host entry point (include/ss_verify.h section 4): `form` in {"records",
        "shared_records", "minimal_records", "text"} says what one input is, `source` in {"host", "pinned", "files"} where the
        inputs lie; cfg given = stwo (the config the caller expects), else stark101 (`shape` = (max_layers, max_path) for its
        records).  host: `inputs` = uint32 arrays (records) or bytes (texts); files: paths; pinned: `blob` (page-locked array:
        pinned_buffer / pinned_text_blob / register_host) with `offsets` (n + 1; none for per-query records) and, for texts,
        `lengths`.  The named methods (verify_stwo_records, verify_stwo_texts_pinned, ...) are this call with the descriptor
        their name says.  -> (status, stats)"""
        d = B.InputDesc()
        d.family = B.FAMILY_STWO if cfg is not None else B.FAMILY_STARK101
        d.form = {"records": B.FORM_RECORDS, "shared_records": B.FORM_SHARED_RECORDS, "minimal_records": B.FORM_MINIMAL_RECORDS,
                  "text": B.FORM_TEXT}[form]
        d.source = {"host": B.SRC_HOST, "pinned": B.SRC_PINNED, "files": B.SRC_FILES}[source]
        d.text_fmt = fmt
        keep = []  # ctypes objects the descriptor points to
        if cfg is not None:
            cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
            keep.append(cs)
            d.cfg = C.addressof(cs)
        if shape is not None:
            sh = B.S101Shape(*shape)
            keep.append(sh)
            d.shape = C.addressof(sh)
        if source == "pinned":
            d.blob = blob.ctypes.data
            if offsets is not None:
                offs = np.ascontiguousarray(offsets, dtype=np.uint64)
                keep.append(offs)
                d.offs = offs.ctypes.data
                n = offs.size - 1
            elif cfg is None or form != "records":
                raise ValueError("a pinned buffer needs n + 1 offsets (only stwo per-query records have a fixed stride)")
            else:
                n = blob.view(np.uint32).size // B.lib().ss_stwo_record_words(C.byref(cs))
            if lengths is not None:
                lens = (C.c_size_t * n)(*[int(x) for x in lengths])
                keep.append(lens)
                d.lens = C.addressof(lens)
        else:
            n = len(inputs)
            if source == "files":
                arr = (C.c_char_p * n)(*[os.fsencode(p) for p in inputs])
            elif form == "text":
                bufs = [bytes(t) for t in inputs]
                arr = (C.c_char_p * n)(*bufs)
                lens = (C.c_size_t * n)(*[len(b) for b in bufs])
                keep += [bufs, lens]
                d.lens = C.addressof(lens)
            else:
                recs = [np.ascontiguousarray(r, dtype=np.uint32) for r in inputs]
                arr = (C.c_void_p * n)(*[r.ctypes.data for r in recs])
                lens = (C.c_size_t * n)(*[r.size for r in recs])
                keep += [recs, lens]
                d.lens = C.addressof(lens)
            keep.append(arr)
            d.items = C.addressof(arr)
        d.n = n
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n:
            B.check(B.lib().ss_verify_inputs(self.ctx, C.byref(d), status.ctypes.data, C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def _ingest(self, fn, head_args, items, fmt):
        n = len(items)
        status = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        stats = B.IngestStats()
        if n == 0:
            return status, {k: 0 for k, _ in B.IngestStats._fields_}
        if items and isinstance(items[0], (bytes, bytearray, memoryview)):
            bufs = [bytes(t) for t in items]
            arr = (C.c_char_p * n)(*bufs)
            lens = (C.c_size_t * n)(*[len(b) for b in bufs])
            B.check(fn(self.ctx, *head_args, n, arr, lens, fmt, status.ctypes.data, C.byref(stats)))
        else:
            arr = (C.c_char_p * n)(*[os.fsencode(p) for p in items])
            B.check(fn(self.ctx, *head_args, n, arr, fmt, status.ctypes.data, C.byref(stats)))
        return status, {k: getattr(stats, k) for k, _ in B.IngestStats._fields_}

    def verify_stwo_texts(self, cfg: StwoConfig, texts: Sequence[bytes], mode: int = MODE_FIXTURE,
                          fmt: int = B.TEXT_AUTO):
        """proof.json / proof.wit texts (bytes) -> (status, stats): uploaded as text and turned into
        records by the GPU reader (canonical texts) or the library's host reader (all others, which
        alone decides what is readable), verified against `cfg` -- the config the caller expects;
        other shapes / declared parameters get STATUS_CONFIG_MISMATCH, unreadable texts
        STATUS_MALFORMED (ss_stwo_verify_texts).  stats["host_parsed"] counts the host reader's share."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        return self._ingest(B.lib().ss_stwo_verify_texts, (C.byref(cs),), list(texts), fmt)

    def read_stwo_texts(self, cfg: StwoConfig, texts: Sequence[bytes], fmt: int, mode: int = MODE_FIXTURE):
        """The GPU reader alone (ss_stwo_read_texts): -> (records uint32[n, W], outcome uint32[n]) with
        outcome 0 = canonical text, record written by the GPU; 1 = left to the host reader."""
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        n = len(texts)
        # (the minimal proof.json is read into minimal records in capacity form: stwo_minimal_from_capacity)
        W = B.lib().ss_stwo_minimal_max_words(C.byref(cs)) if fmt == B.TEXT_JSON_MINIMAL else B.lib().ss_stwo_record_words(C.byref(cs))
        recs = np.zeros((n, W), dtype=np.uint32)
        outcome = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        bufs = [bytes(t) for t in texts]
        arr = (C.c_char_p * n)(*bufs)
        lens = (C.c_size_t * n)(*[len(b) for b in bufs])
        B.check(B.lib().ss_stwo_read_texts(self.ctx, C.byref(cs), n, arr, lens, fmt, recs.ctypes.data,
                                           outcome.ctypes.data))
        return recs, outcome

    def read_stark101_texts(self, texts: Sequence[bytes], fmt: int):
        """The GPU reader alone on stark101 texts (ss_s101_read_texts): records of shape {10, 13}."""
        n = len(texts)
        W = B.lib().ss_s101_record_words(C.byref(B.S101Shape(10, 13)))
        recs = np.zeros((n, W), dtype=np.uint32)
        outcome = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        bufs = [bytes(t) for t in texts]
        arr = (C.c_char_p * n)(*bufs)
        lens = (C.c_size_t * n)(*[len(b) for b in bufs])
        B.check(B.lib().ss_s101_read_texts(self.ctx, n, arr, lens, fmt, recs.ctypes.data, outcome.ctypes.data))
        return recs, outcome

    def verify_stwo_files(self, cfg: StwoConfig, paths: Sequence[str], mode: int = MODE_FIXTURE,
                          fmt: int = B.TEXT_AUTO):
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        return self._ingest(B.lib().ss_stwo_verify_files, (C.byref(cs),), [str(p) for p in paths], fmt)

    def verify_stwo_files_pinned(self, cfg: StwoConfig, paths: Sequence[str], mode: int = MODE_FIXTURE,
                                 fmt: int = B.TEXT_AUTO, chunk_bytes: int = 1 << 30):
        """Files -> verdicts with no staging threads: the files are read (`readinto`) straight into ONE page-locked buffer,
        each at a multiple of 16, and ss_stwo_verify_texts_pinned lets the DMA engine fetch the texts from there -- `chunk_bytes`
        of files at a time (1 GiB: the buffer is allocated once and reused, so a rank never pins more than that).  What a rank
        of an 8-GPU host should call (distributed.files_verifier): the staged ss_stwo_verify_files needs about eight host
        threads to feed the link and a rank of eight has two.  A file that cannot be read gets
        SS_STATUS_MALFORMED like in ss_stwo_verify_files (simfony-cli/src/main.rs:187-190: a witness that cannot be loaded is
        exit 1, not a crash).  -> (status, stats)."""
        paths = [str(p) for p in paths]
        sizes = []
        for p in paths:
            try:
                sizes.append(os.path.getsize(p))
            except OSError:
                sizes.append(0)
        status = np.full(len(paths), 0xFFFFFFFF, dtype=np.uint32)
        total = {k: 0 for k, _ in B.IngestStats._fields_}
        padded = [(n + 15) & ~15 for n in sizes]
        cap = max([chunk_bytes] + padded)  # (a single file larger than the chunk still goes through, alone)
        blob, lo = None, 0
        while lo < len(paths):
            hi, used = lo, 0
            while hi < len(paths) and (hi == lo or used + padded[hi] <= cap):
                used += padded[hi]
                hi += 1
            if blob is None:  # sized for the largest chunk there will be: this one unless a later single file is larger
                blob = self.pinned_buffer(min(cap, max(sum(padded), 16)) // 4 + 4).view(np.uint8)
            lens = np.zeros(hi - lo, dtype=np.uint64)
            offs = np.zeros(hi - lo + 1, dtype=np.uint64)
            offs[1:] = np.cumsum(padded[lo:hi])
            for i in range(lo, hi):
                got = 0
                if sizes[i]:
                    try:
                        with open(paths[i], "rb", buffering=0) as f:
                            o = int(offs[i - lo])
                            view = memoryview(blob)[o:o + sizes[i]]
                            while got < sizes[i]:
                                k = f.readinto(view[got:])
                                if not k:
                                    break
                                got += k
                    except OSError:
                        got = 0
                lens[i - lo] = got if got == sizes[i] else 0  # an empty text is no witness: SS_STATUS_MALFORMED
            st, stats = self.verify_stwo_texts_pinned(cfg, blob, offs, lens, mode, fmt)
            status[lo:hi] = st
            for k, v in stats.items():
                total[k] = max(total[k], v) if k == "threads" else total[k] + v
            lo = hi
        return status, total

    def verify_stark101_texts(self, texts: Sequence[bytes], fmt: int = B.TEXT_AUTO):
        return self._ingest(B.lib().ss_s101_verify_texts, (), list(texts), fmt)

    def verify_stark101_files(self, paths: Sequence[str], fmt: int = B.TEXT_AUTO):
        return self._ingest(B.lib().ss_s101_verify_files, (), [str(p) for p in paths], fmt)

    def pack_stwo_on_device(self, cfg: StwoConfig, mode: int, records: Sequence[np.ndarray]):
        """Upload records as they are and re-tile them with ss_stwo_pack_dev; returns the batch
        tensor (int32 view of the u32 words)."""
        torch = _torch()
        cs = stwo_cfg_struct(cfg, mode, self.stwo_flags)
        flat = np.ascontiguousarray(np.stack(records), dtype=np.uint32)
        rec_dev = _to_dev(flat.reshape(-1), self.device)
        words = B.lib().ss_stwo_batch_words(C.byref(cs), len(records))
        out = torch.empty(words, dtype=torch.int32, device=self.device)
        B.check(B.lib().ss_stwo_pack_dev(self.ctx, C.byref(cs), len(records), rec_dev.data_ptr(),
                                         out.data_ptr(), int(torch.cuda.current_stream(self.device).cuda_stream)))
        torch.cuda.synchronize(self.device)
        return out

    # -- primitives self-test (tests only) ------------------------------------------------
    def selftest(self, op: int, inputs: np.ndarray) -> np.ndarray:
        in_w, out_w = [16, 2, 8, 1, 2, 8, 13][op], [8, 4, 8, 2, 4, 16, 1][op]
        inputs = np.ascontiguousarray(inputs, dtype=np.uint32).reshape(-1, in_w)
        out = np.empty((inputs.shape[0], out_w), dtype=np.uint32)
        B.check(B.lib().ss_selftest(self.ctx, op, inputs.shape[0], inputs.ctypes.data, out.ctypes.data))
        return out


KAT_WIDTHS = [(97, 8), (267, 9), (34, 17), (4, 10), (8, 16), (4, 4), (3, 9), (18, 16), (93, 18), (19, 19), (15, 5), (12, 10)]


def kat(ver: "Verifier", op: int, items) -> np.ndarray:
    """ss_kat: one reference function per row of `items` on the GPU (include/ss_verify.h lists the ops); rows shorter than
    the op's input width are zero padded.  Tests only."""
    in_w, out_w = KAT_WIDTHS[op]
    rows = np.zeros((len(items), in_w), dtype=np.uint32)
    for i, r in enumerate(items):
        r = [int(v) for v in r]
        if len(r) > in_w or any(not (0 <= v < 1 << 32) for v in r):
            raise ValueError("op %d takes at most %d u32 words" % (op, in_w))
        rows[i, :len(r)] = r
    out = np.zeros((len(items), out_w), dtype=np.uint32)
    B.check(B.lib().ss_kat(ver.ctx, op, len(items), rows.ctypes.data, rows.size, out.ctypes.data, out.size))
    return out


_default: Optional[Verifier] = None


def default_verifier() -> Verifier:
    global _default
    if _default is None:
        _default = Verifier(0)
    return _default


def verify_stark101(proof: Stark101Proof) -> bool:
    """Drop-in for `simfony run stark101/main.simf --witness proof.wit`: True = ACCEPT."""
    return int(default_verifier().verify_stark101([proof])[0]) == 0


def verify_stwo(proof: StwoProof, mode: int = MODE_FIXTURE, cfg=None) -> bool:
    """Drop-in for `simfony run stwo-verifier/main.simf --witness proof.wit`.  The expected config
    defaults to what the reference compiles in without -DTESTING (config.simf:34-52)."""
    from .formats import PRODUCTION_CONFIG
    return int(default_verifier().verify_stwo([proof], mode, cfg=cfg or PRODUCTION_CONFIG)[0]) == 0
