"""Multi-GPU driver: proofs are independent, so a batch shards by proof index.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  Rank r verifies the contiguous slice [lo, hi) of the batch on its own
GPU; the only exchange of the path is the final accept-reduce: one
all-reduce(SUM) of {accepted, total} -- 8 bytes, latency bound -- and, when the caller wants
every verdict everywhere, one all-gather of the status words.  No data-path collective.

The reference has nothing to compare with here: it verifies one proof per process
(simfony-cli/src/main.rs:163-209).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of n items: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group=None) -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


# Host threads the staged host paths (ss_stwo_verify_files / _texts / _records: stager threads copy the caller's bytes into
# the library's pinned chunks) need to keep a 55 GB/s host link busy; measured 2 threads: 37-47 GB/s, 1 thread: 30-34 GB/s,
# 8: 52-54 GB/s (profiles/r05_host_path_threads.txt).  Below that the caller-pinned entry points are faster: the DMA engine
# reads the caller's page-locked buffer and no host thread touches a byte.
STAGED_PATH_THREADS = 8


def host_threads_per_rank(world: int) -> int:
    """Cores one rank of `world` may use on this host: scheduler affinity capped by the cgroup quota, shared evenly (what
    bench.py exports as SS_HOST_THREADS)."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n // max(1, world))


def use_pinned_inputs(world: Optional[int] = None) -> bool:
    """True where a rank should hand the library caller-pinned inputs (`ss_stwo_verify_*_pinned`) instead of the staged
    ones: whenever its share of the host's cores cannot feed the staging copy -- every rank of an 8-GPU node on a 16-core
    grant (2 threads each), not a single process that owns the host."""
    if world is None:
        world = _world()[1]
    return host_threads_per_rank(world) < STAGED_PATH_THREADS


def files_verifier(ver, cfg, mode: Optional[int] = None, world: Optional[int] = None) -> Callable[[Sequence[str]], np.ndarray]:
    """The `verify_files_local` to hand to verify_files_sharded on a GPU box: paths -> status words through THIS rank's
    Verifier, reading the files into one page-locked buffer (Verifier.verify_stwo_files_pinned) when the rank's host-thread
    budget is below what the staged reader needs, else through ss_stwo_verify_files."""
    from . import verifier as V
    m = V.MODE_FIXTURE if mode is None else mode
    if use_pinned_inputs(world):
        return lambda ps: ver.verify_stwo_files_pinned(cfg, ps, m)[0]
    return lambda ps: ver.verify_stwo_files(cfg, ps, m)[0]


def verify_files_sharded(paths: Sequence[str], verify_files_local: Callable[[Sequence[str]], np.ndarray],
                         group=None, gather_status: bool = False, device=None):
    """Rank-local ingest ("each rank reads only its slice"): every rank is handed the same list of file
    NAMES and opens only the files of its own contiguous slice -- the reference's caller hands one file per process
    (stwo-verifier/Makefile:17-18: `simfony run main.simf --witness proof.wit`).  verify_files_local(paths_slice) ->
    uint32 status per file; on the GPU box `files_verifier(Verifier(local_rank), cfg)`: the caller-pinned entry point when
    the rank shares the host's cores with seven others, ss_stwo_verify_files (the library reads, stages and uploads the text
    itself) when it owns them.  The only exchange is the final
    accept-reduce, as in verify_sharded.  Returns (local_status, accepted_total, n_total[, all_status])."""
    rank, world = _world(group)
    lo, hi = shard_range(len(paths), rank, world)
    mine = [str(p) for p in paths[lo:hi]]  # the only names this rank ever passes to open(2)
    return verify_sharded(list(paths), lambda _slice: verify_files_local(mine), group=group, gather_status=gather_status,
                          device=device)


def verify_sharded(proofs: Sequence, verify_local: Callable[[Sequence], np.ndarray],
                   group=None, gather_status: bool = False, device=None):
    """Verify `proofs` (the same list on every rank) with each rank doing its own slice.

    verify_local(slice) -> uint32 status per proof (0 = ACCEPT); on the GPU box this is
    `Verifier(local_rank).verify_stwo` / `.verify_stark101`.
    Returns (local_status, accepted_total, n_total[, all_status])."""
    import torch
    import torch.distributed as dist
    rank, world = _world(group)
    # a process group that exists is used whatever its size: a one-rank "nccl" group sends the same two calls through
    # RCCL as eight ranks do (the single-GPU test box's only way to run them)
    grouped = dist.is_available() and dist.is_initialized()
    lo, hi = shard_range(len(proofs), rank, world)
    local = np.asarray(verify_local(proofs[lo:hi]) if hi > lo else np.zeros(0, np.uint32),
                       dtype=np.uint32)
    if device is None:
        backend = dist.get_backend(group) if grouped else "gloo"
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    counts = torch.tensor([int((local == 0).sum()), int(local.size)], dtype=torch.int64, device=device)
    if grouped:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    accepted, total = int(counts[0].item()), int(counts[1].item())
    if not gather_status:
        return local, accepted, total
    if not grouped:
        return local, accepted, total, local.copy()
    # ranks may own slices that differ by one proof: pad to the longest
    longest = (len(proofs) + world - 1) // world
    mine = torch.full((longest,), 0xFFFFFFFF, dtype=torch.int64, device=device)
    mine[:local.size] = torch.from_numpy(local.astype(np.int64)).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out: List[np.ndarray] = []
    for r, t in enumerate(parts):
        a, b = shard_range(len(proofs), r, world)
        out.append(t[:b - a].cpu().numpy().astype(np.uint32))
    return local, accepted, total, np.concatenate(out)
