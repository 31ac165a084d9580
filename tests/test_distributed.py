"""World-size-2 (and 3) `gloo` test of the multi-GPU driver on CPU.

The sharding + accept-reduce logic is exercised with the oracle standing in for the
per-rank GPU verifier (tests may use the oracle as a checker; the product default is the HIP
path, covered by test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    import stark_symphony_amd as ss
    from stark_symphony_amd import distributed, formats
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = ss.stark101_from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json"))))
    rng = np.random.default_rng(1234)
    proofs = [base if i % 3 else formats.stark101_corrupt(base, rng)[0] for i in range(n)]
    local, accepted, total, allst = distributed.verify_sharded(
        proofs, lambda sl: O.s101_verify_batch(list(sl), 1), gather_status=True)
    lo, hi = distributed.shard_range(n, rank, world)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), local=local, accepted=accepted, total=total,
             allst=allst, lo=lo, hi=hi)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 11), (3, 10), (2, 1)])
def test_sharded_verify_gloo(tmp_path, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, "r%d.npz" % r)) for r in range(world)]
    want_accept = sum(1 for i in range(n) if i % 3)
    covered = []
    for r in res:
        assert int(r["total"]) == n and int(r["accepted"]) == want_accept
        assert len(r["local"]) == int(r["hi"]) - int(r["lo"])
        covered += list(range(int(r["lo"]), int(r["hi"])))
        assert np.array_equal(r["allst"], res[0]["allst"])
    assert covered == list(range(n))
    allst = res[0]["allst"]
    assert [(s == 0) for s in allst] == [bool(i % 3) for i in range(n)]


def _index_worker(rank: int, world: int, port: int, n: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from stark_symphony_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def verdicts(sl):  # a verdict that names its proof: status = index + 1 for every seventh, else ACCEPT
        seen.append((sl[0], sl[-1], len(sl)))
        idx = np.asarray(sl, dtype=np.int64)
        return np.where(idx % 7 == 0, idx + 1, 0).astype(np.uint32)
    local, accepted, total, allst = distributed.verify_sharded(range(n), verdicts, gather_status=True)
    np.savez(os.path.join(out_dir, "i%d.npz" % rank), local=local, accepted=accepted, total=total, allst=allst,
             seen=np.array(seen, dtype=np.int64))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [65536, 65539, 5])
def test_sharded_verify_at_the_real_world_size(tmp_path, n):
    """BASELINE configs[3] as it is sharded on an 8-GPU node: world size 8, 65 536 proof indices (8 192 per rank), plus a
    batch that does not divide (ranks differ by one proof: the gather pads) and one with fewer proofs than ranks (three
    ranks own nothing and still take part in both collectives)."""
    from stark_symphony_amd.distributed import shard_range
    world, port = 8, _free_port()
    mp.spawn(_index_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, "i%d.npz" % r)) for r in range(world)]
    want = np.where(np.arange(n) % 7 == 0, np.arange(n) + 1, 0).astype(np.uint32)
    for r, d in enumerate(res):
        lo, hi = shard_range(n, r, world)
        assert np.array_equal(d["local"], want[lo:hi])
        assert d["seen"].tolist() == ([[lo, hi - 1, hi - lo]] if hi > lo else [])  # its slice, once, nothing else
        assert int(d["total"]) == n and int(d["accepted"]) == int((want == 0).sum())
        assert np.array_equal(d["allst"], want)
    if n == 65536:
        assert all(len(d["local"]) == 8192 for d in res)


def _files_worker(rank: int, world: int, port: int, paths, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    import builtins
    import json
    import torch.distributed as dist
    import stark_symphony_amd as ss
    from stark_symphony_amd import distributed
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opened = []

    def verify_files(ps):  # stands in for Verifier.verify_stwo_files: reads each file it is given, once
        out = []
        for p in ps:
            opened.append(p)
            try:
                proof = ss.stark101_from_json(json.load(builtins.open(p)))
                out.append(int(O.s101_verify_batch([proof], 1)[0]))
            except (OSError, ValueError, ss.MalformedProof):
                out.append(2)
        return np.array(out, dtype=np.uint32)
    local, accepted, total, allst = distributed.verify_files_sharded(paths, verify_files, gather_status=True)
    np.savez(os.path.join(out_dir, "f%d.npz" % rank), local=local, accepted=accepted, total=total, allst=allst,
             opened=np.array(opened, dtype=object))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_files_each_rank_opens_only_its_slice(tmp_path, world):
    """verify_files_sharded: the names are global, the reads are rank-local (stwo-verifier/Makefile:17-18
    hands one file per process).  Files of OTHER ranks' slices do not even have to exist on this rank's box: here each
    rank's foreign files are checked never to have been opened."""
    import json
    import stark_symphony_amd as ss
    from stark_symphony_amd import formats
    from stark_symphony_amd.distributed import shard_range
    base = ss.stark101_from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json"))))
    rng = np.random.default_rng(77)
    n, paths, want = 11, [], []
    for i in range(n):
        p = tmp_path / ("p%02d.json" % i)
        if i == 4:
            want.append(2)                                  # absent file: stage-0 verdict, no exception
        else:
            proof = base if i % 3 else formats.stark101_corrupt(base, rng)[0]
            p.write_text(json.dumps(ss.stark101_to_json(proof)))
            want.append(None)
        paths.append(str(p))
    port = _free_port()
    mp.spawn(_files_worker, args=(world, port, paths, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, "f%d.npz" % r), allow_pickle=True) for r in range(world)]
    for r, d in enumerate(res):
        lo, hi = shard_range(n, r, world)
        assert list(d["opened"]) == paths[lo:hi]
        assert int(d["total"]) == n and np.array_equal(d["allst"], res[0]["allst"])
    allst = res[0]["allst"]
    assert allst[4] == 2 and all((allst[i] == 0) == bool(i % 3) for i in range(n) if i != 4)
    assert int(res[0]["accepted"]) == int((allst == 0).sum())


def test_shard_range_is_a_partition():
    from stark_symphony_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_ranks_that_share_a_host_are_told_to_use_pinned_inputs(monkeypatch):
    """distributed.use_pinned_inputs / files_verifier: a rank whose share of the host's cores is below what the staged host
    paths need (8 threads for a 55 GB/s link, profiles/r05_host_path_threads.txt) gets the caller-pinned entry points --
    every rank of an 8-GPU node on a 16-core grant; a process that owns 16 cores keeps the staged ones."""
    from stark_symphony_amd import distributed

    class Ver:
        def verify_stwo_files(self, cfg, ps, mode):
            return ("staged", list(ps)), {}

        def verify_stwo_files_pinned(self, cfg, ps, mode):
            return ("pinned", list(ps)), {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda _: set(range(16)))  # (the cgroup quota may still cap it below)
    assert distributed.host_threads_per_rank(1) <= 16 and distributed.host_threads_per_rank(8) <= 2
    assert distributed.use_pinned_inputs(8) and distributed.use_pinned_inputs(4)
    if distributed.host_threads_per_rank(1) >= distributed.STAGED_PATH_THREADS:
        assert not distributed.use_pinned_inputs(1)
        assert distributed.files_verifier(Ver(), None, world=1)(["a"])[0] == "staged"
    assert distributed.files_verifier(Ver(), None, world=8)(["a", "b"]) == ("pinned", ["a", "b"])
