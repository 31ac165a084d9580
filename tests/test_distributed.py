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


def test_shard_range_is_a_partition():
    from stark_symphony_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
