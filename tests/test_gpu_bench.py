"""bench.py end to end on the GPU box: the JSON contract at N=1 and the multi-rank control flow
(two ranks sharing the one GPU over gloo -- RCCL refuses two ranks on one device; the nccl backend
itself is covered by test_rccl_two_ranks_*, which enables itself on any box with two GPUs)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stwo_fixture",
                        "--proofs-per-gpu", "2048", "--steps", "5", "--warmup", "1", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1
    assert d["value"] > 0 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["kernel_launches"] == 5
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_bench_two_ranks_share_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "stwo_fixture", "--proofs-per-gpu", "512", "--steps", "4",
                        "--warmup", "1", "--e2e", "301"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["proofs_per_gpu"] == 512 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # rank-local ingest under world > 1: every rank its share of the 301 inputs at the same time, the
    # host's cores divided between the ranks, what each rank moved over its link reported
    e = d["e2e"]
    assert e["ranks"] == 2 and e["proofs"] == 301 and e["host_threads_per_rank"] >= 2
    for kind in ("json", "wit", "json_shared", "records", "shared_records", "minimal_records", "records_pinned",
                 "shared_records_pinned", "minimal_records_pinned"):
        assert e[kind]["proofs_per_s"] > 0 and len(e[kind]["per_rank_link_GB_s"]) == 2, kind
    assert e["json"]["host_threads"] == e["host_threads_per_rank"] and e["json_shared"]["host_parsed_texts"] == 0
    assert e["shared_records"]["bytes_per_proof"] < e["records"]["bytes_per_proof"]


@pytest.mark.parametrize("n", [2, pytest.param(4, marks=pytest.mark.launcher_extra)])  # the pytest process + 4 ranks: the GPU box allows 6 processes on its card
def test_bench_spawns_its_own_ranks(n):
    """`python bench.py --gpus N` exactly as the scaling driver may type it (no torchrun, no WORLD_SIZE):
    the process turns into a launcher before touching torch / HIP, starts N ranks, relays rank 0's
    single JSON line and returns non-zero if any rank does.  Here all ranks share GPU 0 over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload",
                        "stwo_fixture", "--proofs-per-gpu", "256", "--steps", "3", "--warmup", "1", "--e2e", "64"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == n and d["steps"] == 3 and d["config"]["proofs_per_gpu"] == 256
    assert abs(d["value"] - n * 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.parametrize("n,batch", [(2, 1024), pytest.param(3, 1000, marks=pytest.mark.launcher_extra)])
def test_bench_strong_scaling_splits_one_batch(n, batch):
    """--scaling strong (the default): ONE batch of --batch proofs split shard_range-wise over the ranks
    (BASELINE.json configs[3]: 65 536 proofs over 8 GPUs); value counts the batch once per step.  An
    uneven split (1000 over 3) leaves ranks with 334 / 333 / 333 proofs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload",
                        "stwo_fixture", "--batch", str(batch), "--steps", "3", "--warmup", "1", "--e2e", "0"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["proofs_per_step"] == batch
    assert d["config"]["proofs_per_gpu"] == (batch + n - 1) // n
    assert abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "no_such"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "end of its stderr" in r.stderr and "unknown workload" in r.stderr  # the failed rank's own words


def _gpu_count() -> int:
    import torch
    return torch.cuda.device_count()  # does not initialise the GPU on this image


_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
import stark_symphony_amd as ss
from stark_symphony_amd import distributed, formats, records, verifier
from oracle import oracle as O
import json
g = os.path.join({root!r}, "tests", "golden")
base = ss.stwo_from_json(json.load(open(os.path.join(g, "stwo_proof.json"))))
rng = np.random.default_rng(0x5EED2025 + 77)
proofs = [base if i % 3 else formats.stwo_corrupt(base, rng)[0] for i in range(37)]   # uneven split
ver = verifier.Verifier(local)
local_st, acc, tot, all_st = distributed.verify_sharded(
    proofs, lambda sl: ver.verify_stwo(sl, cfg=base.cfg), gather_status=True)
want = O.stwo_verify_batch(proofs)
assert tot == 37 and acc == int((want == 0).sum()), (acc, tot)
assert all_st.tolist() == want.tolist()
lo, hi = distributed.shard_range(37, rank, dist.get_world_size())
assert local_st.tolist() == want[lo:hi].tolist()
dist.barrier()
backend_, world_ = dist.get_backend(), dist.get_world_size()
dist.destroy_process_group()
print("rank %d ok: %d of %d accepted, backend %s, world %d" % (rank, acc, tot, backend_, world_))
"""


def _torchrun(worker, nproc, env, timeout=900):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(worker)],
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_rccl_one_rank_runs_the_exchange(tmp_path):
    """The nccl backend (RCCL) with ONE rank on the one GPU of the test box: distributed.verify_sharded sends its
    accept-count all-reduce and its status all-gather through RCCL (a process group that exists is used whatever its
    size) on a mixed batch, against the oracle.  Not a scaling result -- it shows that RCCL initialises on this image
    and accepts the calls, devices and dtypes the N > 1 path makes; two ranks need two devices (next test)."""
    worker = tmp_path / "rccl_worker.py"
    worker.write_text(_RCCL_WORKER.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["NCCL_DEBUG"] = "VERSION"
    r = _torchrun(worker, 1, env)
    assert r.returncode == 0 and r.stdout.count("ok:") == 1, (r.stdout[-2000:], r.stderr[-3000:])
    assert "backend nccl" in r.stdout
    version = [l for l in (r.stdout + r.stderr).splitlines() if "version" in l.lower() and "ccl" in l.lower()]
    assert version, (r.stdout[-2000:], r.stderr[-2000:])  # NCCL_DEBUG=VERSION: the library that ran says which it is
    print("\n".join(version[:3]), r.stdout.strip().splitlines()[-1])


def test_bench_collectives_through_rccl_with_one_rank():
    """bench.py with its process group formed by ONE rank over the nccl backend (SS_BENCH_GROUP_OF_ONE): the accept
    reduce submitted on the pipeline's stream every step, the barriers around the timed region, the max-over-ranks and
    the e2e gathers all go through RCCL on the one GPU of the test box -- the call sequence of the driver's N > 1 runs,
    which no two-device box has executed yet.  Not a scaling result."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    env.update(SS_BENCH_GROUP_OF_ONE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in (["--batch", "4096"], ["--proofs-per-gpu", "2048", "--scaling", "weak"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stwo_fixture", "--steps", "8",
                            "--warmup", "2", "--no-cpu-baseline", "--e2e", "256"] + extra,
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        d = _last_json(r.stdout)
        assert d["n_gpus"] == 1 and d["steps"] == 8
        assert abs(d["value"] - d["config"]["proofs_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        assert "errors" not in d["e2e"] and d["e2e"]["records"]["proofs_per_s"] > 0
        assert d["config"]["accept_reduce"].startswith("all-reduce(SUM) of every step's accept count over nccl")


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the first RCCL evidence comes from a multi-GPU box")
def test_rccl_two_ranks_verify_sharded_and_bench(tmp_path):
    """Two ranks on two devices over the nccl backend (RCCL over xGMI): distributed.verify_sharded with
    the status all-gather on a mixed batch against the oracle, then bench.py --gpus 2 in both scaling
    modes.  Skipped on the single-GPU test box."""
    worker = tmp_path / "rccl_worker.py"
    worker.write_text(_RCCL_WORKER.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = _torchrun(worker, 2, env)
    assert r.returncode == 0 and r.stdout.count("ok:") == 2, (r.stdout[-2000:], r.stderr[-3000:])
    for extra in (["--proofs-per-gpu", "2048"], ["--batch", "4096"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stwo_fixture",
                            "--steps", "5", "--warmup", "2"] + extra, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        d = _last_json(r.stdout)
        assert d["n_gpus"] == 2 and d["config"]["proofs_per_gpu"] == 2048 and d["config"]["proofs_per_step"] == 4096
        assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        _check_multi_rank_line(d, 2, "nccl")


def _check_multi_rank_line(d: dict, n: int, backend: str) -> None:
    """What the driver's first real N > 1 run must carry (SCALE_rNN.json is built from these lines): the roofline of the
    dominant kernel from the timed steps, how the accept count is reduced and over which backend, and what every rank
    moved over its own host link in the e2e section, for the staged and the caller-pinned entry points."""
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_launches"] >= d["steps"]
    assert d["config"]["accept_reduce"].startswith("all-reduce(SUM) of every step's accept count over " + backend)
    assert d["scaling"] in ("strong", "weak") and "cpu_baseline" not in d
    e = d["e2e"]
    assert e["ranks"] == n and "errors" not in e
    for kind in ("json", "wit", "json_shared", "json_pinned", "wit_pinned", "json_shared_pinned", "records", "shared_records",
                 "minimal_records", "records_pinned", "shared_records_pinned", "minimal_records_pinned"):
        assert e[kind]["proofs_per_s"] > 0 and len(e[kind]["per_rank_link_GB_s"]) == n, kind
    assert e["minimal_records"]["bytes_per_proof"] < e["shared_records"]["bytes_per_proof"] < e["records"]["bytes_per_proof"]


@pytest.mark.launcher_extra
def test_two_ranks_on_one_gpu_over_nccl_or_the_documented_refusal():
    """`bench.py --gpus 2 --scaling strong` with BOTH ranks on the one GPU of the test box over the nccl backend.  RCCL
    either accepts two communicator ranks on one device -- then the line is checked like the first real multi-GPU run
    will be -- or refuses them ("Duplicate GPU detected", ncclInvalidUsage: what DESIGN.md section 6 documents), in which
    case the launcher must report that rank's own words and the same command over gloo must work."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--workload", "stwo_fixture",
           "--batch", "2048", "--steps", "4", "--warmup", "1", "--e2e", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    if r.returncode == 0:
        d = _last_json(r.stdout)
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["proofs_per_step"] == 2048
        _check_multi_rank_line(d, 2, "nccl")
        return
    text = r.stderr.lower()
    assert "end of its stderr" in text and ("duplicate gpu" in text or "invalid usage" in text or "ncclinvalidusage" in text), r.stderr[-3000:]
    env["SS_BENCH_BACKEND"] = "gloo"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_multi_rank_line(_last_json(r.stdout), 2, "gloo")


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "stwo_fixture"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "disagrees" in r.stderr


@pytest.mark.parametrize("how,label", [("auto", "independent streams"), pytest.param("on", "hipGraph", marks=pytest.mark.launcher_extra)])
def test_bench_stark101_small_batch_submission(how, label):
    """BASELINE.json configs[1] (stark101 x 4096) cannot fill the chip with one pass: bench.py overlaps
    whole passes (16 independent streams by default, hipGraph replay with --graph on) for the timed
    steps and takes the kernel durations from a pipelined pass."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stark101", "--steps", "24",
                        "--warmup", "2", "--no-cpu-baseline", "--graph", how], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["steps"] == 24 and d["config"]["proofs_per_gpu"] == 4096 and label in d["config"]["submission"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "s101_merkle" and d["roofline"]["kernel_launches"] == 20
