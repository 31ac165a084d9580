"""bench.py end to end on the GPU box: the JSON contract at N=1 and the multi-rank control flow
(two ranks sharing the one GPU over gloo -- RCCL refuses two ranks on one device, so the nccl backend
itself is the only part of `--gpus N` this cannot cover)."""
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
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["proofs_per_gpu"] == 512 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.parametrize("n", [2, 8])
def test_bench_spawns_its_own_ranks(n):
    """`python bench.py --gpus N` exactly as the scaling driver may type it (no torchrun, no WORLD_SIZE):
    the process turns into a launcher before touching torch / HIP, starts N ranks, relays rank 0's
    single JSON line and returns non-zero if any rank does.  Here all ranks share GPU 0 over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload",
                        "stwo_fixture", "--proofs-per-gpu", "256", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == n and d["steps"] == 3 and d["config"]["proofs_per_gpu"] == 256
    assert abs(d["value"] - n * 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SS_BENCH_SHARE_GPU="1", SS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "no_such"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "stwo_fixture"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "disagrees" in r.stderr


@pytest.mark.parametrize("how,label", [("auto", "independent streams"), ("on", "hipGraph")])
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
