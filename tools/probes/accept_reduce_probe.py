"""What the per-step accept reduce costs (bench.py's only collective) at one GPU's 8 192-proof share of the metric batch.
One rank forms a "nccl" group (RCCL) and runs the bench pipeline without the reduce and with it as bench.py submits it
(every step, a copy of the counter and an all-reduce on the pipeline's communication stream), five interleaved repetitions
each, every one with freshly made streams.  Run it under different GPU_MAX_HW_QUEUES: the streams of a process share the
runtime's hardware queues, and a head stream that shares a queue with a tail stream loses its overlap.
Caveat: torch hands out streams from a pool of 32 per device, so after a few repetitions "freshly made" streams are old
ones on whatever queue they had -- read the FIRST repetition of each variant as the clean one, the spread as what a
process that makes many streams can run into; the bench-level table of profiles/r04_accept_reduce.txt is the reference.
Run on a GPU box:  GPU_MAX_HW_QUEUES=16 python tools/probes/accept_reduce_probe.py [n] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch
import torch.distributed as dist

import bench
from stark_symphony_amd import verifier


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    _, _, proofs, _ = bench.load_workload("stwo_2p20")
    ver = verifier.Verifier(0)
    batch = verifier.StwoDeviceBatch(ver, proofs[0].cfg, verifier.MODE_FIXTURE, [verifier.stwo_record(p) for p in proofs],
                                     index=[i % len(proofs) for i in range(n)])
    slots = [batch] + [batch.sibling() for _ in range(2)]
    accs = [torch.zeros(1, dtype=torch.int32, device=ver.device) for _ in slots]

    def measure(make_cb, tail_streams, on_tail):
        pipe = verifier.Pipeline(slots, tail_streams=tail_streams)
        cb = make_cb(pipe)
        kw = {}
        for i in range(12):
            pipe.submit(cb, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(cb, **kw)
        host = time.perf_counter() - t0
        pipe.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3

    def none(pipe):
        return None

    def every_step(pipe):   # bench.py until round 4: a copy of the counter and a blocking all-reduce on a communication stream
        def cb(k):
            accs[k].copy_(slots[k].accept_dev)
            dist.all_reduce(accs[k], op=dist.ReduceOp.SUM)
        return cb

    def nothing(pipe):      # the communication stream's two event waits only
        return lambda k: None

    def copy_only(pipe):
        def cb(k):
            accs[k].copy_(slots[k].accept_dev)
        return cb

    def reduce_in_place(pipe):
        def cb(k):
            dist.all_reduce(slots[k].accept_dev, op=dist.ReduceOp.SUM)
        return cb

    variants = [("no reduce", none, False), ("communication stream, no work on it", nothing, False),
                ("copy of the counter", copy_only, False), ("all-reduce of the counter in place", reduce_in_place, False),
                ("copy + all-reduce on a communication stream", every_step, False)]
    print("GPU_MAX_HW_QUEUES=%s, %d proofs per step, %d steps, five interleaved repetitions" % (
        os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"), n, steps))
    for ts in ((2,) if n < 65536 else (1,)):
        rows = {v[0]: [] for v in variants}
        for rep in range(5):
            for label, make, on_tail in variants:
                rows[label].append(measure(make, ts, on_tail))
        for label, r in rows.items():
            ms = sorted(x[0] for x in r)
            print("  tail streams %d  %-52s best %.4f  median %.4f ms/step  (%.2f M proofs/s best; submit loop %.3f ms/step)  all: %s" % (
                ts, label, ms[0], ms[2], n / ms[0] / 1e3, min(x[1] for x in r), " ".join("%.3f" % x[0] for x in r)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
