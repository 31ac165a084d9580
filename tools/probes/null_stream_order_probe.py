"""Why GPUTEST_r05 went red: a write on the caller's CURRENT (null) stream against passes on non-blocking side streams.

The pattern of round 5's test (tests/test_gpu_parity.py::test_independent_streams_match_single_stream as it was):
    for s in slots: s.status_dev.fill_(0x55)        # torch's current stream = the legacy null stream
    ind.submit() x 23                               # 5 side streams (torch.cuda.Stream: hipStreamNonBlocking)
    ind.synchronize()                               # waits for the side streams only
torch's side streams are created non-blocking, so NOTHING orders the fills before the passes.  This probe replays exactly
that with raw streams (no helper of verifier.py joins anything here) and, per repetition, records an event after the last
fill and one after each slot's first and last pass, then reports where the fill landed:
    before the slot's first pass / between first and last / AFTER the last pass (the red case: status stays 0x55).
Run it under different queue counts (the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues):
    GPU_MAX_HW_QUEUES=4 python tools/probes/null_stream_order_probe.py 50
    GPU_MAX_HW_QUEUES=24 python tools/probes/null_stream_order_probe.py 50
and with `join` as second argument to see the fix (every side stream waits for the current stream once: Pipeline /
IndependentStreams.wait_current) under the same conditions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import verifier  # noqa: E402


def measure(ver, reps: int, join: bool, busy: int = 0, n_slots: int = 5, passes: int = 23):
    """-> (where the fill event fell per slot, slots whose status ended 0x55)"""
    s101 = ss.stark101_from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json"))))
    batch = ver.stark101_batch([s101], replicate=300)
    slots = [batch.sibling() for _ in range(n_slots)]
    streams = [torch.cuda.Stream() for _ in slots]
    extra = [torch.cuda.Stream() for _ in range(busy)]  # extra streams kept alive and busy (a process with many queues in use)
    junk = torch.zeros(1 << 22, device="cuda")
    torch.cuda.synchronize()
    where = {"before_first": 0, "between": 0, "after_last": 0}
    stale = 0
    late_ms = []
    for r in range(reps):
        for e in extra:
            with torch.cuda.stream(e):
                junk.add_(1.0)
        for s in slots:
            s.status_dev.fill_(0x55)
        filled = torch.cuda.Event(enable_timing=True)
        filled.record()  # null stream, behind the fills
        if join:
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
        first = [None] * n_slots
        last = [None] * n_slots
        for i in range(passes):
            k = i % n_slots
            slots[k].run(streams[k])
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(streams[k])
            if first[k] is None:
                first[k] = ev
            last[k] = ev
        torch.cuda.synchronize()
        for k in range(n_slots):
            if first[k].elapsed_time(filled) <= 0:      # filled is not later than first
                where["before_first"] += 1
            elif last[k].elapsed_time(filled) <= 0:
                where["between"] += 1
            else:
                where["after_last"] += 1
                late_ms.append(last[k].elapsed_time(filled))
            stale += int((slots[k].status_dev == 0x55).any().item())
    return where, stale, late_ms


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    join = len(sys.argv) > 2 and sys.argv[2] == "join"
    busy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    where, stale, late = measure(verifier.Verifier(0), reps, join, busy)
    print("GPU_MAX_HW_QUEUES=%s join=%s busy_streams=%d reps=%d slots=5: fill event %s; slots whose status words ended 0x55: %d of %d" % (
        os.environ.get("GPU_MAX_HW_QUEUES", "unset"), join, busy, reps, where, stale, 5 * reps), flush=True)
