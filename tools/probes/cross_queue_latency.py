"""Does a queue count oversubscribe the hardware?  A one-wave kernel on the caller's null stream (fill of one word + a host
wait) while 16 side streams are kept busy with whole stark101 x 4096 passes (verifier.IndependentStreams, what bench.py runs
for BASELINE configs[1]): the latency a foreign stream of the same process sees, idle and under load.
    GPU_MAX_HW_QUEUES=24 python tools/probes/cross_queue_latency.py
If the streams outnumber the queues the hardware can keep resident, the null stream's packet waits for a time slice instead
of a free CU ("whether 24 oversubscribes the hardware queue slots")."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import verifier  # noqa: E402

ver = verifier.Verifier(0)
s101 = ss.stark101_from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json"))))
batch = ver.stark101_batch([s101], replicate=4096)
slots = [batch.sibling() for _ in range(16)]
ind = verifier.IndependentStreams(slots)
word = torch.zeros(1, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def probe(n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        word.fill_(1)
        torch.cuda.current_stream().synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts


idle = probe(300)
# keep ~40 passes queued ahead on the side streams while probing
for _ in range(64):
    ind.submit()
busy = []
t_end = time.perf_counter() + 1.5
passes = 64
while time.perf_counter() < t_end:
    for _ in range(8):
        ind.submit()
    passes += 8
    busy += probe(1)
ind.synchronize()
busy.sort()
q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
print("GPU_MAX_HW_QUEUES=%s  null-stream fill + wait, us: idle median %.1f p99 %.1f | 16 streams busy (%d passes): median %.1f p90 %.1f p99 %.1f max %.1f (n=%d)" % (
    os.environ.get("GPU_MAX_HW_QUEUES", "unset"), q(idle, .5), q(idle, .99), passes, q(busy, .5), q(busy, .9), q(busy, .99), busy[-1], len(busy)), flush=True)
