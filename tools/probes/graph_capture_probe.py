#!/usr/bin/env python3
"""Step-by-step hipGraph capture probe for the verifier's C-ABI launches (debug aid)."""
import faulthandler
import json
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import verifier  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ver = verifier.Verifier(0)
s101 = ss.stark101_from_json(json.load(open(os.path.join(ROOT, "tests", "golden", "stark101_proof.json"))))
batch = ver.stark101_batch([s101], replicate=256)
batch.run()
torch.cuda.synchronize()
print("eager ok", batch.accepted(), flush=True)

if stage == 1:  # single stream, whole pass
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        batch.run(s)
    print("captured", flush=True)
    batch.accept_dev.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("replay ok", batch.accepted(), flush=True)
elif stage == 2:  # two streams, fork / join
    g = torch.cuda.CUDAGraph()
    s, h = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        e = torch.cuda.Event()
        e.record(s)
        h.wait_event(e)
        batch.run(h, verifier.PHASE_HEAD)
        e2 = torch.cuda.Event()
        e2.record(h)
        s.wait_event(e2)
        batch.run(s, verifier.PHASE_TAIL)
    print("captured", flush=True)
    batch.accept_dev.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("replay ok", batch.accepted(), flush=True)
else:
    nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    slots = [batch] + [batch.sibling() for _ in range(nslots - 1)]
    if rounds > 1:  # the crashing pattern: a head stream re-waits on a tail-stream event in the capture
        pipe = verifier.Pipeline(slots)
        g, cap = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.graph(g, stream=cap):
            e = torch.cuda.Event()
            e.record(cap)
            for s in pipe.head_streams + [pipe.tail_stream]:
                s.wait_event(e)
            for _ in range(rounds * nslots):
                pipe.submit()
            for s in pipe.head_streams + [pipe.tail_stream]:
                j = torch.cuda.Event()
                j.record(s)
                cap.wait_event(j)
        print("captured (rounds > 1 did not crash)", flush=True)
        sys.exit(0)
    gp = verifier.GraphedPipeline(slots)
    print("captured", flush=True)
    gp.replay()
    gp.synchronize()
    print("replay ok", [x.accepted() for x in slots], flush=True)
