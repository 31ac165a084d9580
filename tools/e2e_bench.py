#!/usr/bin/env python3
"""Text -> verdict rate of ss_stwo_verify_texts on the metric shape (2^20-row proofs), alone.

    python tools/e2e_bench.py [--n 1536] [--reps 5] [--fmt json|wit|shared|minimal|both|all] [--workload stwo_trace20.npz]

Prints one JSON object: proofs/s and text GB/s (best and median of --reps), the host staging time,
texts that needed the host reader, and the GPU reader's kernel times per chunk (HIP events)."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import binding, records, verifier  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1536)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fmt", default="both")
    ap.add_argument("--workload", default="stwo_trace20.npz")
    ap.add_argument("--noncanonical", type=float, default=0.0, help="fraction of texts with reversed member order (host reader)")
    ap.add_argument("--python-separators", action="store_true",
                    help="proof.json as json.dumps prints it (', ' and ': ': 27 %% more bytes) instead of the external prover's compact form")
    ap.add_argument("--files", action="store_true", help="also time ss_stwo_verify_files on the same texts written to a temp directory")
    args = ap.parse_args()
    p = records.load_stwo_npz(os.path.join(ROOT, "tests", "golden", args.workload))[0]
    cfg = p.cfg
    ver = verifier.Verifier(0)
    out = {"n": args.n, "workload": args.workload}
    kinds = ["json", "wit"] if args.fmt == "both" else ["json", "wit", "shared", "minimal"] if args.fmt == "all" else [args.fmt]
    for kind in kinds:
        if kind == "json":
            text = (json.dumps(ss.stwo_to_json(p)) if args.python_separators else
                    json.dumps(ss.stwo_to_json(p), separators=(",", ":"))).encode()
            odd = json.dumps(dict(reversed(list(ss.stwo_to_json(p).items())))).encode()
            fmt = binding.TEXT_JSON
        elif kind == "shared":  # proof.json with every distinct Merkle sibling once + "queries" (read and expanded on the GPU)
            obj = ss.stwo_to_json(p, shared=True)
            text = (json.dumps(obj) if args.python_separators else json.dumps(obj, separators=(",", ":"))).encode()
            odd = json.dumps(dict(reversed(list(obj.items())))).encode()
            fmt = binding.TEXT_AUTO
        elif kind == "minimal":  # one sorted, deduplicated decommitment per tree (its list lengths are found by the GPU reader)
            from stark_symphony_amd import formats
            obj = formats.stwo_minimal_to_json(formats.stwo_minimise(p))
            text = (json.dumps(obj) if args.python_separators else json.dumps(obj, separators=(",", ":"))).encode()
            odd = json.dumps(dict(reversed(list(obj.items())))).encode()
            fmt = binding.TEXT_JSON_MINIMAL
        else:
            text = ss.stwo_to_wit(p).encode()
            odd = text.replace(b'"type": "u64"', b'"type":  "u64"')
            fmt = binding.TEXT_WIT
        k_odd = int(args.n * args.noncanonical)
        # every text its own buffer: the staging copy must read host memory, not one cache-resident string
        batch = [(odd if i < k_odd else text)[:1] + (odd if i < k_odd else text)[1:] for i in range(args.n)]
        ver.verify_stwo_texts(cfg, batch[:64], fmt=fmt)
        ver.verify_stwo_texts(cfg, batch, fmt=fmt)
        times, st = [], None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            status, st = ver.verify_stwo_texts(cfg, batch, fmt=fmt)
            times.append(time.perf_counter() - t0)
            assert (status == 0).all()
        ver.set_timing(True)
        ver.collect_timing()
        ver.verify_stwo_texts(cfg, batch, fmt=fmt)
        timing = ver.collect_timing()
        ver.set_timing(False)
        file_rate = None
        if args.files:
            import tempfile
            with tempfile.TemporaryDirectory() as d:
                paths = []
                for i in range(args.n):
                    pth = os.path.join(d, "p%05d.%s" % (i, kind))
                    with open(pth, "wb") as f:
                        f.write(batch[i])
                    paths.append(pth)
                ver.verify_stwo_files(cfg, paths, fmt=fmt)
                ft = []
                for _ in range(args.reps):
                    t0 = time.perf_counter()
                    status, _ = ver.verify_stwo_files(cfg, paths, fmt=fmt)
                    ft.append(time.perf_counter() - t0)
                    assert (status == 0).all()
                file_rate = args.n / min(ft)
        best, med = min(times), statistics.median(times)
        out[kind] = {"text_bytes": len(text), "proofs_per_s_best": args.n / best, "proofs_per_s_median": args.n / med,
                     "text_GB_per_s_best": args.n * len(text) / best / 1e9, "total_ms_best": best * 1e3,
                     "stage_ms": st["read_s"] * 1e3, "host_reader_ms": st["parse_s"] * 1e3, "host_parsed": st["host_parsed"],
                     "kernel_ms_per_call": {k: round(v[0], 3) for k, v in timing.items()},
                     "kernel_launches": {k: v[1] for k, v in timing.items()}}
        if file_rate is not None:
            out[kind]["files_proofs_per_s_best"] = file_rate
    print(json.dumps(out))


if __name__ == "__main__":
    main()
