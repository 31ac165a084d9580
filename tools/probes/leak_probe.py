"""Do repeated calls of the host-buffer entry points leak?  Device memory (hipMemGetInfo through torch) and the
process's resident set after every 50 calls of each of: ss_stwo_verify_texts (json, wit, shared json, mixed with
non-canonical texts; since round 5 the minimal proof.json too), ss_stwo_verify_records, ss_stwo_verify_shared_records,
ss_stwo_verify_minimal_records, across three configs (template cache).
    python tools/probes/leak_probe.py [rounds]     (run on a GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch

import stark_symphony_amd as ss
from stark_symphony_amd import binding, records, verifier


def rss_mb() -> float:
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1]) / 1024.0
    return 0.0


def main() -> None:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    g = os.path.join(ROOT, "tests", "golden")
    ver = verifier.Verifier(0)
    sets = []
    for fn in ("stwo_trace16.npz", "stwo_wide256.npz", "stwo_trace20.npz"):
        proofs = records.load_stwo_npz(os.path.join(g, fn))[:4]
        cfg = proofs[0].cfg
        js = [json.dumps(ss.stwo_to_json(p), separators=(",", ":")).encode() for p in proofs]
        nc = [json.dumps(dict(reversed(list(ss.stwo_to_json(p).items())))).encode() for p in proofs[:1]]  # other member order: host reader
        from stark_symphony_amd import formats
        mins = [verifier.stwo_minimise_record(cfg, verifier.stwo_record(p), formats.stwo_queries(p)) for p in proofs]
        mtexts = [verifier.write_stwo_minimal_text(cfg, m, python_separators=False) for m in mins]
        mobj = formats.stwo_minimal_to_json(formats.stwo_minimise(proofs[0]))
        mtexts = mtexts * 8 + [json.dumps(dict(reversed(list(mobj.items())))).encode()]  # (one for the host readers)
        sets.append((cfg, {"json": js * 8 + nc, "wit": [ss.stwo_to_wit(p).encode() for p in proofs] * 8,
                           "shared": [json.dumps(ss.stwo_to_json(p, shared=True), separators=(",", ":")).encode() for p in proofs] * 8,
                           "minimal": mtexts},
                     np.stack([verifier.stwo_record(p) for p in proofs] * 16), [verifier.stwo_shared_record(p) for p in proofs] * 16,
                     mins * 16))
    base = None
    for r in range(rounds):
        for cfg, texts, recs, shared, mins in sets:
            for _ in range(50 // len(sets) + 1):
                for kind, fmt in (("json", binding.TEXT_AUTO), ("wit", binding.TEXT_WIT), ("shared", binding.TEXT_AUTO),
                                  ("minimal", binding.TEXT_JSON_MINIMAL)):
                    st, _ = ver.verify_stwo_texts(cfg, texts[kind], fmt=fmt)
                    assert (st == 0).all(), (kind, st)
                assert (ver.verify_stwo_records(cfg, recs) == 0).all()
                assert (ver.verify_stwo_shared_records(cfg, shared) == 0).all()
                assert (ver.verify_stwo_minimal_records(cfg, mins) == 0).all()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used = (total - free) / 2**20
        if base is None:
            base = (used, rss_mb())
        print("round %d: device memory in use %.0f MiB (%+.0f since round 0), host RSS %.0f MiB (%+.0f)" % (
            r, used, used - base[0], rss_mb(), rss_mb() - base[1]), flush=True)


if __name__ == "__main__":
    main()
