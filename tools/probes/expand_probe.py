#!/usr/bin/env python3
"""Bandwidth of the shared-record expansion kernel (csrc/ss_shared.hip) with the chip full: n shared records of the 2^20
shape resident in HBM -> per-query records, timed with events around ss_stwo_expand_shared_dev.
    python tools/probes/expand_probe.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from stark_symphony_amd import binding as B, records, verifier  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
proofs = records.load_stwo_npz(os.path.join(ROOT, "tests", "golden", "stwo_trace20.npz"))
ver = verifier.Verifier(0)
cfg = proofs[0].cfg
cs = verifier.stwo_cfg_struct(cfg, verifier.MODE_FIXTURE)
shared = [verifier.stwo_shared_record(p) for p in proofs]
W = B.lib().ss_stwo_record_words(C.byref(cs))
batch = [shared[i % len(shared)] for i in range(n)]
offs = np.zeros(n + 1, dtype=np.uint64)
offs[1:] = np.cumsum([b.size for b in batch])
dev = ver.device
sh_dev = torch.from_numpy(np.concatenate(batch).view(np.int32)).to(dev)
offs_dev = torch.from_numpy(offs.view(np.int64)).to(dev)
rec_dev = torch.empty(n * W, dtype=torch.int32, device=dev)
out_dev = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(dev)
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    B.check(B.lib().ss_stwo_expand_shared_dev(ver.ctx, C.byref(cs), n, sh_dev.data_ptr(), offs_dev.data_ptr(), rec_dev.data_ptr(),
                                              out_dev.data_ptr(), int(s.cuda_stream)))
    e1.record(s)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1)
    moved = sh_dev.numel() * 4 + rec_dev.numel() * 4
    print("expand %d records: %.3f ms, %.2f GB read + written = %.0f GB/s (%.2f of 8 TB/s)" % (n, ms, moved / 1e9, moved / ms / 1e6, moved / ms / 1e6 / 8000))
assert int(out_dev.abs().sum().item()) == 0
want = verifier.stwo_record(proofs[0])
assert np.array_equal(rec_dev[:W].cpu().numpy().view(np.uint32), want)
