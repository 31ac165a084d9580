#!/usr/bin/env python3
"""Where one process-per-proof run spends its time (the reference's `make run` convention): HIP start-up, context,
first and second call of the text entry point.  python tools/probes/latency_probe.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from stark_symphony_amd import binding as B  # noqa: E402
lib = B.lib()
t1 = time.perf_counter()
n = lib.ss_device_count()
t2 = time.perf_counter()
ctx = C.c_void_p()
B.check(lib.ss_ctx_create(0, C.byref(ctx)))
t3 = time.perf_counter()
import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import verifier  # noqa: E402
text = open(os.path.join(ROOT, "tests", "golden", "formats", "stwo_proof.wit"), "rb").read()
cs = verifier.stwo_cfg_struct(ss.PRODUCTION_CONFIG, verifier.MODE_FIXTURE)
import numpy as np  # noqa: E402
status = np.zeros(1, np.uint32)
arr = (C.c_char_p * 1)(text)
lens = (C.c_size_t * 1)(len(text))
stats = B.IngestStats()
ts = []
for _ in range(3):
    a = time.perf_counter()
    B.check(lib.ss_stwo_verify_texts(ctx, C.byref(cs), 1, arr, lens, B.TEXT_WIT, status.ctypes.data, C.byref(stats)))
    ts.append(time.perf_counter() - a)
print("load library %.3f s, first HIP call (device count) %.3f s, ss_ctx_create %.3f s, verify_texts calls: %s ms, status %d"
      % (t1 - t0, t2 - t1, t3 - t2, ["%.2f" % (x * 1e3) for x in ts], int(status[0])))
