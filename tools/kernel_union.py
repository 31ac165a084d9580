"""How busy is the GPU?  From a rocprofv3 --kernel-trace CSV: the wall span between the first kernel's start and the last
kernel's end inside a window, the UNION of the kernels' busy intervals (time at least one kernel was running), and the sum of
their durations (> the union where kernels of different streams overlap).

    python tools/kernel_union.py <dir with *_kernel_trace.csv> [skip_fraction]

skip_fraction (default 0.35): the first part of the trace is dropped (start-up: twiddles, the first proof alone on the chip).
Used for VERDICT r5, 5: while prove_many keeps four proofs in flight, is there idle time that batching launches across proofs
could fill?"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under %s" % d)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + int((t1 - t0) * skip)
    hi = t1 - int((t1 - t0) * 0.05)
    win = [(max(a, lo), min(b, hi), k) for a, b, k in rows if b > lo and a < hi]
    union, cur_a, cur_b = 0, None, None
    for a, b, _ in win:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    total = sum(b - a for a, b, _ in win)
    span = hi - lo
    gaps = span - union
    by = {}
    for a, b, k in win:
        name = k.split("(")[0].replace("void ", "").replace("ss::", "")
        by[name] = by.get(name, 0) + (b - a)
    print("window %.1f ms (%d kernels): at least one kernel running %.1f ms = %.1f %% of the window; idle gaps %.2f ms (%.1f %%); "
          "sum of kernel durations %.1f ms = %.2f x the window" % (span / 1e6, len(win), union / 1e6, 100 * union / span, gaps / 1e6,
                                                                    100 * gaps / span, total / 1e6, total / span))
    for name, t in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print("  %-40s %8.2f ms  %5.1f %% of the kernel time" % (name[:40], t / 1e6, 100 * t / total))


if __name__ == "__main__":
    main()
