#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel and counter, launches and the
average counter value per launch (rocprofv3 sums over XCDs / SEs).

    python tools/pmc_summary.py gpurun_out/a/pmc_valu gpurun_out/a/pmc_wait ... > profiles/r02_pmc.json
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(stwo_\w+|s101_\w+|text_\w+|p_\w+)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


def summarise(dirs, newest_only=False, keep=None):
    """newest_only: of several runs merged into one directory, read only the latest file"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        paths = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        if newest_only and paths:
            paths = [max(paths, key=os.path.getmtime)]
        for path in paths:
            seen = set()
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    dur[(k, path)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    out = {}
    for k, counters in acc.items():
        out[k] = {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in sorted(counters.items())}
        ds = [x for (kk, _), v in dur.items() if kk == k for x in v]
        out[k]["avg_ms_with_counters"] = sum(ds) / len(ds)
    return out


if __name__ == "__main__":
    s = summarise(sys.argv[1:])
    keep = {k: v for k, v in s.items() if k.startswith(("stwo_", "s101_"))}
    json.dump(keep, sys.stdout, indent=1)
    print()
