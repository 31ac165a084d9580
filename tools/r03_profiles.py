#!/usr/bin/env python3
"""Turns the output of tools/r03/gpu_final.sh (gpurun_out/r03final) into the committed profiles/r03_* files.

    python tools/r03_profiles.py [gpurun_out/r03final]
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402

G = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03final"))
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()


def one(pattern):
    """the newest match: gpurun merges every pass into the same directory"""
    return max(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)


shutil.copy(os.path.join(G, "bench_default.json"), os.path.join(P, "r03_default_bench.json"))
shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(P, "r03_default_kernel_stats.csv"))
shutil.copy(one("stats_e2e/*/*_kernel_stats.csv"), os.path.join(P, "r03_e2e_kernel_stats.csv"))
for src, dst in (("pmc_valu", "r03_pmc_valu.csv"), ("pmc_wait", "r03_pmc_wait.csv"),
                 ("pmc_fetch", "r03_pmc_FETCH_SIZE.csv"), ("pmc_write", "r03_pmc_WRITE_SIZE.csv")):
    shutil.copy(one(src + "/*/*_counter_collection.csv"), os.path.join(P, dst))

s = pmc_summary.summarise([os.path.dirname(one(d + "/*/*_counter_collection.csv")) for d in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write")],
                          newest_only=True)
n, alg = 65536, 170296 * 65536
cmd = ("rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
       "--no-cpu-baseline --inflight 1 --distinct 0 --e2e 0 --tail-streams 1   (tools/r03/gpu_final.sh; four separate passes: "
       "the two SQ sets of that script, FETCH_SIZE, WRITE_SIZE)")
TOP = "stwo_top_hash_kernel_sha"  # Q = 16 divides 64: the merkle kernel makes the byte compares, the top kernel only hashes
out = {"what": "PMC counters of the Merkle stage (stwo_merkle_kernel_sha incl. the memoisation's byte compares + stwo_top_hash_kernel_sha, "
               "top siblings stored per proof, cold path in stwo_top_cold_kernel) on bench.py's default workload",
       "commit": commit, "command": cmd, "workload": "stwo_2p20, 65536 proofs per launch", "per_launch_avg": {}}
for k, v in s.items():
    if k.startswith("stwo_"):
        out["per_launch_avg"][k] = {c: (x["avg"] if isinstance(x, dict) else x) for c, x in v.items()}
d = {}
for k in ("stwo_merkle_kernel_sha", TOP):
    m = out["per_launch_avg"][k]
    cyc = m["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
    d[k] = {"gpu_cycles": cyc, "clock_GHz": cyc / (m["avg_ms_with_counters"] * 1e-3) / 1e9,
            "valu_issue_slots (1024 SIMDs x cycles / 4)": 1024 * cyc / 4,
            "valu_utilisation": m["SQ_INSTS_VALU"] / (1024 * cyc / 4),
            "cycles_per_valu_instruction_per_simd": 1024 * cyc / m["SQ_INSTS_VALU"],
            "avg_waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc),
            "hbm_bytes (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024)": m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024}
tot = sum(v["hbm_bytes (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024)"] for v in d.values())
d["merkle_stage"] = {
    "hbm_bytes_per_launch": tot, "algorithmic_bytes_per_launch": alg, "ratio": tot / alg,
    "valu_instructions": sum(out["per_launch_avg"][k]["SQ_INSTS_VALU"] for k in ("stwo_merkle_kernel_sha", TOP)),
    "round 3 before the byte compares moved into the merkle kernel (commit 86a5b0c..ab4f14d)": {
        "ratio": 1.91, "top_kernel": "2.20 G VALU instructions at utilisation 0.965, 3.92-3.97 ms, 11.5 GB fetched",
        "merkle_kernel": "8.85 G at 1.03, 14.7-14.8 ms, 8.12 GB"},
    "round 2 (profiles/r02_pmc_merkle_top.json)": {"ratio": 2.044, "top_kernel_fetch_GB": 12.99}}
out["derived"] = d
json.dump(out, open(os.path.join(P, "r03_pmc_merkle_top.json"), "w"), indent=1)
json.dump({"workload": "stwo_2p20", "proofs_per_launch": n, "kernel": "stwo_merkle+stwo_top", "commit": commit,
           "hbm_bytes_per_launch": tot, "hbm_bytes_per_proof": tot / n, "algorithmic_bytes_per_proof": 170296,
           "ratio_to_algorithmic": tot / alg,
           "correction": "MI355X_MICROARCH.md HBM section: bytes = counter x 1024; on gfx950 FETCH_SIZE reports half of a "
                         "wide coalesced (16 B/lane) read, so the fetch side is doubled; WRITE_SIZE is exact",
           "source": ["profiles/r03_pmc_FETCH_SIZE.csv", "profiles/r03_pmc_WRITE_SIZE.csv"], "command": cmd},
          open(os.path.join(P, "r03_hbm_traffic.json"), "w"), indent=1)

# the GPU text reader: HBM bytes per text byte
try:
    e = pmc_summary.summarise([os.path.dirname(one(dd + "/*/*_counter_collection.csv")) for dd in ("pmc_fetch_e2e", "pmc_write_e2e")],
                              newest_only=True)
    text = {}
    for k, v in e.items():
        if "text_" in k:
            text[k] = {"launches": v["FETCH_SIZE"]["launches"], "fetch_bytes_avg": v["FETCH_SIZE"]["avg"] * 2048,
                       "write_bytes_avg": v.get("WRITE_SIZE", {}).get("avg", 0) * 1024, "avg_ms_with_counters": v["avg_ms_with_counters"]}
    json.dump({"what": "FETCH_SIZE / WRITE_SIZE of the GPU text reader's kernels per launch (one launch = one chunk of up to 64 MiB of "
                       "text), `tools/e2e_bench.py --n 1024 --reps 1 --fmt json` under rocprofv3 --pmc, two passes",
               "commit": commit, "kernels": text}, open(os.path.join(P, "r03_pmc_text_reader.json"), "w"), indent=1)
except (ValueError, KeyError) as ex:
    print("no text reader counters:", ex)

lines = {}
for f in sorted(glob.glob(os.path.join(G, "bench_*.json"))):
    try:
        line = json.load(open(f))
    except ValueError:
        continue
    line["_commit"] = commit
    lines[os.path.basename(f)[6:-5]] = line
json.dump({"note": "one bench.py JSON line per configuration (tools/r03/gpu_final.sh), each stamped with the commit it was measured "
                   "at; *_nodedup = SS_FLAG_NO_DEDUP (every path hashed in full); *_8192 = one GPU's share of the 65 536-proof batch "
                   "split over 8 (--proofs-per-gpu 8192; _ts1 = one Merkle stream, default = two)",
           "lines": lines}, open(os.path.join(P, "r03_bench_configs.json"), "w"), indent=1)

e2e = {}
for f in sorted(glob.glob(os.path.join(G, "e2e_*.json"))):
    try:
        e2e[os.path.basename(f)[:-5]] = json.load(open(f))
    except ValueError:
        pass
json.dump({"note": "tools/e2e_bench.py: text -> verdict through ss_stwo_verify_texts / _files (e2e_4096: 4096 texts of the 2^20 shape, "
                   "with --files; _nc1: 1 % of the texts with reversed member order = host reader; _512: a small batch; _2p16: the "
                   "2^16 / Q=32 shape; _pysep: proof.json with json.dumps' default separators, 27 % more bytes)", "commit": commit, "runs": e2e}, open(os.path.join(P, "r03_e2e.json"), "w"), indent=1)

for name, src, head in (("r03_sha_calibration.txt", "sha_bench.txt", "$ build/sha_bench 512      (tools/sha_bench.hip at %s; MI355X)" % commit),
                        ("r03_host_path.txt", "host_path.txt", "$ python tools/host_path_bench.py 2048; ... 16384   (ss_stwo_verify_records, 2^20 shape, at %s; "
                                                               "the first call allocates the scratch)" % commit)):
    body = [l for l in open(os.path.join(G, src)).read().splitlines() if "amdgpu.ids" not in l]
    open(os.path.join(P, name), "w").write("\n".join([head] + body) + "\n")
print("profiles written at", commit)
