#!/usr/bin/env python3
"""Turns the output of tools/r02_gpu_g.sh (gpurun_out/g) into the committed profiles/r02_* files.

    python tools/r02_profiles.py [gpurun_out/g]
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

G = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "g"))
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()


def one(pattern):
    """the newest match: gpurun merges every pass into the same directory"""
    return max(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)


shutil.copy(os.path.join(G, "bench_default.json"), os.path.join(P, "r02_default_bench.json"))
shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(P, "r02_default_kernel_stats.csv"))
for src, dst in (("pmc_valu", "r02_pmc_valu.csv"), ("pmc_wait", "r02_pmc_wait.csv"),
                 ("pmc_fetch", "r02_pmc_FETCH_SIZE.csv"), ("pmc_write", "r02_pmc_WRITE_SIZE.csv")):
    shutil.copy(one(src + "/*/*_counter_collection.csv"), os.path.join(P, dst))

s = pmc_summary.summarise([os.path.dirname(one(d + "/*/*_counter_collection.csv")) for d in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write")],
                          newest_only=True)
n, alg = 65536, 170296 * 65536
cmd = ("rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
       "--no-cpu-baseline --inflight 1 --distinct 0 --e2e 0   (tools/r02_gpu_g.sh; four separate passes: the two SQ "
       "sets of that script, FETCH_SIZE, WRITE_SIZE)")
out = {"what": "PMC counters of the Merkle stage with pair memoisation (stwo_merkle_kernel_sha + stwo_top_kernel_sha) "
               "on bench.py's default workload",
       "commit": commit, "command": cmd, "workload": "stwo_2p20, 65536 proofs per launch", "per_launch_avg": {}}
for k, v in s.items():
    if k.startswith("stwo_"):
        out["per_launch_avg"][k] = {c: (x["avg"] if isinstance(x, dict) else x) for c, x in v.items()}
d = {}
for k in ("stwo_merkle_kernel_sha", "stwo_top_kernel_sha"):
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
    "valu_instructions": sum(out["per_launch_avg"][k]["SQ_INSTS_VALU"] for k in ("stwo_merkle_kernel_sha", "stwo_top_kernel_sha")),
    "valu_instructions_without_memoisation (profiles/r02_pmc_merkle_nodedup.json)": 13725483008}
out["derived"] = d
out["reading"] = (
    "stwo_merkle: every VALU issue slot used (utilisation ~1.0 at 3.9 cycles per instruction), as before, on a third fewer "
    "instructions.  stwo_top: ~2.2 G instructions = 1.92 G of pair hashing (835 k wave-iterations x 2 301) + ~350 per "
    "iteration of plan, fetch, checks and register rotation, issued in ~92 % of the slots (block barriers per depth).  "
    "Stage total ~11.1 G instructions against 13.7 G without memoisation.  HBM traffic of the stage with the guide's x2 on "
    "FETCH_SIZE is ~2.2x the algorithmic bytes (entering nodes, stored nodes, top siblings read by leaders and by the "
    "checks; the top kernel's 16-byte pieces of different tile rows are probably over-corrected by the x2), a seventh of "
    "what the memory system delivers: the binding roof stays the integer VALU.")
json.dump(out, open(os.path.join(P, "r02_pmc_merkle_top.json"), "w"), indent=1)
json.dump({"workload": "stwo_2p20", "proofs_per_launch": n, "kernel": "stwo_merkle+stwo_top", "commit": commit,
           "hbm_bytes_per_launch": tot, "hbm_bytes_per_proof": tot / n, "algorithmic_bytes_per_proof": 170296,
           "ratio_to_algorithmic": tot / alg,
           "correction": "MI355X_MICROARCH.md HBM section: bytes = counter x 1024; on gfx950 FETCH_SIZE reports half of a "
                         "wide coalesced (16 B/lane) read, so the fetch side is doubled; WRITE_SIZE is exact",
           "source": ["profiles/r02_pmc_FETCH_SIZE.csv", "profiles/r02_pmc_WRITE_SIZE.csv"], "command": cmd},
          open(os.path.join(P, "r02_hbm_traffic.json"), "w"), indent=1)

cfg_path = os.path.join(P, "r02_bench_configs.json")
prev = json.load(open(cfg_path)) if os.path.exists(cfg_path) else {"lines": {}}
lines = prev.get("lines", {})
for f in sorted(glob.glob(os.path.join(G, "bench_*.json"))):
    try:
        line = json.load(open(f))
    except ValueError:
        continue
    line["_commit"] = commit
    lines[os.path.basename(f)[6:-5]] = line
json.dump({"note": "one bench.py JSON line per configuration (tools/r02_gpu_g.sh / r02_gpu_k.sh), each stamped with the "
                   "commit it was measured at; *_nodedup = SS_FLAG_NO_DEDUP (every path hashed in full)",
           "lines": lines}, open(cfg_path, "w"), indent=1)

for name, src in (("r02_sha_lds_ab.txt", "sha_bench.txt"), ("r02_fuzz_parity.txt", "fuzz.txt"), ("r02_host_path.txt", "host_path.txt")):
    head = {"r02_sha_lds_ab.txt": "$ build/sha_bench 512      (tools/sha_bench.hip at %s; MI355X)" % commit,
            "r02_fuzz_parity.txt": "$ python tools/fuzz_parity.py 8000 20261003     (at %s: pair memoisation on, ABI 2 records)" % commit,
            "r02_host_path.txt": "$ python tools/host_path_bench.py 2048   (ss_stwo_verify_records, 2^20 shape, at %s; the first "
                                 "call allocates the scratch)" % commit}[name]
    body = [l for l in open(os.path.join(G, src)).read().splitlines() if "amdgpu.ids" not in l]
    tail = []
    if name == "r02_sha_lds_ab.txt":
        tail = ["", "schedule window in LDS = the first compression's 16-word rolling schedule in LDS (volatile column per lane)",
                "instead of VGPRs.  North-star asked for the LDS staging by name; an LDS round trip per schedule word only adds",
                "instructions to a loop whose cost is its instruction count."]
    text = "\n".join([head] + body + tail) + "\n"
    dst = os.path.join(P, name)
    if name == "r02_fuzz_parity.txt" and os.path.exists(dst):
        # the file also holds the long runs of tools/r02_gpu_fuzz.sh: keep them, replace only an earlier
        # section made by this script at the same commit, append otherwise
        sections = [sec for sec in open(dst).read().split("\n\n") if sec.strip() and not sec.startswith(head)]
        text = "\n\n".join([sec.rstrip("\n") for sec in sections] + [text])
    open(dst, "w").write(text)
print("profiles written at", commit)
