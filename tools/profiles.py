#!/usr/bin/env python3
"""Turns the output of tools/evidence.sh (gpurun_out/<tag>) into the committed profiles/<tag>_* files.

    python tools/profiles.py r04

Every derived file is stamped with the commit and with the digest of the kernel sources it was measured on
(bench.kernel_sources_digest): bench.py uses the counter figures only while that digest is the tree's."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import pmc_summary  # noqa: E402
import bench  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
ONLY = set(sys.argv[2:])  # `profiles.py r04 bench host`: only those (bench stats pmc e2e sha host prover fuzz textfuzz sweep); default: everything


def want(section):
    return not ONLY or section in ONLY

G = os.path.join(ROOT, "gpurun_out", TAG)
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
# The digest the profiles are stamped with is the one evidence.sh computed ON THE GPU BOX, over the sources it measured
# (gpurun_out/<tag>/measured_tree.json) -- never the digest of the tree this script happens to run in: if csrc changed
# between the GPU run and this summary, the counters stay tied to the old sources and bench.py stops using them.
local_digest = bench.kernel_sources_digest()
try:
    digest = json.load(open(os.path.join(G, "measured_tree.json")))["kernel_sources_sha256"]
except (OSError, ValueError, KeyError):
    raise SystemExit("gpurun_out/%s/measured_tree.json is missing: evidence.sh writes it; without it the profiles cannot be "
                     "tied to the sources they were measured on" % TAG)
if digest != local_digest:
    print("WARNING: the measured tree's kernel sources (%s...) are not this tree's (%s...): the profiles are stamped with the "
          "MEASURED digest and commit 'unknown'; bench.py will not use their counter figures for this tree" % (digest[:12], local_digest[:12]))
    commit = "unknown (kernel sources differ from %s)" % commit


def one(pattern):
    """the newest match: gpurun merges every pass into the same directory"""
    m = glob.glob(os.path.join(G, pattern))
    return max(m, key=os.path.getmtime) if m else None


def copy(src, name):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_%s" % (TAG, name)))
        return True
    print("missing:", src or name)
    return False


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


if want("bench") and os.path.exists(os.path.join(G, "bench_default.json")):
    json.dump(last_json(os.path.join(G, "bench_default.json")), open(os.path.join(P, TAG + "_default_bench.json"), "w"))
if want("stats"):
    copy(one("stats/*/*_kernel_stats.csv"), "default_kernel_stats.csv")
    copy(one("stats_ts1/*/*_kernel_stats.csv"), "ts1_kernel_stats.csv")  # one tail stream: the kernels' own durations (the roofline's)
    copy(one("stats_e2e/*/*_kernel_stats.csv"), "e2e_kernel_stats.csv")
    copy(one("stats_prover/*/*_kernel_stats.csv"), "prover_kernel_stats.csv")
have_pmc = want("pmc") and all(one(d + "/*/*_counter_collection.csv") for d in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write"))
if have_pmc:
    for src, dst in (("pmc_valu", "pmc_valu.csv"), ("pmc_wait", "pmc_wait.csv"), ("pmc_fetch", "pmc_FETCH_SIZE.csv"), ("pmc_write", "pmc_WRITE_SIZE.csv")):
        copy(one(src + "/*/*_counter_collection.csv"), dst)
    s = pmc_summary.summarise([os.path.dirname(one(d + "/*/*_counter_collection.csv")) for d in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write")],
                              newest_only=True)
    n, alg = 65536, 170296 * 65536
    cmd = ("rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
           "--inflight 1 --distinct 0 --e2e 0 --tail-streams 1   (tools/evidence.sh, section pmc: four separate passes -- two SQ sets, FETCH_SIZE, WRITE_SIZE)")
    TOP = "stwo_top_hash_kernel_sha"  # Q = 16 divides 64: the merkle kernel makes the byte compares, the top kernel only hashes
    out = {"what": "PMC counters of the verifier kernels on bench.py's default workload (stwo 2^20, 65 536 proofs per launch): "
                   "stwo_merkle_kernel_sha incl. the memoisation's byte compares, stwo_top_hash_kernel_sha, the HEAD half",
           "commit": commit, "kernel_sources_sha256": digest, "command": cmd, "per_launch_avg": {}}
    for k, v in s.items():
        if k.startswith("stwo_"):
            out["per_launch_avg"][k] = {c: (x["avg"] if isinstance(x, dict) else x) for c, x in v.items()}
    d = {}
    for k in ("stwo_merkle_kernel_sha", TOP, "stwo_transcript_kernel_sha", "stwo_query_kernel"):
        m = out["per_launch_avg"].get(k)
        if not m:
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        d[k] = {"gpu_cycles": cyc, "clock_GHz": cyc / (m["avg_ms_with_counters"] * 1e-3) / 1e9,
                "valu_instructions": m["SQ_INSTS_VALU"],
                "valu_utilisation (instructions / (1024 SIMDs x cycles / 4))": m["SQ_INSTS_VALU"] / (1024 * cyc / 4),
                "issue_frac_nominal (instructions x 2 cycles / (1024 SIMDs x cycles))": m["SQ_INSTS_VALU"] * 2 / (1024 * cyc),
                "cycles_per_valu_instruction_per_simd": 1024 * cyc / m["SQ_INSTS_VALU"],
                "avg_waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc),
                "hbm_bytes (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024)": m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024}
    stage = ("stwo_merkle_kernel_sha", TOP)
    tot = sum(d[k]["hbm_bytes (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024)"] for k in stage)
    insts = sum(d[k]["valu_instructions"] for k in stage)
    d["merkle_stage"] = {"hbm_bytes_per_launch": tot, "algorithmic_bytes_per_launch": alg, "ratio": tot / alg, "valu_instructions": insts}
    out["derived"] = d
    json.dump(out, open(os.path.join(P, TAG + "_pmc_merkle_top.json"), "w"), indent=1)
    json.dump({"workload": "stwo_2p20", "proofs_per_launch": n, "kernel": "stwo_merkle+stwo_top", "commit": commit,
               "kernel_sources_sha256": digest,
               "hbm_bytes_per_launch": tot, "hbm_bytes_per_proof": tot / n, "algorithmic_bytes_per_proof": 170296,
               "ratio_to_algorithmic": tot / alg, "valu_instructions_per_launch": insts, "valu_instructions_per_proof": insts / n,
               "correction": "MI355X_MICROARCH.md HBM section: bytes = counter x 1024; on gfx950 FETCH_SIZE reports half of a "
                             "wide coalesced (16 B/lane) read, so the fetch side is doubled; WRITE_SIZE is exact",
               "source": ["profiles/%s_pmc_FETCH_SIZE.csv" % TAG, "profiles/%s_pmc_WRITE_SIZE.csv" % TAG, "profiles/%s_pmc_valu.csv" % TAG],
               "command": cmd}, open(os.path.join(P, TAG + "_hbm_traffic.json"), "w"), indent=1)
    print("merkle stage: %.2f GB per launch = %.3f x algorithmic; %.2f G VALU instructions" % (tot / 1e9, tot / alg, insts / 1e9))

# the GPU text reader: HBM bytes per text byte
if want("pmc") and one("pmc_fetch_e2e/*/*_counter_collection.csv") and one("pmc_write_e2e/*/*_counter_collection.csv"):
    e = pmc_summary.summarise([os.path.dirname(one(dd + "/*/*_counter_collection.csv")) for dd in ("pmc_fetch_e2e", "pmc_write_e2e")],
                              newest_only=True, keep=("text_", "stwo_shared", "stwo_pack"))
    text = {}
    for k, v in e.items():
        if "text_" in k or "shared" in k:
            text[k] = {"launches": v["FETCH_SIZE"]["launches"], "fetch_bytes_avg": v["FETCH_SIZE"]["avg"] * 2048,
                       "write_bytes_avg": v.get("WRITE_SIZE", {}).get("avg", 0) * 1024, "avg_ms_with_counters": v["avg_ms_with_counters"]}
    json.dump({"what": "FETCH_SIZE / WRITE_SIZE of the GPU text reader's kernels and of the shared-record expansion per launch (one launch = "
                       "one chunk of up to 64 MiB of text), `tools/e2e_bench.py --n 1024 --reps 1 --fmt all` under rocprofv3 --pmc, two passes",
               "commit": commit, "kernels": text}, open(os.path.join(P, TAG + "_pmc_text_reader.json"), "w"), indent=1)

lines = {}
for f in sorted(glob.glob(os.path.join(G, "bench_*.json"))) if want("bench") else []:
    try:
        line = last_json(f)
    except (ValueError, IndexError):
        continue
    line["_commit"] = commit
    lines[os.path.basename(f)[6:-5]] = line
if lines:
    json.dump({"note": "one bench.py JSON line per configuration (tools/evidence.sh, sections bench + configs), each stamped with the commit it was "
                       "measured at; *_nodedup = SS_FLAG_NO_DEDUP (every path hashed in full); *_8192 = one GPU's share of the 65 536-proof batch "
                       "split over 8 (--proofs-per-gpu 8192); _ts1 = one tail stream (round 5's submission), default = two (three below 32 768 proofs per rank)",
               "lines": lines}, open(os.path.join(P, TAG + "_bench_configs.json"), "w"), indent=1)
e2e = {}
for f in sorted(glob.glob(os.path.join(G, "e2e_*.json"))) if want("e2e") else []:
    try:
        e2e[os.path.basename(f)[:-5]] = json.load(open(f))
    except ValueError:
        pass
if e2e:
    json.dump({"note": "tools/e2e_bench.py: text -> verdict through ss_stwo_verify_texts / _files, formats json / wit / shared (the shared-path "
                       "proof.json, read and expanded on the GPU) (e2e_4096: 4096 texts of the 2^20 shape, with --files; _nc1: 1 % of the texts with "
                       "reversed member order = host reader; _512: a small batch; _2p16: the 2^16 / Q=32 shape; _pysep: proof.json with json.dumps' "
                       "default separators)", "commit": commit, "runs": e2e}, open(os.path.join(P, TAG + "_e2e.json"), "w"), indent=1)
for sect, name, src, head in (("sha", "sha_calibration.txt", "sha_bench.txt", "$ build/sha_bench 512      (tools/sha_bench.hip at %s; MI355X)" % commit),
                        ("host", "host_path.txt", "host_path.txt", "$ python tools/host_path_bench.py 2048 [distinct]; ... 16384 [distinct]   (ss_stwo_verify_records and "
                                                           "ss_stwo_verify_shared_records, 2^20 shape, at %s; the first call of a run allocates the scratch; "
                                                           "`distinct`: every record its own host buffer)" % commit),
                        ("prover", "prover_bench.txt", "prover_bench.txt", "$ python tools/prover_bench.py 20 3 sha256 1,4,4,3 48; python tools/prover101_bench.py   (at %s)" % commit),
                        ("fuzz", "fuzz_parity.txt", "fuzz_parity.txt", "$ python tools/fuzz_parity.py 20000 20261004   (at %s; GPU status words against the oracle, per-query and shared records)" % commit),
                        ("textfuzz", "text_fuzz.txt", "text_fuzz.txt", "$ python tools/text_fuzz.py 4000 20261004   (at %s; GPU reader against the scalar rule and the host reader, json / wit / json-shared)" % commit),
                        ("sweep", "shape_sweep.txt", "shape_sweep.txt", "$ python tools/shape_sweep.py <shapes> <seed>   (at %s; the last line says how many)" % commit)):
    path = os.path.join(G, src)
    if want(sect) and os.path.exists(path):
        body = [l for l in open(path).read().splitlines() if "amdgpu.ids" not in l]
        open(os.path.join(P, "%s_%s" % (TAG, name)), "w").write("\n".join([head] + body) + "\n")
print("profiles written at", commit, "kernel sources", digest[:12])
