#!/usr/bin/env python3
"""Benchmark of the batch STARK verifier hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 launched by torch.distributed.run; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the environment).  A "step" verifies one batch of stwo
circle-STARK proofs that is already resident in HBM, then all-reduces the accept count over RCCL
(the only exchange of the path).  Rank 0 prints ONE JSON line.

`--scaling strong` (default): the step is ONE batch of `--batch` proofs (65 536 for the metric
config, BASELINE.json configs[3]: "batch of 65536 proofs sharded across 8 x MI355X") split
`distributed.shard_range`-wise over the N ranks -- 8 192 proofs per GPU at N = 8, the whole batch
on the one GPU at N = 1.  `--scaling weak`: every rank owns its own `--proofs-per-gpu` batch
(65 536 by default), total work grows with N.  The JSON line says which.

Submission (round 6): K steps are K passes of the HEAD / TAIL pipeline (verifier.Pipeline) with the TAIL halves
alternating over two streams -- three, with four passes in flight, when a rank's share is below 32 768 proofs -- so that
the Merkle stage of consecutive passes overlaps.  `value` and `ms_per_step` come from that timed region; a kernel's
duration is not its own under overlap, so the durations behind `roofline` / `alu_roofline.frac` are measured by HIP events
in a pass through a ONE-slot pipeline right after it, where nothing overlaps (`roofline.durations_from`), and `roofline.launch_period_ms` /
`alu_roofline.frac_of_step` give the timed region's own figures beside them.  `--tail-streams 1` is round 5's submission.

Workloads (`--workload`):
  stwo_2p20      2^20-row wide-Fibonacci trace, blowup 2^4 (LDE 2^24), 16 queries, 19 inner FRI
                 layers, SHA-256 -- BASELINE.json configs[3], the configuration the metric is
                 quoted on: a batch of 65 536 proofs (11.2 GB of records, fits one GPU) per step and
                 per rank.  Proofs: tests/golden/stwo_trace20.npz (made by tools/stwo_prover.py) plus
                 distinct ones made at start-up by the GPU prover (--distinct).
  stwo_fixture   the reference's own proof (tests/golden/stwo_proof.json: trace 2^9, LDE 2^13,
                 16 queries) replicated -- used when the 2^20 fixture is absent.
  stwo_2p16      BASELINE.json configs[2]: 2^16 trace, 32 queries.
  stwo_wide256   BASELINE.json configs[4]: 256 columns, LDE 2^18.
  stwo_2p16_blake2s, stwo_wide256_blake2s  configs[2] / configs[4] with Blake2s, as BASELINE.json names them.
  stwo_2p20_blake2s  the metric config with Blake2s-256 as the hash (BASELINE.json says "Blake2s
                 Merkle"; the reference has no Blake2s, so this variant's parity is unpinned).
  stark101       BASELINE.json configs[1]: the stark101 proof x 4096.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak
# The roof that actually binds: SHA-256 compressions/s of a register-only pair-hash chain on
# the whole chip (tools/sha_bench.hip, measured on MI355X: 34.7 G/s; rotates are half-rate
# v_alignbit_b32, see tools/valu_bench.hip and DESIGN.md section 4).
SHA_CALIBRATED_PEAK = 34.7e9
B2S_CALIBRATED_PEAK = 38.9e9  # Blake2s-256 compressions/s, same tool (one compression per node)


# Workloads without a committed fixture (none at present): every proof of the batch is then made by the
# GPU prover at start-up.  BASELINE.json configs[2] / configs[4] with the hash they name (Blake2s) have
# fixtures since round 3 (tests/golden/stwo_trace16_blake2s.npz, stwo_wide256_blake2s.npz; parity unpinned,
# DESIGN.md section 1) and `-m gpu` parity tests behind them.
GEN_ONLY: dict = {}


def hash_label(h: str) -> str:
    """The reference has no Blake2s: every line measured with it says that its parity is unpinned."""
    return "blake2s (PARITY UNPINNED: not a hash of the reference; RFC 7693 + prover / oracle / GPU agreement only)" if h == "blake2s" else h


def load_workload(name: str):
    """-> (workload name, family, list of distinct proofs, note)"""
    import stark_symphony_amd as ss
    from stark_symphony_amd import formats
    big = os.path.join(GOLDEN, "stwo_trace20.npz")
    if name == "auto":
        name = "stwo_2p20" if os.path.exists(big) else "stwo_fixture"
    if name == "stwo_2p20":
        from stark_symphony_amd import records
        proofs = records.load_stwo_npz(big)
        return name, "stwo", proofs, "wide-Fibonacci 2^20 x 4, LDE 2^24, Q=16, K=19, SHA-256"
    if name in ("stwo_2p16", "stwo_wide256", "stwo_2p20_blake2s", "stwo_2p16_blake2s", "stwo_wide256_blake2s"):
        from stark_symphony_amd import records
        fn = {"stwo_2p16": "stwo_trace16.npz", "stwo_wide256": "stwo_wide256.npz",
              "stwo_2p20_blake2s": "stwo_trace20_blake2s.npz", "stwo_2p16_blake2s": "stwo_trace16_blake2s.npz",
              "stwo_wide256_blake2s": "stwo_wide256_blake2s.npz"}[name]
        proofs = records.load_stwo_npz(os.path.join(GOLDEN, fn))
        c = proofs[0].cfg
        return name, "stwo", proofs, "wide-Fibonacci 2^%d x %d, LDE 2^%d, Q=%d, K=%d, %s" % (
            c.trace_log, c.n_cols, c.lde_log, c.n_queries, c.n_layers, hash_label(c.hash))
    if name in GEN_ONLY:
        c = formats.StwoConfig(**GEN_ONLY[name])
        return name, "stwo", [], "wide-Fibonacci 2^%d x %d, LDE 2^%d, Q=%d, K=%d, %s" % (
            c.trace_log, c.n_cols, c.lde_log, c.n_queries, c.n_layers, hash_label(c.hash))
    if name == "stwo_fixture":
        p = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
        return name, "stwo", [p], "reference proof.json (trace 2^9, LDE 2^13, Q=16, K=8) replicated"
    if name == "stark101":
        p = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
        return name, "stark101", [p], "reference stark101 proof replicated (only one valid proof exists)"
    raise SystemExit("unknown workload %r" % name)


# (kernels, their headers, and what decides launch geometry and batch layout: ss_api.hip's grids / stream split,
# ss_pack.cpp's lay_of and its flags -- ADVICE r4)
MERKLE_STAGE_SOURCES = ("ss_stwo.hip", "ss_stwo_checks.h", "ss_sha256.h", "ss_hash.h", "ss_layout.h", "ss_fields.h", "ss_channel.h",
                        "ss_api.hip", "ss_pack.cpp", "ss_kernels.h")


def kernel_sources_digest() -> str:
    """sha256 over the sources that define the stwo verifier kernels (the counter profiles are of the stwo metric
    config): what ties a committed counter profile to the code that is running (the GPU box has no .git to ask for a commit)."""
    import hashlib
    h = hashlib.sha256()
    for name in MERKLE_STAGE_SOURCES:
        with open(os.path.join(ROOT, "stark-symphony_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def pmc_profile(wname: str, n_local: int):
    """Counter figures of the dominant kernels per launch.  Counters cannot be read from inside this process: they come
    from separate rocprofv3 `--pmc` passes over this same command (tools/evidence.sh), summarised -- FETCH_SIZE /
    WRITE_SIZE with the guide's gfx950 correction, SQ_INSTS_VALU -- in profiles/*_hbm_traffic.json together with the
    digest of the kernel sources they were taken on.  They are used ONLY when that digest is the one of the sources in
    this tree; otherwise the figures are null.  Both are linear in the proofs per launch.
    -> (hbm bytes or None, VALU wave-instructions or None, provenance string)."""
    import glob
    digest = kernel_sources_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload") != wname:
            continue
        rel = os.path.relpath(path, ROOT)
        if d.get("kernel_sources_sha256") != digest:
            return None, None, "%s was taken on other kernel sources (digest %s..., this tree %s...): not used" % (
                rel, str(d.get("kernel_sources_sha256"))[:12], digest[:12])
        insts = d.get("valu_instructions_per_proof")
        return (d["hbm_bytes_per_proof"] * n_local, insts * n_local if insts else None,
                "%s (rocprofv3 --pmc passes at commit %s, kernel sources %s...: the ones in this tree)" % (
                    rel, d.get("commit", "unrecorded"), digest[:12]))
    return None, None, "no PMC profile of this workload under profiles/"


def cpu_baseline(family, proofs, seconds: float):
    """Oracle (CPU restatement of the reference path) on this box's host cores; bounded sample.
    SURVEY.md 8d: (i) one thread, (ii) all granted cores, (iii) `simfony run` itself when the box
    has it (it cannot be built here: SimplicityHL + Rust git forks, no cargo, no network)."""
    import shutil
    from oracle import oracle as O
    # cores this process may really use (cgroup quota, not the 256 logical CPUs a container sees:
    # tools/probes/oracle_scaling.py -- 256 OpenMP threads on a 16-core quota run 35 % slower than 16)
    threads = O.effective_cpus()

    def timed(nthreads: int, budget: float):
        if family == "stwo":
            chunk = max(nthreads * 16, 16)
            batch = O.StwoBatch([proofs[i % len(proofs)] for i in range(chunk)])
            run = lambda: batch.verify(O.MODE_FIXTURE, nthreads)  # noqa: E731
        else:
            chunk = max(nthreads * 64, 64)
            arr = O.s101_array([proofs[i % len(proofs)] for i in range(chunk)])
            run = lambda: O.s101_verify_batch(arr, nthreads)  # noqa: E731
        st = run()
        assert (st == 0).all(), "oracle rejects the benchmark proofs"
        done, t0 = 0, time.perf_counter()
        while True:
            run()
            done += chunk
            dt = time.perf_counter() - t0
            if dt >= budget:
                return done, dt

    done1, dt1 = timed(1, seconds * 0.25)
    done, dt = timed(threads, seconds * 0.75)
    simfony = shutil.which("simfony")
    out = {"value": done / dt, "unit": "proofs/s", "cores": threads, "kind": "port",
           "sample": "%d proofs of the same workload in %.1f s, C oracle (restatement of the "
                     "SimplicityHL verifier), OpenMP over proofs, %d threads = the cores this process is "
                     "allowed (%d logical CPUs visible)" % (done, dt, threads, O.num_procs()),
           "single_thread": {"value": done1 / dt1, "unit": "proofs/s", "cores": 1,
                             "sample": "%d proofs in %.1f s" % (done1, dt1)},
           "simfony": "absent: `command -v simfony` finds nothing and it cannot be built here "
                      "(no cargo / network); the reference path is timed through its C restatement"}
    if simfony:  # reference CLI present: time the real thing on its own one-proof case (config 1)
        out["simfony"] = time_simfony(simfony)
    return out


def time_simfony(exe: str):
    """`simfony run main.simf --witness proof.wit` (stark101/Makefile:8-9), one process per proof."""
    import subprocess
    import tempfile
    src = os.environ.get("SS_SIMFONY_PROGRAM")  # the mcpp-expanded stark101 program, if the box has one
    if not src or not os.path.exists(src):
        return "present at %s, but no compiled stark101 program given (SS_SIMFONY_PROGRAM)" % exe
    wit = os.path.join(GOLDEN, "formats", "stark101_proof.wit")
    with tempfile.TemporaryDirectory() as d:
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 5.0:
            r = subprocess.run([exe, "run", src, "--witness", wit], cwd=d, capture_output=True)
            if r.returncode != 0:
                return "present at %s, `simfony run` exit %d: %s" % (exe, r.returncode, r.stderr[-200:].decode("replace"))
            n += 1
        return {"value": n / (time.perf_counter() - t0), "unit": "proofs/s", "cores": 1,
                "sample": "%d x `simfony run` of the stark101 proof" % n}


def end_to_end_s101(ver, proof, n: int):
    """stark101 texts (the one valid proof replicated: SURVEY.md F3) -> verdicts through ss_s101_verify_texts."""
    import stark_symphony_amd as ss
    from stark_symphony_amd import binding
    texts = {"json": json.dumps(ss.stark101_to_json(proof)).encode(), "wit": ss.stark101_to_wit(proof).encode()}
    out = {"proofs": n, "note": "stark101 proof text -> verdict through ss_s101_verify_texts (GPU reader, host link included)"}
    for kind, fmt in (("json", binding.TEXT_JSON), ("wit", binding.TEXT_WIT)):
        batch = [texts[kind][:1] + texts[kind][1:] for _ in range(n)]  # distinct buffers
        ver.verify_stark101_texts(batch[:64], fmt=fmt)
        ver.verify_stark101_texts(batch, fmt=fmt)
        runs = []
        for _ in range(5):  # the median of five is reported
            t0 = time.perf_counter()
            status, st = ver.verify_stark101_texts(batch, fmt=fmt)
            dt = time.perf_counter() - t0
            assert (status == 0).all(), "e2e: the stark101 proof was not accepted"
            runs.append((dt, st))
        runs.sort(key=lambda r: r[0])
        dt, st = runs[len(runs) // 2]
        out[kind] = {"proofs_per_s": n / dt, "total_s": dt, "text_GB_per_s": st["text_bytes"] / dt / 1e9,
                     "host_parsed_texts": st["host_parsed"], "text_bytes_per_proof": st["text_bytes"] // n}
    return out


def end_to_end(ver, proofs, n: int, rank: int = 0, world: int = 1, dist=None):
    """Text in, verdicts out (never `value`): what a caller holding proof.json / proof.wit files sees.
    n texts of this workload's proofs go through ss_stwo_verify_texts -- raw bytes staged into pinned memory, uploaded,
    turned into records by the GPU reader, re-tiled, verified, verdicts downloaded -- timed around the call; beside it the
    same texts through the Python reader (formats.py) for a few proofs, one thread.  Also the host-memory record paths
    (ss_stwo_verify_records / ss_stwo_verify_shared_records).  With world > 1 every rank runs its share of the n inputs
    at the same time (rank-local ingest, SURVEY.md 8e: the ranks share the host's cores and its links): the rates are
    total inputs / the slowest rank's time, and `per_rank_link_GB_s` lists what each rank moved."""
    import numpy as np
    import stark_symphony_amd as ss
    from stark_symphony_amd import binding, distributed, verifier
    cfg = proofs[0].cfg
    distinct = proofs[:8]
    lo, hi = distributed.shard_range(n, rank, world)
    n_local = hi - lo
    # proof.json as the external prover prints it (compact separators, tests/data/proof.json), proof.wit as
    # generate_wit.py prints it, and the shared-path proof.json (every distinct Merkle sibling once; read and expanded
    # on the GPU since round 4)
    texts = {"json": [json.dumps(ss.stwo_to_json(p), separators=(",", ":")).encode() for p in distinct],
             "wit": [ss.stwo_to_wit(p).encode() for p in distinct],
             "json_shared": [json.dumps(ss.stwo_to_json(p, shared=True), separators=(",", ":")).encode() for p in distinct]}
    out = {"proofs": n, "note": "proof text -> verdict through ss_stwo_verify_texts: raw bytes staged into pinned memory, "
                                "uploaded, turned into records by the GPU reader (csrc/ss_textdev.hip), re-tiled, verified; "
                                "bound by the host link, not what `value` measures",
           "parity": "json / wit / records are the forms the reference's adapters emit (generate_wit.py:106-245): pinned as the "
                     "bench line says; json_shared, shared_records, minimal_records, json_minimal have no bytes in the reference: "
                     "unpinned, held to the per-query record they expand to"}
    # which rows a caller on this host should read: with fewer than 8 host threads per rank (every rank of an 8-GPU node on a
    # 16-core grant) the staged entry points cannot feed the link and distributed.files_verifier / use_pinned_inputs pick the
    # caller-pinned ones (the *_pinned rows); a process that owns the host may use either
    out["rows_for_this_host"] = {"ranks": world, "host_threads_per_rank": distributed.host_threads_per_rank(world),
                                 "use": "*_pinned" if distributed.use_pinned_inputs(world) else "staged or *_pinned"}
    if world > 1:
        out["ranks"] = world
        out["host_threads_per_rank"] = int(os.environ.get("SS_HOST_THREADS", "0")) or None

    errors = []

    def timed(call):
        """MEDIAN of five; with several ranks: all start together, the slowest one's time counts.  A failure on this rank
        is recorded, not raised: the other ranks are waiting in the next collective."""
        runs = []
        for _ in range(5):
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            try:
                status, st = call()
                if not (status == 0).all():
                    raise AssertionError("e2e: a benchmark proof was not accepted")
            except Exception as e:  # noqa: BLE001
                errors.append("rank %d: %r" % (rank, e))
                status, st = None, None
            dt = time.perf_counter() - t0
            if st is None and status is None and errors:
                dt = float("inf")
            runs.append((dt, st))
        runs.sort(key=lambda r: r[0])
        return runs[len(runs) // 2]

    def across_ranks(dt: float, link_bytes: int):
        """-> (slowest rank's time, per-rank GB/s on the host link)"""
        if dist is None:
            return dt, [link_bytes / dt / 1e9]
        import torch
        mine = torch.tensor([dt if dt != float("inf") else -1.0, float(link_bytes)], dtype=torch.float64, device=ver.device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rows = [p.cpu().tolist() for p in parts]
        if any(r[0] < 0 for r in rows):
            return float("inf"), [0.0 if r[0] < 0 else r[1] / r[0] / 1e9 for r in rows]
        return max(r[0] for r in rows), [r[1] / r[0] / 1e9 for r in rows]
    for kind, fmt in (("json", binding.TEXT_JSON), ("wit", binding.TEXT_WIT), ("json_shared", binding.TEXT_AUTO)):
        # every text its own buffer (a copy): the staging copy then reads host memory, not eight cache-resident strings
        batch = [texts[kind][(lo + i) % len(distinct)][:1] + texts[kind][(lo + i) % len(distinct)][1:] for i in range(n_local)]
        try:
            ver.verify_stwo_texts(cfg, batch[:64], fmt=fmt)  # warm-up: scratch allocation, templates
            ver.verify_stwo_texts(cfg, batch, fmt=fmt)
        except Exception as e:  # noqa: BLE001  (recorded again by the timed calls)
            errors.append("rank %d warm-up: %r" % (rank, e))
        dt, st = timed(lambda: ver.verify_stwo_texts(cfg, batch, fmt=fmt))
        text_bytes = sum(len(b) for b in batch)
        slowest, links = across_ranks(dt, text_bytes)
        st = st or {"read_s": 0.0, "parse_s": 0.0, "total_s": 1.0, "host_parsed": -1, "threads": 0}
        failed = slowest == float("inf")
        row = {"proofs_per_s": n / slowest, "total_s": None if failed else slowest, "text_GB_per_s": text_bytes * (n / max(n_local, 1)) / slowest / 1e9,
               "stage_s": st["read_s"], "host_reader_s": st["parse_s"],
               "parse_share": st["parse_s"] / st["total_s"], "host_parsed_texts": st["host_parsed"],
               "host_threads": st["threads"], "text_bytes_per_proof": text_bytes // max(n_local, 1)}
        if world > 1:
            row["per_rank_link_GB_s"] = links
        if rank == 0 and kind != "json_shared":
            t1 = time.perf_counter()
            k = 0
            while k < 3 or time.perf_counter() - t1 < 0.5:
                t = texts[kind][k % len(distinct)]
                p = ss.stwo_from_json(json.loads(t), expect=cfg) if kind == "json" else \
                    ss.stwo_from_wit(t.decode(), cfg.trace_log, cfg.pow_bits, cfg.hash)
                verifier.stwo_record(p)
                k += 1
            row["python_reader_proofs_per_s_one_thread"] = k / (time.perf_counter() - t1)
        out[kind] = row
        # the same texts lying in ONE caller-pinned buffer (ss_stwo_verify_texts_pinned): no staging copy, no stager threads
        try:
            blob, boffs, blens = ver.pinned_text_blob(batch)
            ver.verify_stwo_texts_pinned(cfg, blob, boffs, blens, fmt=fmt)
            dt, st = timed(lambda: ver.verify_stwo_texts_pinned(cfg, blob, boffs, blens, fmt=fmt))
            slowest, links = across_ranks(dt, text_bytes)
            prow = {"proofs_per_s": n / slowest, "total_s": None if slowest == float("inf") else slowest,
                    "text_GB_per_s": text_bytes * (n / max(n_local, 1)) / slowest / 1e9,
                    "host_parsed_texts": (st or {}).get("host_parsed", -1), "stage_threads": 0}
            if world > 1:
                prow["per_rank_link_GB_s"] = links
            out[kind + "_pinned"] = prow
            del blob
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d pinned %s: %r" % (rank, kind, e))
        del batch
    # the minimal proof.json (one decommitment per tree, 30 % of the per-query text): the GPU reader finds each text's list
    # lengths from the member names next to the lists, reads it into a capacity-form minimal record and ss_minimal.hip
    # verifies from there (csrc/ss_text.h, ss_textdev.hip); host readers only for texts it does not take
    try:
        from stark_symphony_amd import formats as _f
        mtexts = [json.dumps(_f.stwo_minimal_to_json(_f.stwo_minimise(p)), separators=(",", ":")).encode() for p in distinct]
        batch = [mtexts[(lo + i) % len(distinct)][:1] + mtexts[(lo + i) % len(distinct)][1:] for i in range(n_local)]
        ver.verify_stwo_minimal_texts(cfg, batch[:64])
        dt, st = timed(lambda: ver.verify_stwo_minimal_texts(cfg, batch))
        text_bytes = sum(len(b) for b in batch)
        slowest, links = across_ranks(dt, text_bytes)
        st = st or {"parse_s": 0.0, "total_s": 1.0, "threads": 0}
        row = {"proofs_per_s": n / slowest, "total_s": None if slowest == float("inf") else slowest,
               "text_GB_per_s": text_bytes * (n / max(n_local, 1)) / slowest / 1e9, "host_reader_s": st["parse_s"],
               "parse_share": st["parse_s"] / st["total_s"], "host_parsed_texts": st.get("host_parsed", -1), "host_threads": st["threads"],
               "text_bytes_per_proof": text_bytes // max(n_local, 1)}
        if world > 1:
            row["per_rank_link_GB_s"] = links
        out["json_minimal"] = row
        try:
            blob, boffs, blens = ver.pinned_text_blob(batch)
            ver.verify_stwo_minimal_texts_pinned(cfg, blob, boffs, blens)
            dt, st = timed(lambda: ver.verify_stwo_minimal_texts_pinned(cfg, blob, boffs, blens))
            slowest, links = across_ranks(dt, text_bytes)
            prow = {"proofs_per_s": n / slowest, "total_s": None if slowest == float("inf") else slowest,
                    "text_GB_per_s": text_bytes * (n / max(n_local, 1)) / slowest / 1e9,
                    "host_parsed_texts": (st or {}).get("host_parsed", -1), "stage_threads": 0}
            if world > 1:
                prow["per_rank_link_GB_s"] = links
            out["json_minimal_pinned"] = prow
            del blob
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d pinned minimal texts: %r" % (rank, e))
        del batch
    except Exception as e:  # noqa: BLE001
        errors.append("rank %d minimal texts: %r" % (rank, e))
    # records in host memory -> verdicts: per-query records, shared records (19 % fewer bytes at this shape, expanded by
    # the GPU behind the link) and minimal records (one sorted, deduplicated decommitment per tree: 27 % fewer bytes,
    # verified without an expansion pass -- csrc/ss_minimal.hip; parity unpinned, no such bytes in the reference)
    from stark_symphony_amd import formats
    recs = [verifier.stwo_record(p) for p in distinct]
    shared = [verifier.stwo_shared_record(p) for p in distinct]
    minimal = [verifier.stwo_minimise_record(cfg, r, formats.stwo_queries(p)) for p, r in zip(distinct, recs)]
    for kind, src, call in (("records", recs, ver.verify_stwo_records), ("shared_records", shared, ver.verify_stwo_shared_records),
                            ("minimal_records", minimal, ver.verify_stwo_minimal_records)):
        batch = [src[(lo + i) % len(distinct)].copy() for i in range(n_local)]
        if kind == "records":  # one 2-d array, a record per row (every record still has its own memory)
            batch = np.stack(batch)
        else:                  # the variable-length shared / minimal records back to back + their offsets
            offs = np.zeros(len(batch) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([b.size for b in batch])
            batch = (np.concatenate(batch), offs)
        try:
            call(cfg, batch)
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d warm-up: %r" % (rank, e))
        dt, _ = timed(lambda: (call(cfg, batch), {}))
        nbytes = int(batch.nbytes) if isinstance(batch, np.ndarray) else int(batch[0].nbytes)
        slowest, links = across_ranks(dt, nbytes)
        row = {"proofs_per_s": n / slowest, "total_s": None if slowest == float("inf") else slowest,
               "link_GB_per_s": nbytes * (n / max(n_local, 1)) / slowest / 1e9,
               "bytes_per_proof": nbytes // max(n_local, 1)}
        if world > 1:
            row["per_rank_link_GB_s"] = links
        out[kind] = row
        # the same inputs lying in ONE caller-pinned buffer: no staging copy, no host thread per byte (csrc/ss_pinned.hip) --
        # what a rank of an 8-GPU host, with two of the 16 granted cores, should use
        try:
            flat_src, offs = (batch.reshape(-1), None) if isinstance(batch, np.ndarray) else batch
            pinned = ver.pinned_buffer(flat_src.size)
            pinned[:] = flat_src
            pk = {"records": "records", "shared_records": "shared", "minimal_records": "minimal"}[kind]
            ver.verify_stwo_pinned(cfg, pinned, offs, pk)
            dt, _ = timed(lambda: (ver.verify_stwo_pinned(cfg, pinned, offs, pk), {}))
            slowest, links = across_ranks(dt, nbytes)
            prow = {"proofs_per_s": n / slowest, "total_s": None if slowest == float("inf") else slowest,
                    "link_GB_per_s": nbytes * (n / max(n_local, 1)) / slowest / 1e9, "host_threads": 0}
            if world > 1:
                prow["per_rank_link_GB_s"] = links
            out[kind + "_pinned"] = prow
            del pinned
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d pinned %s: %r" % (rank, kind, e))
        del batch
    # ONE input -> verdict, the reference's calling convention (`simfony run` takes one witness): median of 21 calls, rank 0.
    # Dominated by the transcript kernel's dependent hash chain and the launches, not by bytes.
    if rank == 0:
        try:
            one = {}
            cases = (("json", lambda: ver.verify_stwo_texts(cfg, [texts["json"][0]], fmt=binding.TEXT_JSON)[0]),
                     ("wit", lambda: ver.verify_stwo_texts(cfg, [texts["wit"][0]], fmt=binding.TEXT_WIT)[0]),
                     ("json_minimal", lambda: ver.verify_stwo_minimal_texts(cfg, [mtexts[0]])[0]),
                     ("record", lambda: ver.verify_stwo_records(cfg, recs[:1])),
                     ("minimal_record", lambda: ver.verify_stwo_minimal_records(cfg, minimal[:1])))
            for name, call in cases:
                st = call()
                if not (np.asarray(st) == 0).all():
                    raise AssertionError("single %s: not accepted" % name)
                ts = []
                for _ in range(21):
                    t0 = time.perf_counter()
                    call()
                    ts.append(time.perf_counter() - t0)
                one[name + "_ms"] = sorted(ts)[10] * 1e3
            out["single_input_latency"] = one
        except Exception as e:  # noqa: BLE001
            errors.append("single-input latency: %r" % (e,))
    if errors:
        out["errors"] = errors
    return out


N_SIMDS = 1024               # 256 CUs x 4 SIMDs
NOMINAL_CLOCK_HZ = 2.4e9      # MI355X peak engine clock


def granted_cores() -> int:
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (what the library's
    effective_cpus() computes)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def spawn_ranks(n: int) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment,
    the same contract torch.distributed.run provides), same command line.  Rank 0's stdout -- the
    single JSON line -- is relayed; the exit status is non-zero if any rank's is."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    import tempfile
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    kids, errs = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        errs.append(tempfile.TemporaryFile())  # every rank's stderr is kept: a failing rank must be readable
        kids.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                     stderr=errs[r]))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(kids[0].stdout.read()), daemon=True)
    reader.start()
    bad = []
    while not bad and any(k.poll() is None for k in kids):
        time.sleep(0.2)
        bad = [(r, k.returncode) for r, k in enumerate(kids) if k.poll() not in (None, 0)]
    if bad:  # a dead rank leaves the others waiting in a collective: stop exactly the children started here
        time.sleep(2.0)
        for k in kids:
            if k.poll() is None:
                k.kill()
    for k in kids:
        k.wait()
    reader.join(timeout=10)

    def tail(r: int, limit: int = 4000) -> str:
        f = errs[r]
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(max(0, size - limit))
        return f.read().decode("utf-8", "replace")
    if bad:
        print("bench.py: ranks failed (rank, exit status): %s" % bad, file=sys.stderr)
        for r, code in bad:  # what the rank said before it died (the others were stopped by this launcher)
            print("---- rank %d (exit %s), end of its stderr:\n%s" % (r, code, tail(r)), file=sys.stderr)
        return 1
    sys.stderr.write(tail(0))  # warnings of rank 0 stay visible
    sys.stdout.write(b"".join(out).decode())
    sys.stdout.flush()
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: one --batch-proof batch split over the ranks (BASELINE.json configs[3]); "
                         "weak: --proofs-per-gpu proofs on every rank")
    ap.add_argument("--batch", type=int, default=0,
                    help="proofs of the whole job per step under --scaling strong (default 65536 for the 2^20 "
                         "shape, 32768 for the smaller ones, 4096 for stark101)")
    ap.add_argument("--proofs-per-gpu", type=int, default=0,
                    help="proofs per rank and step; giving it selects --scaling weak")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct", type=int, default=64,
                    help="distinct valid proofs in the batch (made by the GPU prover; 0 = fixtures only)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batch passes in flight (one HIP stream each); 0 = auto: 3, or 4 (with three tail streams) when a rank's "
                         "share is below 32 768 stwo proofs per pass: profiles/r06_overlap_ab.txt")
    ap.add_argument("--e2e", type=int, default=4096,
                    help="proof.json / proof.wit texts for the end-to-end (text -> verdict) figures; 0 = skip")
    ap.add_argument("--tail-streams", type=int, default=0,
                    help="streams the Merkle halves alternate over; 2: the Merkle stage of consecutive passes overlaps "
                         "(+7 %% at 8 192 proofs per GPU, +1.6 %% at 65 536: profiles/r06_overlap_ab.txt) and the kernel "
                         "durations for the roofline come from a separate non-overlapping pass.  0 = auto: 2 for stwo, "
                         "1 for stark101 (whose small batches run on independent streams anyway); 1 = durations measured in "
                         "the timed region itself")
    ap.add_argument("--graph", choices=["auto", "on", "off", "streams"], default="auto",
                    help="small batches (auto: stark101): 'streams' = whole passes on 16 independent streams, "
                         "'on' = hipGraph replay of independent slots; kernel durations for the roofline then "
                         "come from a separate pipelined pass.  'off' = the HEAD/TAIL pipeline")
    ap.add_argument("--streams", type=int, default=16, help="independent streams of --graph streams")
    ap.add_argument("--top-checks", action="store_true",
                    help="SS_FLAG_TOP_CHECKS: the pair memoisation's byte compares run in the top kernel (the round-3 split); "
                         "less HBM traffic, 2.5 %% more time (DESIGN.md 4)")
    ap.add_argument("--no-dedup", action="store_true",
                    help="SS_FLAG_NO_DEDUP: hash every query's Merkle path in full (A/B of the pair memoisation)")
    args = ap.parse_args()

    if args.proofs_per_gpu:
        args.scaling = "weak"
    # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that
    # share a queue serialize.  Must be in the environment before the runtime initialises.  stark101: 16 overlapping
    # passes need their own queues (measured: 44.7 M -> 60.6 M proofs/s).  stwo: the pipeline's head streams must not
    # share a queue with its tail streams, and its two tail streams not with each other, or the overlap they exist for
    # is lost -- which queue a stream gets depends on how many streams the process made before it, so the accept
    # reduce's communication stream alone moved one GPU's 8 192-proof share from 2.32 to 2.55-2.60 ms per step at 4 AND
    # at 16 queues (rocprofv3 Queue_Id: both tail streams on one queue); with 24 every stream has its own: 2.38 ms
    # (profiles/r04_accept_reduce.txt).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed as is: this process becomes the launcher.  It has not
        # imported torch or touched HIP, so starting children is safe; it never verifies anything.
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    # Test hooks (1-GPU boxes): SS_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and SS_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device -- the multi-rank control flow of this
    # file then runs end to end on a single GPU.  Never set by the driver.
    dev_index = 0 if os.environ.get("SS_BENCH_SHARE_GPU") else local
    backend = os.environ.get("SS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    # Test hook: SS_BENCH_GROUP_OF_ONE=1 makes a single rank form its process group too, so that every collective of
    # this file (accept reduce on its stream, barriers, max-over-ranks, the e2e gathers) goes through RCCL on a one-GPU
    # box.  Never set by the driver.
    grouped = world > 1 or os.environ.get("SS_BENCH_GROUP_OF_ONE") == "1"
    if grouped:
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29531")):
            os.environ.setdefault(k_, v_)
        # the ranks of one host share its granted cores: each library instance gets its share for staging / host reading
        os.environ.setdefault("SS_HOST_THREADS", str(max(2, granted_cores() // world)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from stark_symphony_amd import verifier
    wname, family, proofs, note = load_workload(args.workload)
    ver = verifier.Verifier(dev_index)
    if args.no_dedup:
        ver.stwo_flags = verifier.FLAG_NO_DEDUP
    if args.top_checks:  # the memoisation's byte compares in the top kernel instead of lane against lane in the merkle kernel
        ver.stwo_flags = verifier.FLAG_TOP_CHECKS

    if family == "stwo" and args.distinct > len(proofs) and wname != "stwo_fixture":
        # More distinct valid proofs of the same configuration, made on this GPU by the prover
        # of SURVEY.md 8f-1 (different trace seeds; seed 0 must reproduce the committed proof,
        # which the numpy prover made -- a full-size byte-for-byte self-check).
        import stark_symphony_amd as ss
        from stark_symphony_amd import prover
        from stark_symphony_amd import formats
        c = proofs[0].cfg if proofs else formats.StwoConfig(**GEN_ONLY[wname])
        gp = prover.GpuProver(ver)
        made = gp.prove_many([s + rank * args.distinct for s in range(args.distinct)], workers=4, n_cols=c.n_cols,
                             trace_log=c.trace_log, log_blowup=c.log_blowup, n_queries=c.n_queries, pow_bits=c.pow_bits,
                             hash=c.hash)
        if rank == 0 and proofs:
            assert ss.stwo_to_json(made[0]) == ss.stwo_to_json(proofs[0]), "GPU prover != committed proof"
        note += "; %d distinct proofs made by the GPU prover%s" % (
            len(made), " (seed 0 == committed fixture)" if proofs else "")
        proofs = made
        del gp
        torch.cuda.empty_cache()

    if not proofs:
        raise SystemExit("workload %s has no committed proof: needs --distinct >= 1" % wname)
    from stark_symphony_amd import distributed

    def local_share(default_total: int) -> int:
        """Proofs this rank verifies per step.  strong: its shard_range slice of the one batch."""
        if args.scaling == "weak":
            return args.proofs_per_gpu or default_total
        lo, hi = distributed.shard_range(args.batch or default_total, rank, world)
        if hi <= lo:
            raise SystemExit("--batch %d leaves rank %d of %d without a proof" % (args.batch, rank, world))
        return hi - lo

    if family == "stwo":
        cfg = proofs[0].cfg
        # BASELINE.json configs[3] is "batch of 65536 proofs": 11.2 GB of records, which fits one
        # GPU, so that batch is the step at N=1; at N>1 it is split over the ranks (strong) or every
        # rank gets one of its own (weak).
        per_gpu = local_share(65536 if cfg.lde_log >= 24 else 32768)
        reps = (per_gpu + len(proofs) - 1) // len(proofs)
        index = [i % len(proofs) for i in range(per_gpu)]
        batch = verifier.StwoDeviceBatch(ver, cfg, verifier.MODE_FIXTURE, [verifier.stwo_record(p) for p in proofs],
                                         index=index)
        bytes_per_proof, compr_per_proof = cfg.packed_bytes, cfg.compressions
        dominant = "stwo_merkle"  # + "stwo_top": the Merkle stage is these two kernels
        hash_name = cfg.hash
        alu_peak = B2S_CALIBRATED_PEAK if cfg.hash == "blake2s" else SHA_CALIBRATED_PEAK
    else:
        per_gpu = local_share(4096)
        batch = ver.stark101_batch(proofs, replicate=per_gpu)
        bytes_per_proof, compr_per_proof = 7176, 480
        dominant = "s101_merkle"
        hash_name, alu_peak = "sha256", SHA_CALIBRATED_PEAK
    n_local = batch.n
    if args.scaling == "strong":  # the whole job's proofs per step: the one batch
        job_proofs = sum(b - a for a, b in (distributed.shard_range(
            args.batch or (4096 if family != "stwo" else 65536 if cfg.lde_log >= 24 else 32768), r, world)
            for r in range(world)))
    else:
        job_proofs = n_local * world

    # `--inflight` run slots over the same resident batch, pipelined on a head and a tail stream:
    # the latency-bound transcript kernel of step i+1 overlaps the ALU-bound Merkle kernel of
    # step i; Merkle kernels themselves stay serialized on the tail stream.
    # one GPU's share of the batch under strong scaling: a few milliseconds per pass.  Three tail streams and FOUR passes in
    # flight hold both ways it runs (profiles/r06_overlap_ab.txt): alone 3.50 M proofs/s (2 / 3: 3.47; 3 / 6: 3.52), with the
    # per-step accept reduce through RCCL on the slots' head streams 3.48 M (2 / 3: 3.43) -- where SIX passes lose (2.98 M):
    # collectives of one communicator complete in issue order, which chains the head streams to each other.
    small_share = family == "stwo" and n_local < 32768
    if args.inflight == 0:
        args.inflight = 4 if small_share else 3
    nslot = max(1, args.inflight)
    slots = [batch] + [batch.sibling() for _ in range(nslot - 1)]
    pipe = verifier.Pipeline(slots)
    if args.tail_streams == 0:
        # two tail streams: the Merkle stage of pass i + 1 fills the issue slots the top kernel of pass i leaves empty (its
        # half-empty last turns and block barriers: 0.87 of the SHA roof alone) and the drain of every launch.  Measured on the
        # metric batch 3.58 -> 3.64 M proofs/s, at 8 192 proofs per pass 3.24 -> 3.47 M; whole passes on 2-4 independent streams
        # 3.59-3.61 M (profiles/r06_overlap_ab.txt).  Kernel durations measured under that overlap are not the kernels' own, so
        # the roofline's come from a separate pass through a one-slot pipeline (same batch, same kernels, HIP events).
        # (a small share -- a few milliseconds per pass -- gains from a third tail stream and a fourth pass in flight; 65 536 does not)
        args.tail_streams = (3 if small_share else 2) if family == "stwo" else 1
    timed_pipe = verifier.Pipeline(slots, tail_streams=args.tail_streams) if args.tail_streams > 1 else pipe
    accs = [torch.zeros(1, dtype=torch.int32, device=ver.device) for _ in range(nslot)]
    acc = accs[0]
    torch.cuda.synchronize()  # the fills ran on the current stream; the reduces write `accs` on the pipelines' head streams

    def reduce_accepts(k: int) -> None:
        if grouped:  # the path's only exchange: accept-count reduce over xGMI
            accs[k].copy_(slots[k].accept_dev)
            dist.all_reduce(accs[k], op=dist.ReduceOp.SUM)

    # Where the reduce of a pass is enqueued.  "reuse" (default): on the slot's head stream when the slot comes round
    # again -- one pipeline depth later, flushed before the clock stops -- so the exchange needs no stream and no events
    # of its own; "stream": at once, on the pipeline's communication stream (Pipeline.submit).
    lazy = os.environ.get("SS_BENCH_REDUCE", "reuse") != "stream"

    def step(i: int, p=None) -> None:
        if not grouped:
            (p or pipe).submit()
        elif lazy:
            (p or pipe).submit(on_reuse=reduce_accepts)
        else:
            (p or pipe).submit(reduce_accepts)

    def flush(p=None) -> None:
        if grouped and lazy:
            (p or pipe).flush(reduce_accepts)

    def own_durations():
        """The kernels' OWN durations for the roofline, where the timed region overlaps launches (two or more tail streams,
        independent streams, graph replays): up to 20 passes through a ONE-slot pipeline -- HEAD, then TAIL, then the next
        pass -- so that nothing shares the chip with a kernel while HIP events on its stream time it."""
        iso = verifier.Pipeline(slots[:1])
        ver.set_timing(True)
        ver.collect_timing()
        for i in range(min(args.steps, 20)):
            step(i, iso)
        flush(iso)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    flush()
    torch.cuda.synchronize()
    if args.warmup:
        assert batch.accepted() == n_local, "benchmark proofs must all be ACCEPT (%d of %d)" % (
            batch.accepted(), n_local)
    small = family == "stark101" and not grouped
    streams = args.graph == "streams" or (args.graph == "auto" and small)
    graphed = args.graph == "on"
    if streams:
        # Too small to fill the chip alone, short enough that submission order matters: 16 slots, each
        # on its own stream, whole passes back to back (verifier.IndependentStreams).
        islots = [batch.sibling() for _ in range(max(1, args.streams))]
        ind = verifier.IndependentStreams(islots)
        for _ in range(2 * len(islots)):
            ind.submit()
        ind.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ind.submit()
        ind.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert all(s.accepted() == n_local for s in islots), "independent streams: not every proof accepted"
        graphed = True  # (for the reporting below: per-kernel durations come from the eager pass)
        own_durations()
        timing = ver.collect_timing()
        ver.set_timing(False)
    elif graphed:
        # A pass over a few thousand stark101 proofs is shorter than the ~10 launches and event
        # operations that enqueue it, and too small to fill the chip: the K timed steps are replayed
        # as hipGraphs of S independent slots each (S | K), two graphs alternating.  Events cannot be
        # captured, so the per-kernel durations of the roofline come from an eager pass afterwards.
        S = max(d for d in range(1, 9) if args.steps % d == 0)
        gslots = [batch.sibling() for _ in range(2 * S)]
        graphs = [verifier.GraphedPipeline(gslots[:S], concurrent_tails=True),
                  verifier.GraphedPipeline(gslots[S:], concurrent_tails=True)]
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(args.steps // S):
            graphs[r & 1].replay()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert all(s.accepted() == n_local for s in gslots[:S]), "graphed pass: not every proof accepted"
        own_durations()
        timing = ver.collect_timing()
        ver.set_timing(False)
    else:
        overlapped = timed_pipe is not pipe
        if not overlapped:
            ver.set_timing(True)
            ver.collect_timing()
        else:
            for i in range(2 * nslot):
                step(i, timed_pipe)
            flush(timed_pipe)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, timed_pipe)
        flush(timed_pipe)  # (the reduces of the last passes: inside the timed region)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if overlapped:  # the Merkle stages of consecutive passes overlap: their durations come from a pass where nothing does
            own_durations()
        timing = ver.collect_timing()
        ver.set_timing(False)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ver.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        total_accept = int(acc.item())
    else:
        total_accept = batch.accepted()
    assert total_accept == job_proofs, "accept-reduce mismatch (%d of %d)" % (total_accept, job_proofs)

    # text / records in host memory -> verdicts, every rank its share at the same time (never `value`)
    e2e = end_to_end(ver, proofs, args.e2e, rank, world, dist if grouped else None) if family == "stwo" and args.e2e > 0 else None
    if rank == 0:
        total = job_proofs * args.steps
        value = total / elapsed
        k_ms, k_n = timing.get(dominant, (0.0, 0))
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        executed = compr_per_proof
        if family == "stwo":
            # Merkle stage = stwo_merkle (every chain up to the shared levels) + stwo_top (each distinct
            # pair of the shared levels once): one duration, the sum of the two kernels' averages.
            t_ms, t_n = timing.get("stwo_top", (0.0, 0))
            k_avg_s += (t_ms / max(t_n, 1)) * 1e-3
            if t_n:
                dominant = "stwo_merkle+stwo_top"
            # compressions the kernels really execute: the reference's count minus the pairs memoised,
            # exact for this batch (from the queries of its distinct proofs)
            per_node = 1 if cfg.hash == "blake2s" else 2
            saved = []
            for i in range(len(proofs)):
                full, done = verifier.merkle_node_hashes(cfg, batch.intermediates(i)["queries"],
                                                         0 if args.no_dedup else None)
                saved.append((full - done) * per_node)
            executed = compr_per_proof - sum(saved) / len(saved)
        launch_bytes = bytes_per_proof * n_local
        traffic, valu_insts, traffic_src = pmc_profile(wname, n_local)
        achieved = launch_bytes / k_avg_s / 1e9 if k_avg_s else 0.0
        compr_s = executed * n_local / k_avg_s if k_avg_s else 0.0
        alu_note = "compressions the Merkle stage executes per launch / its kernel time"
        if graphed:  # slots overlap inside a graph: the meaningful rate is the whole job's
            compr_s = executed * value / world
            alu_note = "compressions executed per proof x proofs/s (graphed, overlapping slots: whole-job rate)"
        out = {
            "metric": "proofs verified/sec (batch), stwo 2^20-domain circle-STARK"
                      if family == "stwo" else "proofs verified/sec (batch), stark101",
            "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": wname, "note": note, "proofs_per_gpu": n_local,
                       "proofs_per_step": job_proofs,
                       "batch": ("one batch of %d proofs split over %d GPU(s) (distributed.shard_range), %d on rank 0"
                                 % (job_proofs, world, n_local)) if args.scaling == "strong" else
                                ("%d proofs on each of %d GPU(s)" % (n_local, world)),
                       "distinct_proofs": len(proofs), "bytes_per_proof": bytes_per_proof,
                       "hash_compressions_per_proof": compr_per_proof,
                       "hash_compressions_executed_per_proof": executed,
                       "pair_memoisation": family == "stwo" and not args.no_dedup, "hash": hash_name,
                       # what "pinned" covers: leaf functions by the reference's 86 fn test_* KATs (oracle and
                       # device); end to end stark101 by the reference prover's proof, stwo by the reference's two shipped
                       # proofs -- which only FIXTURE mode accepts (the .simf text rejects its own proofs)
                       "parity": "unpinned (Blake2s is not in the reference: RFC 7693 vectors + prover / oracle / GPU agreement)" if hash_name == "blake2s" else
                                 "pinned: SHA-256, stark101 verify_proof (86 reference KATs + the reference prover's proof)" if family != "stwo" else
                                 "pinned: SHA-256, FIXTURE mode (86 reference KATs + 3 reference proofs); LITERAL mode by the .simf text only",
                       "mode": "fixture_correct", "inflight_streams": nslot,
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "GPU_MAX_HW_QUEUES_matters_for": "value, ms_per_step and kernels_ms_per_step of every pipelined / multi-stream "
                                                        "submission (each stream its own hardware queue; with the runtime's 4 the "
                                                        "8 192-proof share loses up to 10 % and stark101 x 4 096 half its rate: "
                                                        "profiles/r04_queue_robustness.txt); e2e, cpu_baseline and the status words do not depend on it",
                       "submission": "%d independent streams, whole passes" % args.streams if streams else
                                     "hipGraph replay, independent slots" if graphed else
                                     "eager, HEAD/TAIL pipelined" + (", Merkle halves alternating over %d streams (kernel "
                                                                     "durations from a separate non-overlapping pass)"
                                                                     % args.tail_streams if args.tail_streams > 1 else ""),
                       "parallelism": "proofs sharded over %d GPU(s)" % world,
                       "accept_reduce": None if not grouped else
                       ("all-reduce(SUM) of every step's accept count over %s, " % backend) +
                       ("enqueued on the slot's head stream when the slot is used again (one pipeline depth later), the last ones "
                        "flushed inside the timed region" if lazy else "at once on the pipeline's communication stream")},
            "hbm_gb_s": value * bytes_per_proof / 1e9,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_avg_ms": k_avg_s * 1e3, "kernel_launches": k_n,
                         # the rate the timed region itself sustains: one launch of the stage completes every ms_per_step
                         # (below the sum of the two kernels' own durations when consecutive launches overlap)
                         "launch_period_ms": elapsed / args.steps * 1e3,
                         "achieved_per_launch_period": launch_bytes / (elapsed / args.steps) / 1e9,
                         "durations_from": ("up to 20 passes through a one-slot pipeline after the timed region (HIP events on the launch "
                                            "stream; nothing else on the chip): launches of the timed region overlap, a kernel's duration "
                                            "there is not its own")
                         if (streams or graphed or (family == "stwo" and args.tail_streams > 1)) else
                         "HIP events on the launch stream, in the timed region",
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "note": "integer-ALU bound by construction (2 SHA-256 / 1 Blake2s compression "
                                 "per 32-byte sibling); see alu_roofline"},
            "alu_roofline": {"hash_compressions_per_s": compr_s,
                             "reference_equivalent_compressions_per_s":
                                 compr_per_proof * n_local / k_avg_s if k_avg_s else 0.0,
                             "calibrated_peak_compressions_per_s": alu_peak,
                             "frac": compr_s / alu_peak,
                             # the same over the WHOLE step (HEAD kernels, launch edges and all): compressions a pass executes
                             # / ms_per_step -- what the timed region itself sustains against the calibrated roof
                             "frac_of_step": executed * n_local / (elapsed / args.steps) / alu_peak,
                             # the absolute figure beside the calibrated one: VALU wave-instructions the Merkle stage issues
                             # (SQ_INSTS_VALU of the committed counter passes) x 2 cycles -- what a wave64 VALU operation
                             # occupies a SIMD for (MI355X_MICROARCH.md) -- over SIMD-cycles at the nominal clock
                             "issue_frac": (valu_insts * 2 / (N_SIMDS * NOMINAL_CLOCK_HZ * k_avg_s)) if valu_insts and k_avg_s else None,
                             "issue_note": "SQ_INSTS_VALU x 2 cycles / (%d SIMDs x %.1f GHz x kernel time); the rotates and 3-input "
                                           "adds SHA-256 is made of issue at half that rate on this chip (profiles/r01_valu_mixing.txt, "
                                           "r03_sha_formulations.txt), which is why `frac` is taken against a measured calibration"
                                           % (N_SIMDS, NOMINAL_CLOCK_HZ / 1e9),
                             "note": alu_note + "; peak = tools/sha_bench.hip (registers only) on MI355X, "
                                     "profiles/r01_sha_calibration.txt (34.7 G), re-measured in r03_sha_calibration.txt (34.4-34.6 G)"},
            "kernels_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in timing.items()},
        }
        if e2e is not None:
            out["e2e"] = e2e
        elif family == "stark101" and args.e2e > 0 and world == 1:
            out["e2e"] = end_to_end_s101(ver, proofs[0], max(args.e2e, 16384))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(family, proofs, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
