"""`simfony run`-compatible command line (exit-status contract of simfony-cli/src/main.rs:254-257).

    python -m stark_symphony_amd.cli verify --family stark101 --witness target/proof.wit
    python -m stark_symphony_amd.cli verify --family stwo --proof tests/data/proof.json
    python -m stark_symphony_amd.cli verify --family stwo --witness a.wit b.wit            # config.simf, production
    python -m stark_symphony_amd.cli verify --family stwo --config testing --witness t.wit  # config.simf -DTESTING
    python -m stark_symphony_amd.cli verify --family stwo --trace-log 20 --lde-log 24 --n-layers 19 --proof p.json
    python -m stark_symphony_amd.cli convert --family stwo --to wit tests/data/proof.json     # generate_wit.py
    python -m stark_symphony_amd.cli convert --family stwo --to simf tests/data/proof.json    # generate_simf.py
    python -m stark_symphony_amd.cli convert --family stwo --to json-shared tests/data/proof.json   # shared Merkle paths
    python -m stark_symphony_amd.cli convert --family stwo --to json-minimal tests/data/proof.json  # one decommitment per tree
    python -m stark_symphony_amd.cli verify --family stwo --minimal-proof minimal.json               # ... verified as such
    python -m stark_symphony_amd.cli prove --family stark101 --out target/proof.json      # `make proof` (python -m fibsquare)
    python -m stark_symphony_amd.cli prove --family stwo --trace-log 20 --seed 0 --to wit --out target/proof.wit

Exit 0 when every input is ACCEPTed, 1 otherwise (REJECT or malformed witness, like the
reference, whose type errors also end in exit 1: main.rs:77-81,187-190).  Runs on GPU 0.

The stwo parameters (columns, sizes, queries, FRI layers, PoW bits, hash) are the VERIFIER's: the
reference compiles them in (stwo-verifier/src/config.simf:10-51), this command takes them from
`--config production|testing` plus explicit overrides -- never from the proof.  A proof that
declares or has any other shape is rejected like a witness that fails typing.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

from . import formats, verifier


def convert(args) -> int:
    """Format adapters without a GPU: prints what the reference's generate_*.py print."""
    try:
        text = open(args.path).read()
        kind = "wit" if args.path.endswith(".wit") else "simf" if ".simf" in os.path.basename(args.path) else "json"
        if args.family == "stark101":
            p = {"wit": formats.stark101_from_wit, "simf": formats.stark101_from_simf,
                 "json": formats.stark101_from_json}[kind](text)
            out = {"wit": formats.stark101_to_wit, "simf": formats.stark101_to_simf,
                   "json": lambda q: json.dumps(formats.stark101_to_json(q))}[args.to](p)
        else:
            if kind == "json":
                p = formats.stwo_from_json(text, args.trace_log)
            else:
                if args.trace_log is None:
                    raise formats.MalformedProof("--trace-log is required for stwo .wit / .simf input")
                reader = formats.stwo_from_wit if kind == "wit" else formats.stwo_from_simf
                p = reader(text, args.trace_log, args.pow_bits)
            out = {"wit": formats.stwo_to_wit, "simf": formats.stwo_to_simf,
                   "json": lambda q: json.dumps(formats.stwo_to_json(q)),
                   # every distinct Merkle sibling once + the query positions (formats.shared_path_order)
                   "json-shared": lambda q: json.dumps(formats.stwo_to_json(q, shared=True)),
                   # one sorted, deduplicated decommitment per tree, as upstream stwo's prover sends it (formats.minimal_order)
                   "json-minimal": lambda q: json.dumps(formats.stwo_minimal_to_json(formats.stwo_minimise(q)))}[args.to](p)
    except (formats.MalformedProof, OSError, ValueError) as e:
        print("Error: %s" % e, file=sys.stderr)
        return 1
    print(out)
    return 0


def prove(args) -> int:
    """`make proof` on the GPU.  stark101: what `cd scripts && python -m fibsquare` writes to target/proof.json
    (stark101/Makefile:14-17; only the reference seed satisfies the verifier's hard-coded boundary value); stwo: a
    wide-Fibonacci proof in the proof.json schema of tests/data/proof.json (the reference ships two proofs and no
    prover; seed 0 with the reference's sizes reproduces them byte for byte).  --to wit / simf prints what
    generate_wit.py / generate_simf.py make of it."""
    from . import binding
    try:
        ver = verifier.Verifier(args.device)
        if args.family == "stark101":
            from . import prover101
            gp = prover101.Stark101GpuProver(ver)
            res = gp.prove() if args.seed is None else gp.prove(seed=args.seed)
            p = formats.stark101_from_json(res)
            out = {"json": lambda: json.dumps(res), "wit": lambda: formats.stark101_to_wit(p),
                   "simf": lambda: formats.stark101_to_simf(p)}[args.to]()
        else:
            from . import prover as stwo_prover
            gp = stwo_prover.GpuProver(ver)
            proof = gp.prove_proof(n_cols=args.n_cols, trace_log=args.trace_log, log_blowup=args.log_blowup,
                                   n_queries=args.n_queries, pow_bits=args.pow_bits, seed=args.seed or 0, hash=args.hash)
            out = {"json": lambda: json.dumps(formats.stwo_to_json(proof)), "wit": lambda: formats.stwo_to_wit(proof),
                   "simf": lambda: formats.stwo_to_simf(proof),
                   "json-minimal": lambda: json.dumps(formats.stwo_minimal_to_json(formats.stwo_minimise(proof, gp.queries)))}[args.to]()
    except (binding.SsError, ValueError) as e:
        print("Error: %s" % e, file=sys.stderr)
        return 1
    if args.out:
        with open(args.out, "w") as f:
            f.write(out + "\n")
    else:
        print(out)
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="stark_symphony_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    v = sub.add_parser("verify", help="verify witnesses / proofs on the GPU")
    v.add_argument("--family", choices=["stark101", "stwo"], required=True)
    v.add_argument("--witness", nargs="*", default=[], help=".wit files (formats B / D)")
    v.add_argument("--proof", nargs="*", default=[], help="proof.json files (formats A / C)")
    v.add_argument("--minimal-proof", nargs="*", default=[], help="stwo: minimal proof.json files (one sorted, deduplicated "
                                                                  "decommitment per tree, as upstream stwo's prover sends it)")
    v.add_argument("--config", choices=["production", "testing"], default="production",
                   help="stwo: the config.simf profile the verifier enforces (default: production, "
                        "i.e. the reference built without -DTESTING)")
    v.add_argument("--n-cols", type=int, default=None, help="override NUM_COLUMNS")
    v.add_argument("--trace-log", type=int, default=None, help="override TRACE_LOG_SIZE")
    v.add_argument("--lde-log", type=int, default=None, help="override LDE_LOG_SIZE")
    v.add_argument("--n-queries", type=int, default=None, help="override NUM_FRI_QUERIES")
    v.add_argument("--n-layers", type=int, default=None, help="override NUM_FRI_LAYERS")
    v.add_argument("--pow-bits", type=int, default=None, help="override the PoW bits of POW_TARGET_64")
    v.add_argument("--hash", choices=["sha256", "blake2s"], default=None, help="override the hash family")
    v.add_argument("--mode", choices=["fixture", "literal"], default="fixture")
    v.add_argument("--device", type=int, default=0)
    c = sub.add_parser("convert", help="proof.json -> .wit / .simf snippet (the reference's "
                                       "scripts/generate_wit.py and generate_simf.py, same text), or back")
    c.add_argument("--family", choices=["stark101", "stwo"], required=True)
    c.add_argument("--to", choices=["wit", "simf", "json", "json-shared", "json-minimal"], required=True)
    c.add_argument("path", help="proof.json, .wit or .simf snippet (by extension; anything else = json)")
    c.add_argument("--trace-log", type=int, default=None)
    c.add_argument("--pow-bits", type=int, default=5)
    g = sub.add_parser("prove", help="make a proof on the GPU (`make proof`: python -m fibsquare / the stwo prover fork)")
    g.add_argument("--family", choices=["stark101", "stwo"], required=True)
    g.add_argument("--to", choices=["json", "wit", "simf", "json-minimal"], default="json")
    g.add_argument("--out", default=None, help="file to write (default: stdout)")
    g.add_argument("--seed", type=int, default=None, help="stark101: a_1 of the trace (default: the reference's); stwo: r of "
                                                          "the rows [1, r, ...] (default 0 = the reference's proofs)")
    g.add_argument("--n-cols", type=int, default=4)
    g.add_argument("--trace-log", type=int, default=9, help="stwo: log2 of the trace rows (9 = tests/data/proof.json)")
    g.add_argument("--log-blowup", type=int, default=4)
    g.add_argument("--n-queries", type=int, default=16)
    g.add_argument("--pow-bits", type=int, default=5)
    g.add_argument("--hash", choices=["sha256", "blake2s"], default="sha256")
    g.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    if args.cmd == "convert":
        return convert(args)
    if args.cmd == "prove":
        return prove(args)

    import dataclasses
    from . import binding
    expected = formats.PRODUCTION_CONFIG if args.config == "production" else formats.TESTING_CONFIG
    over = {k: getattr(args, k) for k in ("n_cols", "trace_log", "lde_log", "n_queries", "n_layers",
                                          "pow_bits", "hash") if getattr(args, k) is not None}
    expected = dataclasses.replace(expected, **over)
    if not args.witness and not args.proof and not args.minimal_proof:
        print("Error: nothing to verify", file=sys.stderr)
        return 1
    if args.minimal_proof and args.family != "stwo":
        print("Error: --minimal-proof is an stwo form", file=sys.stderr)
        return 1
    # The files go to the library as they are: its native readers (csrc/ss_ingest.cpp) parse them on
    # host threads into the upload staging; nothing is parsed in Python on this path.
    try:
        ver = verifier.Verifier(args.device)
        names, status = [], []
        for paths, fmt in ((args.witness, binding.TEXT_WIT), (args.proof, binding.TEXT_JSON)):
            if not paths:
                continue
            if args.family == "stark101":
                st, _ = ver.verify_stark101_files(paths, fmt)
            else:
                mode = verifier.MODE_FIXTURE if args.mode == "fixture" else verifier.MODE_LITERAL
                st, _ = ver.verify_stwo_files(expected, paths, mode, fmt)
            names += list(paths)
            status += st.tolist()
        if args.minimal_proof:
            mode = verifier.MODE_FIXTURE if args.mode == "fixture" else verifier.MODE_LITERAL
            # (files the library reads itself, like the other forms; unreadable = not a witness: malformed, exit 1, main.rs:77-81)
            st, _ = ver.verify_stwo_files(expected, args.minimal_proof, mode, binding.TEXT_JSON_MINIMAL)
            names += list(args.minimal_proof)
            status += st.tolist()
    except binding.SsError as e:  # no GPU / unsupported config: an error, never a verdict
        print("Error: %s" % e, file=sys.stderr)
        return 1
    bad = 0
    for name, st in zip(names, status):
        if st == 0:
            print("%s: ACCEPT" % name)
            continue
        bad += 1
        if st == binding.STATUS_MALFORMED:  # main.rs:77-81,187-190: the witness does not type-check
            print("Error: %s: malformed witness (not a value of the program's witness types)" % name,
                  file=sys.stderr)
        else:
            why = ("witness does not have the shape of the expected config" if st == binding.STATUS_CONFIG_MISMATCH
                   else "first failing assert 0x%08x" % st)
            print("Error: Failed to run program: %s: REJECT (%s)" % (name, why), file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
