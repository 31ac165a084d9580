"""`simfony run`-compatible command line (exit-status contract of simfony-cli/src/main.rs:254-257).

    python -m stark_symphony_amd.cli verify --family stark101 --witness target/proof.wit
    python -m stark_symphony_amd.cli verify --family stwo --proof tests/data/proof.json
    python -m stark_symphony_amd.cli verify --family stwo --witness a.wit b.wit --trace-log 9
    python -m stark_symphony_amd.cli convert --family stwo --to wit tests/data/proof.json     # generate_wit.py
    python -m stark_symphony_amd.cli convert --family stwo --to simf tests/data/proof.json    # generate_simf.py

Exit 0 when every input is ACCEPTed, 1 otherwise (REJECT or malformed witness, like the
reference, whose type errors also end in exit 1: main.rs:77-81,187-190).  Runs on GPU 0.
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
                   "json": lambda q: json.dumps(formats.stwo_to_json(q))}[args.to](p)
    except (formats.MalformedProof, OSError, ValueError) as e:
        print("Error: %s" % e, file=sys.stderr)
        return 1
    print(out)
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="stark_symphony_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    v = sub.add_parser("verify", help="verify witnesses / proofs on the GPU")
    v.add_argument("--family", choices=["stark101", "stwo"], required=True)
    v.add_argument("--witness", nargs="*", default=[], help=".wit files (formats B / D)")
    v.add_argument("--proof", nargs="*", default=[], help="proof.json files (formats A / C)")
    v.add_argument("--trace-log", type=int, default=None,
                   help="TRACE_LOG_SIZE for stwo .wit files (config.simf:17,35)")
    v.add_argument("--pow-bits", type=int, default=5)
    v.add_argument("--mode", choices=["fixture", "literal"], default="fixture")
    v.add_argument("--device", type=int, default=0)
    c = sub.add_parser("convert", help="proof.json -> .wit / .simf snippet (the reference's "
                                       "scripts/generate_wit.py and generate_simf.py, same text), or back")
    c.add_argument("--family", choices=["stark101", "stwo"], required=True)
    c.add_argument("--to", choices=["wit", "simf", "json"], required=True)
    c.add_argument("path", help="proof.json, .wit or .simf snippet (by extension; anything else = json)")
    c.add_argument("--trace-log", type=int, default=None)
    c.add_argument("--pow-bits", type=int, default=5)
    args = ap.parse_args(argv)
    if args.cmd == "convert":
        return convert(args)

    proofs, names = [], []
    try:
        for path in args.witness:
            text = open(path).read()
            if args.family == "stark101":
                proofs.append(formats.stark101_from_wit(text))
            else:
                if args.trace_log is None:
                    raise formats.MalformedProof("--trace-log is required for stwo .wit files")
                proofs.append(formats.stwo_from_wit(text, args.trace_log, args.pow_bits))
            names.append(path)
        for path in args.proof:
            obj = json.load(open(path))
            proofs.append(formats.stark101_from_json(obj) if args.family == "stark101"
                          else formats.stwo_from_json(obj, args.trace_log))
            names.append(path)
    except (formats.MalformedProof, OSError, ValueError) as e:
        print("Error: %s" % e, file=sys.stderr)
        return 1
    if not proofs:
        print("Error: nothing to verify", file=sys.stderr)
        return 1
    ver = verifier.Verifier(args.device)
    if args.family == "stark101":
        status = ver.verify_stark101(proofs)
    else:
        mode = verifier.MODE_FIXTURE if args.mode == "fixture" else verifier.MODE_LITERAL
        status = ver.verify_stwo(proofs, mode)
    bad = 0
    for name, st in zip(names, status.tolist()):
        if st == 0:
            print("%s: ACCEPT" % name)
        else:
            bad += 1
            print("Error: Failed to run program: %s: REJECT (first failing assert 0x%08x)" % (name, st),
                  file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
