#!/usr/bin/env python3
"""Runs the reference's own format adapters (stark101/scripts/generate_{wit,simf}.py,
stwo-verifier/scripts/generate_{wit,simf}.py) on the committed proof JSONs and stores what they
print under tests/golden/formats/.  Needs /root/reference (this container only); the outputs are
data fixtures: tests compare our writers with them byte for byte and parse them back.

    python tests/golden/make_format_golden.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SS_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "formats")
os.makedirs(OUT, exist_ok=True)

JOBS = [
    ("stark101/scripts/generate_wit.py", "stark101_proof.json", "stark101_proof.wit"),
    ("stark101/scripts/generate_simf.py", "stark101_proof.json", "stark101_proof.simf.txt"),
    ("stwo-verifier/scripts/generate_wit.py", "stwo_proof.json", "stwo_proof.wit"),
    ("stwo-verifier/scripts/generate_simf.py", "stwo_proof.json", "stwo_proof.simf.txt"),
    ("stwo-verifier/scripts/generate_wit.py", "stwo_proof_test.json", "stwo_proof_test.wit"),
    ("stwo-verifier/scripts/generate_simf.py", "stwo_proof_test.json", "stwo_proof_test.simf.txt"),
]
for script, src, dst in JOBS:
    res = subprocess.run([sys.executable, os.path.join(REF, script), os.path.join(HERE, src)],
                         check=True, capture_output=True)
    with open(os.path.join(OUT, dst), "wb") as f:
        f.write(res.stdout)
    print("%-28s %7d bytes" % (dst, len(res.stdout)))
