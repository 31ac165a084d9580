#!/usr/bin/env python3
"""The minimal proof.json (one sorted, deduplicated decommitment per tree: formats.stwo_minimal_to_json) of the reference's
two proofs, as THIS repository's writer prints it -- tests/golden/formats/stwo_proof*.minimal.json.  No bytes of that form
exist in the reference (parity unpinned: upstream stwo's prover, which emits it, is not in /root/reference); the files pin
the form against drift between rounds: the Python writer, the native writer and the readers are held to them
(tests/test_minimal.py), and the verifiers accept them.  Needs nothing but this repository.

    python tests/golden/make_minimal_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import formats  # noqa: E402

for src in ("stwo_proof.json", "stwo_proof_test.json"):
    with open(os.path.join(HERE, src)) as f:
        p = ss.stwo_from_json(json.load(f))
    text = json.dumps(formats.stwo_minimal_to_json(formats.stwo_minimise(p)), separators=(",", ":"))
    dst = os.path.join(HERE, "formats", src.replace(".json", ".minimal.json"))
    with open(dst, "w") as f:
        f.write(text)
    print("%-32s %7d bytes (per-query proof.json: %d)" % (os.path.basename(dst), len(text), os.path.getsize(os.path.join(HERE, src))))
