#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's own unit tests into kats.json.

Runs only in the build container (reads /root/reference); never on the GPU box.
For every `fn test_*` in stark101/src/*.simf and stwo-verifier/src/**/*.simf
(the 86 tests scripts/unit_tests.sh:27-108 runs) it records the ordered list of
integer literals that appear in the test body (comments stripped, identifiers such
as `u32`, `eq_256`, `qm31` excluded).  kats.json holds DATA only -- inputs and
expected outputs -- keyed by "<file>::<test>" with the line number of the test;
tests/test_oracle_kats.py (CPU oracle) and tests/test_gpu_kats.py (device) give every position its meaning.
"""
import json, os, re, sys

REF = os.environ.get("SS_REFERENCE", "/root/reference")
NUM = re.compile(r"(?<![A-Za-z_0-9])(0x[0-9a-fA-F]+|\d+)(?![A-Za-z_0-9])")
out = {}
for sub in ("stark101/src", "stwo-verifier/src"):
    for root, _, files in os.walk(os.path.join(REF, sub)):
        for fn in sorted(files):
            if not fn.endswith(".simf"):
                continue
            path = os.path.join(root, fn)
            rel = os.path.relpath(path, REF)
            lines = open(path).read().split("\n")
            i = 0
            while i < len(lines):
                m = re.match(r"fn (test_\w+)\(", lines[i])
                if not m:
                    i += 1
                    continue
                name, start = m.group(1), i
                depth, body = 0, []
                while True:
                    ln = lines[i].split("//")[0]
                    body.append(ln)
                    depth += ln.count("{") - ln.count("}")
                    i += 1
                    if depth == 0:
                        break
                text = "\n".join(body[1:])  # skip the signature line
                vals = [int(x, 16) if x.startswith("0x") else int(x) for x in NUM.findall(text)]
                out["%s::%s" % (rel, name)] = {"line": start + 1, "values": vals}
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "kats.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote %d tests" % len(out))
