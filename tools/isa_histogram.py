#!/usr/bin/env python3
"""Instruction histogram of a gfx950 kernel, whole body and innermost loop, from hipcc's assembly.

    python tools/isa_histogram.py [kernel-substring ...]     (default: the two Merkle kernels)

Compiles stark-symphony_amd/csrc/ss_stwo.hip with `--cuda-device-only -S` (no GPU needed) and counts
mnemonics.  The "loop" of the Merkle kernels is one sibling level of 64 chains: the basic blocks
between the backward branch's target label and the branch.  DESIGN.md section 4 quotes these counts."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "stark-symphony_amd", "csrc", "ss_stwo.hip")


def assembly(extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                        "--cuda-device-only", "-S", "-o", out, SRC, *extra], check=True, stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def kernels(lines):
    """name -> list of (label or None, mnemonic) in order"""
    out, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if re.match(r"^\.Lfunc_end", ln):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur.append((m.group(1), None, None))
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(?:\s*;.*)?$", ln)
        if m and not m.group(1).startswith("."):
            cur.append((None, m.group(1), m.group(2)))
    return out


def innermost_loop(body):
    """instructions of the largest block range closed by a backward branch"""
    pos = {lab: i for i, (lab, _, _) in enumerate(body) if lab}
    best = []
    for i, (lab, op, args) in enumerate(body):
        if op and op.startswith("s_cbranch") and args.strip() in pos and pos[args.strip()] < i:
            seg = [x for x in body[pos[args.strip()]:i + 1] if x[1]]
            if len(seg) > len(best):
                best = seg
    return best


def report(name, body):
    ops = [x for x in body if x[1]]
    loop = innermost_loop(body)
    print("== %s: %d instructions, largest loop body %d" % (name, len(ops), len(loop)))
    for title, seq in (("whole kernel", ops), ("loop body (one sibling level)", loop)):
        h = collections.Counter(op for _, op, _ in seq)
        valu = sum(c for op, c in h.items() if op.startswith("v_"))
        print("  -- %s: %d VALU, %d SALU, %d memory/other" % (
            title, valu, sum(c for op, c in h.items() if op.startswith("s_")),
            sum(c for op, c in h.items() if not op.startswith(("v_", "s_")))))
        for op, c in h.most_common(14):
            print("     %-24s %5d" % (op, c))


if __name__ == "__main__":
    want = sys.argv[1:] or ["stwo_merkle_kernel_sha", "stwo_merkle_kernel_b2s"]
    ks = kernels(assembly())
    for name, body in ks.items():
        if any(w in name for w in want):
            report(name, body)
