#!/usr/bin/env python3
"""Differential fuzzing of the GPU text reader (run on the GPU box).

    python tools/text_fuzz.py [mutants_per_case] [seed]

For several configurations and both text formats it mutates canonical texts (byte flips, deletions, duplications,
truncations, injected digits / brackets / escapes, numbers replaced by other spellings and values, blanks shifted
against the window grid) and requires, for every mutant:
    GPU outcome == scalar rule (ss_stwo_text_is_canonical);
    where taken: GPU record == scalar rule's record == host reader's record (and the host reader parses it);
and, through the whole entry point, status words == host reader + record path on a sample.  The minimal proof.json gets
the same treatment plus mutants whose lists have other lengths (its lengths are data), and the streaming host reader is held
to the general one on every mutant it takes.
Exit status 0 = no disagreement."""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import stark_symphony_amd as ss  # noqa: E402
from stark_symphony_amd import binding, records, verifier  # noqa: E402
from test_ingest import _text_mutant  # noqa: E402
from test_text_fastpath import _number_mutant, canonical, write_text, s101_canonical  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 20261007
GOLDEN = os.path.join(ROOT, "tests", "golden")
JSON, WIT, SHARED = binding.TEXT_JSON, binding.TEXT_WIT, binding.TEXT_JSON_SHARED


HOST_FMT = {JSON: JSON, WIT: WIT, SHARED: JSON}  # the host reader knows the shared form by its "queries" member


def cases():
    yield "reference test config", ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))
    yield "reference production", ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    for npz in ("stwo_trace16.npz", "stwo_trace16_blake2s.npz", "stwo_wide256.npz"):
        yield npz, records.load_stwo_npz(os.path.join(GOLDEN, npz))[0]


def minimal_leg(ver, rnd, name, p):
    from stark_symphony_amd import formats
    MIN = binding.TEXT_JSON_MINIMAL
    cfg = p.cfg
    m = formats.stwo_minimise(p)
    rec = verifier.stwo_minimal_record(m)
    obj = formats.stwo_minimal_to_json(m)
    base = (json.dumps(obj) if rnd.randrange(2) else json.dumps(obj, separators=(",", ":"))).encode()
    n = N if len(base) < 300000 else max(200, N // 8)
    K = cfg.n_layers

    def lists(o):
        out = [o["decommitments"][1]["hash_witness"], o["decommitments"][2]["hash_witness"]]
        for l in range(K + 1):
            lay = o["fri_proof"]["first_layer"] if l == 0 else o["fri_proof"]["inner_layers"][l - 1]
            out += [lay["fri_witness"], lay["decommitment"]["hash_witness"]]
        return out

    texts = []
    for i in range(n):
        if i % 4 == 0:  # another length of some lists (entries removed / repeated; both value lists by whole rows or not)
            o = json.loads(base)
            for _ in range(rnd.randrange(1, 4)):
                lst = rnd.choice(lists(o))
                k = rnd.randrange(4)
                if k == 0 and lst:
                    del lst[rnd.randrange(len(lst))]
                elif k == 1 and lst:
                    lst.insert(rnd.randrange(len(lst) + 1), lst[rnd.randrange(len(lst))])
                elif k == 2:
                    del lst[:]
                else:
                    rows = rnd.randrange(0, 3)
                    for which, per in ((1, cfg.n_cols), (2, 16)):
                        if rnd.randrange(8):
                            del o["queried_values"][which][:rows * per]
            t = json.dumps(o, separators=(",", ":") if i % 8 else None).encode()
        else:
            t = _text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)
        if i % 7 == 0:
            t = b" " * rnd.randrange(1, 1100) + t
        texts.append(t)
    texts = [t for t in texts if len(t)]
    bad = taken = changed = 0
    recs, outcome = ver.read_stwo_texts(cfg, texts, MIN)
    for i, t in enumerate(texts):
        want, srec = verifier.stwo_minimal_text_is_canonical(cfg, t)
        if (outcome[i] == 0) != want:
            bad += 1
            print("OUTCOME", name, "json-minimal", i, int(outcome[i]), want, t[:100])
            continue
        if want:
            taken += 1
            got, hrec = verifier.parse_stwo_minimal_text(cfg, t, reader=verifier.READER_GENERAL)
            if got != 0 or not np.array_equal(verifier.stwo_minimal_from_capacity(cfg, recs[i]), srec) or not np.array_equal(hrec, srec):
                bad += 1
                print("RECORD", name, "json-minimal", i, got, t[:100])
            changed += not np.array_equal(srec, rec)
        s_rc, s_rec = verifier.parse_stwo_minimal_text(cfg, t, reader=verifier.READER_STREAM)
        if s_rc == 0:  # the streaming host reader against the general one
            g_rc, g_rec = verifier.parse_stwo_minimal_text(cfg, t, reader=verifier.READER_GENERAL)
            if g_rc != 0 or not np.array_equal(s_rec, g_rec):
                bad += 1
                print("STREAM", name, i, g_rc, t[:100])
    # the entry point on a sample: status words of the text path == of the record path for what parses
    sample = texts[:300]
    status, stats = ver.verify_stwo_minimal_texts(cfg, sample)
    ok_recs, ok_idx = [], []
    for i, t in enumerate(sample):
        got, hrec = verifier.parse_stwo_minimal_text(cfg, t)
        if got == 0:
            ok_recs.append(hrec); ok_idx.append(i)
        elif status[i] != got:
            bad += 1
            print("STAGE0", name, "json-minimal", i, int(status[i]), got)
    if ok_recs:
        st2 = ver.verify_stwo_minimal_records(cfg, ok_recs)
        if status[ok_idx].tolist() != st2.tolist():
            bad += 1
            print("STATUS", name, "json-minimal")
    print("%-24s %-4s: %6d mutants, %5d taken by the GPU reader (%5d with a changed record), disagreements %d"
          % (name, "json-minimal", len(texts), taken, changed, bad), flush=True)
    return bad


def main():
    ver = verifier.Verifier(0)
    rnd = random.Random(SEED)
    bad = 0
    for name, p in cases():
        rec = verifier.stwo_record(p)
        for fmt, kind in ((JSON, "json"), (WIT, "wit"), (SHARED, "json-shared")):
            if fmt == SHARED:  # every distinct Merkle sibling once + "queries": read and expanded on the GPU (round 4)
                obj = ss.stwo_to_json(p, shared=True)
                base = (json.dumps(obj) if rnd.randrange(2) else json.dumps(obj, separators=(",", ":"))).encode()
            else:
                base = write_text(p.cfg, rec, fmt, rnd.randrange(2))
            n = N if len(base) < 300000 else max(200, N // 8)
            texts = []
            for i in range(n):
                t = _text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)
                if i % 5 == 0:
                    t = _number_mutant(rnd, t)
                if i % 7 == 0 and fmt != WIT:
                    t = b" " * rnd.randrange(1, 1100) + t
                if fmt == SHARED and i % 9 == 0:  # another position in the hint (the lists then rarely fit it)
                    try:
                        o = json.loads(base)
                        o["queries"][rnd.randrange(p.cfg.n_queries)] = rnd.randrange(1 << p.cfg.lde_log)
                        t = json.dumps(o, separators=(",", ":")).encode()
                    except ValueError:
                        pass
                texts.append(t)
            recs, outcome = ver.read_stwo_texts(p.cfg, texts, fmt)
            taken = changed = 0
            for i, t in enumerate(texts):
                want, srec = canonical(p.cfg, t, fmt)
                if (outcome[i] == 0) != want:
                    bad += 1
                    print("OUTCOME", name, kind, i, int(outcome[i]), want, t[:100])
                    continue
                if want:
                    taken += 1
                    got, hrec = verifier.parse_stwo_text(p.cfg, t, fmt=HOST_FMT[fmt])
                    if got != 0 or not np.array_equal(recs[i], srec) or not np.array_equal(hrec, srec):
                        bad += 1
                        print("RECORD", name, kind, i, got, t[:100])
                    changed += not np.array_equal(srec, rec)
            # the entry point on a sample: status words of the text path == of the record path for what parses
            sample = texts[:300]
            status, stats = ver.verify_stwo_texts(p.cfg, sample, fmt=fmt)
            ok_recs, ok_idx = [], []
            for i, t in enumerate(sample):
                got, hrec = verifier.parse_stwo_text(p.cfg, t, fmt=HOST_FMT[fmt])
                if got == 0:
                    ok_recs.append(hrec); ok_idx.append(i)
                elif status[i] != got:
                    bad += 1
                    print("STAGE0", name, kind, i, int(status[i]), got)
            if ok_recs:
                st2 = ver.verify_stwo_records(p.cfg, ok_recs)
                if status[ok_idx].tolist() != st2.tolist():
                    bad += 1
                    print("STATUS", name, kind)
            print("%-24s %-4s: %6d mutants, %5d taken by the GPU reader (%5d with a changed record), disagreements so far %d"
                  % (name, kind, n, taken, changed, bad), flush=True)
        # the minimal proof.json (round 5): its list lengths are found in the text (landmarks), so the mutants also delete and
        # duplicate whole list entries -- still this form, another length
        bad += minimal_leg(ver, rnd, name, p)
    # stark101
    for fn, fmt in (("stark101_proof.json", JSON), (os.path.join("formats", "stark101_proof.wit"), WIT)):
        base = open(os.path.join(GOLDEN, fn), "rb").read()
        texts = [(_text_mutant(rnd, base) if i % 3 == 0 else _number_mutant(rnd, base)) for i in range(N)]
        recs, outcome = ver.read_stark101_texts(texts, fmt)
        taken = 0
        for i, t in enumerate(texts):
            want, srec = s101_canonical(t, fmt)
            if (outcome[i] == 0) != want:
                bad += 1
                print("OUTCOME stark101", i)
                continue
            if want:
                taken += 1
                pad = srec == 0xEEEEEEEE
                rc, shape, hrec = verifier.parse_s101_text(t, fmt=fmt)
                if rc != 0 or not np.array_equal(recs[i][~pad], srec[~pad]) or recs[i][pad].any() or not np.array_equal(hrec, recs[i]):
                    bad += 1
                    print("RECORD stark101", i)
        print("stark101 %-16s: %6d mutants, %5d taken, disagreements so far %d" % ("json" if fmt == JSON else "wit", N, taken, bad), flush=True)
    print("TOTAL disagreements:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
