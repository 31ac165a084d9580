#!/usr/bin/env python3
"""Generate tests/golden/stark101_proof.json by IMPORTING the reference prover.

Runs only in the build container (needs /root/reference); never on the GPU box.
It calls the reference's own `fibsquare.prover.prove()`
(/root/reference/stark101/scripts/fibsquare/prover.py:94-171) and dumps its `res`
dict exactly as the reference's `__main__.py:8-11` does, plus the prover-side
channel transcript summary we use as stage-level golden values:

  * res               -> stark101_proof.json   (format A of SURVEY.md 8b)
  * channel.proof     -> stark101_transcript.json (59 messages; bytes as hex,
                         FieldElement as int, list[bytes] as list of hex)

The reference prover is deterministic (trace seed 1, 3141592), so this file is a
pure function of the reference sources.
"""
import contextlib, io, json, os, sys

REF = os.environ.get("SS_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "stark101", "scripts"))

from fibsquare.prover import prove          # noqa: E402
from fibsquare.field import FieldElement    # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
with contextlib.redirect_stdout(io.StringIO()):
    proof, res = prove()


def enc(m):
    if isinstance(m, bytes):
        return {"bytes": m.hex()}
    if isinstance(m, FieldElement):
        return {"felt": m.val}
    if isinstance(m, list):
        return {"path": [x.hex() for x in m]}
    raise TypeError(type(m))


with open(os.path.join(here, "stark101_proof.json"), "w") as f:
    json.dump(res, f, indent=1)
with open(os.path.join(here, "stark101_transcript.json"), "w") as f:
    json.dump([enc(m) for m in proof], f, indent=0)
print("wrote stark101_proof.json (%d layers) and transcript (%d msgs)"
      % (len(res["fri_layers"]), len(proof)))
