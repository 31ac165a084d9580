"""The Python calls INTEGRATION.md section 1 shows, executed as written on the reference's own files (tests/golden holds
them byte for byte): documentation that drifts from the API fails here."""
import json
import os

import numpy as np
import pytest

import stark_symphony_amd as ss
from stark_symphony_amd import formats, prover, prover101, verifier

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_integration_md_python_calls(tmp_path):
    import torch
    # stark101: proof.json / proof.wit / the prover's return value
    proof = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    assert verifier.verify_stark101(proof)
    wit = ss.stark101_to_wit(proof)
    assert verifier.verify_stark101(ss.stark101_from_wit(wit))
    # stwo: tests/data/proof.json
    p = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof.json"))))
    assert verifier.verify_stwo(p)
    ver = verifier.Verifier(device=0)
    bad = formats.stwo_corrupt(p, np.random.default_rng(1))[0]
    status = ver.verify_stwo([p, bad, p], cfg=ss.PRODUCTION_CONFIG)
    assert status.dtype == np.uint32 and status[0] == 0 and status[1] != 0 and status[2] == 0
    assert ver.verify_stwo([p], cfg=[ss.PRODUCTION_CONFIG, p.cfg]).tolist() == [0]
    # files / texts straight into the library
    a, b = tmp_path / "a.wit", tmp_path / "b.json"
    a.write_text(ss.stwo_to_wit(p))
    b.write_text(json.dumps(ss.stwo_to_json(p)))
    status, stats = ver.verify_stwo_files(ss.PRODUCTION_CONFIG, [str(a), str(b)])
    assert status.tolist() == [0, 0] and "host_parsed" in stats
    status, stats = ver.verify_stark101_texts([wit.encode()])
    assert status.tolist() == [0] and stats["host_parsed"] == 0
    # shared records
    shared = [verifier.stwo_shared_record(q, queries=None) for q in (p, p)]
    assert ver.verify_stwo_shared_records(p.cfg, shared).tolist() == [0, 0]
    # minimal records, the minimal proof.json, a caller-pinned buffer
    minimal = [verifier.stwo_minimal_record(formats.stwo_minimise(q)) for q in (p, p, bad)]
    assert ver.verify_stwo_minimal_records(p.cfg, minimal)[:2].tolist() == [0, 0]
    text = json.dumps(formats.stwo_minimal_to_json(formats.stwo_minimise(p)))
    status, stats = ver.verify_stwo_minimal_texts(p.cfg, [text.encode()])
    assert status.tolist() == [0]
    flat = ver.pinned_buffer(sum(r.size for r in minimal)); flat[:] = np.concatenate(minimal)
    offs = np.concatenate([[0], np.cumsum([r.size for r in minimal])]).astype(np.uint64)
    assert ver.verify_stwo_pinned(p.cfg, flat, offs, "minimal")[:2].tolist() == [0, 0]
    blob, boffs, blens = ver.pinned_text_blob([a.read_bytes(), b.read_bytes()])
    status, stats = ver.verify_stwo_texts_pinned(p.cfg, blob, boffs, blens)
    assert status.tolist() == [0, 0]
    # section 4: rank-local file ingest through the entry point the rank's host-thread budget calls for (one rank here)
    from stark_symphony_amd import distributed
    local, accepted, total = distributed.verify_files_sharded([str(a), str(b)], distributed.files_verifier(ver, ss.PRODUCTION_CONFIG))
    assert local.tolist() == [0, 0] and (accepted, total) == (2, 2)
    assert distributed.files_verifier(ver, ss.PRODUCTION_CONFIG, world=8)([str(a), str(b)]).tolist() == [0, 0]   # the pinned route
    # resident batches, pipelined; the accept reduce hook; a hipGraph replay
    batch = ver.stwo_batch([p, bad])
    pipe = verifier.Pipeline([batch, batch.sibling(), batch.sibling()])
    pipe.submit(); pipe.synchronize()
    assert batch.status()[0] == 0 and batch.status()[1] != 0
    assert set(batch.intermediates(0)) >= {"queries"}
    seen = []
    reduce = lambda k: seen.append(int(pipe.slots[k].accept_dev.item()))  # (a caller would all_reduce here)
    for _ in range(5):
        pipe.submit(on_reuse=reduce)
    pipe.flush(reduce); pipe.synchronize()
    assert seen and all(v == 1 for v in seen)
    graph = verifier.GraphedPipeline([batch, batch.sibling()], concurrent_tails=True); graph.replay(); graph.synchronize()
    assert batch.accepted() == 1
    # the GPU provers: byte-identical to the reference's proofs
    res = prover101.Stark101GpuProver(ver).prove()
    assert res is not None
    pj = prover.GpuProver(ver).prove(n_cols=4, trace_log=9, log_blowup=4, n_queries=16)
    assert pj == json.load(open(os.path.join(GOLDEN, "stwo_proof.json")))
    torch.cuda.synchronize()
