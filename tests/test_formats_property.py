"""Property tests of the format layer (formats A-D plus the .simf snippets) on random proofs of
random shapes: every writer/reader pair is the identity, the packed record has the size the C ABI
announces, and no malformed text escapes as anything but MalformedProof."""
import json

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import stark_symphony_amd as ss
from stark_symphony_amd import binding, formats, records, verifier

U32 = 1 << 32


def _rand_stwo(seed: int, n_cols: int, lde_log: int, n_queries: int, n_layers: int, ragged: bool) -> ss.StwoProof:
    rng = np.random.default_rng(seed)
    cfg = ss.StwoConfig(n_cols, max(1, lde_log - 1), lde_log, n_queries, n_layers, int(rng.integers(0, 20)))
    w = lambda *shape: rng.integers(0, U32, size=shape, dtype=np.uint64).astype(np.uint32)  # noqa: E731
    h = lambda n: rng.integers(0, 256, size=(n, 32), dtype=np.uint8)  # noqa: E731

    def plen(n: int) -> int:
        return int(rng.integers(0, 32)) if ragged and rng.integers(4) == 0 else n
    return ss.StwoProof(cfg, h(3), w(n_cols, 4), w(16, 4), w(n_queries, n_cols), w(n_queries, 16),
                        [h(plen(lde_log)) for _ in range(n_queries)], [h(plen(lde_log)) for _ in range(n_queries)],
                        h(n_layers + 1), w(4), w(n_layers + 1, n_queries, 4),
                        [[h(plen(lde_log - 1 - l)) for _ in range(n_queries)] for l in range(n_layers + 1)],
                        int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(2)))


shapes = st.tuples(st.integers(0, 2 ** 31), st.integers(1, 9), st.integers(3, 12), st.integers(1, 5),
                   st.integers(0, 2)).filter(lambda t: t[4] + 1 < t[2])


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(shapes)
def test_stwo_every_format_is_the_identity(t):
    seed, n_cols, lde_log, n_queries, n_layers = t
    p = _rand_stwo(seed, n_cols, lde_log, n_queries, n_layers, ragged=False)
    want = ss.stwo_to_json(p)
    c = p.cfg
    assert ss.stwo_to_json(ss.stwo_from_json(json.dumps(want), c.trace_log)) == want
    assert ss.stwo_to_json(ss.stwo_from_wit(ss.stwo_to_wit(p), c.trace_log, c.pow_bits)) == want
    assert ss.stwo_to_json(ss.stwo_from_simf(ss.stwo_to_simf(p), c.trace_log, c.pow_bits)) == want
    rec = verifier.stwo_record(p)
    cs = verifier.stwo_cfg_struct(c, verifier.MODE_FIXTURE)
    import ctypes as C
    trailer = (c.n_layers + 3) * c.n_queries  # path_len words
    assert rec.size == binding.lib().ss_stwo_record_words(C.byref(cs)) == c.packed_bytes // 4 + trailer
    assert records.stwo_from_record(c, rec).trace_paths[0].shape == (c.lde_log, 32)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(shapes)
def test_stwo_ragged_paths_survive_the_text_formats_and_are_reported(t):
    seed, n_cols, lde_log, n_queries, n_layers = t
    p = _rand_stwo(seed, n_cols, lde_log, n_queries, n_layers, ragged=True)
    c = p.cfg
    back = ss.stwo_from_wit(ss.stwo_to_wit(p), c.trace_log, c.pow_bits)
    same_len = all(len(a) == len(b) for a, b in zip(p.trace_paths + p.cp_paths, back.trace_paths + back.cp_paths))
    assert same_len and all(len(a) == len(b) for la, lb in zip(p.fri_paths, back.fri_paths) for a, b in zip(la, lb))
    if len(p.trace_paths[0]) == c.lde_log:  # the .wit has no config: the first trace path fixes LDE_LOG_SIZE
        rec = verifier.stwo_record(back)
        lens = rec[c.packed_bytes // 4:].reshape(c.n_layers + 3, c.n_queries)
        assert lens[0].tolist() == [len(x) for x in p.trace_paths] and lens[1].tolist() == [len(x) for x in p.cp_paths]
        for l, layer in enumerate(p.fri_paths):
            assert lens[2 + l].tolist() == [len(x) for x in layer]


def _rand_s101(seed: int, n_layers: int) -> ss.Stark101Proof:
    rng = np.random.default_rng(seed)
    ev = lambda: formats.Stark101Eval(int(rng.integers(0, U32)),  # noqa: E731
                                      rng.integers(0, 256, size=(int(rng.integers(0, 32)), 32), dtype=np.uint8))
    layers = [formats.Stark101Layer(bytes(rng.integers(0, 256, size=32, dtype=np.uint8)), int(rng.integers(0, U32)),
                                    ev(), ev()) for _ in range(n_layers)]
    return ss.Stark101Proof(bytes(rng.integers(0, 256, size=32, dtype=np.uint8)), [ev(), ev(), ev()], layers,
                            int(rng.integers(0, U32)))


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(0, 31))
def test_stark101_every_format_is_the_identity(seed, n_layers):
    p = _rand_s101(seed, n_layers)
    want = ss.stark101_to_json(p)
    assert ss.stark101_to_json(ss.stark101_from_json(json.dumps(want))) == want
    assert ss.stark101_to_json(ss.stark101_from_wit(ss.stark101_to_wit(p))) == want
    assert ss.stark101_to_json(ss.stark101_from_simf(ss.stark101_to_simf(p))) == want
    ml, pm = verifier.s101_shape_of([p])
    import ctypes as C
    sh = binding.S101Shape(ml, pm)
    assert verifier.s101_record(p, ml, pm).size == binding.lib().ss_s101_record_words(C.byref(sh))


@settings(max_examples=300, deadline=None)
@given(st.text(alphabet="()[],0123456789xabf_ list!qm31\n", max_size=60))
def test_literal_parser_never_crashes(text):
    try:
        v = formats.parse_literal(text)
    except ss.MalformedProof:
        return
    assert isinstance(v, (int, list))


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(0, 400))
def test_truncated_witness_text_is_malformed_not_a_crash(seed, cut):
    p = _rand_stwo(seed, 2, 4, 2, 1, ragged=False)
    text = ss.stwo_to_simf(p)
    cut = min(cut, len(text) - 2)
    with pytest.raises(ss.MalformedProof):
        ss.stwo_from_simf(text[:cut], 3)
