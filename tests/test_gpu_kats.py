"""The reference's 86 `fn test_*` known-answer tests replayed ON THE DEVICE ("tests of its CPU
restatement and its HIP kernels"; Appendix A).  Every key of tests/golden/kats.json -- the literals of one reference test,
extracted by tests/golden/make_kats.py -- has a handler below that feeds the test's inputs to the GPU (ss_kat: one reference
function per item, through the device functions the kernels are built from; or the kernels themselves for the two
end-to-end proofs) and compares with the test's EXPECTED LITERALS directly -- not through the oracle.  The test is
parametrised over the keys of the file, so a vector without a handler fails.  The position of every literal inside a
test body is the one tests/test_oracle_kats.py spells out for the same vector."""
import json
import os

import numpy as np
import pytest

from stark_symphony_amd import verifier

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "kats.json")))
S, W = "stark101/src/", "stwo-verifier/src/"
P101 = 3221225473
HANDLERS = {}


def kat(*keys):
    def deco(fn):
        for k in keys:
            assert k not in HANDLERS, k
            HANDLERS[k] = fn
        return fn
    return deco


def h8(v: int):
    """u256 -> 8 hash words (word j = big-endian bytes 4j..4j+3)."""
    return [(v >> (32 * (7 - j))) & 0xFFFFFFFF for j in range(8)]


def u256(words) -> int:
    out = 0
    for w in words:
        out = (out << 32) | int(w)
    return out


@pytest.fixture(scope="module")
def K():
    ver = verifier.Verifier(0)

    def run(op, items):
        return verifier.kat(ver, op, items)
    run.ver = ver
    return run


def sha_words(K, words):
    return [int(x) for x in K(0, [[len(words)] + list(words)])[0]]


def merkle(K, family, auth, leaf, root, nodes):
    row = [family, auth, len(nodes)] + list(leaf) + h8(root)
    for n in nodes:
        row += h8(n)
    return int(K(1, [row])[0][0])


# ------------------------------------------------------------------------------------ stark101 field / channel / sha256
@kat(S + "field.simf::test_endianness")
def _(v, K, key):
    assert (v[0] >> 32, v[0] & 0xFFFFFFFF) == (v[1], v[2])  # a statement about jets, no function of the path


@kat(*(S + "field.simf::" + n for n in ("test_add_mod", "test_sub_mod", "test_mul_mod", "test_mul_mod_2", "test_exp_mod", "test_exp_mod_2")))
def _(v, K, key):
    a, b, c = v
    col = {"add": 0, "sub": 1, "mul": 2, "exp": 4}[key.split("::test_")[1][:3]]
    assert int(K(11, [[0, a, b]])[0][col]) == c


@kat(S + "field.simf::test_div_mod")
def _(v, K, key):
    a, b = v
    q = int(K(11, [[0, a, b]])[0][3])
    assert q != 0xFFFFFFFF and int(K(11, [[0, q, b]])[0][2]) == a


@kat(S + "field.simf::test_div_mod_2")
def _(v, K, key):
    a, b, c = v
    assert int(K(11, [[0, a, b]])[0][3]) == c


@kat(S + "channel.simf::test_channel_draw_32")
def _(v, K, key):
    state, mx, val, nxt = v
    out = K(11, [[1] + h8(state) + [mx]])[0]
    assert int(out[0]) == val and u256(out[1:9]) == nxt


@kat(S + "sha256.simf::test_sha256", W + "hasher.simf::test_sha256")
def _(v, K, key):
    assert u256(sha_words(K, h8(v[0]))) == v[1]


@kat(S + "sha256.simf::test_sha256_32", W + "hasher.simf::test_sha256_32")
def _(v, K, key):
    assert u256(sha_words(K, [v[0]])) == v[1]


@kat(S + "merkle.simf::test_merkle", W + "merkle.simf::test_merkle")
def _(v, K, key):
    root, leaf_in, n0, n1, auth = v
    fam = 1 if key.startswith(W) else 0
    leaf = sha_words(K, h8(leaf_in))
    assert merkle(K, fam, auth, leaf, root, [n0, n1]) == 0
    assert merkle(K, fam, auth ^ 1, leaf, root, [n0, n1]) != 0
    if fam:  # one sibling short: the path ends at 2, not 1 (merkle.simf:42)
        assert merkle(K, 1, auth, leaf, root, [n0]) == 1 and merkle(K, 1, auth, leaf, root, [n1, n0]) == 2


@kat(S + "merkle.simf::test_decommitment", W + "merkle.simf::test_decommitment")
def _(v, K, key):
    root, ev, leaf_id, nodes, n_leaves = v[0], v[1], v[2], v[3:16], v[16]
    fam = 1 if key.startswith(W) else 0
    leaf = sha_words(K, [ev])
    assert merkle(K, fam, leaf_id + n_leaves, leaf, root, nodes) == 0
    bad = list(nodes)
    bad[5] ^= 1
    assert merkle(K, fam, leaf_id + n_leaves, leaf, root, bad) != 0


# --------------------------------------------------------------------------------------------------- stark101 air / fri
@kat(S + "air.simf::test_fibsquare_calc_x")
def _(v, K, key):
    assert int(K(11, [[3, v[0], 0, 0]])[0][0]) == v[1]


@kat(S + "air.simf::test_fibsquare_eval_p0")
def _(v, K, key):
    x, f_x, p0 = v
    assert int(K(11, [[3, 0, x, f_x]])[0][1]) == p0


@kat(S + "air.simf::test_fibsquare_eval_cp")
def _(v, K, key):
    a0, a1, a2, f_x, f_gx, f_ggx, x, cp = v
    assert int(K(11, [[4, a0, a1, a2, f_x, f_gx, f_ggx, x]])[0][0]) == cp


@kat(S + "air.simf::test_fibsquare_read_coefficients")
def _(v, K, key):
    state, a0, a1, a2 = v
    assert [int(x) for x in K(11, [[2] + h8(state)])[0][:3]] == [a0, a1, a2]


@kat(S + "fri.simf::test_fri_eval_cp_next")
def _(v, K, key):
    cpa, cpb, x, beta, nxt = v
    assert int(K(11, [[5, cpa, cpb, x, beta]])[0][0]) == nxt


@kat(S + "fri.simf::test_compute_auth_path")
def _(v, K, key):
    out = K(11, [[6, v[i], v[i + 1]] for i in range(0, 16, 4)])
    assert [[int(r[0]), int(r[1])] for r in out] == [[v[i + 2], v[i + 3]] for i in range(0, 16, 4)]


def _layer(v):
    return v[0], v[1], v[2], v[3:16], v[16], v[17:30]


@kat(S + "fri.simf::test_fri_verify_layer")
def _(v, K, key):
    root, beta, cpa, pa, cpb, pb = _layer(v)
    idx, x, cp_ev, dom = v[30:34]
    exp_idx, exp_x, exp_cp, exp_dom = v[34:38]
    assert cp_ev == cpa
    a, b = (int(t) for t in K(11, [[6, idx, dom]])[0][:2])
    assert merkle(K, 0, a, sha_words(K, [cpa]), root, pa) == 0
    assert merkle(K, 0, b, sha_words(K, [cpb]), root, pb) == 0
    nxt = int(K(11, [[5, cpa, cpb, x, beta]])[0][0])
    assert (idx, int(K(11, [[0, x, x]])[0][2]), nxt, dom // 2) == (exp_idx, exp_x, exp_cp, exp_dom)


@kat(S + "fri.simf::test_fri_read_commitment")
def _(v, K, key):
    root, beta = v[0], v[1]
    state, expect = v[30], v[31]
    mixed = sha_words(K, h8(state) + h8(root))  # channel_mix_256: sha256(state || root)
    out = K(11, [[1] + mixed + [P101]])[0]
    assert int(out[0]) == beta and u256(out[1:9]) == expect


@kat(S + "verifier.simf::test_verifier")
def _(v, K, key):
    """verifier.simf:44-388: the literal IS the committed proof (checked word by word) and the KERNELS accept it."""
    import stark_symphony_amd as ss
    proof = ss.stark101_from_json(json.load(open(os.path.join(GOLDEN, "stark101_proof.json"))))
    j = ss.stark101_to_json(proof)
    flat = [j["p_mt_root"]]
    for ev, pth in j["evals"]:
        flat += [ev] + pth
    for l in j["fri_layers"]:
        flat += [l[0], l[1], l[2]] + l[3] + [l[4]] + l[5]
    flat.append(j["fri_last_layer"])
    assert flat == v
    b = K.ver.stark101_batch([proof])
    b.run()
    assert b.status().tolist() == [0] and b.intermediates(0)["idx"] == 6160


# ---------------------------------------------------------------------------------------------------------- stwo fields
@kat(W + "fields/m31.simf::test_m31_inv")
def _(v, K, key):
    a, e = v
    inv = int(K.ver.selftest(1, np.array([[a, 0]], dtype=np.uint32))[0][3])
    # a^e by squaring on the device's m31_mul (selftest op 1, column 2)
    res, base, k = 1, a, e
    while k:
        if k & 1:
            res = int(K.ver.selftest(1, np.array([[res, base]], dtype=np.uint32))[0][2])
        base = int(K.ver.selftest(1, np.array([[base, base]], dtype=np.uint32))[0][2])
        k >>= 1
    assert inv == res and int(K.ver.selftest(1, np.array([[0, 0]], dtype=np.uint32))[0][3]) == 0xFFFFFFFF


@kat(W + "fields/m31.simf::test_m31_add", W + "fields/m31.simf::test_m31_sub")
def _(v, K, key):
    a, b, c = v
    assert int(K.ver.selftest(1, np.array([[a, b]], dtype=np.uint32))[0][0 if key.endswith("add") else 1]) == c


@kat(*(W + "fields/cm31.simf::test_cm31_" + n for n in ("add", "sub", "mul")))
def _(v, K, key):
    col = {"add": 0, "sub": 2, "mul": 4}[key[-3:]]
    assert [int(x) for x in K(3, [v[0:4]])[0][col:col + 2]] == v[4:6]


@kat(W + "fields/cm31.simf::test_cm31_mul_2")
def _(v, K, key):
    ab = [int(x) for x in K(3, [v[0:4]])[0][4:6]]
    assert [int(x) for x in K(3, [ab + v[4:6]])[0][4:6]] == v[6:8]


@kat(W + "fields/cm31.simf::test_cm31_div")
def _(v, K, key):
    assert [int(x) for x in K(3, [v[0:4]])[0][6:8]] == v[4:6]


@kat(W + "fields/cm31.simf::test_cm31_inv")
def _(v, K, key):
    inv = [int(x) for x in K(3, [v[0:2] + [1, 0]])[0][8:10]]
    assert [int(x) for x in K(3, [v[0:2] + inv])[0][4:6]] == v[2:4]


@kat(W + "fields/qm31.simf::test_qm31_inv")
def _(v, K, key):
    out = K.ver.selftest(2, np.array([v + [1, 0, 0, 0]], dtype=np.uint32))[0]
    inv = [int(x) for x in out[4:8]]
    assert [int(x) for x in K.ver.selftest(2, np.array([v + inv], dtype=np.uint32))[0][:4]] == [1, 0, 0, 0]


@kat(W + "fields/qm31.simf::test_qm31_add", W + "fields/qm31.simf::test_qm31_sub")
def _(v, K, key):
    col = 0 if key.endswith("add") else 4
    assert [int(x) for x in K(4, [v[0:8]])[0][col:col + 4]] == v[8:12]


@kat(W + "fields/qm31.simf::test_qm31_mul")
def _(v, K, key):
    assert [int(x) for x in K.ver.selftest(2, np.array([v[0:8]], dtype=np.uint32))[0][:4]] == v[8:12]


@kat(W + "fields/qm31.simf::test_qm31_mul_m31")
def _(v, K, key):
    assert [int(x) for x in K(4, [v[0:4] + [v[4], 0, 0, 0]])[0][8:12]] == v[5:9]


@kat(W + "fields/qm31.simf::test_qm31_mul_cm31")
def _(v, K, key):
    a, b, c = v[0:4], v[4:6], v[6:10]
    got = [int(x) for x in K(4, [a + b + [0, 0]])[0][12:16]]
    assert got == [int(x) for x in K.ver.selftest(2, np.array([a + c], dtype=np.uint32))[0][:4]]


# ---------------------------------------------------------------------------------------------------------- stwo groups
@kat(W + "groups/m31_point.simf::test_m31_point_add_1")
def _(v, K, key):
    assert [int(x) for x in K(5, [v[0:2] + v[0:2]])[0][0:2]] == v[2:4]


@kat(W + "groups/m31_point.simf::test_m31_point_add_2")
def _(v, K, key):
    assert [int(x) for x in K(5, [v[0:4]])[0][0:2]] == v[4:6]


@kat(W + "groups/m31_point.simf::test_m31_point_zero")
def _(v, K, key):
    assert [int(x) for x in K.ver.selftest(3, np.array([[0]], dtype=np.uint32))[0]] == v


@kat(W + "groups/m31_point.simf::test_m31_point_add_zero")
def _(v, K, key):
    assert [int(x) for x in K(5, [v + [1, 0]])[0][0:2]] == v


@kat(W + "groups/m31_point.simf::test_m31_point_dbl")
def _(v, K, key):
    assert [int(x) for x in K(5, [v[0:2] + [0, 0]])[0][2:4]] == v[2:4]


@kat(W + "groups/m31_point.simf::test_circle_point_index_to_m31_point")
def _(v, K, key):
    assert [int(x) for x in K.ver.selftest(3, np.array([[v[0]]], dtype=np.uint32))[0]] == v[1:3]


QM31_GEN = [1, 0, 478637715, 513582971, 992285211, 649143431, 740191619, 1186584352]  # groups/qm31_point.simf:14


@kat(W + "groups/qm31_point.simf::test_add_circle_point_m31")
def _(v, K, key):
    """identity test (no literals): G + m computed with the M31 coordinates == G + the embedded point."""
    m = [2, 1268011823]  # groups/m31_point.simf:13
    emb = [m[0], 0, 0, 0, m[1], 0, 0, 0]
    out = K(7, [QM31_GEN + emb + m])[0]
    assert out[0:8].tolist() == out[8:16].tolist()


@kat(W + "groups/qm31_point.simf::test_m31_point_neg")
def _(v, K, key):
    """identity test: 3G + (-(3G)) == zero (qm31_neg is the wrap-around P - a on every coordinate of y)."""
    g2 = [int(x) for x in K(7, [QM31_GEN + QM31_GEN + [0, 0]])[0][0:8]]
    g3 = [int(x) for x in K(7, [g2 + QM31_GEN + [0, 0]])[0][0:8]]
    neg = g3[0:4] + [int(K.ver.selftest(1, np.array([[0, c]], dtype=np.uint32))[0][1]) if c else 2147483647 for c in g3[4:8]]
    # (m31_neg(c) = P - c: for c != 0 that is the canonical 0 - c; no coordinate of 3G.y is zero)
    assert all(g3[4:8])
    assert [int(x) for x in K(7, [g3 + neg + [0, 0]])[0][0:8]] == [1, 0, 0, 0, 0, 0, 0, 0]


@kat(W + "groups/coset.simf::test_bit_reverse_position")
def _(v, K, key):
    i, log, r = v
    assert int(K(6, [[i, 0, log]])[0][0]) == r


@kat(W + "groups/coset.simf::test_circle_point_index_add", W + "groups/coset.simf::test_circle_point_index_mul")
def _(v, K, key):
    a, b, c = v
    assert int(K(6, [[a, b, 1]])[0][1 if key.endswith("add") else 2]) == c


@kat(W + "groups/coset.simf::test_circle_point_index_neg")
def _(v, K, key):
    assert int(K(6, [[v[0], 0, 1]])[0][3]) == v[1]


@kat(W + "groups/circle_domain.simf::test_circle_domain")
def _(v, K, key):
    assert [int(x) for x in K(6, [[0, 0, v[0]]])[0][4:7]] == v[1:4]


@kat(W + "groups/circle_domain.simf::test_circle_position_to_point_index",
     W + "groups/circle_domain.simf::test_circle_position_to_point_index_2")
def _(v, K, key):
    log, pos, idx = v
    assert int(K(6, [[pos, 0, log]])[0][7]) == idx


# --------------------------------------------------------------------------------------------------------- stwo channel
def chan(K, digest, ctr, k, payload=()):
    out = K(2, [h8(digest) + [ctr, k] + list(payload)])[0]
    return u256(out[0:8]), int(out[8]), [int(x) for x in out[9:17]]


@kat(W + "channel.simf::test_channel_draw_qm31")
def _(v, K, key):
    _, ctr, r = chan(K, v[0], v[1], 0)
    assert r == v[2:10] and ctr == v[1] + 2


@kat(W + "channel.simf::test_channel_draw_qm31_point")
def _(v, K, key):
    assert chan(K, v[0], v[1], 1)[2] == v[2:10]


@kat(W + "pow.simf::test_reverse_bytes_32")
def _(v, K, key):
    assert chan(K, 0, 0, 3, [0, 0, 0, 0, v[0]])[2][1] == v[1]


@kat(W + "pow.simf::test_check_proof_of_work")
def _(v, K, key):
    digest, ctr, nonce, expect = v
    tgt = 0x07FFFFFFFFFFFFFF
    d, _, r = chan(K, digest, ctr, 3, [nonce >> 32, nonce & 0xFFFFFFFF, tgt >> 32, tgt & 0xFFFFFFFF])
    assert d == expect and r[0] == 1
    assert chan(K, digest, ctr, 3, [(nonce + 1) >> 32, (nonce + 1) & 0xFFFFFFFF, tgt >> 32, tgt & 0xFFFFFFFF])[2][0] == 0


@kat(W + "evals/commit.simf::test_evals_commit")
def _(v, K, key):
    d, _, r = chan(K, v[0], v[1], 5, h8(v[2]) + h8(v[3]) + h8(v[4]))
    assert d == v[5] and r[0:4] == v[6:10]


@kat(W + "fri/queries.simf::test_channel_draw_queries_8")
def _(v, K, key):
    assert chan(K, v[0], v[1], 4, [v[2]])[2] == v[3:11]


@kat(W + "fri/commit.simf::test_fri_commit")
def _(v, K, key):
    d, ctr = v[0], v[1]
    alphas = []
    for root in v[2:5]:
        d, ctr, r = chan(K, d, ctr, 6, h8(root))
        alphas.append(r[0:4])
    d, _, _ = chan(K, d, ctr, 7, v[5:9])
    assert d == v[9] and alphas[0] == v[10:14]


# ------------------------------------------------------------------------------------------------- stwo evals / oods
def comp(K, log_size, pt, cols, alpha, cp16):
    row = [log_size] + list(pt) + [x for c in cols for x in c] + list(alpha) + [x for c in cp16 for x in c]
    out = [int(x) for x in K(8, [row])[0]]
    return {"van": out[0:4], "eval": out[4:8], "dec": out[8:12], "part": out[12:16], "abort": out[16]}


Z4, Z16 = [[0] * 4] * 4, [[0] * 4] * 16


@kat(W + "evals/composition_poly.simf::test_composition_poly_eval_from_partitions")
def _(v, K, key):
    parts = [v[0:4], v[4:8], v[8:12], v[12:16]] + [[0] * 4] * 12
    assert comp(K, 1, [0] * 8, Z4, [0] * 4, parts)["part"] == v[16:20]


@kat(W + "evals/composition_poly.simf::test_vanishing_poly_eval")
def _(v, K, key):
    assert comp(K, v[0], v[1:9], Z4, [0] * 4, Z16)["van"] == v[9:13]


@kat(W + "constraints/wide_fibonacci.simf::test_eval_composition_poly")
def _(v, K, key):
    out = comp(K, v[0], v[1:9], [v[9 + 4 * i:13 + 4 * i] for i in range(4)], v[25:29], Z16)
    assert out["abort"] == 0 and out["eval"] == v[29:33]


@kat(W + "evals/trace_poly.simf::test_col_evals_qm31_get", W + "evals/trace_poly.simf::test_col_evals_m31_get")
def _(v, K, key):
    """MAX_COLUMN_OFFSET = 1: `get(col, 0)` is the identity (layout only; no device function)."""
    if key.endswith("qm31_get"):
        assert v[0:4] == v[5:9] and v[4] == 0
    else:
        assert v == [1, 0, 1]


def _oods_inputs(v, off):
    return [v[off + 4 * i:off + 4 * i + 4] for i in range(4)], [v[off + 16 + 4 * i:off + 20 + 4 * i] for i in range(16)]


@kat(W + "deep/oods.simf::test_channel_mix_oods_evals")
def _(v, K, key):
    trace, cp = _oods_inputs(v, 2)
    words = h8(v[0]) + [x for c in trace for x in c] + [x for c in cp for x in c]
    assert u256(sha_words(K, words)) == v[82]


@kat(W + "deep/oods.simf::test_oods")
def _(v, K, key):
    """deep/oods.simf:68-100 step by step (oods :44-64) on the device functions."""
    log_size, alpha = v[2], v[3:7]
    trace, cp = _oods_inputs(v, 7)
    d, ctr, pt = chan(K, v[0], v[1], 1)
    d = u256(sha_words(K, h8(d) + [x for c in trace for x in c] + [x for c in cp for x in c]))
    out = comp(K, log_size, pt, trace, alpha, cp)
    assert out["abort"] == 0 and out["eval"] == out["dec"]  # the OODS check itself (deep/oods.simf:58)
    d2, _, r = chan(K, d, 0, 0)
    assert d2 == v[87] and r[0:4] == v[88:92]


@kat(W + "evals/verify.simf::test_verify_query")
def _(v, K, key):
    roots = v[0:3]
    trace_vals, trace_path = v[3:7], v[7:18]
    cp_vals, cp_path = v[18:34], v[34:45]
    query, domain = v[45], v[46]
    assert merkle(K, 1, query + domain, sha_words(K, trace_vals), roots[1], trace_path) == 0
    assert merkle(K, 1, query + domain, sha_words(K, cp_vals), roots[2], cp_path) == 0


# ------------------------------------------------------------------------------------------------ stwo deep quotients
@kat(W + "deep/quotients.simf::test_quotient_denominator_inverse")
def _(v, K, key):
    out = K(9, [v[0:8] + [0] * 8 + v[8:10] + [0]])[0]
    assert [int(x) for x in out[0:2]] == v[10:12] and out[18] == 0


@kat(W + "deep/quotients.simf::test_deep_quotient_nominator")
def _(v, K, key):
    """the nominator from GIVEN coefficients: b * value - (a * q.y + c) over the device's qm31_mul_m31 / add / sub"""
    a, b, c = v[0:4], v[4:8], v[8:12]
    qy, value = v[13], v[14]
    bv = [int(x) for x in K(4, [b + [value, 0, 0, 0]])[0][8:12]]
    ay = [int(x) for x in K(4, [a + [qy, 0, 0, 0]])[0][8:12]]
    s = [int(x) for x in K(4, [ay + c])[0][0:4]]
    assert [int(x) for x in K(4, [bv + s])[0][4:8]] == v[15:19]


@kat(W + "deep/quotients.simf::test_deep_quotient_interpolant_coefficients")
def _(v, K, key):
    out = K(9, [v[12:20] + v[20:24] + v[24:28] + [0, 0, 0]])[0]
    assert [int(x) for x in out[2:14]] == v[0:12]


# ------------------------------------------------------------------------------------------------------------ stwo fri
@kat(W + "fri/folding.simf::test_circle_fold", W + "fri/folding.simf::test_line_fold")
def _(v, K, key):
    kind = 0 if key.endswith("circle_fold") else 1
    out = K(10, [[kind, v[0]] + v[1:5] + v[5:9] + [v[9]] + v[10:14]])[0]
    assert out[0] == 0 and [int(x) for x in out[1:5]] == v[14:18]


def fri_decommit(K, position, e0, e1, log_size, nodes, root):
    """verify_decommitment, fri/layers.simf:40-48"""
    node = sha_words(K, sha_words(K, e0) + sha_words(K, e1))
    return merkle(K, 1, (position + (1 << log_size)) // 2, node, root, nodes)


@kat(W + "fri/layers.simf::test_verify_decommitment")
def _(v, K, key):
    assert fri_decommit(K, v[0], v[1:5], v[5:9], v[9], v[10:13], v[13]) == 0


@kat(W + "fri/layers.simf::test_fri_verify_first_layer")
def _(v, K, key):
    query, ev, wit, nodes, root, alpha, log_size = v[0], v[1:5], v[5:9], v[9:12], v[12], v[13:17], v[17]
    assert query % 2 == 0 and fri_decommit(K, query, ev, wit, log_size, nodes, root) == 0
    assert K(10, [[0, query] + ev + wit + [log_size] + alpha])[0][0] == 0


@kat(W + "fri/layers.simf::test_fri_verify_inner_layer")
def _(v, K, key):
    query, ev, wit, nodes, root, alpha, log_size = v[0], v[1:5], v[5:9], v[9:11], v[11], v[12:16], v[16]
    assert fri_decommit(K, query, ev, wit, log_size, nodes, root) == 0
    assert K(10, [[1, query] + ev + wit + [log_size] + alpha])[0][0] == 0


@kat(W + "verifier.simf::test_verify_proof")
def _(v, K, key):
    """verifier.simf:62-108 builds this proof and never verifies it; the literal IS tests/golden/stwo_proof_test.json
    (word by word) and the KERNELS accept it in FIXTURE mode / reject it where the oracle says in LITERAL mode; the
    intermediates of the same run are the constants of the folding / layers KATs."""
    import stark_symphony_amd as ss
    p = ss.stwo_from_json(json.load(open(os.path.join(GOLDEN, "stwo_proof_test.json"))))
    flat = [int.from_bytes(bytes(r), "big") for r in p.roots]
    flat += [int(x) for x in p.trace_vals[0]] + [int.from_bytes(bytes(n), "big") for n in p.trace_paths[0]]
    flat += [int(x) for x in p.cp_vals[0]] + [int.from_bytes(bytes(n), "big") for n in p.cp_paths[0]]
    flat += [int(x) for x in p.oods_trace.reshape(-1)] + [int(x) for x in p.oods_cp.reshape(-1)]
    flat += [int.from_bytes(bytes(r), "big") for r in p.fri_roots] + [int(x) for x in p.last_layer]
    for l in range(3):
        flat += [int(x) for x in p.fri_witness[l, 0]]
        flat += [int.from_bytes(bytes(n), "big") for n in p.fri_paths[l][0]]
    flat.append(p.pow_nonce)
    assert flat == v
    b = K.ver.stwo_batch([p], verifier.MODE_FIXTURE)
    b.run()
    assert b.status().tolist() == [0]
    got = b.intermediates(0)
    fold = KATS[W + "fri/folding.simf::test_circle_fold"]["values"]
    line = KATS[W + "fri/folding.simf::test_line_fold"]["values"]
    # the kernels' own values on this proof against the literals of the reference's fold tests: the query, the value the
    # first fold takes in (fri_answer at query 8) and the first two fold alphas
    assert got["queries"].tolist() == [fold[0]]
    assert got["fri_answers"][0].tolist() == fold[1:5]
    assert got["fold_alphas"][0].tolist() == fold[10:14] and got["fold_alphas"][1].tolist() == line[10:14]
    lit = K.ver.stwo_batch([p], verifier.MODE_LITERAL)
    lit.run()
    assert lit.status().tolist() == [(7 << 24) | 1]


# ---------------------------------------------------------------------------------- macro tests (no device functions)
@kat(W + "macros/array_fold.simf::test_fold_arr_8", W + "macros/array_map.simf::test_map_arr_8",
     W + "macros/array_zip.simf::test_zip_arr_8", W + "fri/answers.simf::test_trace_evals_zip_arr_4",
     W + "fri/answers.simf::test_cp_evals_zip_arr_4")
def _(v, K, key):
    """macros/array_{fold,map,zip}.simf unroll fixed-size arrays; the kernels loop.  The literals are held only."""
    if key.endswith("fold_arr_8"):
        assert sum(v[1:9]) + v[9] == v[10]
    elif key.endswith("arr_8"):
        assert v[1:9] == list(range(1, 9))
    else:
        assert v[1:5] == [1, 2, 3, 4]


@pytest.mark.parametrize("key", sorted(KATS))
def test_reference_kat_on_the_device(K, key):
    assert key in HANDLERS, "no device replay for %s" % key
    HANDLERS[key]([int(x) for x in KATS[key]["values"]], K, key)

# (tests/test_oracle_kats.py::test_gpu_replay_has_a_handler_for_every_vector checks on the CPU that HANDLERS covers the file)
