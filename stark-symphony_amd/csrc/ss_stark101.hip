// stark101 (FibonacciSq over p = 3 * 2^30 + 1) batch verifier kernels for gfx950.
//
// Reference path: `verify_proof`, stark101/src/verifier.simf:24-42.
//
//   s101_transcript_kernel   one lane per PROOF: the Fiat-Shamir chain (verifier.simf:27-33,
//        fri.simf:37-54), the composition polynomial (air.simf:58-101) and the FRI fold chain
//        (fri.simf:58-91 minus its Merkle checks).  There is a single query per proof, so the
//        algebra is per proof too.
//   s101_merkle_kernel       one lane per Merkle CHAIN (3 trace + 2 per FRI layer per proof),
//        one wavefront per 64 chains of the same kind (merkle.simf:22-43).  87 % of the work.
//
// The three `channel_mix_32(state, p_ev)` calls of air.simf:42 only update a channel state
// that verify_proof never reads again (verifier.simf:35 binds it and drops it), so they do
// not influence accept/reject and are not executed here.
#include <hip/hip_runtime.h>

#include "ss_fields.h"
#include "ss_layout.h"
#include "ss_s101.h"
#include "ss_sha256.h"

namespace ss {

__global__ void __launch_bounds__(64)
s101_transcript_kernel(S101Layout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                       uint32_t *__restrict__ status, uint32_t *__restrict__ accept_count)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= lay.n) return;
    const uint32_t np = lay.np;
    const uint32_t *head = batch + lay.off_head;
    const uint32_t *leaf = batch + lay.off_leaf;
    auto H = [&](uint32_t w) { return head[(size_t)w * np + p]; };
    auto EV = [&](uint32_t t) { return leaf[(size_t)t * np + p]; };
    uint32_t fail = 0xffffffffu;
    auto FAIL = [&](uint32_t code) { fail = code < fail ? code : fail; };
    uint32_t nl = H(lay.h_nlayers);
    if (nl > lay.ML) nl = lay.ML;

    // verifier.simf:27  state = sha256(p_mt_root)
    Dig101 st;
    for (int i = 0; i < 8; i++) st.v[i] = H(lay.h_root + i);
    st = s101_hash_state<0>(st, 0);
    // :29  fibsquare_read_coefficients (air.simf:30-35)
    const uint32_t a0 = s101_draw<S101_P>(st), a1 = s101_draw<S101_P>(st), a2 = s101_draw<S101_P>(st);
    // :31  fri_read_commitments_32 (fri.simf:37-54)
    for (uint32_t i = 0; i < nl; i++) {
        W16101 w;
#pragma unroll
        for (int j = 0; j < 8; j++) { w.v[j] = st.v[j]; w.v[8 + j] = H(lay.h_layer + 9 * i + j); }
        Dig101 iv;
        sha_iv(iv.v);
        st = s101_compress_pad64_call(s101_compress_call(iv, w));  // channel_mix_256
        const uint32_t random = s101_draw<S101_P>(st);
        if (random != H(lay.h_layer + 9 * i + 8)) FAIL(s101_code(1, i));
    }
    const uint32_t last = H(lay.h_last);
    st = s101_hash_state<1>(st, last);  // channel_mix_32(state, last_layer)
    // per-stage values for ss_s101_read_intermediates (the reference's counterpart: what prover_test.py:32-104 recomputes
    // and simfony's dbg! tracker prints, simfony-cli/src/tracker.rs:48-80)
    uint32_t *ints = ws + lay.ws_int + p;
    auto INT = [&](uint32_t row, uint32_t v) { ints[(size_t)row * np] = v; };
#pragma unroll
    for (int i = 0; i < 8; i++) INT(5 + lay.ML + 1 + i, st.v[i]);
    // :33  random query
    const uint32_t idx = s101_draw<8192u>(st);
    ws[p] = idx;
    INT(0, a0); INT(1, a1); INT(2, a2);

    // :37-39  x = 5 * h^idx, composition polynomial (air.simf:58-101)
    // Every divisor of the path is a canonical field element (an output of sub_mod / mul_mod),
    // so div_mod aborts exactly when it is 0 and otherwise multiplies by the unique inverse
    // (see f101_div).  The four independent divisors x-1, x-g^1022, x^1024-1 and x share ONE
    // inversion (Montgomery's trick); the fold divisors 2*x^(2^i) are inverted as
    // inv2 * (1/x)^(2^i), and "/2" is a multiplication by (p+1)/2.
    const uint32_t f_x = EV(0), f_gx = EV(1), f_ggx = EV(2);
    const uint32_t x = f101_mul(5u, f101_pow(1734477367u, idx));
    const uint32_t d0 = f101_sub(x, 1), d1 = f101_sub(x, 2450347685u);
    const uint32_t d2 = f101_sub(f101_pow(x, 1024), 1);
    if (d0 == 0) FAIL(s101_code(3, 0));
    if (d1 == 0) FAIL(s101_code(3, 1));
    if (d2 == 0) FAIL(s101_code(3, 2));
    const uint32_t e0 = d0 ? d0 : 1, e1 = d1 ? d1 : 1, e2 = d2 ? d2 : 1, e3 = x ? x : 1;
    const uint32_t p01 = f101_mul(e0, e1), p012 = f101_mul(p01, e2), p0123 = f101_mul(p012, e3);
    uint32_t r = f101_pow(p0123, S101_P - 2);
    const uint32_t x_inv = f101_mul(r, p012);
    r = f101_mul(r, e3);
    const uint32_t i2 = f101_mul(r, p01);
    r = f101_mul(r, e2);
    const uint32_t i1 = f101_mul(r, e0), i0 = f101_mul(r, e1);
    uint32_t cp;
    {
        const uint32_t p0 = f101_mul(f101_sub(f_x, 1), i0);
        const uint32_t p1 = f101_mul(f101_sub(f_x, 2338775057u), i1);
        const uint32_t num0 = f101_sub(f_ggx, f101_add(f101_mul(f_x, f_x), f101_mul(f_gx, f_gx)));
        const uint32_t num1 = f101_mul(f101_mul(f101_sub(x, 2342081930u), f101_sub(x, 2450347685u)),
                                       f101_sub(x, 532203874u));
        const uint32_t p2 = f101_mul(f101_mul(num0, num1), i2);
        cp = f101_add(f101_add(f101_mul(p0, a0), f101_mul(p1, a1)), f101_mul(p2, a2));
    }
    INT(3, x); INT(4, cp);
    // :41  fri_verify_32 without the Merkle checks (fri.simf:58-62,74-91)
    constexpr uint32_t kInv2 = (S101_P + 1) / 2;
    uint32_t xx = x, xx_inv = x_inv, cur = cp;
    for (uint32_t i = 0; i < nl; i++) {
        const uint32_t cpa = EV(3 + 2 * i), cpb = EV(4 + 2 * i), beta = H(lay.h_layer + 9 * i + 8);
        INT(5 + i, cur);
        if (cur != cpa) FAIL(s101_code(4, 4 * i + 0));
        if (f101_mul(xx, 2) == 0) FAIL(s101_code(4, 4 * i + 3));
        const uint32_t op0 = f101_mul(f101_add(cpa, cpb), kInv2);
        const uint32_t op1 = f101_mul(f101_sub(cpa, cpb), f101_mul(kInv2, xx_inv));
        cur = f101_add(op0, f101_mul(op1, beta));
        xx = f101_mul(xx, xx);
        xx_inv = f101_mul(xx_inv, xx_inv);
    }
    INT(5 + nl, cur);
    if (cur != last) FAIL(s101_code(5, 0));
    // the first kernel of a pass: it also RESETS the pass's status words and accept count (no memset dispatches: a stark101 x
    // 4 096 pass is 60 us, each memset in front of it cost 6 %, profiles/r06_reset_in_kernel_ab.txt)
    status[p] = fail;
    if (p == 0 && accept_count) *accept_count = 0;
}

__device__ __forceinline__ uint32_t jet_div(uint32_t a, uint32_t b) { return b ? a / b : 0; }
__device__ __forceinline__ uint32_t jet_mod(uint32_t a, uint32_t b) { return b ? a % b : a; }

__global__ void __launch_bounds__(256)
s101_merkle_kernel(S101Layout lay, const uint32_t *__restrict__ batch, const uint32_t *__restrict__ ws,
                   uint32_t *__restrict__ status)
{
    const uint32_t tiles_per_type = lay.np >> 6;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t type = tile / tiles_per_type;  // wave-uniform
    if (type >= lay.n_types) return;
    const uint32_t g = tile - type * tiles_per_type;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p = g * 64 + lane;
    const uint32_t np = lay.np;
    const uint32_t *head = batch + lay.off_head;
    bool live = p < lay.n;
    const uint32_t pc = live ? p : 0;
    const uint32_t ev = batch[lay.off_leaf + (size_t)type * np + pc];
    uint32_t len = batch[lay.off_len + (size_t)type * np + pc];
    if (len > lay.PM) len = lay.PM;
    const uint32_t idx = ws[pc];
    uint32_t auth, root_w, code;
    if (type < 3) {
        // read_eval_checked (air.simf:38-43): auth = idx + 8k + 8192
        auth = idx + 8 * type + 8192u;
        root_w = lay.h_root;
        code = s101_code(2, type);
    } else {
        // compute_auth_path (fri.simf:66-71) with domain_size = 8192 / 2^layer (divide_32 chain)
        const uint32_t i = (type - 3) >> 1, b = (type - 3) & 1;
        const uint32_t dom = i < 14 ? (8192u >> i) : 0u;
        auth = b ? jet_mod(idx + jet_div(dom, 2), dom) + dom : jet_mod(idx, dom) + dom;
        root_w = lay.h_layer + 9 * i;
        code = s101_code(4, 4 * i + 1 + b);
        uint32_t nl = head[(size_t)lay.h_nlayers * np + pc];
        live = live && i < nl;
    }
    if (!live) len = 0;
    // wave-uniform trip count: the longest chain of the tile
    uint32_t maxlen = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t other = __shfl_xor(maxlen, o);
        maxlen = other > maxlen ? other : maxlen;
    }

    uint32_t node[8];
    {
        uint32_t m[1] = {ev};
        sha256_words<1>(m, node);  // sha256_32 (sha256.simf:18)
    }
    const uint4 *tp = reinterpret_cast<const uint4 *>(batch + lay.off_path + type * lay.path_stride) +
                      ((size_t)g * lay.PM * 2) * 64 + lane;
    uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0;
    if (maxlen) { s0 = tp[0]; s1 = tp[64]; }
    for (uint32_t lvl = 0; lvl < maxlen; lvl++) {
        uint4 n0 = s0, n1 = s1;
        if (lvl + 1 < maxlen) {
            n0 = tp[(size_t)(lvl + 1) * 128];
            n1 = tp[(size_t)(lvl + 1) * 128 + 64];
        }
        const uint32_t sib[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const bool right = auth & 1;
        uint32_t w[16], nxt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            w[j] = right ? sib[j] : node[j];
            w[8 + j] = right ? node[j] : sib[j];
        }
        sha_iv(nxt);
        sha256_compress(nxt, w);
        sha256_compress_pad64(nxt);
        if (lvl < len) {
#pragma unroll
            for (int j = 0; j < 8; j++) node[j] = nxt[j];
            auth = jet_div(auth, 2);
        }
        s0 = n0; s1 = n1;
    }
    if (!live) return;
    bool same = true;
#pragma unroll
    for (int j = 0; j < 8; j++) same &= node[j] == head[(size_t)(root_w + j) * np + p];
    if (!same) atomicMin(&status[p], code);  // assert!(eq_256(computed_root, root)) merkle.simf:42
}

}  // namespace ss
