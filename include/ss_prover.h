/*
 * ss_prover.h -- C ABI of the GPU building blocks of the wide-Fibonacci circle-STARK prover
 * (libss_verify.so, same library as the verifier).
 *
 * SURVEY.md 8(f) row 1.  The reference repository has NO stwo prover: its two proofs
 * (stwo-verifier/tests/data/proof.json, proof_test.json) came from an external forked stwo.
 * tools/stwo_prover.py re-derives that prover (numpy) and reproduces both files byte for
 * byte; the kernels below are its data-parallel steps on the GPU, driven by
 * stark-symphony_amd/prover.py, whose output must equal the numpy prover's byte for byte
 * (tests/test_gpu_prover.py) -- and therefore the reference's fixtures.
 *
 * Verifier-side definitions the kernels mirror: circle domain / line domain
 * (stwo-verifier/src/groups/circle_domain.simf:17-43, line_domain.simf:18-31), leaf and node
 * hashes (hasher.simf:27-104), wide-Fibonacci constraints (constraints/wide_fibonacci.simf:24-62),
 * DEEP quotients (deep/quotients.simf:15-44), folds (fri/folding.simf:15-41), proof of work
 * (pow.simf:22-36).
 *
 * All pointers are device memory on the context's GPU; every call is asynchronous on `stream`.
 * Arrays of field elements are uint32 words (canonical, < 2^31 - 1).  A "column" is 2^m words in
 * the bit-reversed storage order of the canonic coset: value[i] = f(domain.at(bitrev(i))).
 * QM31 arrays are [n][4] words.  Hashes are [n][8] stored words (include/ss_verify.h).
 */
#ifndef SS_PROVER_H
#define SS_PROVER_H

#include <stddef.h>
#include <stdint.h>

#include "ss_verify.h"

#ifdef __cplusplus
extern "C" {
#endif

/* row r of the trace = [1, r + seed_term, c2, c3, ...], c_k = c_{k-1}^2 + c_{k-2}^2;
 * cols_out[k][r], 2^n_log rows.                                                         */
int ss_p_trace(ss_ctx *ctx, uint32_t n_log, uint32_t n_cols, uint32_t seed_term,
               uint32_t *cols_out, void *stream);

/* Twiddles of the canonic coset of log size m, 2^m words each (layer 0 first):
 *   layer 0: y of storage pair h (2^(m-1) words); layer i >= 1: pi^(i-1)(x_{h 2^i}) (2^(m-1-i)
 *   words, at word offset 2^m - 2^(m-i)); the last word is unused.  itw = element-wise inverses.
 *   hx_out (may be NULL): x of storage pair h, 2^(m-1) words.                               */
int ss_p_twiddles(ss_ctx *ctx, uint32_t m, uint32_t *tw_out, uint32_t *itw_out, uint32_t *hx_out,
                  void *stream);

/* In-place circle FFT of `ncols` columns of 2^m words (stride 2^m): inverse = evaluations ->
 * coefficients in the basis y^k0 x^k1 pi(x)^k2 ... (scaled by 2^-m) with tw = the INVERSE
 * twiddles, forward = the opposite with the plain twiddles.                                 */
int ss_p_fft(ss_ctx *ctx, uint32_t m, uint32_t ncols, uint32_t *data, const uint32_t *tw,
             int inverse, void *stream);
/* Low-degree extension: `ncols` polynomials of 2^k coefficients each (coefs: stride 2^k) -> their evaluations on the
 * canonic coset of log size m >= k (out: stride 2^m); tw = the plain twiddles of size m.  Equal to ss_p_fft (forward)
 * of the coefficients followed by zeros, without writing or reading the zeros.                                     */
int ss_p_lde(ss_ctx *ctx, uint32_t k, uint32_t m, uint32_t ncols, const uint32_t *coefs, uint32_t *out,
             const uint32_t *tw, void *stream);

/* out[i] = H(be4(cols[0][i]) || ... || be4(cols[w-1][i])), cols stride = col_stride words. */
int ss_p_hash_rows(ss_ctx *ctx, uint32_t hash, size_t n, uint32_t w, const uint32_t *cols,
                   size_t col_stride, uint32_t *out, void *stream);
/* out[i] = H(be4 a, b, c, d of vals[i]) for QM31 rows. */
int ss_p_hash_qm31(ss_ctx *ctx, uint32_t hash, size_t n, const uint32_t *vals, uint32_t *out,
                   void *stream);
/* Merkle tree over n = 2^k leaf hashes: levels[0 .. 2n-1) = leaves (n), then n/2 parents, ...,
 * root last; node = H(left || right).  `levels` already holds the leaves in [0, n).          */
int ss_p_merkle(ss_ctx *ctx, uint32_t hash, size_t n_leaves, uint32_t *levels, void *stream);
/* out[i] = H(leaf[i] || leaf[i]), n hashes: the first node level of a tree whose leaves come in equal pairs (the 16
 * composition columns are polynomials in x alone, so the two points (x, +-y) of a storage pair have the same row and
 * the same leaf hash): the 2n duplicated leaves need not exist in memory.                                          */
int ss_p_merkle_dup(ss_ctx *ctx, uint32_t hash, size_t n, const uint32_t *leaf, uint32_t *out, void *stream);

/* Composition polynomial on the canonic coset of log size n+1 (hx_c = its pair x coordinates):
 * F = (sum_k alpha^(N-1-k) (c_k - c_{k-1}^2 - c_{k-2}^2)) / pi^(n-1)(x), out[coord][i].     */
int ss_p_composition(ss_ctx *ctx, uint32_t n_log, uint32_t n_cols, const uint32_t *ev,
                     const uint32_t *hx_c, const uint32_t alpha[4], uint32_t *out, void *stream);

/* out[0..4) = sum_k coeffs[k] y^k0 x^k1 pi(x)^k2 ... ; factors_host = the m QM31 factors
 * (y, x, pi(x), ...) in HOST memory; scratch holds 3 * 2^m words.                          */
int ss_p_eval_at_point(ss_ctx *ctx, uint32_t m, const uint32_t *coeffs, const uint32_t *factors_host,
                       uint32_t *scratch, uint32_t *out, void *stream);

/* The same for `ncols` columns (stride col_stride words) at ONE point: m launches in all, not m
 * per column; scratch holds ncols * 3 * 2^m words, out[ncols][4].                              */
int ss_p_eval_at_point_batch(ss_ctx *ctx, uint32_t m, uint32_t ncols, const uint32_t *coeffs,
                             size_t col_stride, const uint32_t *factors_host, uint32_t *scratch,
                             uint32_t *out, void *stream);

/* DEEP quotient row of every LDE position (two batches: trace columns sampled at P, the 16
 * composition columns at 2P; row = b1 alpha^16 + b2).  hx_hy = pair x then pair y of the LDE
 * coset (2^(lde_log-1) words each); bcoef = the b line coefficient of every column already
 * multiplied by alpha^i ((n_cols + 16) QM31); sums_alpha16 = A1, C1, A2, C2, alpha^16 (host).
 * cp_log = lde_log: cp_lde holds one value per LDE position (stride 2^lde_log); cp_log = lde_log - 1: the
 * composition columns depend on x only, so the two positions of a storage pair (x, +-y) carry the same
 * value and cp_lde holds one per pair (stride 2^(lde_log-1)).                                          */
int ss_p_quotients(ss_ctx *ctx, uint32_t lde_log, uint32_t n_cols, const uint32_t *trace_lde,
                   const uint32_t *cp_lde, uint32_t cp_log, const uint32_t *hx_hy, const uint32_t *bcoef,
                   const uint32_t p[8], const uint32_t p2[8], const uint32_t sums_alpha16[20],
                   uint32_t *out, void *stream);

/* One FRI fold: out[i] = (v[2i] + v[2i+1]) + alpha * (v[2i] - v[2i+1]) * coord_inv[i].       */
int ss_p_fri_fold(ss_ctx *ctx, size_t n_out, const uint32_t *in, const uint32_t *coord_inv,
                  const uint32_t alpha[4], uint32_t *out, void *stream);

/* The FRI commit loop without host round trips.  state_dev = the Fiat-Shamir channel in device
 * memory: 8 stored digest words + the draw counter (channel.simf:31-172).
 * ss_p_channel_fri_layer = fri_layer_commit (fri/commit.simf:34-46): digest <- H(digest || root),
 * alpha <- channel_draw_qm31; root_dev points at the tree's root node (8 stored words), which is
 * also copied to root_out_dev; alpha_out_dev receives the 4 words ss_p_fri_fold_dev reads.      */
int ss_p_channel_fri_layer(ss_ctx *ctx, uint32_t hash, uint32_t *state_dev, const uint32_t *root_dev,
                           uint32_t *alpha_out_dev, uint32_t *root_out_dev, void *stream);
int ss_p_fri_fold_dev(ss_ctx *ctx, size_t n_out, const uint32_t *in, const uint32_t *coord_inv,
                      const uint32_t *alpha_dev, uint32_t *out, void *stream);

/* Smallest nonce >= start with LE64(last 8 bytes of H(digest || be8(nonce))) < target, searched
 * in [start, start + count); *nonce_out = UINT64_MAX if none.  digest: 8 stored words.       */
int ss_p_pow(ss_ctx *ctx, uint32_t hash, const uint32_t digest[8], uint64_t target, uint64_t start,
             uint64_t count, uint64_t *nonce_out_dev, void *stream);

/* ------------------------------------------------------------------ stark101 prover (8f row 2)
 * Data-parallel steps of stark101/scripts/fibsquare/prover.py:25-171 (csrc/ss_s101_prover.hip);
 * the host side is stark-symphony_amd/prover101.py.  Sizes are the reference's: 1023-step trace,
 * size-1024 subgroup, 8192-point coset 5 * <h>, natural order.                                 */

/* trace_out[1023]: a_0 = 1, a_1 = seed, a_i = a_{i-2}^2 + a_{i-1}^2 (prover.py:25-30; the reference's
 * seed is 3141592); coef_out[1024]: the interpolant of degree < 1023 through (g^i, a_i) (prover.py:111). */
int ss_p101_trace_poly(ss_ctx *ctx, uint32_t seed, uint32_t *trace_out, uint32_t *coef_out, void *stream);
/* out[j] = p(5 h^j), j < 8192 (prover.py:112).                                               */
int ss_p101_lde(ss_ctx *ctx, const uint32_t *coef, uint32_t *out, void *stream);
/* Composition polynomial on the coset from the trace LDE (prover.py:42-65,117-118); alphas = the
 * three channel coefficients (host); claim = a_1022 (2338775057 for the reference's seed).     */
int ss_p101_composition(ss_ctx *ctx, const uint32_t *p_ev, const uint32_t alphas[3], uint32_t claim,
                        uint32_t *out, void *stream);
/* FRI layer `layer` (len = 8192 >> layer evaluations) -> the next layer (len / 2) with the
 * channel's beta (prover.py:68-88).                                                           */
int ss_p101_fold(ss_ctx *ctx, uint32_t layer, uint32_t len, uint32_t beta, const uint32_t *in,
                 uint32_t *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SS_PROVER_H */
