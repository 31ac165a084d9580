/*
 * ss_verify_test.h -- tests and diagnosis only: what the parity suite drives that no caller of the verifier needs.
 *   - device replay of the reference's known-answer tests (ss_kat) and of single primitives (ss_selftest)
 *   - the GPU text reader alone, and the scalar statement of its rule
 *   - the workspace layout behind ss_stwo_read_intermediates
 */
#ifndef SS_VERIFY_TEST_H
#define SS_VERIFY_TEST_H

#include "ss_verify.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Would the text entry points read this text on the GPU (1) or hand it to the host reader (0)?  fmt is SS_TEXT_JSON,
 * SS_TEXT_WIT, SS_TEXT_JSON_SHARED (record_out: the per-query record it expands to) or SS_TEXT_JSON_MINIMAL (record_out:
 * ss_stwo_minimal_max_words words, the minimal record in CAPACITY form -- the fixed words, then every list at the base it
 * has when all lists have their largest length, the first n entries of each filled -- which is what the GPU reader writes
 * for such texts, csrc/ss_text.h).  Scalar statement of the GPU reader's rule; when it returns 1 and record_out is not
 * NULL, record_out holds the record.  No GPU involved.                                                                */
int ss_stwo_text_is_canonical(const ss_stwo_cfg *cfg, const char *text, size_t len, int fmt, uint32_t *record_out);
int ss_s101_text_is_canonical(const char *text, size_t len, int fmt, uint32_t *record_out);
int ss_stwo_minimal_from_capacity(const ss_stwo_cfg *cfg, const uint32_t *capacity, uint32_t *minimal_out, size_t cap_words,
                                  size_t *words_out);
int ss_stwo_minimal_to_capacity(const ss_stwo_cfg *cfg, const uint32_t *minimal, size_t words, uint32_t *capacity_out);

/* The GPU reader alone: n texts of format fmt -> records_host (n * ss_stwo_record_words words; SS_TEXT_JSON_SHARED: read
 * into shared records and expanded, all on the GPU; SS_TEXT_JSON_MINIMAL: n * ss_stwo_minimal_max_words words, capacity
 * form) and outcome_host[i] = 0 (canonical: record i written by the GPU) or 1 (left to the host reader; record i
 * unspecified).  Synchronous; outcome equals ss_*_text_is_canonical.  stark101: records of shape {10, 13}.            */
int ss_stwo_read_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts, const size_t *lens,
                       int fmt, uint32_t *records_host, uint32_t *outcome_host);
int ss_s101_read_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                       uint32_t *records_host, uint32_t *outcome_host);

/* Word offsets inside the workspace of a stwo pass, for callers that read it themselves. */
typedef struct ss_stwo_ws_layout {
    uint64_t np, nip;                 /* proofs / instances padded to 64                        */
    uint64_t ctx, alpha, leaf;        /* section word offsets: ctx[w][np], alpha[proof][n_pow][4],
                                         leaf[layer][8][nip] (even, odd member of the leaf pair) */
    uint64_t total_words;
    uint32_t c_queries, c_p, c_p2, c_fold, c_m1, n_pow;   /* ctx word indices */
    /* pair memoisation: levels below the root it covers (0 = off), and -- when the query count divides 64 -- the
     * per-query plan the query kernel leaves for the Merkle kernel's byte compares: plan[instance][4 words], byte
     * d-1 of words 0..1 = the query of the proof that leads this query's position at depth d (root = 0; possibly
     * itself), of words 2..3 = the one that leads the sibling position, 0xff = none.  has_plan = 0: no such section. */
    uint32_t top_levels, has_plan;
    uint64_t plan;
} ss_stwo_ws_layout;
int ss_stwo_ws_layout_of(const ss_stwo_cfg *cfg, size_t n, ss_stwo_ws_layout *out);

/* Device self-test of the primitives: runs `op` over `n` inputs.
 *   op 0  sha256 of 64-byte messages: in 16 words/item, out 8 words/item
 *   op 1  m31: in (a, b) -> out (add, sub, mul, inv(a) or 0xffffffff when a == 0)
 *   op 2  qm31: in (a[4], b[4]) -> out (mul[4], inv(a)[4] or all-ones on abort)
 *   op 3  circle point of index: in idx -> out (x, y)
 *   op 4  stark101 field: in (a, b) -> out (add, sub, mul, div(a,b) or 0xffffffff on abort)
 *   op 5  lazily reduced M31 forms on words in [0, P]: in (a[4], b[4]) -> out 16 words
 *         (a*b [4], a*a [4], a*(0 + im(b) u) [4], then for x = a[0] mod P, y = b[0] mod P:
 *         x+y, x-y, x*y, and (a[1] * 2^32 + b[1]) mod P)
 *   op 6  the asserts behind the FRI layer loop as the query kernel evaluates them (fri/verify.simf:124-128,
 *         fri/layers.simf:73-78): in (mode, lde_log, n_layers, query, folded position, folded value[4], last layer[4])
 *         -> out the first failing status code, 0 = none (stages 8 / 9 of the stwo status codes)            */
int ss_selftest(ss_ctx *ctx, int op, size_t n, const uint32_t *in_host, uint32_t *out_host);

/* Device replay of the reference's known-answer tests -- the `fn test_...` bodies of every .simf file under stark101/src and
 * stwo-verifier/src, SURVEY.md Appendix A -- (csrc/ss_kat.hip): one reference function per item, evaluated ON THE GPU
 * through the device functions the kernels are built from; tests/test_gpu_kats.py feeds the literals of the reference's
 * `fn test_*` bodies (tests/golden/kats.json) and compares with the expected literals directly.
 * in_words / out_words = n x the op's widths.  Hashes are 8 words (word j = big-endian bytes 4j..4j+3).
 *   op 0  (97 -> 8)   SHA-256 of in[0] <= 96 big-endian words in[1..]: sha256, sha256_32, sha256_pair, the leaf hashers
 *                     (hasher.simf:34-104), channel_mix_256 / _mix_oods_evals as digest || values
 *   op 1  (267 -> 9)  merkle_verify_32: family (0 stark101 merkle.simf:22-43 | 1 stwo :22-44), auth, len, leaf[8], root[8],
 *                     path[31][8] -> rc (0 | 1 `path == 1` fails | 2 root differs), computed root[8]
 *   op 2  (34 -> 17)  stwo channel (channel.simf:31-172): digest[8], counter, k, payload[24] -> digest', counter', result[8];
 *                     k = 0 two draw_qm31 | 1 draw_qm31_point | 2 mix_u256 | 3 check_proof_of_work (nonce hi lo, target hi
 *                     lo; result[1] = reverse_bytes_32(payload[4])) | 4 draw_queries_8 (mask) | 5 evals_commit (3 roots) |
 *                     6 mix_u256 + draw_qm31 | 7 mix_line_poly (4 words)
 *   op 3  (4 -> 10)   cm31: a, b -> add, sub, mul, a / b, inv(a)            (all-ones where the reference aborts)
 *   op 4  (8 -> 16)   qm31: a, b -> add, sub, a * m31(b[0]), a * cm31(b[0], b[1])
 *   op 5  (4 -> 4)    m31 points: p, q -> p + q, 2p
 *   op 6  (3 -> 9)    a, b, log -> bit_reverse_position(a, log), index add / mul / neg(a), circle_domain(log)[3],
 *                     circle position a -> point index, line position a -> x coordinate
 *   op 7  (18 -> 16)  qm31 points: P, Q, m -> P + Q, P + m (qm31_point_add_m31_point)
 *   op 8  (93 -> 18)  log_size, P[8], 4 columns[16], alpha[4], 16 cp parts[64] -> vanishing_poly_eval[4],
 *                     eval_composition_poly[4], composition_poly_eval_from_decomposed[4], .._from_partitions(parts 0..3)[4],
 *                     abort flag, 0
 *   op 9  (19 -> 19)  deep/quotients.simf: sample point[8], value[4], alpha_i[4], domain point[2], queried value ->
 *                     denominator inverse[2], interpolant coefficients[12], nominator[4], abort flag
 *   op 10 (15 -> 5)   kind (0 circle_fold | 1 line_fold), position, f_p[4], f_neg_p[4], log_size, alpha[4] -> abort flag, folded[4]
 *   op 11 (12 -> 10)  stark101: k, args[11]; k = 0 field (a, b -> add sub mul div exp) | 1 channel_draw_32 (state[8], max ->
 *                     value, state') | 2 read_coefficients (state -> 3 draws, 7 state words) | 3 calc_x / eval_p0 (idx, x, f_x)
 *                     | 4 eval_cp (a0 a1 a2 f_x f_gx f_ggx x) | 5 fri_eval_cp_next (cpa cpb x beta) | 6 compute_auth_path
 *                     (idx, domain) | 7 channel_mix_32 (state[8], m -> state')                                        */
int ss_kat(ss_ctx *ctx, int op, size_t n, const uint32_t *in_host, size_t in_words, uint32_t *out_host, size_t out_words);

#ifdef __cplusplus
}
#endif
#endif /* SS_VERIFY_TEST_H */
