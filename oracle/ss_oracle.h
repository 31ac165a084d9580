/*
 * ss_oracle.h -- CPU restatement of the stark-symphony verifier hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (stark-symphony_amd/) never links or calls it.
 *
 * It restates, in plain sequential C, the SimplicityHL programs
 *   /root/reference/stark101/src/{field,channel,sha256,merkle,air,fri,verifier}.simf
 *   /root/reference/stwo-verifier/src/ (every .simf below it)
 * function by function (each C function cites the file:line it follows).  The
 * Simplicity jets those programs call (add_32, multiply_32, modulo_64, the
 * sha_256_ctx_8_* family ...) live in the un-vendored dependency
 * simplicity-sys 0.4.0 @ m-kus/rust-simplicity 7c43d07c (Cargo.lock:1406-1477);
 * they are restated from their published semantics (wrapping unsigned
 * arithmetic, FIPS 180-4 SHA-256) and pinned by the reference's own `fn test_*`
 * known-answer tests (tests/golden/kats.json) plus the three end-to-end proofs.
 *
 * Parity status: stark101 fully pinned (leaf KATs + the reference Python prover's
 * proof == verifier.simf:44-388 literal).  stwo leaf functions pinned by KATs;
 * stwo verify_proof end-to-end is NOT executed by any reference test
 * (verifier.simf:62-108 builds the proof and never calls verify_proof) -- the
 * `SO_MODE_LITERAL` path follows the .simf text, the `SO_MODE_FIXTURE` path is
 * what the reference's two proof fixtures actually satisfy.
 */
#ifndef SS_ORACLE_H
#define SS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ sha256 */
void so_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
/* Blake2s-256 (RFC 7693): the hash of the BASELINE.json "Blake2s Merkle" variant.  The
 * reference has no Blake2s, so this variant's parity is UNPINNED: only the
 * hash itself is checked against RFC 7693 vectors.  Same byte strings, other hash function. */
void so_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]);
void so_set_hash(int kind); /* 0 sha256, 1 blake2s: for calling stwo leaf functions one by one */
/* number of compression-function calls since the last reset (for work counts) */
uint64_t so_sha256_blocks(void);
void so_sha256_blocks_reset(void);

/* ----------------------------------------------------------------- stark101 */
#define SO_S101_P 3221225473u
#define SO_MAX_LIST 31 /* SimplicityHL List<T,32> holds at most 31 elements */

uint32_t so_s101_add_mod(uint32_t a, uint32_t b);
uint32_t so_s101_sub_mod(uint32_t a, uint32_t b);
uint32_t so_s101_mul_mod(uint32_t a, uint32_t b);
int so_s101_div_mod(uint32_t a, uint32_t b, uint32_t *out); /* 0 ok, 1 abort */
uint32_t so_s101_exp_mod(uint32_t a, uint32_t b);
uint32_t so_s101_reduce_256_mod_32(const uint8_t v[32], uint32_t modulo);
uint32_t so_s101_channel_draw_32(uint8_t state[32], uint32_t max);
void so_s101_channel_mix_32(uint8_t state[32], uint32_t input);
void so_s101_channel_mix_256(uint8_t state[32], const uint8_t input[32]);
int so_s101_merkle_verify(const uint8_t leaf[32], uint32_t auth_path,
                          const uint8_t *proof, uint32_t len,
                          const uint8_t root[32]);
uint32_t so_s101_calc_x(uint32_t idx);
int so_s101_eval_p0(uint32_t x, uint32_t f_x, uint32_t *out);
int so_s101_eval_cp(uint32_t x, uint32_t a0, uint32_t a1, uint32_t a2,
                    uint32_t f_x, uint32_t f_gx, uint32_t f_ggx, uint32_t *out);
int so_s101_fri_eval_cp_next(uint32_t cpa, uint32_t cpb, uint32_t x,
                             uint32_t beta, uint32_t *out);
void so_s101_compute_auth_path(uint32_t idx, uint32_t domain_size,
                               uint32_t *cpa_path, uint32_t *cpb_path);

typedef struct {
    uint32_t ev;
    uint32_t len;
    uint8_t path[SO_MAX_LIST][32]; /* leaf -> root order */
} so_s101_eval;

typedef struct {
    uint8_t root[32];
    uint32_t beta;
    so_s101_eval cpa;
    so_s101_eval cpb;
} so_s101_layer;

typedef struct {
    uint8_t root[32];
    so_s101_eval evals[3];
    uint32_t n_layers;
    so_s101_layer layers[SO_MAX_LIST];
    uint32_t last;
} so_s101_proof;

/* optional stage-level intermediates */
typedef struct {
    uint32_t alpha[3];
    uint32_t idx;
    uint32_t x;
    uint32_t cp;
    uint32_t fold[SO_MAX_LIST + 1]; /* cp value entering layer i; [n] = final */
    uint8_t state_after_commit[32];
} so_s101_trace;

/* status: 0 = ACCEPT, else (stage << 8) | sub of the first failing assert in the
 * reference's evaluation order:
 *   stage 1 beta mismatch           sub = layer            (fri.simf:43)
 *   stage 2 trace Merkle root       sub = k in 0..2        (air.simf:41 -> merkle.simf:42)
 *   stage 3 composition div abort   sub = 0 p0,1 p1,2 p2   (field.simf:46 via air.simf:63-80)
 *   stage 4 FRI layer               sub = 4*layer + {0 chain fri.simf:77,
 *                                     1 cpa Merkle :79, 2 cpb Merkle :80, 3 fold div :58-60}
 *   stage 5 last layer value        sub = 0                (fri.simf:90)          */
uint32_t so_s101_verify(const so_s101_proof *p, so_s101_trace *tr);

/* ------------------------------------------------------------------- stwo */
#define SO_M31_P 2147483647u

typedef struct { uint32_t a, b; } so_cm31;
typedef struct { uint32_t a, b, c, d; } so_qm31;
typedef struct { uint32_t x, y; } so_m31_point;
typedef struct { so_qm31 x, y; } so_qm31_point;

uint32_t so_m31_add(uint32_t a, uint32_t b);
uint32_t so_m31_neg(uint32_t a);
uint32_t so_m31_sub(uint32_t a, uint32_t b);
uint32_t so_m31_mul(uint32_t a, uint32_t b);
uint32_t so_m31_exp(uint32_t a, uint32_t b);
int so_m31_inv(uint32_t a, uint32_t *out);
so_cm31 so_cm31_add(so_cm31 a, so_cm31 b);
so_cm31 so_cm31_sub(so_cm31 a, so_cm31 b);
so_cm31 so_cm31_mul(so_cm31 a, so_cm31 b);
int so_cm31_inv(so_cm31 a, so_cm31 *out);
int so_cm31_div(so_cm31 a, so_cm31 b, so_cm31 *out);
so_qm31 so_qm31_add(so_qm31 a, so_qm31 b);
so_qm31 so_qm31_sub(so_qm31 a, so_qm31 b);
so_qm31 so_qm31_mul(so_qm31 a, so_qm31 b);
so_qm31 so_qm31_mul_m31(so_qm31 a, uint32_t b);
so_qm31 so_qm31_mul_cm31(so_qm31 a, so_cm31 b);
int so_qm31_inv(so_qm31 a, so_qm31 *out);
so_m31_point so_m31_point_add(so_m31_point a, so_m31_point b);
so_m31_point so_m31_point_dbl(so_m31_point a);
so_m31_point so_circle_point_index_to_m31_point(uint32_t index);
so_qm31_point so_qm31_point_add(so_qm31_point a, so_qm31_point b);
so_qm31_point so_qm31_point_add_m31_point(so_qm31_point a, so_m31_point b);
uint32_t so_bit_reverse_position(uint32_t position, uint8_t log_size);
uint32_t so_circle_point_index_add(uint32_t a, uint32_t b);
uint32_t so_circle_point_index_mul(uint32_t a, uint32_t b);
uint32_t so_circle_point_index_neg(uint32_t a);
void so_circle_domain(uint8_t log_size, uint32_t out[3]);
uint32_t so_circle_position_to_point_index(uint8_t log_size, uint32_t position);
uint32_t so_line_position_to_x_coord(uint8_t log_size, uint32_t position);

typedef struct { uint8_t digest[32]; uint32_t counter; } so_channel;
void so_channel_init(so_channel *s);
int so_channel_draw_qm31(so_channel *s, so_qm31 *out);
int so_channel_draw_qm31_point(so_channel *s, so_qm31_point *out);
void so_channel_mix_u256(so_channel *s, const uint8_t in[32]);
void so_channel_mix_u64(so_channel *s, uint64_t in);
void so_channel_draw_queries_8(so_channel *s, uint32_t mask, uint32_t out[8]);
uint32_t so_reverse_bytes_32(uint32_t v);
int so_check_proof_of_work(so_channel *s, uint64_t nonce, uint64_t target);
void so_hash_u32s(const uint32_t *vals, size_t n, uint8_t out[32]);
/* returns 0 ok, 1 path!=1, 2 root mismatch (merkle.simf:42-43) */
int so_stwo_merkle_verify(const uint8_t leaf[32], uint32_t auth_path,
                          const uint8_t *proof, uint32_t len,
                          const uint8_t root[32]);
int so_evals_commit(so_channel *s, const uint8_t roots[3][32], so_qm31 *cp_alpha);
so_qm31 so_composition_poly_eval_from_partitions(const so_qm31 p[4]);
so_qm31 so_composition_poly_eval_from_decomposed(const so_qm31 d[16], so_qm31_point pt);
so_qm31 so_vanishing_poly_eval(uint8_t log_size, so_qm31_point pt);
int so_eval_composition_poly(uint8_t log_size, so_qm31_point pt,
                             const so_qm31 *trace_evals, uint32_t n_cols,
                             so_qm31 alpha, so_qm31 *out);
void so_channel_mix_oods_evals(so_channel *s, const so_qm31 *trace, uint32_t n_cols,
                               const so_qm31 cp[16]);
int so_deep_quotient_denominator_inverse(so_qm31_point sample, so_m31_point q, so_cm31 *out);
void so_deep_quotient_interpolant_coefficients(so_qm31_point sample, so_qm31 value,
                                               so_qm31 alpha_i, so_qm31 out[3]);
so_qm31 so_deep_quotient_nominator(const so_qm31 coeffs[3], so_m31_point q, uint32_t value);
int so_circle_fold(uint32_t position, so_qm31 f_p, so_qm31 f_neg_p, uint8_t log_size_ex,
                   so_qm31 alpha, so_qm31 *out);
int so_line_fold(uint32_t position, so_qm31 f_p, so_qm31 f_neg_p, uint8_t log_size_ex,
                 so_qm31 alpha, so_qm31 *out);

typedef struct {
    uint32_t len;
    const uint8_t *nodes; /* len x 32 bytes, leaf -> root order */
} so_path;

typedef struct {
    uint32_t n_cols;       /* NUM_COLUMNS      config.simf:14 */
    uint32_t trace_log;    /* TRACE_LOG_SIZE   config.simf:17,35 */
    uint32_t lde_log;      /* LDE_LOG_SIZE     config.simf:21,39 */
    uint32_t n_queries;    /* NUM_FRI_QUERIES  config.simf:25,43 */
    uint32_t n_layers;     /* NUM_FRI_LAYERS   config.simf:29,47 (inner layers) */
    uint64_t pow_target;   /* POW_TARGET_64    config.simf:32,51 */
    uint32_t hash;         /* 0 SHA-256 (reference), 1 Blake2s-256 (extension, unpinned) */
} so_stwo_cfg;

typedef struct {
    uint8_t roots[3][32];        /* Commitments           evals/commit.simf:16 */
    const so_qm31 *oods_trace;   /* [n_cols]              deep/oods.simf:20 */
    so_qm31 oods_cp[16];
    const uint32_t *trace_vals;  /* [n_queries][n_cols]   evals/verify.simf:20-33 */
    const uint32_t *cp_vals;     /* [n_queries][16] */
    const so_path *trace_paths;  /* [n_queries] */
    const so_path *cp_paths;     /* [n_queries] */
    const uint8_t *fri_roots;    /* [(1+n_layers)][32]    fri/commit.simf:19 */
    so_qm31 last_layer;
    const so_qm31 *fri_witness;  /* [(1+n_layers)][n_queries]  fri/verify.simf:15-20 */
    const so_path *fri_paths;    /* [(1+n_layers)][n_queries] */
    uint64_t pow_nonce;
} so_stwo_proof;

#define SO_MODE_LITERAL 0 /* the .simf text, incl. D1-D3 */
#define SO_MODE_FIXTURE 1 /* two-batch DEEP quotient, no log_size_ex==0 / folded_query==0 asserts */

typedef struct {
    so_qm31 cp_alpha, deep_alpha;
    so_qm31_point oods_point;
    so_qm31 fold_alpha[SO_MAX_LIST + 1];
    uint8_t digest_after[6][32]; /* after stages I, II, III, IV, V(queries), unused */
    uint32_t queries[64];
    so_qm31 answers[64];
    so_qm31 folded[64];        /* final folded evaluation per query */
    uint32_t folded_query[64];
    uint32_t final_log_size;
} so_stwo_trace;

/* status: 0 = ACCEPT else (stage<<24)|(layer<<16)|(query<<4)|sub, first failing
 * assert in reference evaluation order:
 *   stage 1 channel draw exhausted (for_while u8, channel.simf:127-135)   sub = draw ordinal
 *   stage 2 OODS: sub 1 point inverse abort (channel.simf:147), 2 vanishing inverse abort
 *           (wide_fibonacci.simf:61), 3 CP mismatch (deep/oods.simf:58)
 *   stage 4 proof of work (pow.simf:33)
 *   stage 5 decommit, query q: sub 0 trace path!=1, 1 trace root, 2 cp path!=1, 3 cp root
 *           (evals/verify.simf:47-69 -> merkle.simf:42-43)
 *   stage 6 DEEP denominator inverse abort, query q (deep/quotients.simf:22; sub = batch)
 *   stage 7 FRI layer l (0 first, 1.. inner), query q: sub 0 path!=1, 1 root, 2 fold inverse abort
 *           (fri/layers.simf:43-68)
 *   stage 8 log_size_ex != 0 (fri/verify.simf:127)                [LITERAL only]
 *   stage 9 last layer, query q: sub 0 folded_query != 0 (fri/layers.simf:75) [LITERAL only],
 *           sub 1 value mismatch (fri/layers.simf:76)                              */
uint32_t so_stwo_verify(const so_stwo_cfg *cfg, const so_stwo_proof *p, int mode,
                        so_stwo_trace *tr);

/* The asserts behind the FRI layer loop for one query (fri/verify.simf:127, fri/layers.simf:75-76): first failing code,
 * 0 = none.  Exported because no honest proof reaches them in LITERAL mode (stage 7 fails first), so end-to-end batches
 * compare them only vacuously; tests/test_gpu_intermediates.py feeds the GPU's twin the same random cases.        */
uint32_t so_stwo_fri_tail(int mode, uint32_t lde_log, uint32_t n_layers, uint32_t q, uint32_t folded_query, so_qm31 eval,
                          so_qm31 last);

/* ----------------------------------------------------- shared records (ss_oracle_shared.c)
 * The reference never deduplicates Merkle siblings (fri/queries.simf:41; generate_wit.py:36-42 splits per query).
 * These two restate the DEFINITION of the product's shared-record order (first use in a walk over query 0, 1, ..
 * leaf -> root) so that the closed form the GPU uses can be checked against it.                                   */
uint32_t so_shared_walk(uint32_t lde_log, uint32_t tree, uint32_t n_queries, const uint32_t *queries, uint32_t *plan);
int so_shared_expand(uint32_t n_cols, uint32_t lde_log, uint32_t n_queries, uint32_t n_layers, const uint32_t *shared,
                     size_t words, uint32_t *record);

/* ----------------------------------------------------- minimal decommitment (ss_oracle.c, last section)
 * Upstream stwo's one-decommitment-per-tree form (queries sorted and deduplicated, only the siblings / fold-pair
 * evaluations the verifier cannot compute): PARITY UNPINNED, the reference holds no bytes of it (fri/queries.simf:41).
 * `rec` is a minimal record (include/ss_verify.h).  so_stwo_verify_minimal restates the published layer-by-layer walk;
 * so_stwo_minimal_expand also writes R(M), the per-query record in which every omitted value is the one the walk
 * computes -- status(M) is defined as so_stwo_verify(R(M)).  2 = no minimal record of the config.             */
uint32_t so_stwo_verify_minimal(const so_stwo_cfg *cfg, const uint32_t *rec, size_t words, int mode);
uint32_t so_stwo_minimal_expand(const so_stwo_cfg *cfg, const uint32_t *rec, size_t words, int mode, uint32_t *record_out);

#ifdef __cplusplus
}
#endif
#endif
