/*
 * ss_verify_forms.h -- the input forms beyond the per-query record, and the named entry point of every (form, source)
 * pair that ss_verify_inputs (ss_verify.h, section 4) dispatches to.  A binding that binds ss_verify_inputs needs this
 * header only for the layouts and converters of the shared and minimal forms.
 *
 * Neither form has bytes in the reference: PARITY UNPINNED.  Both are defined through the reference's own per-query path:
 * a shared input verifies exactly as the per-query record it expands to, a minimal input M exactly as R(M), the per-query
 * record in which every omitted value is the one the verifier computes.
 */
#ifndef SS_VERIFY_FORMS_H
#define SS_VERIFY_FORMS_H

#include "ss_verify.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ============================================================================================ shared records
 * The same proof with every DISTINCT sibling of a tree stored once.  The reference presents one full path per query and
 * hashes all of them (fri/queries.simf:41 "we do not sort and remove duplicates"; scripts/generate_wit.py:36-42 splits the
 * prover's lists per query), so the Q paths of a tree repeat the nodes where they meet: 9-21 % of a record.  Layout
 * (ss_stwo_shared_fixed_words words, then the nodes):
 *   roots[3][8]  oods_trace[n_cols][4]  oods_cp[16][4]  fri_roots[1+n_layers][8]  last_layer[4]  pow_nonce_hi  _lo
 *   n_queries x { trace_vals[n_cols], cp_vals[16] }
 *   (1+n_layers) x n_queries x witness[4]
 *   queries[n_queries]   positions in the LDE domain -- an UNTRUSTED hint that only says which siblings coincide
 *   count[3+n_layers]    distinct siblings per tree (kind 0 trace, 1 cp, 2+l FRI layer l)
 *   nodes                tree by tree, count[t] x 8 words, in the order a walk over query 0, 1, .. leaf -> root first
 *                        needs them (csrc/ss_shared.h states the closed form)
 * Expansion is a gather without hashing; the verifier then draws its own queries and checks every expanded path in full,
 * so a wrong hint can only make a proof fail.  A record whose positions leave the domain, whose counts are not what its
 * positions imply or whose size is not fixed + 8 * sum(count) is SS_STATUS_MALFORMED.  Only proofs whose paths all have
 * the config's lengths and agree wherever they meet have a shared form.
 * The shared-path proof.json (SS_TEXT_JSON_SHARED) is the same thing as text: proof.json whose hash lists hold the
 * distinct siblings and whose last member "queries" holds the positions (at most 64).                                 */
size_t ss_stwo_shared_fixed_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_shared_max_words(const ss_stwo_cfg *cfg);   /* fixed + 8 * n_queries * sum of the path lengths */
/* counts[3+n_layers] for these positions; SS_ERR_ARG when one lies outside the LDE domain.  Pure. */
int ss_stwo_shared_counts(const ss_stwo_cfg *cfg, const uint32_t *queries, uint32_t *counts);
/* per-query record + the positions its prover drew -> shared record.  *words_out receives its size; written when it fits
 * cap_words (SS_ERR_ARG otherwise).  Returns 0, or 1 = this proof has no shared form; SS_ERR_ARG when a position lies
 * outside the LDE domain (the caller's error -- not "no shared form").  Pure.                                         */
int ss_stwo_share_record(const ss_stwo_cfg *cfg, const uint32_t *record, const uint32_t *queries, uint32_t *shared_out,
                         size_t cap_words, size_t *words_out);
/* shared -> per-query record on the host (what the GPU does in ss_stwo_expand_shared_dev).  Returns 0 or
 * SS_STATUS_MALFORMED (record_out zeroed).  Pure.                                                                     */
int ss_stwo_unshare_record(const ss_stwo_cfg *cfg, const uint32_t *shared, size_t words, uint32_t *record_out);
/* ... on the GPU (csrc/ss_shared.hip): shared_dev holds n shared records, record i at word offset offs_dev[i] (n + 1
 * offsets, device memory); records_dev receives n * ss_stwo_record_words words, outcome_dev[i] = 0 or SS_STATUS_MALFORMED
 * (record i zeroed).  Asynchronous on `stream`; feed records_dev to ss_stwo_pack_dev.                                  */
int ss_stwo_expand_shared_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *shared_dev,
                              const uint64_t *offs_dev, uint32_t *records_dev, uint32_t *outcome_dev, void *stream);
/* Shared record -> the shared-path proof.json, byte for byte what json.dumps prints for formats.stwo_to_json(proof,
 * shared=True).  0 = `shared` is no shared record of the config.  No GPU involved.                                    */
size_t ss_stwo_write_shared_text(const ss_stwo_cfg *cfg, const uint32_t *shared, size_t words, int python_separators,
                                 char *buf, size_t cap);

/* =========================================================================================== minimal records
 * One decommitment per TREE instead of one path per query -- what upstream stwo's prover sends (MerkleDecommitment /
 * FriLayerProof of starkware-libs/stwo, a dependency that is not in the reference's repository) before the reference's
 * adapter cuts it per query (scripts/generate_wit.py:36-42; fri/queries.simf:41; merkle.simf:22-44 folds one path).  With
 * Nodes(a) = the distinct positions `query >> a`, ascending, and Lone(a) = those whose sibling `x ^ 1` is not among them
 * (a = 0 .. lde_log - 1 counts from the leaves):
 *   head                       roots[3][8] oods_trace[n_cols][4] oods_cp[16][4] fri_roots[1+n_layers][8] last_layer[4] nonce_hi _lo
 *   n_vals[2]  n_fw[1+n_layers]  n_hw[3+n_layers]     the lengths of the lists below (data, like a path's length)
 *   trace_vals[n_vals[0]][n_cols]  cp_vals[n_vals[1]][16]   once per node of Nodes(0)
 *   fri_wit[l][n_fw[l]][4]     layer l: the fold partners of Lone(l), i.e. the members of the layer's pairs that are not
 *                              queried themselves
 *   hash_wit[t][n_hw[t]][8]    tree t (0 trace, 1 cp, 2+l FRI layer l): the siblings of Lone(a) for a = first ..
 *                              lde_log-1 (first = 0, 0, l+1), level by level
 * Every other sibling / partner is a value the verifier computes from another query's chain, and the library takes it from
 * there: no expansion pass, no hint -- the queries are the verifier's own.  A tree whose lists do not have the lengths the
 * queries imply fails like a path of the wrong length (sub 0 of stage 5 / 7, query 0); a record whose size is not what its
 * counts give, or whose counts exceed n_queries (x the tree's depth), is SS_STATUS_MALFORMED.  The published algorithm is
 * restated in oracle/ss_oracle.c and held against the per-query path through R(M).
 * The minimal proof.json (SS_TEXT_JSON_MINIMAL) is the same thing in the schema of proof.json -- the lists as upstream
 * stwo's prover fills them; with one query the two forms are the same bytes.                                          */
size_t ss_stwo_minimal_fixed_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_minimal_max_words(const ss_stwo_cfg *cfg);
/* list lengths for these positions: counts[0..1] = n_vals, [2 .. 2+n_layers] = n_fw, [3+n_layers .. 5+2 n_layers] = n_hw.  Pure. */
int ss_stwo_minimal_counts(const ss_stwo_cfg *cfg, const uint32_t *queries, uint32_t *counts);
/* per-query record + the positions its prover drew -> minimal record (a selection: nothing is hashed or checked beyond
 * "queries that present the same thing present the same words").  Returns 0, or 1 = no minimal form.  Pure.            */
int ss_stwo_minimise_record(const ss_stwo_cfg *cfg, const uint32_t *record, const uint32_t *queries, uint32_t *minimal_out,
                            size_t cap_words, size_t *words_out);
/* Minimal records resident in HBM: min_dev holds n minimal records, record i at word offset offs_dev[i]; batch_dev
 * (ss_stwo_minimal_batch_words words) and the workspace (ss_stwo_minimal_workspace_bytes) are scratch the call fills.
 * SS_PHASE_HEAD = read the records, transcript, plan + gather, query kernel; SS_PHASE_TAIL = merkle / top / finalize, as
 * for ss_stwo_verify_phase_dev.  Asynchronous, no allocation, graph-capturable.                                        */
size_t ss_stwo_minimal_batch_words(const ss_stwo_cfg *cfg, size_t n);
size_t ss_stwo_minimal_workspace_bytes(const ss_stwo_cfg *cfg, size_t n);
int ss_stwo_verify_minimal_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *min_dev,
                               const uint64_t *offs_dev, uint32_t *batch_dev, void *workspace_dev, size_t workspace_bytes,
                               uint32_t *status_dev, uint32_t *accept_count_dev, int phases, void *stream);
/* text -> minimal record (0 / SS_STATUS_CONFIG_MISMATCH / SS_STATUS_MALFORMED; *words_out = its size, written when it fits
 * cap_words, SS_ERR_ARG otherwise) and back (the text's length, 0 = no minimal record of the config).  No GPU involved.
 * The host has two readers of this form: a streaming one (one pass, no tree) for texts in the writers' member order that
 * declare the expected config, and the general one (any member order; the one that judges a text malformed or
 * mismatching).  ss_stwo_parse_minimal tries the streaming reader and gives the general one what it declines;
 * ss_stwo_parse_minimal_route picks one, for tests and diagnosis: SS_READER_STREAM returns SS_READER_DECLINED for a text it
 * does not take.  What the streaming reader takes it reads as the general one does (tests/test_minimal.py).            */
#define SS_READER_AUTO 0
#define SS_READER_GENERAL 1
#define SS_READER_STREAM 2
#define SS_READER_DECLINED 3
int ss_stwo_parse_minimal(const ss_stwo_cfg *cfg, const char *text, size_t len, uint32_t *minimal_out, size_t cap_words,
                          size_t *words_out);
int ss_stwo_parse_minimal_route(const ss_stwo_cfg *cfg, const char *text, size_t len, int reader, uint32_t *minimal_out,
                                size_t cap_words, size_t *words_out);
size_t ss_stwo_write_minimal_text(const ss_stwo_cfg *cfg, const uint32_t *minimal, size_t words, int python_separators,
                                  char *buf, size_t cap);

/* ================================================= the (form, source) pairs of ss_verify_inputs under their names
 * Each is ss_verify_inputs with the descriptor its name says (family / form / source; `fmt` = text_fmt) -- kept because
 * ABI 2.0-2.3 callers bind them; same verdicts, same scratch, same lock.
 *   records         SS_FORM_RECORDS          x SS_SRC_HOST | SS_SRC_PINNED
 *   shared_records  SS_FORM_SHARED_RECORDS   x SS_SRC_HOST (shared[i] has words[i] words) | SS_SRC_PINNED (flat, offs)
 *   minimal_records SS_FORM_MINIMAL_RECORDS  x SS_SRC_HOST | SS_SRC_PINNED
 *   texts / files   SS_FORM_TEXT             x SS_SRC_HOST | SS_SRC_PINNED | SS_SRC_FILES; minimal_texts = fmt
 *                                              SS_TEXT_JSON_MINIMAL (ss_stwo_verify_files takes that fmt too)          */
int ss_s101_verify_records(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const uint32_t *const *records,
                           uint32_t *status_host);
int ss_stwo_verify_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *records,
                           uint32_t *status_host);
int ss_stwo_verify_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *records, uint32_t *status_host);
int ss_stwo_verify_shared_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *shared,
                                  const size_t *words, uint32_t *status_host);
int ss_stwo_verify_shared_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *flat,
                                         const uint64_t *offs, uint32_t *status_host);
int ss_stwo_verify_minimal_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *minimal,
                                   const size_t *words, uint32_t *status_host);
int ss_stwo_verify_minimal_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *flat,
                                          const uint64_t *offs, uint32_t *status_host);
int ss_stwo_verify_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts, const size_t *lens,
                         int fmt, uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_files(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *paths, int fmt,
                         uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *blob, const uint64_t *offs,
                                const size_t *lens, int fmt, uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_minimal_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts,
                                 const size_t *lens, uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_minimal_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const void *blob, const uint64_t *offs,
                                        const size_t *lens, uint32_t *status_host, ss_ingest_stats *stats);
int ss_s101_verify_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                         uint32_t *status_host, ss_ingest_stats *stats);
int ss_s101_verify_texts_pinned(ss_ctx *ctx, size_t n, const char *blob, const uint64_t *offs, const size_t *lens, int fmt,
                                uint32_t *status_host, ss_ingest_stats *stats);
int ss_s101_verify_files(ss_ctx *ctx, size_t n, const char *const *paths, int fmt, uint32_t *status_host,
                         ss_ingest_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* SS_VERIFY_FORMS_H */
