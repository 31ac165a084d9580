/*
 * ss_verify.h -- C ABI of libss_verify.so, the MI355X (gfx950) batch STARK verifier.
 *
 * What this replaces.  The reference has no plugin/FFI seam for its verifier: the
 * boundary is "one process per proof", `simfony run main.simf --witness proof.wit`
 *   - stark101/Makefile:8-9, stwo-verifier/Makefile:17-18        (the call)
 *   - simfony-cli/src/main.rs:163-209 `handle_run`               (compile + satisfy + run)
 *   - simfony-cli/src/main.rs:205 `run_program(..)`              (the verifier executes here)
 *   - simfony-cli/src/main.rs:254-257                            (exit 0 = ACCEPT, 1 = REJECT)
 * whose body is `verify_proof` (stark101/src/verifier.simf:24-42,
 * stwo-verifier/src/verifier.simf:32-58).  The entry points below are what a cgo / JNI /
 * ctypes / Rust-FFI binding for "verify these N witnesses" would bind instead
 * (INTEGRATION.md shows the binding).  Plain pointers and sizes only.
 *
 * No CPU fallback exists in this library: every verify entry point runs HIP kernels and
 * returns SS_ERR_NO_DEVICE / SS_ERR_HIP if it cannot.
 *
 * ---------------------------------------------------------------------------------------
 * Data conventions
 *   word      little-endian uint32_t.
 *   hash      8 words; word j is the big-endian integer of digest bytes 4j..4j+3, i.e.
 *             the SHA-256 state word (the u256 the reference prints, most significant
 *             word first).
 *   QM31      4 words (a, b, c, d) of a + bi + (c + di)j  (fields/qm31.simf:15).
 *   record    one proof in natural order (layouts below); what a caller produces.
 *   batch     N records re-tiled for the GPU by ss_*_pack (SoA over proofs / queries,
 *             Merkle paths in 64-chain tiles [level][half][lane][4 words] so that a
 *             wavefront reads one sibling level as two contiguous 1 KiB bursts).  A batch
 *             is a pure permutation (+ zero padding) of its records: no hashing, no
 *             arithmetic.  Its size is ss_*_batch_words().
 *   status    one uint32_t per proof: 0 = ACCEPT, otherwise the code of the FIRST assert
 *             that fails in the reference's evaluation order (codes below).  The
 *             reference only exposes accept/reject; the code is extra information.
 */
#ifndef SS_VERIFY_H
#define SS_VERIFY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_VERSION 0x00020003 /* 2.0: stwo records carry their Merkle path lengths; shape_status is gone.
                                 2.1: GPU text reader behind the text entry points, ss_stwo_write_text, thread rules
                                 2.2: shared records (every distinct Merkle sibling once), expanded on the GPU
                                 2.3: minimal records (one sorted, deduplicated decommitment per tree), verified without
                                      expansion; ss_selftest replays the reference's known-answer tests; stark101
                                      intermediates; caller-pinned host buffers */

/* return codes (all < 0 are errors; verdicts live in the status array) */
#define SS_OK 0
#define SS_ERR_ARG (-1)       /* null pointer, bad size, unsupported config, n == 0 (an empty batch is the caller's no-op) */
#define SS_ERR_HIP (-2)       /* a HIP call failed; see ss_last_error() */
#define SS_ERR_NO_DEVICE (-3) /* no usable gfx950 device */
#define SS_ERR_WORKSPACE (-4) /* workspace too small */
#define SS_ERR_NOMEM (-5)     /* the host ran out of memory inside the call (nothing is left half done: the call failed as a whole) */

int ss_version(void);
const char *ss_last_error(void); /* thread-local text of the last error */
int ss_device_count(void);       /* number of visible HIP devices, <0 on error */
/* sizeof(ss_stwo_cfg) == 40 and sizeof(ss_s101_shape) == 8 as this library was compiled: a
 * binding written in another language checks its own struct against these once at start-up. */
size_t ss_abi_sizeof_cfg(void);
size_t ss_abi_sizeof_shape(void);

/* ======================================================================== stark101
 * Record (ss_s101_record_words(max_layers, max_path) words, zero padded):
 *   root[8]  n_layers  last
 *   3 x { ev, len, path[max_path][8] }                        (air.simf:24-27)
 *   max_layers x { root[8], beta,
 *                  cpa_ev, cpa_len, cpa_path[max_path][8],
 *                  cpb_ev, cpb_len, cpb_path[max_path][8] }   (fri.simf:30)
 * paths are leaf -> root, as the prover emits them (prover.py:144-146,162-164).
 * Status codes: (stage << 8) | sub
 *   1 beta mismatch (fri.simf:43) sub=layer | 2 trace Merkle (air.simf:41) sub=k |
 *   3 composition division abort (field.simf:46) sub=0..2 |
 *   4 FRI layer, sub = 4*layer + {0 chain fri.simf:77, 1 cpa Merkle :79, 2 cpb Merkle :80,
 *     3 fold division :58-60} | 5 last value (fri.simf:90)                                  */
typedef struct ss_s101_shape {
    uint32_t max_layers; /* <= 31 */
    uint32_t max_path;   /* <= 31 */
} ss_s101_shape;

size_t ss_s101_record_words(const ss_s101_shape *shape);
size_t ss_s101_batch_words(const ss_s101_shape *shape, size_t n);
size_t ss_s101_workspace_bytes(const ss_s101_shape *shape, size_t n);
/* records[i] -> proof i (pointers may repeat).  batch_host receives ss_s101_batch_words words. */
int ss_s101_pack(const ss_s101_shape *shape, size_t n, const uint32_t *const *records,
                 uint32_t *batch_host);

/* ============================================================================ stwo
 * Runtime form of the compile-time macros of stwo-verifier/src/config.simf:10-51.       */
typedef struct ss_stwo_cfg {
    uint32_t n_cols;     /* NUM_COLUMNS      (1..1024)                     */
    uint32_t trace_log;  /* TRACE_LOG_SIZE                                 */
    uint32_t lde_log;    /* LDE_LOG_SIZE     (<= 31)                       */
    uint32_t n_queries;  /* NUM_FRI_QUERIES  (1..64)                       */
    uint32_t n_layers;   /* NUM_FRI_LAYERS, inner layers (<= 30, < lde_log) */
    uint32_t mode;       /* SS_MODE_*                                      */
    uint64_t pow_target; /* POW_TARGET_64: digest value must be < target   */
    uint32_t hash;       /* SS_HASH_*                                      */
    uint32_t flags;      /* SS_FLAG_*; 0 = defaults (explicit tail word: sizeof == 40) */
} ss_stwo_cfg;

/* SS_FLAG_NO_DEDUP: hash every query's Merkle path in full, as the reference does
 * (fri/queries.simf:41 notes that it does not deduplicate).  By default the library hashes each
 * distinct (left, right) pair of the top levels of a tree once per proof and checks byte for byte
 * that every query presenting that node presents the same pair; trees where two queries disagree
 * are re-hashed query by query, so the status words are the reference's either way.        */
#define SS_FLAG_NO_DEDUP 1u
/* SS_FLAG_TOP_CHECKS: make those byte compares in the top kernel for every query count.  By default, when the
 * query count divides 64, they are made by the merkle kernel (a proof's chains are lanes of one wavefront there);
 * other query counts always take the top-kernel path.  Same status words; for A/B runs and tests.             */
#define SS_FLAG_TOP_CHECKS 2u

/* SS_HASH_SHA256 is the reference (hasher.simf:13-104, channel.simf:36-172).  SS_HASH_BLAKE2S is
 * the "Blake2s Merkle" variant BASELINE.json names: the same protocol over the same byte
 * strings with Blake2s-256 (RFC 7693) as the hash.  The reference has no Blake2s, so that
 * variant's parity is unpinned (RFC vectors + prover/oracle/GPU agreement only).          */
#define SS_HASH_SHA256 0u
#define SS_HASH_BLAKE2S 1u

/* SS_MODE_LITERAL follows the .simf text (single DEEP batch fri/answers.simf:97-130,
 * `log_size_ex == 0` fri/verify.simf:127, `folded_query == 0` fri/layers.simf:75).
 * SS_MODE_FIXTURE is what the reference's own proofs (tests/data/proof*.json) satisfy:
 * trace columns sampled at P and the 16 composition columns at 2P as two DEEP batches
 * (docs/batching_samples.md:62-70), and neither of those two asserts.   */
#define SS_MODE_LITERAL 0u
#define SS_MODE_FIXTURE 1u

/* Record (ss_stwo_record_words words):
 *   roots[3][8]  oods_trace[n_cols][4]  oods_cp[16][4]  fri_roots[1+n_layers][8]
 *   last_layer[4]  pow_nonce_hi  pow_nonce_lo
 *   n_queries x { trace_vals[n_cols], cp_vals[16], trace_path[lde_log][8], cp_path[lde_log][8] }
 *   (1+n_layers) x n_queries x { witness[4], path[lde_log-1-layer][8] }
 *   path_len[3+n_layers][n_queries]      kind 0 trace, 1 cp, 2+l FRI layer l
 * The reference types every Merkle path List<u256, 32> (scripts/generate_wit.py:70-103): its
 * length is data.  path_len carries the length the proof really has; the fixed slots above hold
 * the first min(length, slot) siblings, zero padded.  A path whose length differs from the slot
 * cannot verify in the reference whatever it contains (`path == 1`, merkle.simf:42) and the
 * library reports the code of exactly that assert -- the caller computes no part of the verdict.
 * Status codes: (stage << 24) | (layer << 16) | (query << 4) | sub
 *   0/1 the proof does not have the shape of the expected config (host side, before any kernel) |
 *   1 channel draw exhausted | 2 OODS (sub 1 point inverse, 2 vanishing inverse, 3 CP mismatch
 *   deep/oods.simf:58) | 4 proof of work (pow.simf:33) | 5 decommit (sub 0 trace path, 1 trace
 *   root, 2 cp path, 3 cp root) | 6 DEEP denominator abort (sub = batch) | 7 FRI layer (sub 0
 *   path, 1 root, 2 fold inverse) | 8 log_size_ex != 0 [LITERAL] | 9 last layer (sub 0
 *   folded_query != 0 [LITERAL], 1 value mismatch fri/layers.simf:76)                       */
/* Pure functions of their arguments (no environment, no process state): sizes computed in one process hold in another. */
size_t ss_stwo_record_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_batch_words(const ss_stwo_cfg *cfg, size_t n);
size_t ss_stwo_workspace_bytes(const ss_stwo_cfg *cfg, size_t n);
int ss_stwo_pack(const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *records,
                 uint32_t *batch_host);

/* Shared record: the same proof with every DISTINCT sibling of a tree stored once.  The reference presents one
 * full path per query and hashes all of them (fri/queries.simf:41 "we do not sort and remove duplicates";
 * scripts/generate_wit.py:36-42 splits the prover's lists per query), so the Q paths of a tree repeat the nodes
 * where they meet: 9-21 % of a record.  Layout (ss_stwo_shared_fixed_words words, then the nodes):
 *   roots[3][8]  oods_trace[n_cols][4]  oods_cp[16][4]  fri_roots[1+n_layers][8]  last_layer[4]  pow_nonce_hi  _lo
 *   n_queries x { trace_vals[n_cols], cp_vals[16] }
 *   (1+n_layers) x n_queries x witness[4]
 *   queries[n_queries]       positions in the LDE domain -- an UNTRUSTED hint that only says which siblings coincide
 *   count[3+n_layers]        distinct siblings per tree (kind 0 trace, 1 cp, 2+l FRI layer l)
 *   nodes                    tree by tree, count[t] x 8 words, in the order a walk over query 0, 1, .. leaf -> root
 *                            first needs them (csrc/ss_shared.h states the closed form)
 * Expansion is a gather without hashing; the verifier then draws its own queries and checks every expanded path in
 * full, so a wrong hint can only make a proof fail.  A record whose positions leave the domain, whose counts are not
 * what its positions imply or whose size is not fixed + 8 * sum(count) is no shared record of the config:
 * status SS_STATUS_MALFORMED.  Only proofs whose paths all have the config's lengths and agree wherever they meet
 * have a shared form.  No bytes of such a format exist in the reference: parity is "verifies exactly as the
 * per-query record it expands to".                                                                              */
size_t ss_stwo_shared_fixed_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_shared_max_words(const ss_stwo_cfg *cfg);   /* fixed + 8 * n_queries * sum of the path lengths */
/* counts[3+n_layers] for these positions; SS_ERR_ARG when one lies outside the LDE domain.  Pure. */
int ss_stwo_shared_counts(const ss_stwo_cfg *cfg, const uint32_t *queries, uint32_t *counts);
/* per-query record + the positions its prover drew -> shared record.  *words_out receives its size; written when it
 * fits cap_words (SS_ERR_ARG otherwise).  Returns 0, or 1 = this proof has no shared form; SS_ERR_ARG when a position
 * lies outside the LDE domain (the caller's error, as in ss_stwo_shared_counts -- not "no shared form").  Pure.  */
int ss_stwo_share_record(const ss_stwo_cfg *cfg, const uint32_t *record, const uint32_t *queries, uint32_t *shared_out,
                         size_t cap_words, size_t *words_out);
/* shared -> per-query record on the host (what the GPU does in ss_stwo_expand_shared_dev).  Returns 0 or
 * SS_STATUS_MALFORMED (record_out zeroed).  Pure.                                                                */
int ss_stwo_unshare_record(const ss_stwo_cfg *cfg, const uint32_t *shared, size_t words, uint32_t *record_out);

/* Minimal record: one decommitment per TREE instead of one path per query -- what upstream stwo's prover sends
 * (MerkleDecommitment / FriLayerProof of starkware-libs/stwo, a dependency that is not in the reference's repository)
 * before the reference's adapter cuts it per query (scripts/generate_wit.py:36-42; fri/queries.simf:41 "we do not sort and
 * remove duplicates"; merkle.simf:22-44 folds one path).  With Nodes(a) = the distinct positions `query >> a`, ascending,
 * and Lone(a) = those whose sibling `x ^ 1` is not among them (a = 0 .. lde_log - 1 counts from the leaves):
 *   head                       roots[3][8] oods_trace[n_cols][4] oods_cp[16][4] fri_roots[1+n_layers][8] last_layer[4] nonce_hi _lo
 *   n_vals[2]  n_fw[1+n_layers]  n_hw[3+n_layers]         the lengths of the lists below (data, like a path's length)
 *   trace_vals[n_vals[0]][n_cols]  cp_vals[n_vals[1]][16]  once per node of Nodes(0)
 *   fri_wit[l][n_fw[l]][4]                                 layer l: the fold partners of Lone(l), i.e. the members of the
 *                                                          layer's pairs that are not queried themselves
 *   hash_wit[t][n_hw[t]][8]                                tree t (0 trace, 1 cp, 2+l FRI layer l): the siblings of Lone(a)
 *                                                          for a = first .. lde_log-1 (first = 0, 0, l+1), level by level
 * Every other sibling / partner is a value the verifier computes from another query's chain, and the library takes it
 * from there: no expansion pass, no hint -- the queries are the verifier's own.  Verdicts: a minimal record M verifies
 * exactly as R(M), the per-query record in which each omitted value is the computed one; a tree whose lists do not have
 * the lengths the queries imply fails like a path of the wrong length (sub 0 of stage 5 / 7, query 0).  A record whose
 * size is not what its counts give, or whose counts exceed n_queries (x the tree's depth), is SS_STATUS_MALFORMED.
 * No bytes of this form exist in the reference: PARITY UNPINNED; the published algorithm is restated in
 * oracle/ss_oracle.c and held against the per-query path through R(M).  Only proofs whose paths all have the config's
 * lengths and whose queries agree wherever they present the same thing have a minimal form.                       */
size_t ss_stwo_minimal_fixed_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_minimal_max_words(const ss_stwo_cfg *cfg);
/* list lengths for these positions: counts[0..1] = n_vals, [2 .. 2+n_layers] = n_fw, [3+n_layers .. 5+2 n_layers] = n_hw.  Pure. */
int ss_stwo_minimal_counts(const ss_stwo_cfg *cfg, const uint32_t *queries, uint32_t *counts);
/* per-query record + the positions its prover drew -> minimal record (a selection: nothing is hashed or checked beyond
 * "queries that present the same thing present the same words").  Returns 0, or 1 = no minimal form.  Pure.        */
int ss_stwo_minimise_record(const ss_stwo_cfg *cfg, const uint32_t *record, const uint32_t *queries, uint32_t *minimal_out,
                            size_t cap_words, size_t *words_out);

/* ======================================================================= execution
 * One context per process and GPU (one process per GPU is the intended deployment).
 * Threads.  A context may be shared by threads.  Entry points that use the context's own scratch
 * -- ss_*_verify_records, ss_*_verify_texts / _files, ss_selftest -- serialize on a lock inside the
 * context: concurrent calls are safe and run one after the other.  The device entry points
 * (ss_*_verify_batch_dev / _phase_dev, ss_stwo_pack_dev) take caller-owned buffers and a caller-owned
 * stream and run concurrently from any number of threads (the timing list has its own lock); what they
 * require is the usual HIP rule that the buffers of two in-flight calls are distinct.  Functions
 * without a context (sizes, packers, parsers, writers) are pure.
 * Process environment.  When the library is LOADED it puts GPU_MAX_HW_QUEUES=24 into the process environment unless the
 * variable is already set or SS_KEEP_ENV is (csrc/ss_env.cpp): the HIP runtime reads it when it initialises -- at the
 * process's first HIP call -- and its default of 4 hardware queues serialises streams that share one, which is what the
 * pipelined entry points and every caller that keeps several passes in flight on streams of its own must avoid
 * (stark101 x 4 096, 16 passes in flight: 40.9 M proofs/s at the default, 63.7 M with 24).                            */
typedef struct ss_ctx ss_ctx;
int ss_ctx_create(int device, ss_ctx **out);
void ss_ctx_destroy(ss_ctx *ctx);

/* Same permutation as ss_stwo_pack, done by the GPU: records_dev holds the n records back to
 * back (n * ss_stwo_record_words words, device memory), batch_dev receives
 * ss_stwo_batch_words words.  Asynchronous on `stream`.  This is what the host-buffer entry
 * point uses after uploading the raw records, so the host never re-tiles 170 KB proofs.    */
int ss_stwo_pack_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *records_dev,
                     uint32_t *batch_dev, void *stream);

/* Shared records -> per-query records on the GPU (csrc/ss_shared.hip): shared_dev holds n shared records, record i
 * at word offset offs_dev[i] with offs_dev[i + 1] - offs_dev[i] words (n + 1 offsets, device memory); records_dev
 * receives n * ss_stwo_record_words words, outcome_dev[i] = 0 or SS_STATUS_MALFORMED (record i zeroed).
 * Asynchronous on `stream`; feed records_dev to ss_stwo_pack_dev.                                                */
int ss_stwo_expand_shared_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *shared_dev,
                              const uint64_t *offs_dev, uint32_t *records_dev, uint32_t *outcome_dev, void *stream);

/* The same for stark101: records_dev holds n records of `shape` back to back, batch_dev receives
 * ss_s101_batch_words words (the permutation of ss_s101_pack).                                      */
int ss_s101_pack_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const uint32_t *records_dev,
                     uint32_t *batch_dev, void *stream);

/* Device-resident entry points: every pointer is device memory on ctx's GPU, `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous; status_dev is valid once the stream
 * has drained.  accept_count_dev (may be NULL) receives the
 * number of accepted proofs (one uint32_t) -- the value a multi-GPU caller all-reduces.
 * No allocation and no synchronisation happen inside, so the call is hipGraph-capturable.  */
int ss_s101_verify_batch_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n,
                             const uint32_t *batch_dev, void *workspace_dev,
                             size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, void *stream);
int ss_stwo_verify_batch_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n,
                             const uint32_t *batch_dev, void *workspace_dev, size_t workspace_bytes,
                             uint32_t *status_dev, uint32_t *accept_count_dev, void *stream);

/* The same work in two separately enqueueable halves, for callers that pipeline batches:
 *   SS_PHASE_HEAD  reset status, transcript kernel (Fiat-Shamir chain, latency bound) and
 *                  query kernel  -> writes the workspace
 *   SS_PHASE_TAIL  Merkle kernel (ALU bound) and finalize -> reads the workspace
 * HEAD of batch i+1 on one stream overlaps TAIL of batch i on another (each batch needs its
 * own workspace / status; order HEAD -> TAIL of one batch with an event).  SS_PHASE_ALL on one
 * stream is exactly ss_*_verify_batch_dev.                                                 */
#define SS_PHASE_HEAD 1
#define SS_PHASE_TAIL 2
#define SS_PHASE_ALL 3
int ss_s101_verify_phase_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n,
                             const uint32_t *batch_dev, void *workspace_dev,
                             size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, int phases, void *stream);
int ss_stwo_verify_phase_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n,
                             const uint32_t *batch_dev, void *workspace_dev, size_t workspace_bytes,
                             uint32_t *status_dev, uint32_t *accept_count_dev, int phases, void *stream);

/* Host-buffer convenience: pack + H2D + verify + D2H, synchronous.  Scratch (pinned staging and
 * device buffers) belongs to the context and only grows.  PCIe-inclusive; not what bench.py times. */
int ss_s101_verify_records(ss_ctx *ctx, const ss_s101_shape *shape, size_t n,
                           const uint32_t *const *records, uint32_t *status_host);
int ss_stwo_verify_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n,
                           const uint32_t *const *records, uint32_t *status_host);

/* Minimal records on the device: min_dev holds n minimal records, record i at word offset offs_dev[i] with offs_dev[i+1]
 * - offs_dev[i] words; batch_dev (ss_stwo_minimal_batch_words words) and the workspace (ss_stwo_minimal_workspace_bytes)
 * are scratch the call fills.  SS_PHASE_HEAD = read the records, transcript, plan + gather, query kernel; SS_PHASE_TAIL =
 * merkle / top / finalize, as for ss_stwo_verify_phase_dev.  Asynchronous, no allocation, graph-capturable.        */
size_t ss_stwo_minimal_batch_words(const ss_stwo_cfg *cfg, size_t n);
size_t ss_stwo_minimal_workspace_bytes(const ss_stwo_cfg *cfg, size_t n);
int ss_stwo_verify_minimal_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *min_dev,
                               const uint64_t *offs_dev, uint32_t *batch_dev, void *workspace_dev, size_t workspace_bytes,
                               uint32_t *status_dev, uint32_t *accept_count_dev, int phases, void *stream);
/* The same from host memory (minimal[i] has words[i] words): the fewest bytes on the host link of all input forms. */
int ss_stwo_verify_minimal_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *minimal,
                                   const size_t *words, uint32_t *status_host);

/* The three host-buffer paths WITHOUT the staging copy (what replaces, for a batch caller, the reference's "one file per
 * process" hand-over: stwo-verifier/Makefile:17-18, simfony-cli/src/main.rs:163-209): the inputs lie back to back in ONE buffer
 * of page-locked host memory -- allocated by hipHostMalloc, or any memory registered with ss_host_register (hipHostRegister) -- and the DMA
 * engine reads it directly, chunk by chunk, while the previous chunk is verified.  No host thread touches the bytes
 * (the staged entry points need about four cores to feed the link; a rank of an 8-GPU host has two).  records: n x
 * ss_stwo_record_words words; shared / minimal: record i at word offset offs[i], offs[n] = the total (n + 1 ascending
 * offsets, ordinary memory).  SS_ERR_ARG when the buffer is not page-locked.  Verdicts are those of the staged twins. */
int ss_host_register(ss_ctx *ctx, void *ptr, size_t bytes);
int ss_host_unregister(ss_ctx *ctx, void *ptr);
int ss_stwo_verify_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *records, uint32_t *status_host);
int ss_stwo_verify_shared_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *flat,
                                         const uint64_t *offs, uint32_t *status_host);
int ss_stwo_verify_minimal_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *flat,
                                          const uint64_t *offs, uint32_t *status_host);

/* The same from shared records (shared[i] has words[i] words): fewer bytes on the host link, expanded behind it. */
int ss_stwo_verify_shared_records(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *shared,
                                  const size_t *words, uint32_t *status_host);

/* ===================================================================== text ingestion
 * The reference's callers hand the verifier TEXT: proof.json (stwo: the schema read by
 * stwo-verifier/scripts/generate_wit.py:106-245; stark101: fibsquare/prover.py:108,143-167) or
 * proof.wit (generate_wit.py:218-243, stark101/scripts/generate_wit.py:13-30), the file `simfony run
 * --witness` consumes (simfony-cli/src/main.rs:163-209).  These entry points read that text natively
 * (host threads, straight into the pinned staging buffers of the upload), so the drop-in path is
 * not bound by a Python parser.  `cfg` is the config the CALLER expects: a proof that declares, or
 * has the shape of, any other config is not verified against it.
 * Status words of stage 0 (host side, before any kernel; smaller than every assert code):
 *   SS_STATUS_CONFIG_MISMATCH  well-formed witness of another shape / declared parameters -- the
 *                              reference would fail to type it (main.rs:77-81,187-190)
 *   SS_STATUS_MALFORMED        not a witness of the reference's types at all                     */
#define SS_TEXT_AUTO 0 /* sniff: a .wit is a JSON object with a COMMITMENTS / P_MT_ROOT member */
#define SS_TEXT_JSON 1
#define SS_TEXT_WIT 2
/* The shared-path proof.json: proof.json with every DISTINCT Merkle sibling of a tree once (in the order of the shared
 * record above) and a last member "queries" with the positions (formats.stwo_to_json(shared=True), `cli convert --to
 * json-shared`; at most 64 positions).  The text entry points recognise it by that member under SS_TEXT_AUTO and
 * SS_TEXT_JSON; the constant is for the writer / diagnostic calls that want that form explicitly.               */
#define SS_TEXT_JSON_SHARED 3
#define SS_STATUS_CONFIG_MISMATCH 1u
#define SS_STATUS_MALFORMED 2u

/* The minimal proof.json: proof.json with one decommitment per tree (minimal record above; formats.stwo_minimal_to_json,
 * `cli convert --to json-minimal`) -- the lists as upstream stwo's prover fills them before the reference's adapter cuts them
 * per query (scripts/generate_wit.py:36-42).  Nothing in the text names its form; with two or more queries its lists are
 * shorter than n_queries equal shares, with one query the two forms are the same bytes.                                   */
#define SS_TEXT_JSON_MINIMAL 4
/* text -> minimal record (0 / SS_STATUS_CONFIG_MISMATCH / SS_STATUS_MALFORMED; *words_out = its size, written when it fits
 * cap_words, SS_ERR_ARG otherwise) and back (the text's length, 0 = no minimal record of the config).  No GPU involved. */
int ss_stwo_parse_minimal(const ss_stwo_cfg *cfg, const char *text, size_t len, uint32_t *minimal_out, size_t cap_words,
                          size_t *words_out);
size_t ss_stwo_write_minimal_text(const ss_stwo_cfg *cfg, const uint32_t *minimal, size_t words, int python_separators,
                                  char *buf, size_t cap);
/* The host has two readers of this form: a streaming one (one pass, no tree) for texts in the writers' member order that
 * declare the expected config, and the general one (any member order; the one that judges a text malformed or mismatching).
 * ss_stwo_parse_minimal and ss_stwo_verify_minimal_texts try the streaming reader and give the general one what it declines;
 * this call picks one, for tests and diagnosis: SS_READER_STREAM returns SS_READER_DECLINED for a text it does not take.
 * What the streaming reader takes it reads as the general one does (tests/test_minimal.py).                              */
#define SS_READER_AUTO 0
#define SS_READER_GENERAL 1
#define SS_READER_STREAM 2
#define SS_READER_DECLINED 3
int ss_stwo_parse_minimal_route(const ss_stwo_cfg *cfg, const char *text, size_t len, int reader, uint32_t *minimal_out,
                                size_t cap_words, size_t *words_out);

/* One text -> one record (ss_stwo_record_words words).  Returns 0 = parsed, SS_STATUS_CONFIG_MISMATCH,
 * SS_STATUS_MALFORMED (record_out untouched or zeroed), or < 0 on a bad argument.  No GPU involved. */
int ss_stwo_parse(const ss_stwo_cfg *cfg, const char *text, size_t len, int fmt, uint32_t *record_out);
/* stark101: the shape is data.  ss_s101_parse returns 0 / SS_STATUS_MALFORMED and the proof's shape;
 * record_out (may be NULL) receives ss_s101_record_words(shape_inout) words when the proof fits the
 * shape passed in (layers / path lengths up to 31 always fit {31, 31}).                           */
int ss_s101_parse(const char *text, size_t len, int fmt, ss_s101_shape *shape_inout, uint32_t *record_out);

typedef struct ss_ingest_stats {
    double read_s;    /* wall time staging the raw bytes (copy / file read into pinned memory)  */
    double parse_s;   /* wall time in the host reader (texts the GPU reader did not take)       */
    double total_s;   /* the whole call: stage + upload + GPU read + verify + download          */
    uint64_t text_bytes, record_bytes;
    uint32_t threads; /* host threads used (scheduler affinity capped by the cgroup quota) */
    uint32_t host_parsed; /* texts that went through the host reader (0 for canonical texts) */
} ss_ingest_stats;

/* Texts / files -> verdicts, synchronous.  The raw bytes are uploaded in pinned chunks and turned
 * into records ON THE GPU (csrc/ss_textdev.hip): a text that is, byte for byte, what the reference's
 * producers write for the expected config -- proof.json as the external prover / json.dumps prints it
 * (tests/data/proof.json), proof.wit as generate_wit.py:218-243 prints it -- except for its numbers and
 * for whitespace outside JSON strings, is read without a parse tree; every other text (other key
 * order, escapes, non-canonical numbers, another shape, not a witness at all) is handed to the host
 * reader behind ss_stwo_parse, which alone decides parsed / SS_STATUS_CONFIG_MISMATCH /
 * SS_STATUS_MALFORMED.  Chunks are staged, uploaded, read, re-tiled and verified in a pipeline.
 * stark101 has canonical texts for the protocol's proof shape (10 layers, paths of 13 / 13 - layer siblings:
 * prover.py:108,143-167 and stark101/scripts/generate_wit.py:13-30); proofs of other shapes -- which no honest
 * prover makes -- are read by the host reader and verified in a batch of their own shape.
 * status_host[i] is the verdict of input i (stage-0 codes above included).  stats may be NULL.      */
int ss_stwo_verify_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts,
                         const size_t *lens, int fmt, uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_files(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *paths, int fmt,
                         uint32_t *status_host, ss_ingest_stats *stats);
/* ss_stwo_verify_texts with the texts in ONE page-locked buffer of the caller's (hipHostMalloc / ss_host_register): text i at
 * byte offs[i] -- multiples of 16, ascending, at least lens[i] apart, offs[n] = the end -- with lens[i] bytes.  Nothing is
 * staged: the DMA engine reads the texts where they are (the bytes between a text's end and the next offset are uploaded with
 * it and ignored); non-canonical texts are read by the host reader from the same buffer.  Same verdicts as ss_stwo_verify_texts. */
int ss_stwo_verify_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *blob, const uint64_t *offs,
                                const size_t *lens, int fmt, uint32_t *status_host, ss_ingest_stats *stats);
/* minimal proof.json texts -> verdicts through the same pipeline: the GPU reader finds each text's list lengths from the
 * member names next to the lists, compares the text with the template those lengths imply and fills a minimal record in
 * capacity form (csrc/ss_text.h, ss_textdev.hip); ss_minimal.hip verifies from there (no per-query record is ever made); texts
 * the GPU reader does not take go to the host readers above.  _pinned: as ss_stwo_verify_texts_pinned.                    */
/* (the same as ss_stwo_verify_texts / _texts_pinned with fmt = SS_TEXT_JSON_MINIMAL; ss_stwo_verify_files takes that fmt too.
 * SS_TEXT_AUTO never picks this form: nothing in such a text names it.)                                                  */
int ss_stwo_verify_minimal_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts,
                                 const size_t *lens, uint32_t *status_host, ss_ingest_stats *stats);
int ss_stwo_verify_minimal_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const void *blob, const uint64_t *offs,
                                        const size_t *lens, uint32_t *status_host, ss_ingest_stats *stats);
int ss_s101_verify_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                         uint32_t *status_host, ss_ingest_stats *stats);
int ss_s101_verify_texts_pinned(ss_ctx *ctx, size_t n, const char *blob, const uint64_t *offs, const size_t *lens, int fmt,
                                uint32_t *status_host, ss_ingest_stats *stats);  /* as ss_stwo_verify_texts_pinned */
int ss_s101_verify_files(ss_ctx *ctx, size_t n, const char *const *paths, int fmt, uint32_t *status_host,
                         ss_ingest_stats *stats);

/* Record -> text, byte for byte what the reference's adapter prints for that proof
 * (stwo-verifier/scripts/generate_wit.py:218-243 for SS_TEXT_WIT; the proof.json schema it reads, :106-245,
 * for SS_TEXT_JSON with `python_separators` 0 = "," ":" as in tests/data/proof.json, 1 = ", " ": " as
 * json.dumps prints).  Returns the text's length; the text is written when it fits `cap` (no terminator).
 * 0 = not writable: unsupported config, a Merkle path whose length is not the config's, or (JSON) a
 * pow_target that is no 2^(64-bits) - 1.  No GPU involved.                                           */
size_t ss_stwo_write_text(const ss_stwo_cfg *cfg, const uint32_t *record, int fmt, int python_separators,
                          char *buf, size_t cap);
/* Shared record -> the shared-path proof.json (SS_TEXT_JSON_SHARED), byte for byte what json.dumps prints for
 * formats.stwo_to_json(proof, shared=True).  0 = `shared` is no shared record of the config.  No GPU involved. */
size_t ss_stwo_write_shared_text(const ss_stwo_cfg *cfg, const uint32_t *shared, size_t words, int python_separators,
                                 char *buf, size_t cap);
/* Diagnostic: would ss_stwo_verify_texts read this text on the GPU (1) or hand it to the host reader (0)?
 * fmt is SS_TEXT_JSON, SS_TEXT_WIT or SS_TEXT_JSON_SHARED (record_out: the per-query record it expands to).  Scalar statement of the GPU reader's rule (ss_text.h); when it
 * returns 1 and record_out is not NULL, record_out holds the record.  No GPU involved.              */
int ss_stwo_text_is_canonical(const ss_stwo_cfg *cfg, const char *text, size_t len, int fmt, uint32_t *record_out);
/* ... fmt SS_TEXT_JSON_MINIMAL: record_out (ss_stwo_minimal_max_words words) receives the minimal record in CAPACITY form --
 * the fixed words, then every list at the base it has when all lists have their largest length, the first n entries of
 * each filled -- which is what the GPU reader writes for such texts (csrc/ss_text.h).  The two forms into each other:  */
int ss_stwo_minimal_from_capacity(const ss_stwo_cfg *cfg, const uint32_t *capacity, uint32_t *minimal_out, size_t cap_words,
                                  size_t *words_out);
int ss_stwo_minimal_to_capacity(const ss_stwo_cfg *cfg, const uint32_t *minimal, size_t words, uint32_t *capacity_out);

/* stark101 twins.  The protocol fixes the shape of a stark101 proof (10 FRI layers, Merkle paths of 13 and
 * 13 - layer siblings: stark101/scripts/fibsquare/prover.py:94-171), so canonical texts exist for that shape only;
 * records here have shape {max_layers 10, max_path 13}.  ss_s101_write_text prints what prover.py's proof.json
 * (json.dumps, `python_separators` as above) resp. stark101/scripts/generate_wit.py:13-30 print.                  */
size_t ss_s101_write_text(const uint32_t *record, int fmt, int python_separators, char *buf, size_t cap);
int ss_s101_text_is_canonical(const char *text, size_t len, int fmt, uint32_t *record_out);

/* The GPU reader alone (diagnostic): n texts of format fmt (SS_TEXT_JSON / SS_TEXT_WIT / SS_TEXT_JSON_SHARED: read
 * into shared records and expanded, all on the GPU) -> records_host
 * (n * ss_stwo_record_words words) and outcome_host[i] = 0 (canonical: record i written by the GPU) or 1
 * (left to the host reader; record i unspecified).  Synchronous; outcome equals ss_stwo_text_is_canonical.
 * SS_TEXT_JSON_MINIMAL: records_host holds n * ss_stwo_minimal_max_words words and receives capacity-form minimal records. */
int ss_stwo_read_texts(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const char *const *texts, const size_t *lens,
                       int fmt, uint32_t *records_host, uint32_t *outcome_host);

int ss_s101_read_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                       uint32_t *records_host, uint32_t *outcome_host); /* records of shape {10, 13} */

/* Per-stage intermediates of one proof after a verify call (the reference's counterpart is the
 * debug tracker of `simfony run`, simfony-cli/src/tracker.rs:48-80, which prints the values a
 * program passes to dbg!).  ss_stwo_ws_layout_of gives the word offsets inside the workspace for
 * callers that read it themselves; ss_stwo_read_intermediates copies the usual ones to the host
 * (synchronises `stream` first; any out pointer may be NULL):
 *   queries[n_queries]         fri_generate_queries (fri/queries.simf:29-43)
 *   oods_point[8]              x.a..x.d, y.a..y.d (channel_draw_qm31_point, channel.simf:143-151)
 *   deep_alpha[4]              the DEEP random coefficient (deep/oods.simf:62)
 *   fold_alphas[4*(1+n_layers)]  fri_commit (fri/commit.simf:70-85)
 *   fri_answers[4*n_queries]   fri_answer of every query (fri/answers.simf:97-130)                */
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
int ss_stwo_read_intermediates(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const void *workspace_dev,
                               size_t proof, void *stream, uint32_t *queries, uint32_t *oods_point,
                               uint32_t *deep_alpha, uint32_t *fold_alphas, uint32_t *fri_answers);

/* The stark101 twin (any out pointer may be NULL; `workspace_dev` as passed to ss_s101_verify_*_dev):
 *   alphas[3]            the composition coefficients (air.simf:30-35)
 *   idx, x, cp           the query (verifier.simf:33), its domain point (:37), the composition value (air.simf:94-101)
 *   folds[max_layers+1]  the value entering FRI layer i -- what fri.simf:77 compares with the layer's cpa -- and, at
 *                        [n_layers], the final one (fri.simf:90); what stark101/scripts/fibsquare/prover_test.py:32-104
 *                        recomputes as `rhs`
 *   state[8]             the channel state after the commitments, before the query draw (verifier.simf:31-33)   */
int ss_s101_read_intermediates(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const void *workspace_dev, size_t proof,
                               void *stream, uint32_t *alphas, uint32_t *idx, uint32_t *x, uint32_t *cp, uint32_t *folds,
                               uint32_t *state);

/* Kernel timing.  With timing enabled every *_verify_batch_dev call records a HIP event pair
 * around each kernel ON THE CALLER'S STREAM (no synchronisation; not graph-capturable, so off
 * by default).  ss_ctx_collect_timing waits for the recorded events, sums them per kernel
 * name (static strings) and clears the list: names[i], total_ms[i], launches[i] for i <
 * return value (<= cap).                                                                  */
int ss_ctx_set_timing(ss_ctx *ctx, int enabled);
int ss_ctx_collect_timing(ss_ctx *ctx, int cap, const char **names, float *total_ms,
                          uint32_t *launches);

/* Device self-test of the primitives (tests only): runs `op` over `n` inputs.
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
 * stwo-verifier/src, SURVEY.md Appendix A -- (tests only; csrc/ss_kat.hip): one reference function per item,
 * evaluated ON THE GPU through the device functions the kernels are built from; tests/test_gpu_kats.py feeds the literals
 * of the reference's `fn test_*` bodies (tests/golden/kats.json) and compares with the expected literals directly.
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
#endif /* SS_VERIFY_H */
