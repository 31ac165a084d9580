/*
 * ss_verify.h -- C ABI of libss_verify.so, the MI355X (gfx950) batch STARK verifier: the core.
 *
 * What this replaces.  The reference has no plugin / FFI seam for its verifier: the boundary is "one process per
 * proof", `simfony run main.simf --witness proof.wit`
 *   - stark101/Makefile:8-9, stwo-verifier/Makefile:17-18        (the call)
 *   - simfony-cli/src/main.rs:163-209 `handle_run`               (compile + satisfy + run)
 *   - simfony-cli/src/main.rs:205 `run_program(..)`              (the verifier executes here)
 *   - simfony-cli/src/main.rs:254-257                            (exit 0 = ACCEPT, 1 = REJECT)
 * whose body is `verify_proof` (stark101/src/verifier.simf:24-42, stwo-verifier/src/verifier.simf:32-58).  What a cgo /
 * JNI / ctypes / Rust-FFI binding for "verify these N witnesses" binds instead is ONE host call, ss_verify_inputs
 * (section 4), and -- for callers whose proofs already live in HBM -- the device entry points of section 3.
 * INTEGRATION.md shows the bindings.  Plain pointers and sizes only.
 *
 * Two more headers hold what a binding does not need first:
 *   ss_verify_forms.h   the shared and minimal input forms (their layouts, sizes, converters) and the named entry
 *                       point of every (form, source) pair ss_verify_inputs dispatches to
 *   ss_verify_test.h    tests and diagnosis only: device replay of the reference's known-answer tests, the GPU text
 *                       reader alone, workspace layout
 *
 * No CPU fallback exists in this library: every verify entry point runs HIP kernels and returns SS_ERR_NO_DEVICE /
 * SS_ERR_HIP if it cannot.
 *
 * Data conventions
 *   word      little-endian uint32_t.
 *   hash      8 words; word j is the big-endian integer of digest bytes 4j..4j+3, i.e. the SHA-256 state word (the
 *             u256 the reference prints, most significant word first).
 *   QM31      4 words (a, b, c, d) of a + bi + (c + di)j  (fields/qm31.simf:15).
 *   record    one proof in natural order (layouts below); what a caller produces.
 *   batch     N records re-tiled for the GPU by ss_*_pack[_dev] (SoA over proofs / queries, Merkle paths in 64-chain
 *             tiles so that a wavefront reads one sibling level as two contiguous 1 KiB bursts): a pure permutation
 *             (+ zero padding) of its records, no hashing, no arithmetic.
 *   status    one uint32_t per proof: 0 = ACCEPT, otherwise the code of the FIRST assert that fails in the reference's
 *             evaluation order (codes below).  The reference only exposes accept / reject; the code is extra.
 */
#ifndef SS_VERIFY_H
#define SS_VERIFY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ================================================================================================ 1. library
 * 2.0 stwo records carry their Merkle path lengths | 2.1 GPU text reader, writers, thread rules | 2.2 shared records |
 * 2.3 minimal records, device KAT replay, stark101 intermediates, caller-pinned buffers | 2.4 ss_verify_inputs (one
 * descriptor entry point), ss_process_defaults is an explicit call (nothing happens at load time any more)            */
#define SS_VERSION 0x00020004

/* return codes (all < 0 are errors; verdicts live in the status array) */
#define SS_OK 0
#define SS_ERR_ARG (-1)       /* null pointer, bad size, unsupported config or combination, n == 0 (an empty batch is the caller's no-op) */
#define SS_ERR_HIP (-2)       /* a HIP call failed; see ss_last_error() */
#define SS_ERR_NO_DEVICE (-3) /* no usable gfx950 device */
#define SS_ERR_WORKSPACE (-4) /* workspace too small */
#define SS_ERR_NOMEM (-5)     /* the host ran out of memory inside the call (the call failed as a whole) */

int ss_version(void);
const char *ss_last_error(void); /* thread-local text of the last error */
int ss_device_count(void);       /* number of visible HIP devices, < 0 on error */
/* sizeof(ss_stwo_cfg) == 40 and sizeof(ss_s101_shape) == 8 as this library was compiled: a binding written in another
 * language checks its own structs against these once at start-up.                                                   */
size_t ss_abi_sizeof_cfg(void);
size_t ss_abi_sizeof_shape(void);

/* Process default, an EXPLICIT call (until 2.3 a constructor did this when the library was loaded): puts
 * GPU_MAX_HW_QUEUES=24 into the process environment unless the variable is already set (the caller's value wins) or
 * SS_KEEP_ENV is.  The HIP runtime reads the variable once, at the process's first HIP call, so call this before that --
 * and before other threads exist: setenv is not thread-safe.  Its default of 4 hardware queues serialises streams that
 * share one, which is what callers keeping several passes in flight (section 3) must avoid; 24 is the smallest count
 * that holds every measured rate (profiles/r06_hw_queues_sweep.txt: stark101 x 4 096 on 16 streams 44.7 M proofs/s at
 * 4, 51 M at 16, 61 M at 24 and 32; one batch of 65 536 stwo proofs per pass does not depend on it).  Verdicts never
 * depend on it.  Returns the queue count now in the environment (0 = unset).  The Python binding applies the same rule on
 * import (in Python: loading this library before torch would bind the system's HIP runtime ahead of torch's).         */
int ss_process_defaults(void);

/* ============================================================================================= 2. proofs
 * ---- stark101.  Record (ss_s101_record_words(shape) words, zero padded):
 *   root[8]  n_layers  last
 *   3 x { ev, len, path[max_path][8] }                        (air.simf:24-27)
 *   max_layers x { root[8], beta,
 *                  cpa_ev, cpa_len, cpa_path[max_path][8],
 *                  cpb_ev, cpb_len, cpb_path[max_path][8] }   (fri.simf:30)
 * paths are leaf -> root, as the prover emits them (prover.py:144-146,162-164).
 * Status codes: (stage << 8) | sub
 *   1 beta mismatch (fri.simf:43) sub=layer | 2 trace Merkle (air.simf:41) sub=k |
 *   3 composition division abort (field.simf:46) sub=0..2 |
 *   4 FRI layer, sub = 4*layer + {0 chain fri.simf:77, 1 cpa Merkle :79, 2 cpb Merkle :80, 3 fold division :58-60} |
 *   5 last value (fri.simf:90)                                                                                        */
typedef struct ss_s101_shape {
    uint32_t max_layers; /* <= 31 */
    uint32_t max_path;   /* <= 31 */
} ss_s101_shape;

/* ---- stwo.  Runtime form of the compile-time macros of stwo-verifier/src/config.simf:10-51.  The config is the
 * CALLER's expectation: a proof that declares, or has the shape of, another one is never verified against its own.   */
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

/* SS_MODE_LITERAL follows the .simf text (single DEEP batch fri/answers.simf:97-130, `log_size_ex == 0`
 * fri/verify.simf:127, `folded_query == 0` fri/layers.simf:75).  SS_MODE_FIXTURE is what the reference's own proofs
 * (tests/data/proof*.json) satisfy: trace columns sampled at P and the 16 composition columns at 2P as two DEEP batches
 * (docs/batching_samples.md:62-70), and neither of those two asserts.                                */
#define SS_MODE_LITERAL 0u
#define SS_MODE_FIXTURE 1u
/* SS_HASH_SHA256 is the reference (hasher.simf:13-104, channel.simf:36-172).  SS_HASH_BLAKE2S is the "Blake2s Merkle"
 * variant BASELINE.json names: the same protocol over the same byte strings with Blake2s-256 (RFC 7693).  The reference
 * has no Blake2s: that variant's parity is unpinned (RFC vectors + prover / oracle / GPU agreement only).             */
#define SS_HASH_SHA256 0u
#define SS_HASH_BLAKE2S 1u
/* SS_FLAG_NO_DEDUP: hash every query's Merkle path in full, as the reference does (fri/queries.simf:41 notes that it does
 * not deduplicate).  By default each distinct (left, right) pair of the top levels of a tree is hashed once per proof and
 * every query presenting that node is checked byte for byte to present the same pair; trees where two queries disagree
 * are re-hashed query by query, so the status words are the reference's either way.
 * SS_FLAG_TOP_CHECKS: make those byte compares in the top kernel for every query count (default: in the merkle kernel
 * when the query count divides 64).  Same status words; for A/B runs and tests.                                        */
#define SS_FLAG_NO_DEDUP 1u
#define SS_FLAG_TOP_CHECKS 2u

/* Record (ss_stwo_record_words words):
 *   roots[3][8]  oods_trace[n_cols][4]  oods_cp[16][4]  fri_roots[1+n_layers][8]  last_layer[4]  pow_nonce_hi  _lo
 *   n_queries x { trace_vals[n_cols], cp_vals[16], trace_path[lde_log][8], cp_path[lde_log][8] }
 *   (1+n_layers) x n_queries x { witness[4], path[lde_log-1-layer][8] }
 *   path_len[3+n_layers][n_queries]      kind 0 trace, 1 cp, 2+l FRI layer l
 * The reference types every Merkle path List<u256, 32> (scripts/generate_wit.py:70-103): its length is data.  path_len
 * carries the length the proof really has; the fixed slots hold the first min(length, slot) siblings, zero padded.  A
 * path whose length differs from the slot cannot verify in the reference whatever it contains (`path == 1`,
 * merkle.simf:42) and the library reports the code of exactly that assert -- the caller computes no part of the verdict.
 * Status codes: (stage << 24) | (layer << 16) | (query << 4) | sub
 *   stage 0 is decided on the host, before any kernel (smaller than every assert code):
 *     SS_STATUS_CONFIG_MISMATCH  a well-formed witness of another shape / declared parameters -- the reference would
 *                                fail to type it (main.rs:77-81,187-190)
 *     SS_STATUS_MALFORMED        not a witness of the reference's types at all
 *   1 channel draw exhausted | 2 OODS (sub 1 point inverse, 2 vanishing inverse, 3 CP mismatch deep/oods.simf:58) |
 *   4 proof of work (pow.simf:33) | 5 decommit (sub 0 trace path, 1 trace root, 2 cp path, 3 cp root) | 6 DEEP
 *   denominator abort (sub = batch) | 7 FRI layer (sub 0 path, 1 root, 2 fold inverse) | 8 log_size_ex != 0 [LITERAL] |
 *   9 last layer (sub 0 folded_query != 0 [LITERAL], 1 value mismatch fri/layers.simf:76)                             */
#define SS_STATUS_CONFIG_MISMATCH 1u
#define SS_STATUS_MALFORMED 2u

/* Sizes and host packers: pure functions of their arguments (no environment, no process state).  records[i] -> proof i
 * (pointers may repeat); batch_host receives ss_*_batch_words words.                                                   */
size_t ss_s101_record_words(const ss_s101_shape *shape);
size_t ss_s101_batch_words(const ss_s101_shape *shape, size_t n);
size_t ss_s101_workspace_bytes(const ss_s101_shape *shape, size_t n);
int ss_s101_pack(const ss_s101_shape *shape, size_t n, const uint32_t *const *records, uint32_t *batch_host);
size_t ss_stwo_record_words(const ss_stwo_cfg *cfg);
size_t ss_stwo_batch_words(const ss_stwo_cfg *cfg, size_t n);
size_t ss_stwo_workspace_bytes(const ss_stwo_cfg *cfg, size_t n);
int ss_stwo_pack(const ss_stwo_cfg *cfg, size_t n, const uint32_t *const *records, uint32_t *batch_host);

/* ========================================================================================== 3. execution
 * One context per process and GPU (one process per GPU is the intended deployment; a process may hold one per GPU:
 * every entry point makes its context's device current for its duration and restores the caller's).
 * Threads.  A context may be shared by threads.  The host entry points (section 4) use the context's own scratch and
 * serialise on a lock inside it.  The device entry points below take caller-owned buffers and a caller-owned stream and
 * run concurrently from any number of threads; what they require is the usual HIP rule that the buffers of two
 * in-flight calls are distinct.
 * Streams.  A device entry point orders its work on `stream` and on nothing else.  In particular the status words and
 * the accept count of a pass are written on `stream` (reset, kernels, finalize) and are the verifier's alone from the
 * call until `stream` has drained: a write the caller enqueues on another stream -- the null stream included, which
 * non-blocking streams do not wait for -- is ordered against the pass only by the caller's own events
 * (verifier.Pipeline / IndependentStreams join the caller's current stream for that reason).                         */
typedef struct ss_ctx ss_ctx;
int ss_ctx_create(int device, ss_ctx **out);
void ss_ctx_destroy(ss_ctx *ctx);

/* The permutation of ss_*_pack done by the GPU: records_dev holds the n records back to back (device memory),
 * batch_dev receives ss_*_batch_words words.  Asynchronous on `stream`.                                              */
int ss_stwo_pack_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *records_dev, uint32_t *batch_dev,
                     void *stream);
int ss_s101_pack_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const uint32_t *records_dev,
                     uint32_t *batch_dev, void *stream);

/* Device-resident verification (what bench.py times): every pointer is device memory on ctx's GPU, `stream` is a
 * hipStream_t (NULL = the null stream).  Asynchronous; status_dev is valid once the stream has drained.
 * accept_count_dev (may be NULL) receives the number of accepted proofs (one uint32_t) -- the value a multi-GPU caller
 * all-reduces.  No allocation and no synchronisation happen inside, so the call is hipGraph-capturable.
 * The same work in two separately enqueueable halves, for callers that pipeline batches:
 *   SS_PHASE_HEAD  reset status, transcript kernel (Fiat-Shamir chain, latency bound) and query kernel -> workspace
 *   SS_PHASE_TAIL  Merkle kernels (ALU bound) and finalize <- workspace
 * HEAD of batch i+1 on one stream overlaps TAIL of batch i on another (each batch needs its own workspace / status;
 * order HEAD -> TAIL of one batch with an event).  SS_PHASE_ALL on one stream is exactly ss_*_verify_batch_dev.       */
#define SS_PHASE_HEAD 1
#define SS_PHASE_TAIL 2
#define SS_PHASE_ALL 3
int ss_s101_verify_batch_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const uint32_t *batch_dev,
                             void *workspace_dev, size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, void *stream);
int ss_stwo_verify_batch_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *batch_dev,
                             void *workspace_dev, size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, void *stream);
int ss_s101_verify_phase_dev(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const uint32_t *batch_dev,
                             void *workspace_dev, size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, int phases, void *stream);
int ss_stwo_verify_phase_dev(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const uint32_t *batch_dev,
                             void *workspace_dev, size_t workspace_bytes, uint32_t *status_dev,
                             uint32_t *accept_count_dev, int phases, void *stream);

/* ================================================================================== 4. inputs in host memory
 * ONE entry point for every way the reference's callers hold a witness: synchronous, PCIe-inclusive (never what
 * bench.py's `value` times), scratch (pinned staging, device buffers) owned by the context and only growing.  The
 * descriptor says which proof family, what ONE input is, and where the n inputs lie:
 *
 *   form  SS_FORM_RECORDS          per-query records, the layouts of section 2
 *         SS_FORM_SHARED_RECORDS   every distinct Merkle sibling of a tree once          (stwo; ss_verify_forms.h)
 *         SS_FORM_MINIMAL_RECORDS  one sorted, deduplicated decommitment per tree        (stwo; ss_verify_forms.h)
 *         SS_FORM_TEXT             the TEXT the reference's callers hand over, text_fmt = SS_TEXT_*: proof.json (stwo: the
 *                                  schema read by stwo-verifier/scripts/generate_wit.py:106-245; stark101:
 *                                  fibsquare/prover.py:108,143-167) or proof.wit (generate_wit.py:218-243,
 *                                  stark101/scripts/generate_wit.py:13-30), the file `simfony run --witness` consumes
 *                                  (main.rs:163-209).  The raw bytes are uploaded in pinned chunks and turned into
 *                                  records ON THE GPU: a text that is, byte for byte, what the reference's producers write
 *                                  for the expected config except for its numbers and for whitespace outside strings is
 *                                  read without a parse tree; every other text goes to the host reader behind
 *                                  ss_stwo_parse, which alone decides parsed / CONFIG_MISMATCH / MALFORMED.
 *   source SS_SRC_HOST    items[i] -> input i in ordinary memory; lens[i] = its words (shared / minimal records) or bytes
 *                         (texts); per-query records have the size of their config / shape.  Stager threads copy the
 *                         bytes into pinned chunks (about eight cores feed a 55 GB/s link).
 *          SS_SRC_PINNED  all inputs in ONE page-locked buffer `blob` (hipHostMalloc, or registered with
 *                         ss_host_register): the DMA engine reads them where they are, no host thread touches a byte --
 *                         what a rank of an 8-GPU host should use.  Records: back to back (per-query: fixed stride, offs
 *                         unused; shared / minimal: record i at WORD offset offs[i], offs[n] = the total).  Texts: text i
 *                         at BYTE offset offs[i] (multiples of 16, ascending, offs[n] = the end) with lens[i] bytes.
 *                         SS_ERR_ARG when the buffer is not page-locked.
 *          SS_SRC_FILES   items[i] = path of a text file (SS_FORM_TEXT); an unreadable file is SS_STATUS_MALFORMED
 *                         (main.rs:187-190: a witness that cannot be loaded is exit 1).
 * stark101 takes SS_FORM_RECORDS from SS_SRC_HOST and SS_FORM_TEXT from all three; stwo takes every pair except files
 * of records.  status_host[i] is the verdict of input i, stage-0 codes included; verdicts do not depend on the source.
 * stats (may be NULL) is filled for texts and zeroed otherwise.                                                       */
#define SS_FAMILY_STARK101 1u
#define SS_FAMILY_STWO 2u
#define SS_FORM_RECORDS 0u
#define SS_FORM_SHARED_RECORDS 1u
#define SS_FORM_MINIMAL_RECORDS 2u
#define SS_FORM_TEXT 3u
#define SS_SRC_HOST 0u
#define SS_SRC_PINNED 1u
#define SS_SRC_FILES 2u
#define SS_TEXT_AUTO 0         /* sniff: a .wit is a JSON object with a COMMITMENTS / P_MT_ROOT member */
#define SS_TEXT_JSON 1
#define SS_TEXT_WIT 2
#define SS_TEXT_JSON_SHARED 3  /* the shared-path proof.json (ss_verify_forms.h); AUTO and JSON recognise it by its "queries" member */
#define SS_TEXT_JSON_MINIMAL 4 /* the minimal proof.json (ss_verify_forms.h); nothing in such a text names it: AUTO never picks it */

typedef struct ss_input_desc {
    uint32_t family;            /* SS_FAMILY_*                                                  */
    uint32_t form;              /* SS_FORM_*                                                    */
    uint32_t source;            /* SS_SRC_*                                                     */
    uint32_t text_fmt;          /* SS_TEXT_* (SS_FORM_TEXT)                                     */
    const ss_stwo_cfg *cfg;     /* stwo: the config the caller expects                          */
    const ss_s101_shape *shape; /* stark101 records: their shape (stark101 texts carry their own) */
    size_t n;                   /* inputs                                                       */
    const void *const *items;   /* SS_SRC_HOST: n inputs; SS_SRC_FILES: n C strings             */
    const size_t *lens;         /* see `source`                                                 */
    const void *blob;           /* SS_SRC_PINNED                                                */
    const uint64_t *offs;       /* SS_SRC_PINNED: n + 1 offsets                                 */
} ss_input_desc;

typedef struct ss_ingest_stats {
    double read_s;    /* wall time staging the raw bytes (copy / file read into pinned memory)  */
    double parse_s;   /* wall time in the host reader (texts the GPU reader did not take)       */
    double total_s;   /* the whole call: stage + upload + GPU read + verify + download          */
    uint64_t text_bytes, record_bytes;
    uint32_t threads; /* host threads used (scheduler affinity capped by the cgroup quota) */
    uint32_t host_parsed; /* texts that went through the host reader (0 for canonical texts) */
} ss_ingest_stats;

int ss_verify_inputs(ss_ctx *ctx, const ss_input_desc *in, uint32_t *status_host, ss_ingest_stats *stats);
int ss_host_register(ss_ctx *ctx, void *ptr, size_t bytes); /* page-lock existing memory in place (hipHostRegister) */
int ss_host_unregister(ss_ctx *ctx, void *ptr);

/* =================================================================== 5. the reference's text formats, no GPU
 * One text -> one record.  ss_stwo_parse returns 0 = parsed, SS_STATUS_CONFIG_MISMATCH, SS_STATUS_MALFORMED (record_out
 * untouched or zeroed), or < 0 on a bad argument.  stark101: the shape is data -- ss_s101_parse returns 0 /
 * SS_STATUS_MALFORMED and the proof's shape; record_out (may be NULL) receives ss_s101_record_words(shape_inout) words
 * when the proof fits the shape passed in (layers / path lengths up to 31 always fit {31, 31}).                      */
int ss_stwo_parse(const ss_stwo_cfg *cfg, const char *text, size_t len, int fmt, uint32_t *record_out);
int ss_s101_parse(const char *text, size_t len, int fmt, ss_s101_shape *shape_inout, uint32_t *record_out);
/* Record -> text, byte for byte what the reference's adapter prints for that proof (stwo-verifier/scripts/
 * generate_wit.py:218-243 for SS_TEXT_WIT; the proof.json schema it reads, :106-245, for SS_TEXT_JSON with
 * `python_separators` 0 = "," ":" as in tests/data/proof.json, 1 = ", " ": " as json.dumps prints).  Returns the text's
 * length; the text is written when it fits `cap` (no terminator).  0 = not writable: unsupported config, a Merkle path
 * whose length is not the config's, or (JSON) a pow_target that is no 2^(64-bits) - 1.  The stark101 twin prints what
 * prover.py's proof.json resp. stark101/scripts/generate_wit.py:13-30 print, for the protocol's one proof shape (10 FRI
 * layers, paths of 13 and 13 - layer siblings: fibsquare/prover.py:94-171; record of shape {10, 13}).                 */
size_t ss_stwo_write_text(const ss_stwo_cfg *cfg, const uint32_t *record, int fmt, int python_separators, char *buf,
                          size_t cap);
size_t ss_s101_write_text(const uint32_t *record, int fmt, int python_separators, char *buf, size_t cap);

/* ================================================================================== 6. intermediates, timing
 * Per-stage values of one proof after a verify call -- the reference's counterpart is the debug tracker of `simfony
 * run` (simfony-cli/src/tracker.rs:48-80), which prints the values a program passes to dbg!.  Synchronises `stream`
 * first; any out pointer may be NULL; `workspace_dev` as passed to the device entry point.
 *   stwo      queries[n_queries]           fri_generate_queries (fri/queries.simf:29-43)
 *             oods_point[8]                x.a..x.d, y.a..y.d (channel_draw_qm31_point, channel.simf:143-151)
 *             deep_alpha[4]                the DEEP random coefficient (deep/oods.simf:62)
 *             fold_alphas[4*(1+n_layers)]  fri_commit (fri/commit.simf:70-85)
 *             fri_answers[4*n_queries]     fri_answer of every query (fri/answers.simf:97-130)
 *   stark101  alphas[3]                    the composition coefficients (air.simf:30-35)
 *             idx, x, cp                   the query (verifier.simf:33), its domain point (:37), the composition value
 *                                          (air.simf:94-101)
 *             folds[max_layers+1]          the value entering FRI layer i -- what fri.simf:77 compares with the layer's
 *                                          cpa -- and, at [n_layers], the final one (fri.simf:90); what
 *                                          stark101/scripts/fibsquare/prover_test.py:32-104 recomputes as `rhs`
 *             state[8]                     the channel state after the commitments, before the query draw
 *                                          (verifier.simf:31-33)                                                     */
int ss_stwo_read_intermediates(ss_ctx *ctx, const ss_stwo_cfg *cfg, size_t n, const void *workspace_dev, size_t proof,
                               void *stream, uint32_t *queries, uint32_t *oods_point, uint32_t *deep_alpha,
                               uint32_t *fold_alphas, uint32_t *fri_answers);
int ss_s101_read_intermediates(ss_ctx *ctx, const ss_s101_shape *shape, size_t n, const void *workspace_dev,
                               size_t proof, void *stream, uint32_t *alphas, uint32_t *idx, uint32_t *x, uint32_t *cp,
                               uint32_t *folds, uint32_t *state);
/* Kernel timing.  With timing enabled every device entry point records a HIP event pair around each kernel ON THE
 * CALLER'S STREAM (no synchronisation; not graph-capturable, so off by default).  ss_ctx_collect_timing waits for the
 * recorded events, sums them per kernel name (static strings) and clears the list: names[i], total_ms[i], launches[i]
 * for i < return value (<= cap).                                                                                      */
int ss_ctx_set_timing(ss_ctx *ctx, int enabled);
int ss_ctx_collect_timing(ss_ctx *ctx, int cap, const char **names, float *total_ms, uint32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* SS_VERIFY_H */
