/*
 * ss_verify_file.c -- the C ABI of libss_verify.so from plain C (no Python, no torch).
 *
 * What a non-Python host (the reference's Rust CLI through FFI) does:
 * read raw records, call ss_stwo_verify_records, map the status words to the exit status of
 * `simfony run` (simfony-cli/src/main.rs:254-257: 0 = every proof accepted, 1 otherwise).
 *
 *   ss_verify_file stwo <n_cols> <trace_log> <lde_log> <n_queries> <n_layers> <pow_bits> <hash 0|1> \
 *                  <mode 0|1> records.bin
 *   ss_verify_file stwo-shared <the same nine numbers> shared.bin      (ABI 2.2: SHARED records back to back -- every
 *                  distinct Merkle sibling once; a record says how long it is: fixed words + 8 * sum of its counts)
 *   ss_verify_file stwo-minimal <the same nine numbers> minimal.bin    (ABI 2.3: MINIMAL records back to back -- one sorted,
 *                  deduplicated decommitment per tree; the record's count words say how long its lists are)
 *   ss_verify_file stwo-text <the same nine numbers> <fmt 0..4> file...      (texts: the library reads the files itself and
 *                  parses them on the GPU; fmt = SS_TEXT_AUTO / _JSON / _WIT / _JSON_SHARED / _JSON_MINIMAL)
 *   ss_verify_file stark101 <max_layers> <max_path> records.bin
 *
 * records.bin = the records back to back, little-endian u32 words (include/ss_verify.h; written by
 * stark_symphony_amd.records / verifier.stwo_record).  Build:
 *   gcc -O2 -Iinclude examples/ss_verify_file.c -o build/ss_verify_file -Lstark-symphony_amd -lss_verify \
 *       -Wl,-rpath,$PWD/stark-symphony_amd
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ss_verify.h"
#include "ss_verify_forms.h" /* the layouts of shared / minimal records */

static uint32_t *read_words(const char *path, size_t *n_words)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (bytes <= 0 || bytes % 4) { fprintf(stderr, "%s: not a whole number of words\n", path); fclose(f); return NULL; }
    uint32_t *w = (uint32_t *)malloc((size_t)bytes);
    if (!w || fread(w, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "%s: read failed\n", path); fclose(f); free(w); return NULL; }
    fclose(f);
    *n_words = (size_t)bytes / 4;
    return w;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: see the header of examples/ss_verify_file.c\n"); return 2; }
    const int shared = strcmp(argv[1], "stwo-shared") == 0, minimal = strcmp(argv[1], "stwo-minimal") == 0;
    const int text = strcmp(argv[1], "stwo-text") == 0;
    const int stwo = shared || minimal || text || strcmp(argv[1], "stwo") == 0;
    if ((stwo && !text && argc != 11) || (text && argc < 12) || (!stwo && (strcmp(argv[1], "stark101") != 0 || argc != 5))) {
        fprintf(stderr, "usage: see the header of examples/ss_verify_file.c\n");
        return 2;
    }
    ss_stwo_cfg cfg;
    ss_s101_shape shape;
    size_t rec_words;
    const char *path;
    memset(&cfg, 0, sizeof cfg);
    if (stwo) {
        cfg.n_cols = (uint32_t)atoi(argv[2]);
        cfg.trace_log = (uint32_t)atoi(argv[3]);
        cfg.lde_log = (uint32_t)atoi(argv[4]);
        cfg.n_queries = (uint32_t)atoi(argv[5]);
        cfg.n_layers = (uint32_t)atoi(argv[6]);
        const int pow_bits = atoi(argv[7]);
        /* POW_TARGET_64 = 2^(64 - bits) - 1, compared with lt_64 (config.simf:32, pow.simf:32) */
        cfg.pow_target = pow_bits <= 0 ? UINT64_MAX : (pow_bits >= 64 ? 0 : ((uint64_t)1 << (64 - pow_bits)) - 1);
        cfg.hash = (uint32_t)atoi(argv[8]);
        cfg.mode = (uint32_t)atoi(argv[9]);
        rec_words = ss_stwo_record_words(&cfg);
        path = argv[10];
    } else {
        shape.max_layers = (uint32_t)atoi(argv[2]);
        shape.max_path = (uint32_t)atoi(argv[3]);
        rec_words = ss_s101_record_words(&shape);
        path = argv[4];
    }
    if (!rec_words) { fprintf(stderr, "unsupported configuration\n"); return 2; }
    if (text) {  /* files of text: nothing is read or parsed here */
        const size_t nf = (size_t)argc - 11;
        uint32_t *st = (uint32_t *)malloc(nf * sizeof *st);
        ss_ingest_stats stats;
        ss_ctx *tctx = NULL;
        int trc = ss_ctx_create(0, &tctx);
        if (trc == SS_OK) {
            ss_input_desc in;
            memset(&in, 0, sizeof in);
            in.family = SS_FAMILY_STWO; in.form = SS_FORM_TEXT; in.source = SS_SRC_FILES; in.text_fmt = (uint32_t)atoi(argv[10]);
            in.cfg = &cfg; in.n = nf; in.items = (const void *const *)(argv + 11);
            trc = ss_verify_inputs(tctx, &in, st, &stats);
        }
        if (trc != SS_OK) { fprintf(stderr, "libss_verify: %s (code %d)\n", ss_last_error(), trc); return 2; }
        size_t bad = 0;
        for (size_t i = 0; i < nf; i++) {
            if (st[i]) { bad++; printf("%s: REJECT (0x%08x)\n", argv[11 + i], st[i]); }
            else printf("%s: ACCEPT\n", argv[11 + i]);
        }
        printf("%u of %zu texts went through the host reader\n", stats.host_parsed, nf);
        ss_ctx_destroy(tctx);
        free(st);
        return bad ? 1 : 0;
    }
    size_t n_words = 0;
    uint32_t *words = read_words(path, &n_words);
    if (!words) return 2;
    size_t n = 0;
    const uint32_t **recs;
    size_t *lens = NULL;
    if (shared) {
        /* split the file: a shared record is ss_stwo_shared_fixed_words words -- the last 3 + n_layers of them its
         * counts -- followed by 8 words per counted node.  Whether the counts are the ones its positions imply is
         * the library's business (SS_STATUS_MALFORMED), not this program's. */
        const size_t fixed = ss_stwo_shared_fixed_words(&cfg), trees = 3 + cfg.n_layers, max_words = ss_stwo_shared_max_words(&cfg);
        if (fixed == 0) { fprintf(stderr, "unsupported stwo config\n"); return 2; }  /* (the size functions return 0 for one) */
        recs = (const uint32_t **)malloc((n_words / fixed + 1) * sizeof *recs);
        lens = (size_t *)malloc((n_words / fixed + 1) * sizeof *lens);
        for (size_t o = 0; o < n_words;) {
            size_t len = fixed;
            if (o + fixed > n_words) len = n_words - o;  /* a truncated tail: handed over as it is */
            else {
                for (size_t t = 0; t < trees; t++) len += 8 * (size_t)words[o + fixed - trees + t];
                if (len > max_words || o + len > n_words) len = n_words - o;
            }
            recs[n] = words + o;
            lens[n++] = len;
            o += len;
        }
    } else if (minimal) {
        /* a minimal record: ss_stwo_minimal_fixed_words words, the last 2 + (n_layers + 1) + (n_layers + 3) of them its list
         * lengths -- value rows of the two trees, fri_witness entries per layer, hashes per tree -- then the lists */
        const size_t fixed = ss_stwo_minimal_fixed_words(&cfg), K = cfg.n_layers, nc = 2 + (K + 1) + (K + 3);
        const size_t max_words = ss_stwo_minimal_max_words(&cfg);
        if (fixed == 0) { fprintf(stderr, "unsupported stwo config\n"); return 2; }
        recs = (const uint32_t **)malloc((n_words / fixed + 1) * sizeof *recs);
        lens = (size_t *)malloc((n_words / fixed + 1) * sizeof *lens);
        for (size_t o = 0; o < n_words;) {
            size_t len = fixed;
            if (o + fixed > n_words) len = n_words - o;
            else {
                const uint32_t *c = words + o + fixed - nc;
                len += (size_t)c[0] * cfg.n_cols + (size_t)c[1] * 16;
                for (size_t l = 0; l <= K; l++) len += 4 * (size_t)c[2 + l];
                for (size_t t = 0; t < K + 3; t++) len += 8 * (size_t)c[3 + K + t];
                if (len > max_words || o + len > n_words) len = n_words - o;
            }
            recs[n] = words + o;
            lens[n++] = len;
            o += len;
        }
    } else {
        if (n_words % rec_words) { fprintf(stderr, "%s: %zu words is not a multiple of the %zu-word record\n", path, n_words, rec_words); return 2; }
        n = n_words / rec_words;
        recs = (const uint32_t **)malloc(n * sizeof *recs);
        for (size_t i = 0; i < n; i++) recs[i] = words + i * rec_words;
    }
    uint32_t *status = (uint32_t *)malloc((n ? n : 1) * sizeof *status);

    ss_ctx *ctx = NULL;
    int rc = ss_ctx_create(0, &ctx);
    if (rc == SS_OK) {
        /* one descriptor for all four: the form of one input, where the n inputs lie (include/ss_verify.h section 4) */
        ss_input_desc in;
        memset(&in, 0, sizeof in);
        in.family = stwo ? SS_FAMILY_STWO : SS_FAMILY_STARK101;
        in.form = shared ? SS_FORM_SHARED_RECORDS : minimal ? SS_FORM_MINIMAL_RECORDS : SS_FORM_RECORDS;
        in.source = SS_SRC_HOST;
        in.cfg = stwo ? &cfg : NULL;
        in.shape = stwo ? NULL : &shape;
        in.n = n;
        in.items = (const void *const *)recs;
        in.lens = lens;  /* words of every shared / minimal record; per-query records have their config's size */
        rc = ss_verify_inputs(ctx, &in, status, NULL);
    }
    if (rc != SS_OK) {  /* no CPU fallback: a missing GPU is an error, never a verdict */
        fprintf(stderr, "libss_verify: %s (code %d)\n", ss_last_error(), rc);
        return 2;
    }
    size_t rejected = 0;
    for (size_t i = 0; i < n; i++) {
        if (status[i]) { rejected++; printf("proof %zu: REJECT (first failing assert 0x%08x)\n", i, status[i]); }
        else printf("proof %zu: ACCEPT\n", i);
    }
    ss_ctx_destroy(ctx);
    free(status); free(recs); free(lens); free(words);
    return rejected ? 1 : 0;
}
