/*
 * ss_run.c -- `simfony run`'s calling convention on top of libss_verify.so, in plain C.
 *
 * The reference verifies with
 *     simfony run <program.simf> --witness <proof.wit>          (stark101/Makefile:8-9, stwo-verifier/Makefile:17-18)
 * and its callers read the exit status: 0 = the program ran (ACCEPT), 1 = `Error: Failed to run program
 * ...` (a failed assert) or a witness that does not type-check (simfony-cli/src/main.rs:77-81,187-190,
 * 205-206,254-257).  This program takes the same arguments, hands the .wit file to the library's native
 * reader (ss_verify_inputs: SS_FORM_TEXT from SS_SRC_FILES) and returns the same exit status -- a one-line change in a Makefile.  The
 * program file is not executed (the verifier it contains is the library's kernels); it only selects
 * the witness family when --family is not given: from its path ("stark101" / "stwo"), else from its
 * text (`witness::P_MT_ROOT` is read by stark101/src/main.simf:13, `witness::COMMITMENTS` by
 * stwo-verifier/src/main.simf:10).  The family is NEVER taken from a witness -- that is the untrusted
 * input -- and a witness of the other family is malformed (exit 1), as it fails typing in the reference.
 * If the family cannot be determined the exit status is 2.  For stwo the config is the one the program
 * was compiled for:
 * production by default, --config testing for `mcpp -DTESTING` builds (config.simf:10-51), or explicit
 * --n-cols / --trace-log / --lde-log / --n-queries / --n-layers / --pow-bits overrides.
 *
 *   ss_run run <program.simf> --witness <proof.wit> [--witness more.wit ...]
 *              [--family stark101|stwo] [--config production|testing] [--mode fixture|literal] [--device N]
 *
 * Build:  gcc -O2 -Iinclude examples/ss_run.c -o build/ss_run -Lstark-symphony_amd -lss_verify \
 *             -Wl,-rpath,$PWD/stark-symphony_amd
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ss_verify.h"

/* Does the (trusted) program text contain `needle`?  Reads the whole file in 64 KiB pieces. */
static int file_mentions(const char *path, const char *needle)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    static char buf[(1 << 16) + 64];
    const size_t nl = strlen(needle);
    size_t keep = 0, n;
    int found = 0;
    while (!found && (n = fread(buf + keep, 1, (1 << 16), f)) > 0) {
        const size_t have = keep + n;
        buf[have] = 0;
        for (size_t i = 0; i < have; i++)  /* NUL bytes would end strstr early */
            if (!buf[i]) buf[i] = ' ';
        found = strstr(buf, needle) != NULL;
        keep = have < nl - 1 ? have : nl - 1;  /* a needle may straddle two reads */
        memmove(buf, buf + have - keep, keep);
    }
    fclose(f);
    return found;
}

/* 1 = stark101, 0 = stwo, -1 = cannot tell.  Only trusted input decides: --family, the program's
 * path, the program's text.  (The witness is the untrusted input and never selects the statement.) */
static int family_of(const char *family, const char *program)
{
    if (family) return !strcmp(family, "stark101") ? 1 : !strcmp(family, "stwo") ? 0 : -1;
    const int p101 = strstr(program, "stark101") != NULL, pstwo = strstr(program, "stwo") != NULL;
    if (p101 != pstwo) return p101;
    const int t101 = file_mentions(program, "witness::P_MT_ROOT"), tstwo = file_mentions(program, "witness::COMMITMENTS");
    if (t101 != tstwo) return t101;
    return -1;
}

int main(int argc, char **argv)
{
    if (argc < 3 || strcmp(argv[1], "run") != 0) {
        fprintf(stderr, "usage: ss_run run <program.simf> --witness <proof.wit> [...]  (see the header of examples/ss_run.c)\n");
        return 2;
    }
    const char *program = argv[2], *family = NULL, *profile = "production", *mode = "fixture";
    const char **wits = (const char **)malloc((size_t)argc * sizeof *wits);  /* every --witness is verified */
    if (!wits) { fprintf(stderr, "Error: out of memory\n"); return 2; }
    size_t n = 0;
    int device = 0;
    long over[6] = {-1, -1, -1, -1, -1, -1};  /* n_cols trace_log lde_log n_queries n_layers pow_bits */
    static const char *over_names[6] = {"--n-cols", "--trace-log", "--lde-log", "--n-queries", "--n-layers", "--pow-bits"};
    for (int i = 3; i < argc; i++) {
        const char *a = argv[i];
        const char *v = i + 1 < argc ? argv[i + 1] : NULL;
        int taken = 0;
        if (!strcmp(a, "--witness") && v) { wits[n++] = v; taken = 1; }
        else if (!strcmp(a, "--family") && v) { family = v; taken = 1; }
        else if (!strcmp(a, "--config") && v) { profile = v; taken = 1; }
        else if (!strcmp(a, "--mode") && v) { mode = v; taken = 1; }
        else if (!strcmp(a, "--device") && v) { device = atoi(v); taken = 1; }
        else {
            for (int k = 0; k < 6; k++)
                if (!strcmp(a, over_names[k]) && v) { over[k] = atol(v); taken = 1; }
        }
        if (!taken) { fprintf(stderr, "Error: unknown or incomplete argument %s\n", a); free(wits); return 2; }
        i++;
    }
    if (!n) { fprintf(stderr, "Error: no --witness given\n"); free(wits); return 1; }
    const int s101 = family_of(family, program);
    if (s101 < 0) {
        fprintf(stderr, "Error: cannot tell which verifier %s is (stark101 or stwo): pass --family\n", program);
        free(wits);
        return 2;
    }

    ss_stwo_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    if (!strcmp(profile, "testing")) { cfg.n_cols = 4; cfg.trace_log = 3; cfg.lde_log = 4; cfg.n_queries = 1; cfg.n_layers = 2; }
    else { cfg.n_cols = 4; cfg.trace_log = 9; cfg.lde_log = 13; cfg.n_queries = 16; cfg.n_layers = 8; }   /* config.simf:34-52 */
    long pow_bits = 5;
    if (over[0] >= 0) cfg.n_cols = (uint32_t)over[0];
    if (over[1] >= 0) cfg.trace_log = (uint32_t)over[1];
    if (over[2] >= 0) cfg.lde_log = (uint32_t)over[2];
    if (over[3] >= 0) cfg.n_queries = (uint32_t)over[3];
    if (over[4] >= 0) cfg.n_layers = (uint32_t)over[4];
    if (over[5] >= 0) pow_bits = over[5];
    cfg.pow_target = pow_bits <= 0 ? UINT64_MAX : (pow_bits >= 64 ? 0 : ((uint64_t)1 << (64 - pow_bits)) - 1);
    cfg.mode = !strcmp(mode, "literal") ? SS_MODE_LITERAL : SS_MODE_FIXTURE;
    cfg.hash = SS_HASH_SHA256;

    if (ss_abi_sizeof_cfg() != sizeof cfg) { fprintf(stderr, "Error: libss_verify ABI mismatch\n"); return 2; }
    ss_ctx *ctx = NULL;
    uint32_t *status = (uint32_t *)malloc(n * sizeof *status);
    if (!status) { fprintf(stderr, "Error: out of memory\n"); free(wits); return 2; }
    ss_process_defaults();  /* before the first HIP call (ss_ctx_create below); one witness needs no extra queues, a batch caller does */
    int rc = ss_ctx_create(device, &ctx);
    if (rc == SS_OK) {
        /* the library's ONE host entry point: n witness FILES of text, read, uploaded and parsed by the library itself */
        ss_input_desc in;
        memset(&in, 0, sizeof in);
        in.family = s101 ? SS_FAMILY_STARK101 : SS_FAMILY_STWO;
        in.form = SS_FORM_TEXT;
        in.source = SS_SRC_FILES;
        in.text_fmt = SS_TEXT_WIT;
        in.cfg = s101 ? NULL : &cfg;
        in.n = n;
        in.items = (const void *const *)wits;
        rc = ss_verify_inputs(ctx, &in, status, NULL);
    }
    if (rc != SS_OK) {  /* no GPU, unsupported config: an error, never a verdict */
        fprintf(stderr, "Error: libss_verify: %s (code %d)\n", ss_last_error(), rc);
        ss_ctx_destroy(ctx);
        free(status);
        free(wits);
        return 2;
    }
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        if (status[i] == 0) { printf("%s: ACCEPT\n", wits[i]); continue; }
        bad++;
        if (status[i] == SS_STATUS_MALFORMED)
            fprintf(stderr, "Error: %s: malformed witness (not a value of the program's witness types)\n", wits[i]);
        else if (status[i] == SS_STATUS_CONFIG_MISMATCH)
            fprintf(stderr, "Error: Failed to run program: %s: the witness does not have the shape the program was compiled for\n", wits[i]);
        else
            fprintf(stderr, "Error: Failed to run program: %s: assertion failed (first failing assert 0x%08x)\n", wits[i], status[i]);
    }
    ss_ctx_destroy(ctx);
    free(status);
    free(wits);
    return bad ? 1 : 0;
}
