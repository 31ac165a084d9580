/* Sanitizer driver of the CPU oracle (oracle/ss_oracle.c).  Built by tests/test_sanitizers.py with
 *   gcc -fsanitize=address,undefined -fno-sanitize-recover=all
 * Reads stwo records (layout of include/ss_verify.h, written by the test) from a file, verifies each in both
 * modes and prints the status words; the test compares them with the production build of the oracle.
 *
 *   oracle_san <n_cols> <trace_log> <lde_log> <n_queries> <n_layers> <hash> <records.bin>               */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ss_oracle.h"

static void be_bytes(const uint32_t *w, size_t nwords, uint8_t *out)
{
    for (size_t i = 0; i < nwords; i++) {
        out[4 * i] = (uint8_t)(w[i] >> 24); out[4 * i + 1] = (uint8_t)(w[i] >> 16);
        out[4 * i + 2] = (uint8_t)(w[i] >> 8); out[4 * i + 3] = (uint8_t)w[i];
    }
}

int main(int argc, char **argv)
{
    if (argc != 8) { fprintf(stderr, "usage: oracle_san N TL L Q K hash records.bin\n"); return 2; }
    so_stwo_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_cols = (uint32_t)atoi(argv[1]); cfg.trace_log = (uint32_t)atoi(argv[2]); cfg.lde_log = (uint32_t)atoi(argv[3]);
    cfg.n_queries = (uint32_t)atoi(argv[4]); cfg.n_layers = (uint32_t)atoi(argv[5]); cfg.hash = (uint32_t)atoi(argv[6]);
    cfg.pow_target = 0x07ffffffffffffffull;
    const uint32_t N = cfg.n_cols, L = cfg.lde_log, Q = cfg.n_queries, K = cfg.n_layers;
    size_t W = 24 + 4 * (size_t)N + 64 + 8 * (size_t)(K + 1) + 4 + 2 + (size_t)Q * (N + 16 + 16 * (size_t)L);
    for (uint32_t l = 0; l <= K; l++) W += (size_t)Q * (4 + 8 * (size_t)(L - 1 - l));
    W += (size_t)(K + 3) * Q;
    FILE *f = fopen(argv[7], "rb");
    if (!f) return 2;
    uint32_t *rec = malloc(W * 4);
    while (fread(rec, 4, W, f) == W) {
        const uint32_t *r = rec, *plen = rec + W - (size_t)(K + 3) * Q;
        so_stwo_proof p;
        memset(&p, 0, sizeof p);
        be_bytes(r, 24, &p.roots[0][0]); r += 24;
        so_qm31 *oods = malloc(N * sizeof *oods);
        for (uint32_t k = 0; k < N; k++, r += 4) oods[k] = (so_qm31){r[0], r[1], r[2], r[3]};
        p.oods_trace = oods;
        for (uint32_t k = 0; k < 16; k++, r += 4) p.oods_cp[k] = (so_qm31){r[0], r[1], r[2], r[3]};
        uint8_t *fri_roots = malloc(32 * (size_t)(K + 1));
        be_bytes(r, 8 * (size_t)(K + 1), fri_roots); r += 8 * (K + 1);
        p.fri_roots = fri_roots;
        p.last_layer = (so_qm31){r[0], r[1], r[2], r[3]}; r += 4;
        p.pow_nonce = ((uint64_t)r[0] << 32) | r[1]; r += 2;
        uint32_t *tv = malloc((size_t)Q * N * 4), *cv = malloc((size_t)Q * 16 * 4);
        so_path *tp = malloc(Q * sizeof *tp), *cp = malloc(Q * sizeof *cp);
        so_path *fp = malloc((size_t)(K + 1) * Q * sizeof *fp);
        so_qm31 *fw = malloc((size_t)(K + 1) * Q * sizeof *fw);
        /* every path gets its own exact-size heap block: reading past a (possibly shortened) path is an ASan report */
        uint8_t **blocks = malloc(((size_t)(K + 3) * Q) * sizeof *blocks);
        size_t nb = 0;
        for (uint32_t q = 0; q < Q; q++) {
            memcpy(tv + (size_t)q * N, r, N * 4); r += N;
            memcpy(cv + (size_t)q * 16, r, 64); r += 16;
            for (int t = 0; t < 2; t++) {
                uint32_t len = plen[t * Q + q];
                if (len > L) len = L;  /* the record's fixed slot keeps at most L siblings */
                uint8_t *b = malloc(len ? 32 * (size_t)len : 1);
                be_bytes(r, 8 * (size_t)len, b); r += 8 * L;
                blocks[nb++] = b;
                so_path *dst = t == 0 ? &tp[q] : &cp[q];
                dst->len = plen[t * Q + q] > L ? L : plen[t * Q + q];
                dst->nodes = b;
            }
        }
        for (uint32_t l = 0; l <= K; l++)
            for (uint32_t q = 0; q < Q; q++) {
                fw[(size_t)l * Q + q] = (so_qm31){r[0], r[1], r[2], r[3]}; r += 4;
                const uint32_t slot = L - 1 - l;
                uint32_t len = plen[(2 + l) * Q + q];
                if (len > slot) len = slot;
                uint8_t *b = malloc(len ? 32 * (size_t)len : 1);
                be_bytes(r, 8 * (size_t)len, b); r += 8 * slot;
                blocks[nb++] = b;
                fp[(size_t)l * Q + q].len = len;
                fp[(size_t)l * Q + q].nodes = b;
            }
        p.trace_vals = tv; p.cp_vals = cv; p.trace_paths = tp; p.cp_paths = cp; p.fri_witness = fw; p.fri_paths = fp;
        so_stwo_trace tr;
        const uint32_t s1 = so_stwo_verify(&cfg, &p, SO_MODE_FIXTURE, &tr);
        const uint32_t s0 = so_stwo_verify(&cfg, &p, SO_MODE_LITERAL, NULL);
        printf("%u %u\n", s1, s0);
        for (size_t i = 0; i < nb; i++) free(blocks[i]);
        free(blocks); free(oods); free(fri_roots); free(tv); free(cv); free(tp); free(cp); free(fp); free(fw);
    }
    free(rec);
    fclose(f);
    return 0;
}
