/*
 * ss_oracle_shared.c -- CPU restatement of the shared-record expansion (TEST INFRASTRUCTURE ONLY, see ss_oracle.h).
 *
 * The reference has no deduplicated decommitment: stwo-verifier/src/fri/queries.simf:41 says "we do not sort and
 * remove duplicates", and its adapter stwo-verifier/scripts/generate_wit.py:36-42 cuts the prover's concatenated
 * witness lists into one full path per query.  A shared record (include/ss_verify.h) stores every distinct sibling
 * of a tree once; expanding it must give back exactly those per-query paths.  No reference bytes exist for the
 * shared form, so this file pins the product's closed-form rule (csrc/ss_shared.h) against the DEFINITION of the
 * order: walk query 0, 1, .. and each of its paths leaf -> root; a sibling position not met before in this tree
 * takes the next index of the tree's node list.
 */
#include <stdlib.h>
#include <string.h>

#include "ss_oracle.h"

static uint32_t tree_len(uint32_t L, uint32_t t) { return t < 2 ? L : L + 1 - t; }
/* FRI layer l (tree 2 + l) is indexed by query >> (l + 1): fri/layers.simf:29-40 halves the position per layer */
static uint32_t tree_shift(uint32_t t) { return t < 2 ? 0 : t - 1; }

/* plan[q * len + lvl] = index of query q's sibling at level lvl in the node list of tree t; returns the list length */
uint32_t so_shared_walk(uint32_t lde_log, uint32_t t, uint32_t n_queries, const uint32_t *queries, uint32_t *plan)
{
    const uint32_t len = tree_len(lde_log, t), shift = tree_shift(t);
    uint32_t *seen_lvl = malloc(sizeof(uint32_t) * (size_t)n_queries * len + 4);
    uint32_t *seen_pos = malloc(sizeof(uint32_t) * (size_t)n_queries * len + 4);
    uint32_t count = 0;
    for (uint32_t q = 0; q < n_queries; q++) {
        const uint32_t idx = queries[q] >> shift;
        for (uint32_t lvl = 0; lvl < len; lvl++) {
            const uint32_t pos = (idx >> lvl) ^ 1; /* the sibling's position: merkle.simf:22-33 pairs a node with it */
            uint32_t id = count;
            for (uint32_t j = 0; j < count; j++)
                if (seen_lvl[j] == lvl && seen_pos[j] == pos) { id = j; break; }
            if (id == count) { seen_lvl[count] = lvl; seen_pos[count] = pos; count++; }
            plan[(size_t)q * len + lvl] = id;
        }
    }
    free(seen_lvl);
    free(seen_pos);
    return count;
}

/* shared record -> per-query record (both as include/ss_verify.h lays them out).  0, or 2 = malformed (record zeroed) */
int so_shared_expand(uint32_t N, uint32_t L, uint32_t Q, uint32_t K, const uint32_t *sh, size_t words, uint32_t *rec)
{
    const uint32_t head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
    const uint32_t qstride = N + 16 + 16 * L;
    const uint32_t fbase = head + Q * qstride;
    uint32_t foff[32], o = 0;
    for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
    const uint32_t tbase = fbase + o, rec_words = tbase + (K + 3) * Q;
    const uint32_t s_vals = head, s_wit = s_vals + Q * (N + 16), s_qry = s_wit + 4 * Q * (K + 1), s_cnt = s_qry + Q;
    const uint32_t s_nodes = s_cnt + K + 3;
    memset(rec, 0, (size_t)rec_words * 4);
    if (words < s_nodes) return 2;
    for (uint32_t q = 0; q < Q; q++)
        if (sh[s_qry + q] >> L) return 2;
    uint32_t *plan = malloc(sizeof(uint32_t) * (size_t)Q * L + 4);
    size_t node0 = s_nodes;
    int bad = 0;
    for (uint32_t t = 0; t < K + 3 && !bad; t++) {
        const uint32_t len = tree_len(L, t);
        const uint32_t count = so_shared_walk(L, t, Q, sh + s_qry, plan);
        if (sh[s_cnt + t] != count || node0 + 8 * (size_t)count > words) { bad = 1; break; }
        for (uint32_t q = 0; q < Q; q++) {
            uint32_t *dst = t < 2 ? rec + head + q * qstride + N + 16 + t * 8 * L
                                  : rec + fbase + foff[t - 2] + q * (4 + 8 * len) + 4;
            for (uint32_t lvl = 0; lvl < len; lvl++) memcpy(dst + 8 * lvl, sh + node0 + 8 * (size_t)plan[(size_t)q * len + lvl], 32);
            rec[tbase + t * Q + q] = len;
        }
        node0 += 8 * (size_t)count;
    }
    free(plan);
    if (bad || node0 != words) { memset(rec, 0, (size_t)rec_words * 4); return 2; }
    memcpy(rec, sh, (size_t)head * 4);
    for (uint32_t q = 0; q < Q; q++) memcpy(rec + head + q * qstride, sh + s_vals + q * (N + 16), (size_t)(N + 16) * 4);
    for (uint32_t l = 0; l <= K; l++)
        for (uint32_t q = 0; q < Q; q++) memcpy(rec + fbase + foff[l] + q * (4 + 8 * (L - 1 - l)), sh + s_wit + (l * Q + q) * 4, 16);
    return 0;
}
