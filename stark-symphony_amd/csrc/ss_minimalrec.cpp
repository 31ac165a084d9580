// Minimal records on the host (include/ss_verify.h, "minimal record"): sizes, the expected list lengths for a set of
// positions, and the writer a prover-side caller uses (per-query record -> minimal record: a selection, no hashing).
// The way back needs hashing -- it IS the verification -- and exists only on the GPU (ss_minimal.hip) and, as the
// checker's definition, in the tests' CPU checker.  Host-only; built with the sanitizers by tests/native/host_san.cpp.
#include "ss_minimal.h"

#include <cstring>

#include "ss_pack.h"

using namespace ss;

extern "C" size_t ss_stwo_minimal_fixed_words(const ss_stwo_cfg *c)
{
    return cfg_ok(c) ? (size_t)min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers).data : 0;
}

extern "C" size_t ss_stwo_minimal_max_words(const ss_stwo_cfg *c)
{
    return cfg_ok(c) ? (size_t)min_max_words(min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers)) : 0;
}

extern "C" int ss_stwo_minimal_counts(const ss_stwo_cfg *c, const uint32_t *queries, uint32_t *counts)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!queries || !counts) return set_err(SS_ERR_ARG, "null argument");
    const MinMap m = min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    MinPlan p;
    if (!min_plan(m.L, m.Q, queries, p)) return set_err(SS_ERR_ARG, "query position outside the LDE domain");
    min_counts(m, p, counts);
    return SS_OK;
}

extern "C" int ss_stwo_minimise_record(const ss_stwo_cfg *c, const uint32_t *rec, const uint32_t *queries, uint32_t *out,
                                       size_t cap_words, size_t *words_out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!rec || !queries || !out || !words_out) return set_err(SS_ERR_ARG, "null argument");
    const uint32_t N = c->n_cols, L = c->lde_log, Q = c->n_queries, K = c->n_layers;
    const MinMap m = min_map(N, L, Q, K);
    MinPlan p;
    if (!min_plan(L, Q, queries, p)) return set_err(SS_ERR_ARG, "query position outside the LDE domain");
    uint32_t counts[2 * kMaxList + 8];
    min_counts(m, p, counts);
    size_t total = m.data + (size_t)counts[0] * (N + kCp);
    for (uint32_t l = 0; l <= K; l++) total += 4 * (size_t)counts[2 + l];
    for (uint32_t t = 0; t < K + 3; t++) total += 8 * (size_t)counts[3 + K + t];
    *words_out = total;
    if (cap_words < total) return set_err(SS_ERR_ARG, "minimal record needs %zu words, %zu given", total, cap_words);
    // per-query record offsets (the layout of ss_stwo_record_words)
    const uint32_t head = m.head, qstride = N + kCp + 16 * L, fbase = head + Q * qstride;
    uint32_t foff[kMaxList + 1], o = 0;
    for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
    const uint32_t tbase = fbase + o;
    for (uint32_t t = 0; t < K + 3; t++)  // only full-length paths have a minimal form
        for (uint32_t q = 0; q < Q; q++)
            if (rec[tbase + t * Q + q] != min_tree_len(L, t)) return 1;
    memcpy(out, rec, (size_t)head * 4);
    out[m.nv] = out[m.nv + 1] = counts[0];
    for (uint32_t l = 0; l <= K; l++) out[m.nfw + l] = counts[2 + l];
    for (uint32_t t = 0; t < K + 3; t++) out[m.nhw + t] = counts[3 + K + t];
    // A slot is written by the first query that owns it; every later owner must present the same words.
    uint8_t seen[kMaxQueries * 32];
    auto put = [&](uint32_t *dst, const uint32_t *src, uint32_t words, uint8_t &mark) {
        if (!mark) { memcpy(dst, src, (size_t)words * 4); mark = 1; return true; }
        return memcmp(dst, src, (size_t)words * 4) == 0;
    };
    uint32_t *tv = out + m.data, *cv = tv + (size_t)counts[0] * N;
    memset(seen, 0, sizeof seen);
    for (uint32_t q = 0; q < Q; q++) {
        const uint32_t *src = rec + head + q * qstride;
        uint8_t twice = seen[p.vidx[q]];
        if (!put(tv + (size_t)p.vidx[q] * N, src, N, seen[p.vidx[q]])) return 1;
        if (!put(cv + (size_t)p.vidx[q] * kCp, src + N, kCp, twice)) return 1;
    }
    uint32_t *fw = cv + (size_t)counts[0] * kCp;
    for (uint32_t l = 0; l <= K; l++) {
        memset(seen, 0, sizeof seen);
        for (uint32_t q = 0; q < Q; q++)
            if (p.sib[q][l] == kMinNone &&
                !put(fw + 4 * (size_t)p.widx[q][l], rec + fbase + foff[l] + q * (4 + 8 * (L - 1 - l)), 4, seen[p.widx[q][l]]))
                return 1;
        fw += 4 * (size_t)counts[2 + l];
    }
    uint32_t *hw = fw;
    for (uint32_t t = 0; t < K + 3; t++) {
        const uint32_t len = min_tree_len(L, t), sh = min_tree_shift(t);
        memset(seen, 0, sizeof seen);
        for (uint32_t q = 0; q < Q; q++)
            for (uint32_t lvl = 0; lvl < len; lvl++) {
                const uint32_t a = sh + lvl;
                if (p.sib[q][a] != kMinNone) continue;  // the verifier computes that sibling: not sent, not looked at
                const uint32_t idx = (uint32_t)(p.cum[a] - p.cum[sh]) + p.widx[q][a];
                const uint32_t *src = t < 2 ? rec + head + q * qstride + N + kCp + t * 8 * L + 8 * lvl
                                            : rec + fbase + foff[t - 2] + q * (4 + 8 * len) + 4 + 8 * lvl;
                if (!put(hw + 8 * (size_t)idx, src, 8, seen[idx])) return 1;
            }
        hw += 8 * (size_t)counts[3 + K + t];
    }
    return 0;
}
