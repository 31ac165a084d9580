// Shared-path records on the host (include/ss_verify.h, "shared record"): sizes, the writer a prover-side
// caller uses (per-query record -> shared record) and the expansion (shared -> per-query), which is the host
// statement of the rule the GPU applies in csrc/ss_shared.hip.  Host-only, no hashing, no arithmetic; built
// with the sanitizers by tests/native/host_san.cpp.
#include "ss_shared.h"

#include <cstring>

#include "ss_pack.h"

using namespace ss;

namespace ss {

// per-query record offsets of the pieces a shared record re-arranges (the layout of ss_stwo_record_words)
struct PerQueryMap {
    uint32_t head, qstride, fbase, tbase, words;
    uint32_t foff[kMaxList + 1];
    PerQueryMap(uint32_t N, uint32_t L, uint32_t Q, uint32_t K)
    {
        head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
        qstride = N + kCp + 16 * L;
        fbase = head + Q * qstride;
        uint32_t o = 0;
        for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
        tbase = fbase + o;
        words = tbase + (K + 3) * Q;
    }
    // word offset of sibling `lvl` of query q's path in tree t
    uint32_t path(uint32_t L, uint32_t N, uint32_t t, uint32_t q, uint32_t lvl) const
    {
        if (t < 2) return head + q * qstride + N + kCp + t * 8 * L + 8 * lvl;
        const uint32_t l = t - 2;
        return fbase + foff[l] + q * (4 + 8 * (L - 1 - l)) + 4 + 8 * lvl;
    }
};

}  // namespace ss

extern "C" size_t ss_stwo_shared_fixed_words(const ss_stwo_cfg *c)
{
    return cfg_ok(c) ? (size_t)shared_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers).nodes : 0;
}

extern "C" size_t ss_stwo_shared_max_words(const ss_stwo_cfg *c)
{
    if (!cfg_ok(c)) return 0;
    const SharedMap m = shared_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    return (size_t)m.nodes + 8 * (size_t)m.max_nodes;
}

extern "C" int ss_stwo_shared_counts(const ss_stwo_cfg *c, const uint32_t *queries, uint32_t *counts)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!queries || !counts) return set_err(SS_ERR_ARG, "null argument");
    const SharedMap m = shared_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    SharedPlan p;
    if (!shared_plan(m, queries, p)) return set_err(SS_ERR_ARG, "query position outside the LDE domain");
    for (uint32_t t = 0; t < m.K + 3; t++) counts[t] = p.base[t][m.Q];
    return SS_OK;
}

extern "C" int ss_stwo_share_record(const ss_stwo_cfg *c, const uint32_t *rec, const uint32_t *queries, uint32_t *out,
                                    size_t cap_words, size_t *words_out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!rec || !queries || !out || !words_out) return set_err(SS_ERR_ARG, "null argument");
    const uint32_t N = c->n_cols, L = c->lde_log, Q = c->n_queries, K = c->n_layers;
    const SharedMap m = shared_map(N, L, Q, K);
    const PerQueryMap r(N, L, Q, K);
    SharedPlan p;
    if (!shared_plan(m, queries, p)) return set_err(SS_ERR_ARG, "query position outside the LDE domain");  // (a caller error, as in ss_stwo_shared_counts: not "no shared form")
    size_t total = m.nodes;
    for (uint32_t t = 0; t < K + 3; t++) total += 8 * (size_t)p.base[t][Q];
    *words_out = total;
    if (cap_words < total) return set_err(SS_ERR_ARG, "shared record needs %zu words, %zu given", total, cap_words);
    for (uint32_t t = 0; t < K + 3; t++)  // only full-length paths have a shared form
        for (uint32_t q = 0; q < Q; q++)
            if (rec[r.tbase + t * Q + q] != shared_tree_len(L, t)) return 1;
    memcpy(out, rec, (size_t)m.head * 4);
    for (uint32_t q = 0; q < Q; q++) memcpy(out + m.vals + q * (N + kCp), rec + r.head + q * r.qstride, (size_t)(N + kCp) * 4);
    for (uint32_t l = 0; l <= K; l++)
        for (uint32_t q = 0; q < Q; q++)
            memcpy(out + m.wit + (l * Q + q) * 4, rec + r.fbase + r.foff[l] + q * (4 + 8 * (L - 1 - l)), 16);
    memcpy(out + m.qry, queries, (size_t)Q * 4);
    uint32_t *node = out + m.nodes;
    for (uint32_t t = 0; t < K + 3; t++) {
        const uint32_t len = shared_tree_len(L, t), sh = shared_tree_shift(t);
        out[m.cnt + t] = p.base[t][Q];
        for (uint32_t q = 0; q < Q; q++) {
            const uint32_t fresh = shared_fresh(p.s[q], L, t);
            for (uint32_t lvl = 0; lvl < len; lvl++) {
                const uint32_t *src = rec + r.path(L, N, t, q, lvl);
                if (lvl < fresh) memcpy(node + 8 * (size_t)(p.base[t][q] + lvl), src, 32);
                else if (memcmp(node + 8 * (size_t)(p.base[t][p.lead[q][sh + lvl]] + lvl), src, 32) != 0)
                    return 1;  // two queries present different bytes for one node
            }
        }
        node += 8 * (size_t)p.base[t][Q];
    }
    return 0;
}

// shared record (compact, or capacity form: tree t's nodes at a fixed base with room for Q * len_t) -> per-query record
void ss::shared_expand_host(const SharedMap &m, const SharedPlan &p, const uint32_t *sh_rec, bool capacity, uint32_t *rec)
{
    const uint32_t N = m.N, L = m.L, Q = m.Q, K = m.K;
    const PerQueryMap r(N, L, Q, K);
    memset(rec, 0, (size_t)r.words * 4);
    memcpy(rec, sh_rec, (size_t)m.head * 4);
    for (uint32_t q = 0; q < Q; q++) memcpy(rec + r.head + q * r.qstride, sh_rec + m.vals + q * (N + kCp), (size_t)(N + kCp) * 4);
    for (uint32_t l = 0; l <= K; l++)
        for (uint32_t q = 0; q < Q; q++)
            memcpy(rec + r.fbase + r.foff[l] + q * (4 + 8 * (L - 1 - l)), sh_rec + m.wit + (l * Q + q) * 4, 16);
    const uint32_t *node = sh_rec + m.nodes;
    for (uint32_t t = 0; t < K + 3; t++) {
        const uint32_t len = shared_tree_len(L, t), sh = shared_tree_shift(t);
        for (uint32_t q = 0; q < Q; q++) {
            for (uint32_t lvl = 0; lvl < len; lvl++)
                memcpy(rec + r.path(L, N, t, q, lvl), node + 8 * (size_t)(p.base[t][p.lead[q][sh + lvl]] + lvl), 32);
            rec[r.tbase + t * Q + q] = len;
        }
        node += 8 * (size_t)(capacity ? Q * len : p.base[t][Q]);
    }
}

extern "C" int ss_stwo_unshare_record(const ss_stwo_cfg *c, const uint32_t *sh_rec, size_t words, uint32_t *rec)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!sh_rec || !rec) return set_err(SS_ERR_ARG, "null argument");
    const uint32_t N = c->n_cols, L = c->lde_log, Q = c->n_queries, K = c->n_layers;
    const SharedMap m = shared_map(N, L, Q, K);
    memset(rec, 0, ss_stwo_record_words(c) * 4);
    if (words < m.nodes) return (int)SS_STATUS_MALFORMED;
    SharedPlan p;
    if (!shared_plan(m, sh_rec + m.qry, p)) return (int)SS_STATUS_MALFORMED;
    size_t total = m.nodes;
    for (uint32_t t = 0; t < K + 3; t++) {
        if (sh_rec[m.cnt + t] != p.base[t][Q]) return (int)SS_STATUS_MALFORMED;
        total += 8 * (size_t)p.base[t][Q];
    }
    if (words != total) return (int)SS_STATUS_MALFORMED;
    shared_expand_host(m, p, sh_rec, false, rec);
    return 0;
}
