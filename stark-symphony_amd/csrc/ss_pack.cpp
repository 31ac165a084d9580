// Sizes and host packers of include/ss_verify.h: record -> batch permutations (no hashing, no arithmetic).
// Host-only code, kept out of the HIP translation units so that the CPU tests can build it with
// -fsanitize=address,undefined (tests/native/host_san.cpp): these functions index caller-owned buffers with
// offsets computed from caller-supplied configs.
#include "ss_pack.h"

#include <cstdlib>
#include <cstring>

#include "ss_pool.h"

using namespace ss;

bool ss::cfg_ok(const ss_stwo_cfg *c)
{
    return c && c->hash <= SS_HASH_BLAKE2S && c->flags <= (SS_FLAG_NO_DEDUP | SS_FLAG_TOP_CHECKS) &&
           stwo_cfg_ok(c->n_cols, c->trace_log, c->lde_log, c->n_queries, c->n_layers, c->mode);
}
// The layout of a batch and of its workspace is a function of (cfg, n) ONLY -- no process state: a caller may size
// buffers in one process and verify in another (the round-3 A/B environment knobs are gone: the byte
// compares' place is cfg.flags & SS_FLAG_TOP_CHECKS, the group size is the compile-time kTopMinGroups).
StwoLayout ss::lay_of(const ss_stwo_cfg *c, size_t n, bool minimal)
{
    return stwo_layout(c->n_cols, c->trace_log, c->lde_log, c->n_queries, c->n_layers, c->mode,
                       c->pow_target, n, !(c->flags & SS_FLAG_NO_DEDUP), c->hash == SS_HASH_BLAKE2S, kTopMinGroups,
                       !(c->flags & SS_FLAG_TOP_CHECKS), minimal);
}

extern "C" size_t ss_stwo_record_words(const ss_stwo_cfg *c)
{
    return cfg_ok(c) ? (size_t)stwo_record_words(c->n_cols, c->lde_log, c->n_queries, c->n_layers) : 0;
}
extern "C" size_t ss_stwo_batch_words(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n).total_words : 0;
}
extern "C" size_t ss_stwo_workspace_bytes(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n).ws_total_words * 4 : 0;
}

extern "C" int ss_stwo_pack(const ss_stwo_cfg *c, size_t n, const uint32_t *const *records,
                            uint32_t *out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !records || !out) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const StwoLayout y = lay_of(c, n);
    memset(out, 0, (size_t)y.total_words * 4);
    const uint32_t N = y.N, L = y.L, Q = y.Q, K = y.K;
    // sibling `lv` (from the leaf) of query q's path in tree `type` of proof p: the lowest len - top levels sit in
    // the 64-chain tiles, the top ones in top[proof][type][level][query][8]  (ss_layout.h)
    auto path_word = [&](uint32_t type, uint32_t len, size_t p, uint32_t q, uint32_t lv, uint32_t w) -> uint32_t & {
        const uint32_t top = y.T < len ? y.T : len, low = len - top;
        if (lv < low) {
            const uint64_t base = type == 0 ? y.off_trace_path : type == 1 ? y.off_cp_path : y.off_fri_path[type - 2];
            return out[tile_word(base, low, (uint64_t)p * Q + q, lv, w)];
        }
        return out[y.off_top + (uint64_t)p * y.top_words + y.top_off[type] + ((uint64_t)(lv - low) * Q + q) * 8 + w];
    };
    parallel_for(n, [&](size_t p) {
        const uint32_t *r = records[p];
        for (uint32_t w = 0; w < y.head_words; w++) out[y.off_head + (uint64_t)w * y.np + p] = r[w];
        r += y.head_words;
        for (uint32_t q = 0; q < Q; q++) {
            const uint64_t inst = (uint64_t)p * Q + q;
            for (uint32_t k = 0; k < N; k++) out[y.off_trace_vals + (uint64_t)k * y.nip + inst] = *r++;
            for (uint32_t k = 0; k < kCp; k++) out[y.off_cp_vals + (uint64_t)k * y.nip + inst] = *r++;
            for (uint32_t type = 0; type < 2; type++)
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t w = 0; w < 8; w++) path_word(type, L, p, q, l, w) = *r++;
        }
        for (uint32_t l = 0; l <= K; l++) {
            const uint32_t len = L - 1 - l;
            for (uint32_t q = 0; q < Q; q++) {
                const uint64_t inst = (uint64_t)p * Q + q;
                for (uint32_t w = 0; w < 4; w++)
                    out[y.off_fri_wit + ((uint64_t)l * 4 + w) * y.nip + inst] = *r++;
                for (uint32_t lv = 0; lv < len; lv++)
                    for (uint32_t w = 0; w < 8; w++) path_word(2 + l, len, p, q, lv, w) = *r++;
            }
        }
        for (uint32_t kind = 0; kind < K + 3; kind++)  // trailer: path lengths
            for (uint32_t q = 0; q < Q; q++) out[y.off_plen + (uint64_t)kind * y.nip + (uint64_t)p * Q + q] = *r++;
    });
    return SS_OK;
}

bool ss::shape_ok(const ss_s101_shape *sh) { return sh && s101_shape_ok(sh->max_layers, sh->max_path); }

extern "C" size_t ss_s101_record_words(const ss_s101_shape *sh)
{
    return shape_ok(sh) ? (size_t)s101_record_words(sh->max_layers, sh->max_path) : 0;
}
extern "C" size_t ss_s101_batch_words(const ss_s101_shape *sh, size_t n)
{
    return shape_ok(sh) && n ? (size_t)s101_layout(sh->max_layers, sh->max_path, n).total_words : 0;
}
extern "C" size_t ss_s101_workspace_bytes(const ss_s101_shape *sh, size_t n)
{
    return shape_ok(sh) && n ? (size_t)s101_layout(sh->max_layers, sh->max_path, n).ws_total_words * 4 : 0;
}

extern "C" int ss_s101_pack(const ss_s101_shape *sh, size_t n, const uint32_t *const *records,
                            uint32_t *out)
{
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n || !records || !out) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const S101Layout y = s101_layout(sh->max_layers, sh->max_path, n);
    memset(out, 0, (size_t)y.total_words * 4);
    const uint32_t ML = y.ML, PM = y.PM;
    parallel_for(n, [&](size_t p) {
        const uint32_t *r = records[p];
        auto head = [&](uint32_t w) -> uint32_t & { return out[y.off_head + (uint64_t)w * y.np + p]; };
        auto chain = [&](uint32_t type, const uint32_t *&rr) {
            out[y.off_leaf + (uint64_t)type * y.np + p] = *rr++;
            uint32_t len = *rr++;
            out[y.off_len + (uint64_t)type * y.np + p] = len;
            const uint64_t base = y.off_path + type * y.path_stride;
            for (uint32_t l = 0; l < PM; l++)
                for (uint32_t w = 0; w < 8; w++) out[tile_word(base, PM, p, l, w)] = *rr++;
        };
        for (uint32_t w = 0; w < 8; w++) head(y.h_root + w) = *r++;
        head(y.h_nlayers) = *r++;
        head(y.h_last) = *r++;
        for (uint32_t k = 0; k < 3; k++) chain(k, r);
        for (uint32_t i = 0; i < ML; i++) {
            for (uint32_t w = 0; w < 9; w++) head(y.h_layer + 9 * i + w) = *r++;  // root[8], beta
            chain(3 + 2 * i, r);
            chain(4 + 2 * i, r);
        }
    });
    return SS_OK;
}

