// Shared-path ("deduplicated decommitment") records: layout and the rule that undoes the sharing.
//
// The reference presents one full authentication path per query and hashes every one of them
// (stwo-verifier/src/fri/queries.simf:41: "we do not sort and remove duplicates";
// stwo-verifier/scripts/generate_wit.py:36-42 splits the prover's witness lists per query).  A shared record
// stores every DISTINCT sibling of a tree once, in the order a walk over query 0, 1, .. leaf -> root first needs
// it (formats.shared_path_order), and names the query positions.  Undoing the sharing is a pure gather -- no
// hashing -- after which the per-query record of include/ss_verify.h exists again and nothing downstream
// knows the difference.  The positions are an UNTRUSTED hint: the verifier draws its own queries and checks
// every expanded path in full, so a wrong hint can only make a proof fail.
//
// Closed form of the first-use order (what the kernel and the host functions compute; the walk itself is
// restated independently by the test checker and by formats.shared_path_order, and compared in the tests):
//   d(q, q')   = bit length of pos[q] ^ pos[q']  = the lowest level (from the leaf of the LDE-sized tree) at
//                which the two queries sit at the same position;
//   s(q)       = min over q' < q of d(q, q')  (32 for q = 0): below level s(q) no earlier query has been where q is;
//   tree t     has shift_t (0 for the trace and composition trees, l + 1 for FRI layer l: it is indexed by
//                pos >> (l + 1)) and len_t levels (L, L, L - 1 - l);
//   fresh_t(q) = clamp(s(q) - shift_t, 0, len_t) siblings of query q are new in tree t -- its levels 0 .. fresh_t(q) - 1;
//   base_t(q)  = sum of fresh_t(q') over q' < q;   count_t = base_t(Q);
//   lead(q, a) = the first query q' <= q with d(q, q') <= a  (a = absolute level shift_t + lvl);
//   node index of (t, q, lvl) = base_t(lead(q, shift_t + lvl)) + lvl.
// (The earliest query at a position is at levels < its own s there, so its sibling at that level is one of its
// fresh ones; all later queries at the position reuse it.)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "ss_layout.h"

namespace ss {

// Word offsets inside a shared record (include/ss_verify.h "shared record").
struct SharedMap {
    uint32_t N, L, Q, K;
    uint32_t head;   // words of the per-proof head (same words as the per-query record's)
    uint32_t vals;   // vals[q][N + 16]
    uint32_t wit;    // wit[l][q][4]
    uint32_t qry;    // queries[Q]
    uint32_t cnt;    // count[K + 3]
    uint32_t nodes;  // first node word = number of fixed words
    uint32_t max_nodes;  // Q * sum of len_t
};

SS_HD inline uint32_t shared_tree_len(uint32_t L, uint32_t t) { return t < 2 ? L : L + 1 - t; }
SS_HD inline uint32_t shared_tree_shift(uint32_t t) { return t < 2 ? 0 : t - 1; }

SS_HD inline SharedMap shared_map(uint32_t N, uint32_t L, uint32_t Q, uint32_t K)
{
    SharedMap m{};
    m.N = N; m.L = L; m.Q = Q; m.K = K;
    m.head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
    m.vals = m.head;
    m.wit = m.vals + Q * (N + kCp);
    m.qry = m.wit + 4 * Q * (K + 1);
    m.cnt = m.qry + Q;
    m.nodes = m.cnt + K + 3;
    uint32_t s = 0;
    for (uint32_t t = 0; t < K + 3; t++) s += shared_tree_len(L, t);
    m.max_nodes = Q * s;
    return m;
}

SS_HD inline uint32_t shared_bitlen(uint32_t v)
{
    uint32_t n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

SS_HD inline uint32_t shared_fresh(uint32_t s, uint32_t L, uint32_t t)
{
    const uint32_t sh = shared_tree_shift(t), len = shared_tree_len(L, t);
    const uint32_t f = s > sh ? s - sh : 0;
    return f < len ? f : len;
}

// The plan of one proof on the host (Q <= kMaxQueries): s(q), lead(q, a) and base_t(q) of the closed form above.
struct SharedPlan {
    uint32_t s[kMaxQueries];
    uint8_t lead[kMaxQueries][32];
    uint32_t base[kMaxList + 3][kMaxQueries + 1];  // base[t][Q] = count_t
};

// false: a position lies outside the LDE domain
inline bool shared_plan(const SharedMap &m, const uint32_t *pos, SharedPlan &p)
{
    for (uint32_t q = 0; q < m.Q; q++) {
        if (pos[q] >> m.L) return false;
        uint32_t s = 32;
        for (uint32_t a = 0; a < 32; a++) p.lead[q][a] = (uint8_t)q;
        for (uint32_t e = q; e-- > 0;) {  // descending, so that the earliest query at a position wins
            const uint32_t d = shared_bitlen(pos[q] ^ pos[e]);
            if (d < s) s = d;
            for (uint32_t a = d; a < 32; a++) p.lead[q][a] = (uint8_t)e;
        }
        p.s[q] = s;
    }
    for (uint32_t t = 0; t < m.K + 3; t++) {
        uint32_t b = 0;
        for (uint32_t q = 0; q < m.Q; q++) { p.base[t][q] = b; b += shared_fresh(p.s[q], m.L, t); }
        p.base[t][m.Q] = b;
    }
    return true;
}

// shared record (compact, or "capacity" form: tree t's nodes at nodes + 8 Q sum_{t' < t} len_t') -> per-query record
void shared_expand_host(const SharedMap &m, const SharedPlan &p, const uint32_t *shared, bool capacity, uint32_t *record);

}  // namespace ss
