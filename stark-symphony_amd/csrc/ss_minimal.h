// Minimal records ("sorted multi-proof Merkle decommitment"): layout and the closed form of the
// order, shared by the host functions (ss_minimalrec.cpp) and the kernels (ss_minimal.hip).
//
// The reference presents one full authentication path per query and folds each on its own
// (stwo-verifier/src/fri/queries.simf:41 "we do not sort and remove duplicates", scripts/generate_wit.py:36-42,
// merkle.simf:22-44).  Upstream stwo -- the prover the reference's two proofs come from; not in /root/reference --
// sends one decommitment per TREE: queries sorted and deduplicated, the queried values once per distinct position, and
// only the siblings (and fold-pair evaluations) the verifier cannot compute from other queried nodes.  No bytes of
// that form exist in the reference: PARITY UNPINNED.  What pins it here is the correspondence with the per-query
// record: M verifies exactly as R(M), the record in which every omitted value is the one the walk computes
// (the test checker's so_stwo_minimal_expand).
//
// Structure.  Absolute level a = 0 .. L-1 counts from the leaves of the LDE-sized trees; every tree of a proof sees
// the queries through the same positions x(q, a) = query >> a (FRI layer l's fold pairs are the nodes of level l + 1).
//   Nodes(a) = the distinct x(q, a), ascending;      Lone(a) = the nodes of Nodes(a) whose sibling x ^ 1 is not in Nodes(a).
//   trace / composition tree:  values once per node of Nodes(0);  witness = siblings of Lone(0), Lone(1), .., Lone(L-1)
//   FRI layer l:               fri_witness = partners of Lone(l);  witness = siblings of Lone(l+1), .., Lone(L-1)
// Closed form per query (what host and device compute; Q <= 64):
//   rep(q, a)  = no e < q with x(e, a) == x(q, a)                       q represents its node
//   sib(q, a)  = the first e with x(e, a) == x(q, a) ^ 1, else none    a chain whose node is q's sibling
//   widx(q, a) = #{ e : rep(e, a), sib(e, a) == none, x(e, a) < x(q, a) }   rank of q's node inside Lone(a)
//   vidx(q)    = #{ e : rep(e, 0), x(e, 0) < x(q, 0) }                  rank of q's position inside Nodes(0)
//   cum(a)     = sum over a' < a of |Lone(a')|
// so query q's sibling at level lvl of tree t (shift_t = 0, 0, l + 1) is node sib(q, a)'s computed node when it exists
// and witness number cum(a) - cum(shift_t) + widx(q, a) of the tree otherwise, a = shift_t + lvl.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "ss_layout.h"

namespace ss {

// Word offsets inside a minimal record (include/ss_verify.h "minimal record").
struct MinMap {
    uint32_t N, L, Q, K;
    uint32_t head;  // per-proof head words (the per-query record's)
    uint32_t nv;    // n_vals[2]
    uint32_t nfw;   // n_fw[K + 1]
    uint32_t nhw;   // n_hw[K + 3]
    uint32_t data;  // first list word = number of fixed words
};

SS_HD inline uint32_t min_tree_len(uint32_t L, uint32_t t) { return t < 2 ? L : L + 1 - t; }
SS_HD inline uint32_t min_tree_shift(uint32_t t) { return t < 2 ? 0 : t - 1; }

SS_HD inline MinMap min_map(uint32_t N, uint32_t L, uint32_t Q, uint32_t K)
{
    MinMap m{};
    m.N = N; m.L = L; m.Q = Q; m.K = K;
    m.head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
    m.nv = m.head;
    m.nfw = m.nv + 2;
    m.nhw = m.nfw + K + 1;
    m.data = m.nhw + K + 3;
    return m;
}

SS_HD inline uint64_t min_max_words(const MinMap &m)
{
    uint64_t w = m.data + (uint64_t)m.Q * (m.N + kCp) + 4ull * m.Q * (m.K + 1);
    for (uint32_t t = 0; t < m.K + 3; t++) w += 8ull * m.Q * min_tree_len(m.L, t);
    return w;
}

// the smallest divisor of 64 that is >= Q: the kernels of the minimal path give every proof Qp chains (the last
// Qp - Q repeat query 0), so that a proof's chains are always lanes of one wavefront
SS_HD inline uint32_t min_pad_queries(uint32_t Q)
{
    uint32_t p = 1;
    while (p < Q) p <<= 1;
    return p;
}

constexpr uint8_t kMinNone = 0xff;

// The plan of one proof on the host.
struct MinPlan {
    uint32_t n_nodes0;                 // |Nodes(0)|
    uint8_t vidx[kMaxQueries];
    uint8_t sib[kMaxQueries][32];      // kMinNone: the sibling is not computed
    uint8_t widx[kMaxQueries][32];
    uint16_t cum[33];
};

// false: a position lies outside the LDE domain
inline bool min_plan(uint32_t L, uint32_t Q, const uint32_t *pos, MinPlan &p)
{
    for (uint32_t q = 0; q < Q; q++)
        if (pos[q] >> L) return false;
    p.cum[0] = 0;
    for (uint32_t a = 0; a < 32; a++) {
        bool rep[kMaxQueries], lone[kMaxQueries];
        uint32_t n_lone = 0, n_rep = 0;
        for (uint32_t q = 0; q < Q; q++) {
            const uint32_t x = a < L ? pos[q] >> a : 0;
            rep[q] = true;
            uint8_t s = kMinNone;
            for (uint32_t e = 0; e < Q; e++) {
                const uint32_t xe = a < L ? pos[e] >> a : 0;
                if (e < q && xe == x) rep[q] = false;
                if (s == kMinNone && xe == (x ^ 1)) s = (uint8_t)e;
            }
            p.sib[q][a] = a < L ? s : kMinNone;
            lone[q] = s == kMinNone;
            n_lone += rep[q] && lone[q];
            n_rep += rep[q];
        }
        for (uint32_t q = 0; q < Q; q++) {
            const uint32_t x = a < L ? pos[q] >> a : 0;
            uint32_t w = 0, v = 0;
            for (uint32_t e = 0; e < Q; e++) {
                const uint32_t xe = a < L ? pos[e] >> a : 0;
                w += rep[e] && lone[e] && xe < x;
                v += rep[e] && xe < x;
            }
            p.widx[q][a] = (uint8_t)w;
            if (a == 0) p.vidx[q] = (uint8_t)v;
        }
        if (a == 0) p.n_nodes0 = n_rep;
        p.cum[a + 1] = (uint16_t)(p.cum[a] + (a < L ? n_lone : 0));
    }
    return true;
}

// expected list lengths for these positions: counts[0..1] values (nodes), [2 .. 2+K] fri witnesses, [3+K .. 5+2K] hashes
inline void min_counts(const MinMap &m, const MinPlan &p, uint32_t *counts)
{
    counts[0] = counts[1] = p.n_nodes0;
    for (uint32_t l = 0; l <= m.K; l++) counts[2 + l] = (uint32_t)(p.cum[l + 1] - p.cum[l]);
    for (uint32_t t = 0; t < m.K + 3; t++) counts[3 + m.K + t] = (uint32_t)(p.cum[m.L] - p.cum[min_tree_shift(t)]);
}

}  // namespace ss
