// Batch layouts shared by the host packers and the kernels (see include/ss_verify.h for the
// record formats and DESIGN.md "data layout in HBM").
//
// Everything is an array of little-endian u32 words in ONE buffer.
//   per-proof "head" words      head[w][proof]              (SoA, proofs padded to 64)
//   per-instance values         vals[column][instance]      (instance = proof * Q + query)
//   Merkle path lengths         plen[kind][instance]        (kind 0 trace, 1 cp, 2 + l FRI layer l)
//   Merkle paths, per chain type: 64-chain tiles
//          tile[g][level][half][lane][4 words],   g = instance / 64, lane = instance % 64
//     so a wavefront fetches one sibling level of its 64 chains as two contiguous 1 KiB reads.
//     With pair memoisation (T > 0) the tiles hold only the levels stwo_merkle_kernel hashes, the lowest
//     len - min(T, len); the top min(T, len) levels of every tree live in
//          top[proof][type][level][query][8 words]
//     -- the Q siblings of one (proof, tree, level) are one contiguous run of Q * 32 bytes, so the leader that
//     hashes a node and the followers whose bytes are compared with its sibling touch the same cache lines.
//     A batch is therefore laid out for the cfg (hash family and flags included) it will be verified with.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef SS_HD
#ifdef __HIPCC__
#define SS_HD __host__ __device__
#else
#define SS_HD
#endif
#endif

namespace ss {

constexpr uint32_t kMaxList = 31;
constexpr uint32_t kMaxQueries = 64;
constexpr uint32_t kCp = 16;  // NUM_CP_PARTITIONS, evals/composition_poly.simf:12

SS_HD inline uint64_t round_up64(uint64_t v) { return (v + 63) & ~(uint64_t)63; }

// word offset of word `w` (0..7) of the sibling at `level` of chain `inst` in a tiled section
SS_HD inline uint64_t tile_word(uint64_t base, uint32_t tile_len, uint64_t inst, uint32_t level,
                                uint32_t w)
{
    return base + (((inst >> 6) * tile_len + level) * 2 + (w >> 2)) * 256 + (inst & 63) * 4 + (w & 3);
}

// ------------------------------------------------------------------------------ stwo
struct StwoLayout {
    uint32_t N, TL, L, Q, K, mode;
    uint64_t pow_target;
    uint32_t n, np;    // proofs, padded to 64
    uint32_t ni, nip;  // instances (n * Q), padded to 64
    // head word indices
    uint32_t h_roots, h_oods_trace, h_oods_cp, h_fri_roots, h_last, h_nonce, head_words;
    // section word offsets inside the batch buffer
    uint64_t off_head, off_trace_vals, off_cp_vals, off_fri_wit, off_plen, off_trace_path, off_cp_path;
    uint64_t off_fri_path[kMaxList + 1];
    uint64_t off_top;                  // top[proof][type][level][query][8]
    uint32_t top_off[kMaxList + 3];    // word offset of a type inside a proof's part of `top`
    uint32_t top_words;                // words per proof in `top`
    uint64_t total_words;
    // workspace word offsets (u32 words)
    uint32_t c_queries, c_p, c_p2, c_b01, c_b02, c_a1, c_c1, c_a2, c_c2, c_m1, c_fold, ctx_words;
    uint32_t n_pow;  // DEEP alpha powers kept per proof
    uint64_t ws_ctx, ws_alpha, ws_leaf, ws_total_words;
    // Merkle pair memoisation (stwo_top_kernel): the top T levels of every tree are hashed once per
    // distinct (left, right) pair of a proof instead of once per query.  T = 0 switches it off.
    uint32_t T;           // levels below the root handled by the top kernel (per tree: min(T, len))
    uint32_t top_G;       // proofs per top-kernel group: top_G * Q <= kTopChains chains
    uint32_t top_blocks;  // persistent blocks of the top kernel (each owns a slice of ws_vals)
    uint64_t ws_top;      // top[type][inst][8]: node of every chain at depth min(T, len), native words
    uint64_t ws_vals;     // vals[block][parity][type][slot][8]: nodes of the distinct pairs, two depths
    uint64_t ws_counter;  // [0] next group of the top kernel, [1] trees it flagged (two words, zeroed before each launch)
    uint64_t ws_flag;     // flag[type][proof]: 1 = the tree's checks failed, stwo_top_cold_kernel re-hashes its chains
    // The byte compares that prove the sharing ("same", "edge", "cross at the edge": stwo_top_kernel) depend on proof
    // bytes only.  When a proof's Q chains sit in one wavefront of the merkle kernel (64 % Q == 0) they are made THERE,
    // lane against lane, from a per-query plan the query kernel leaves in ws_plan; the top kernel then only hashes.
    uint32_t mchk;
    uint64_t ws_plan;     // plan[inst] = 4 words: byte dd-1 of words 0..1 = the query (of its proof) that leads this
                          // chain's position at depth dd, of words 2..3 = the one that leads the sibling position (0xff: none)
    // Minimal records (ss_minimal.h): the batch is filled by stwo_min_expand_kernel from the verifier's OWN queries, the
    // siblings / fold-pair evaluations the record omits are holes that the kernels fill from the chain that computes
    // them.  Q is then the query count padded to a divisor of 64 (chains Qd .. Q-1 of a proof repeat query 0: whatever
    // they fail, chain 0 fails with a smaller code), Qd the count the transcript draws.
    uint32_t minimal, Qd;
    uint64_t ws_sib;      // sib[inst][32 bytes]: byte a = the query of the proof whose node at absolute level a is this
                          // chain's sibling (0xff: none, the sibling comes from the witness)
};

constexpr uint32_t kTopChains = 256;     // chains a top-kernel block plans at once (= its threads)
constexpr uint32_t kTopMaxT = 8;         // ceil_log2(kMaxQueries) + 2 >= T
#ifndef SS_TOP_MIN_GROUPS
#define SS_TOP_MIN_GROUPS 512  // A/B at 512 / 1 024 / 2 048 (profiles/r06_top_min_groups_ab.txt): 8 192 proofs verify 1 % faster in 512 groups of
                               // 16 proofs on 512 blocks (fuller plans; the free block slots go to the next pass's merkle kernel) than in 1 024 of 8
#endif
constexpr uint32_t kTopMinGroups = SS_TOP_MIN_GROUPS;  // a smaller batch is cut into smaller groups until it gives this many
// workspace slices of the persistent top kernel.  The launch uses min(groups, blocks resident at once), and at
// 256 threads and >= 123 VGPRs per lane at most 4 blocks fit a CU (3 for SHA-256): 1024 on the 256 CUs of an MI355X.
constexpr uint32_t kTopMaxBlocks = 1024;

SS_HD inline uint32_t ceil_log2(uint32_t v)
{
    uint32_t r = 0;
    while (r < 31 && (1u << r) < v) r++;
    return r;
}

SS_HD inline bool stwo_cfg_ok(uint32_t N, uint32_t TL, uint32_t L, uint32_t Q, uint32_t K, uint32_t mode)
{
    return N >= 1 && N <= 1024 && L >= 2 && L <= 31 && TL >= 1 && TL <= L && Q >= 1 &&
           Q <= kMaxQueries && K + 1 < L && K <= 30 && mode <= 1;
}

SS_HD inline StwoLayout stwo_layout(uint32_t N, uint32_t TL, uint32_t L, uint32_t Q, uint32_t K,
                                    uint32_t mode, uint64_t pow_target, uint64_t n, bool dedup = true,
                                    bool light_hash = false, uint32_t min_groups = kTopMinGroups,
                                    bool merkle_checks = true, bool minimal = false)
{
    StwoLayout y{};
    y.Qd = Q;
    y.minimal = minimal;
    if (minimal) {  // every proof gets a power-of-two number of chains (<= 64): its chains are lanes of one wavefront
        uint32_t qp = 1;
        while (qp < Q) qp <<= 1;
        Q = qp;
        merkle_checks = true;
    }
    y.N = N; y.TL = TL; y.L = L; y.Q = Q; y.K = K; y.mode = mode; y.pow_target = pow_target;
    y.n = (uint32_t)n;
    y.np = (uint32_t)round_up64(n);
    y.ni = (uint32_t)(n * Q);
    y.nip = (uint32_t)round_up64(n * Q);
    y.h_roots = 0;
    y.h_oods_trace = 24;
    y.h_oods_cp = y.h_oods_trace + 4 * N;
    y.h_fri_roots = y.h_oods_cp + 4 * kCp;
    y.h_last = y.h_fri_roots + 8 * (K + 1);
    y.h_nonce = y.h_last + 4;
    y.head_words = y.h_nonce + 2;
    uint64_t o = 0;
    y.off_head = o;        o += (uint64_t)y.head_words * y.np;
    y.off_trace_vals = o;  o += (uint64_t)N * y.nip;
    y.off_cp_vals = o;     o += (uint64_t)kCp * y.nip;
    y.off_fri_wit = o;     o += (uint64_t)(K + 1) * 4 * y.nip;
    y.off_plen = o;        o += (uint64_t)(K + 3) * y.nip;  // path length per chain: plen[kind][inst]
    // Q random leaves share their ancestors down to about log2(Q) levels below the root; two more
    // levels still merge ~1.5 pairs per tree, below that almost nothing.  With a hash half as long
    // (Blake2s: one compression per node) the second extra level no longer pays for its bookkeeping
    // (measured: configs[2] with Blake2s 4.71 M -> 4.82 M proofs/s at one level less).
#ifndef SS_TOP_EXTRA
#define SS_TOP_EXTRA 2
#endif
    const uint32_t want = ceil_log2(Q) + (light_hash ? SS_TOP_EXTRA - 1 : SS_TOP_EXTRA);
    y.T = (dedup && Q > 1) ? (want < L ? want : L) : 0;
    auto tile_len = [&](uint32_t len) { return len - (y.T < len ? y.T : len); };
    y.off_trace_path = o;  o += (uint64_t)tile_len(L) * 8 * y.nip;
    y.off_cp_path = o;     o += (uint64_t)tile_len(L) * 8 * y.nip;
    for (uint32_t l = 0; l <= K; l++) {
        y.off_fri_path[l] = o;
        o += (uint64_t)tile_len(L - 1 - l) * 8 * y.nip;
    }
    y.off_top = o;
    {
        uint32_t t = 0;
        for (uint32_t type = 0; type < K + 3; type++) {
            const uint32_t len = type < 2 ? L : L - 1 - (type - 2);
            y.top_off[type] = t;
            t += (y.T < len ? y.T : len) * Q * 8;
        }
        y.top_words = t;
    }
    o += (uint64_t)y.top_words * n;
    o = (o + 3) & ~(uint64_t)3;
    y.total_words = o;
    // per-proof context written by the transcript kernel
    uint32_t c = 0;
    y.c_queries = c; c += Q;
    y.c_p = c;       c += 8;            // OODS point P  (x.a..x.d, y.a..y.d)
    y.c_p2 = c;      c += 8;            // 2P
    y.c_b01 = c;     c += 4;            // DEEP b0 = (0, -2 im(P.y)) of the batch sampled at P
    y.c_b02 = c;     c += 4;            // ... at 2P
    y.c_a1 = c;      c += 4;
    y.c_c1 = c;      c += 4;
    y.c_a2 = c;      c += 4;
    y.c_c2 = c;      c += 4;
    y.c_m1 = c;      c += 4;
    y.c_fold = c;    c += 4 * (K + 1);
    y.ctx_words = c;
    uint64_t w = 0;
    y.ws_ctx = w;   w += (uint64_t)y.ctx_words * y.np;
    // alpha[proof][k][4] = deep_alpha^(k+1): one 16-byte load per column in the query kernel
    y.n_pow = N + kCp;
    y.ws_alpha = w; w += (uint64_t)y.n_pow * 4 * y.np;
    y.ws_leaf = w;  w += (uint64_t)(K + 1) * 8 * y.nip;
    y.top_G = kTopChains / Q;  // Q <= kMaxQueries = 64
    // a batch that gives fewer groups than kTopMinGroups (two per CU) is cut into smaller groups: half-empty plans cost less
    // than idle CUs -- but not less than blocks the overlapping next pass could have used, hence 512 and not the 768-1 024 resident slots
    while (y.top_G > 1 && (n + y.top_G - 1) / y.top_G < min_groups) y.top_G = (y.top_G + 1) / 2;
    const uint64_t groups = (n + y.top_G - 1) / y.top_G;
    y.top_blocks = y.T ? (uint32_t)(groups < kTopMaxBlocks ? groups : kTopMaxBlocks) : 0;
    y.ws_top = w;   w += y.T ? (uint64_t)(K + 3) * y.nip * 8 : 0;
    y.ws_vals = w;  w += (uint64_t)y.top_blocks * 2 * (K + 3) * kTopChains * 8;
    y.ws_counter = w; w += 4;
    y.ws_flag = w;  w += y.T ? (uint64_t)(K + 3) * y.np : 0;  // (directly behind the counter: one memset clears both)
    y.mchk = y.T && merkle_checks && 64 % Q == 0;
    y.ws_plan = w;  w += y.mchk ? (uint64_t)y.nip * 4 : 0;
    y.ws_sib = w;   w += minimal ? (uint64_t)y.nip * 8 : 0;
    y.ws_total_words = w;
    return y;
}

SS_HD inline uint64_t stwo_record_words(uint32_t N, uint32_t L, uint32_t Q, uint32_t K)
{
    uint64_t w = 24 + 4 * (uint64_t)N + 64 + 8 * (uint64_t)(K + 1) + 4 + 2;
    w += (uint64_t)Q * (N + kCp + 16 * (uint64_t)L);
    for (uint32_t l = 0; l <= K; l++) w += (uint64_t)Q * (4 + 8 * (uint64_t)(L - 1 - l));
    w += (uint64_t)(K + 3) * Q;  // trailer: path_len[kind][query], kind 0 trace, 1 cp, 2 + l FRI layer l
    return w;
}

// -------------------------------------------------------------------------- stark101
// chain types: 0..2 trace evals; 3 + 2*i cpa of layer i; 4 + 2*i cpb of layer i.
// Every type owns PM levels per tile (a chain reads only its own `len` levels, so the
// zero padding of shorter chains costs capacity, never bandwidth).
struct S101Layout {
    uint32_t ML, PM;  // max layers, max path length of the records
    uint32_t n, np;
    uint32_t n_types;
    uint32_t h_root, h_nlayers, h_last, h_layer, head_words;  // layer: root[8], beta
    uint64_t off_head, off_leaf, off_len, off_path;            // off_path + type * path_stride
    uint64_t path_stride;
    uint64_t total_words;
    // workspace: idx[np] (read by the merkle kernel), then the per-stage values of every proof, int[row][np]:
    // rows 0..2 the composition coefficients (air.simf:30-35), 3 x (verifier.simf:37), 4 cp (air.simf:94-101),
    // 5 .. 5+ML the value entering FRI layer i / the final one (fri.simf:74-91), then 8 rows: the channel state after
    // the commitments, before the query draw (verifier.simf:31-33)
    uint64_t ws_int;
    uint32_t int_rows;
    uint64_t ws_total_words;
};

SS_HD inline bool s101_shape_ok(uint32_t ML, uint32_t PM) { return ML <= kMaxList && PM <= kMaxList; }

SS_HD inline uint64_t s101_record_words(uint32_t ML, uint32_t PM)
{
    return 10 + 3 * (2 + 8 * (uint64_t)PM) + (uint64_t)ML * (9 + 2 * (2 + 8 * (uint64_t)PM));
}

SS_HD inline S101Layout s101_layout(uint32_t ML, uint32_t PM, uint64_t n)
{
    S101Layout y{};
    y.ML = ML; y.PM = PM;
    y.n = (uint32_t)n;
    y.np = (uint32_t)round_up64(n);
    y.n_types = 3 + 2 * ML;
    y.h_root = 0; y.h_nlayers = 8; y.h_last = 9; y.h_layer = 10;
    y.head_words = 10 + 9 * ML;
    uint64_t o = 0;
    y.off_head = o;  o += (uint64_t)y.head_words * y.np;
    y.off_leaf = o;  o += (uint64_t)y.n_types * y.np;  // leaf value (ev) per chain
    y.off_len = o;   o += (uint64_t)y.n_types * y.np;  // path length per chain
    y.off_path = o;
    y.path_stride = (uint64_t)PM * 8 * y.np;
    o += y.path_stride * y.n_types;
    y.total_words = o;
    y.ws_int = y.np;
    y.int_rows = 5 + (ML + 1) + 8;
    y.ws_total_words = (uint64_t)y.np * (1 + y.int_rows);
    return y;
}

}  // namespace ss
