// stwo circle-STARK batch verifier kernels for gfx950.
//
// Reference path: `verify_proof`, stwo-verifier/src/verifier.simf:32-58 and everything it
// calls.  The seven stages of docs/verifier_flow.md:3-36 are regrouped by their parallelism:
//
//   stwo_transcript_kernel   one lane per PROOF.   Stages I-IV + query generation: the
//        Fiat-Shamir chain is ~(3K+25) dependent hash compressions, so lanes (not
//        wavefronts) are the unit; it also hoists everything of stage VI that does not
//        depend on the query (DEEP line coefficients, fri/answers.simf:44-64).
//   stwo_query_kernel        one lane per QUERY.   Stage VI (DEEP quotient at the query
//        point) and the fold chain of stage VII (fri/folding.simf:15-41); emits the leaf
//        pair of every FRI layer for the Merkle kernel.
//   stwo_merkle_kernel       one lane per Merkle CHAIN, one wavefront per 64 chains of the
//        same kind: the trace / composition decommitments of stage V (evals/verify.simf:47-69)
//        and the per-layer FRI decommitments (fri/layers.simf:40-48).  >90 % of the work.
//   stwo_finalize_kernel     one lane per proof: first-failure code -> status, accept count.
//
// The transcript and Merkle kernels exist once per hash family (ss_hash.h): *_sha is the
// reference's SHA-256, *_b2s the Blake2s variant (same byte strings, parity unpinned).
//
// A failed assert never stops a lane: each check contributes its code through atomicMin, and
// because the codes are ordered like the reference's evaluation order the minimum IS the
// first failing assert (every check is a pure function of the proof).
#include <hip/hip_runtime.h>

#include "ss_channel.h"
#include "ss_fields.h"
#include "ss_hash.h"
#include "ss_layout.h"
#include "ss_stwo_checks.h"

namespace ss {

// ========================================================================== transcript
template <int HF>
__device__ __forceinline__ void stwo_transcript_body(const StwoLayout &lay, const uint32_t *__restrict__ batch,
                                                     uint32_t *__restrict__ ws, uint32_t *__restrict__ status,
                                                     uint32_t *__restrict__ accept_count, uint32_t reset)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= lay.n) return;
    const uint32_t *head = batch + lay.off_head;
    const uint32_t np = lay.np;
    auto H = [&](uint32_t w) { return head[(size_t)w * np + p]; };
    auto HQ = [&](uint32_t w) { return QM31{H(w), H(w + 1), H(w + 2), H(w + 3)}; };
    uint32_t *ctx = ws + lay.ws_ctx;
    auto CW = [&](uint32_t w, uint32_t v) { ctx[(size_t)w * np + p] = v; };
    auto CQ = [&](uint32_t w, QM31 q) { CW(w, q.a); CW(w + 1, q.b); CW(w + 2, q.c); CW(w + 3, q.d); };
    uint32_t fail = 0xffffffffu;
    auto FAIL = [&](uint32_t code) { fail = code < fail ? code : fail; };
    // The Fiat-Shamir chain (channel.simf:31-172) is ~3K + 30 hashes, each depending on the one before.  Every one of
    // them is H(digest || a few value words) -- channel_mix_u256 / _u64 / _line_poly / _oods_evals hash the digest
    // followed by their argument, channel_draw_words hashes it followed by the counter -- so the whole transcript is a
    // state machine around ONE inlined compression: a phase names the value words, the block loop hashes them, the
    // phase's tail consumes the digest.  (Round 3 called an out-of-line compression from ~12 sites: 184 VGPRs and an
    // 80-byte call frame in scratch per lane; one site costs neither.)
    enum : uint32_t { kRoot0, kRoot1, kDrawCp, kRoot2, kDrawT, kOods, kDrawDeep, kFriRoot, kFriDraw, kLast, kNonce, kQueries, kEnd };
    uint32_t dig[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // channel_init, channel.simf:31 (native words of the hash family)
    uint32_t ctr = 0, phase = kRoot0, fri_l = 0, tries = 0;  // (fri_l: the FRI layer, afterwards the first query of a draw)
    // cp_alpha, deep_alpha and the OODS point live through dozens of compressions without being touched: parked in the
    // lane's own LDS column instead of 16 VGPRs
    __shared__ uint32_t s_park[16][64];
    auto park = [&](uint32_t k, const QM31 &v) {
        s_park[k][threadIdx.x] = v.a; s_park[k + 1][threadIdx.x] = v.b; s_park[k + 2][threadIdx.x] = v.c; s_park[k + 3][threadIdx.x] = v.d;
    };
    auto parked = [&](uint32_t k) {
        return QM31{s_park[k][threadIdx.x], s_park[k + 1][threadIdx.x], s_park[k + 2][threadIdx.x], s_park[k + 3][threadIdx.x]};
    };
    enum : uint32_t { kParkCpAlpha = 0, kParkDeepAlpha = 4, kParkPx = 8, kParkPy = 12 };
    const uint32_t qmask = shl32(lay.L, 1) - 1;
    while (phase != kEnd) {
        // ---- the value words of this hash (short messages in registers, the OODS values straight from the batch)
        uint32_t v8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nvals = 1;
        switch (phase) {
        case kRoot0: case kRoot1: case kRoot2: case kFriRoot: {  // channel_mix_u256 (channel.simf:154-161)
            const uint32_t o = phase == kFriRoot ? lay.h_fri_roots + 8 * fri_l
                                                 : lay.h_roots + (phase == kRoot0 ? 0 : phase == kRoot1 ? 8 : 16);
#pragma unroll
            for (int i = 0; i < 8; i++) v8[i] = H(o + i);
            nvals = 8;
            break;
        }
        case kOods: nvals = 4 * (lay.N + kCp); break;  // channel_mix_oods_evals (deep/oods.simf:23-39): trace and cp values are adjacent
        case kLast:                                    // channel_mix_line_poly, fri/commit.simf:48-57
#pragma unroll
            for (int i = 0; i < 4; i++) v8[i] = H(lay.h_last + i);
            nvals = 4;
            break;
        case kNonce: v8[0] = H(lay.h_nonce); v8[1] = H(lay.h_nonce + 1); nvals = 2; break;  // channel_mix_u64
        default: v8[0] = ctr; break;                   // channel_draw_words (channel.simf:36-65): digest || be4(counter)
        }
        const bool from_batch = phase == kOods;
        const uint32_t total = 8 + nvals, nblk = Hasher<HF>::n_blocks(total);
        Dig st;
        Hasher<HF>::iv(st.v);
        for (uint32_t blk = 0; blk < nblk; blk++) {
            W16 w;
            Hasher<HF>::fill(w, blk, nblk, total, dig, [&](int j, uint32_t i) { return from_batch ? H(lay.h_oods_trace + i) : v8[j & 7]; });
            Hasher<HF>::compress(st, w, blk, nblk, total);  // <- the kernel's only compression
        }
        // ---- what the digest is for
        const bool is_draw = phase == kDrawCp || phase == kDrawT || phase == kDrawDeep || phase == kFriDraw;
        if (is_draw || phase == kQueries) ctr = ctr + 1;
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) dig[i] = st.v[i];
            ctr = 0;
        }
        QM31 drawn = qm31_zero();
        if (is_draw) {
            // channel_draw_qm31 (channel.simf:115-140): retry while any of the first four words >= 2^32 - 2; the
            // for_while counter is a u8, so at most 256 attempts
            const uint32_t w0 = Hasher<HF>::native(st.v[0]), w1 = Hasher<HF>::native(st.v[1]);
            const uint32_t w2 = Hasher<HF>::native(st.v[2]), w3 = Hasher<HF>::native(st.v[3]);
            const bool ok = w0 < 4294967294u && w1 < 4294967294u && w2 < 4294967294u && w3 < 4294967294u;
            if (!ok && ++tries < 256) continue;
            if (ok) drawn = {m31_red(w0), m31_red(w1), m31_red(w2), m31_red(w3)};
            else  // sub = the ordinal of the draw: cp_alpha, t, deep_alpha, then one per FRI layer
                FAIL(stwo_code(1, 0, 0, phase == kDrawCp ? 0 : phase == kDrawT ? 1 : phase == kDrawDeep ? 2 : 3 + fri_l));
            tries = 0;
        }
        switch (phase) {
        // ---- stage I: evals_commit (evals/commit.simf:20-35)
        case kRoot0: phase = kRoot1; break;
        case kRoot1: phase = kDrawCp; break;
        case kDrawCp: park(kParkCpAlpha, drawn); phase = kRoot2; break;
        case kRoot2: phase = kDrawT; break;
        // ---- stage II: oods (deep/oods.simf:44-64)
        case kDrawT: {  // channel_draw_qm31_point (channel.simf:143-151)
            const QM31 t = drawn;
            QM31 inv;
            QM31 t_sq = qm31_mul(t, t);
            if (!qm31_inv(qm31_add(qm31_one(), t_sq), inv)) FAIL(stwo_code(2, 0, 0, 1));
            park(kParkPx, qm31_mul(qm31_sub(qm31_one(), t_sq), inv));
            park(kParkPy, qm31_mul(qm31_add(t, t), inv));
            phase = kOods;
            break;
        }
        case kOods: {
            // eval_composition_poly (constraints/wide_fibonacci.simf:24-62).  A column value reaches
            // the squares through multiplications only, so it is reduced first and squared once
            // (the reference squares it as `b` and again as `a`); it reaches the subtraction raw.
            const QM31 cp_alpha = parked(kParkCpAlpha);
            const QM31Point P{parked(kParkPx), parked(kParkPy)};
            QM31 acc = qm31_zero(), sq_a = qm31_zero(), sq_b = qm31_zero();
            uint32_t skip = 0;
            for (uint32_t k = 0; k < lay.N; k++) {
                const QM31 c = HQ(lay.h_oods_trace + 4 * k);
                if (skip == 2) {
                    const QM31 constraint = qm31_sub(c, qm31_add(sq_b, sq_a));
                    acc = qm31_add(qm31_mul_c(acc, cp_alpha), constraint);
                } else {
                    skip++;
                }
                sq_a = sq_b;
                sq_b = qm31_sqr_c(qm31_red(c));
            }
            // vanishing_poly_eval (evals/composition_poly.simf:27-35,66-71): u8 loop counter
            QM31 van = P.x;
            {
                const uint32_t n_iter = (lay.TL - 1) & 0xff;
                for (uint32_t counter = 0; counter < 256; counter++) {
                    if (counter == n_iter) break;
                    van = qm31_dbl_x(van);
                }
            }
            QM31 van_inv;
            if (!qm31_inv(van, van_inv)) FAIL(stwo_code(2, 0, 0, 2));
            QM31 cp_eval = qm31_mul(acc, van_inv);
            // composition_poly_eval_from_decomposed (evals/composition_poly.simf:38-59)
            // (one part at a time, the loop kept rolled: sixteen QM31 values in flight at once cost 130 VGPRs)
            QM31 sampled = qm31_zero();
#pragma unroll 1
            for (int j = 0; j < 4; j++) {  // j = a, b, c, d ; index = 4 * coord + j
                QM31 c0 = HQ(lay.h_oods_cp + 4 * (0 + j)), c1 = HQ(lay.h_oods_cp + 4 * (4 + j));
                QM31 c2 = HQ(lay.h_oods_cp + 4 * (8 + j)), c3 = HQ(lay.h_oods_cp + 4 * (12 + j));
                QM31 r = qm31_add(c0, qm31_mul(c1, QM31{0, 1, 0, 0}));
                r = qm31_add(r, qm31_mul(c2, QM31{0, 0, 1, 0}));
                r = qm31_add(r, qm31_mul(c3, QM31{0, 0, 0, 1}));
                // sampled = part[0] + part[1] P.y + part[2] P.x + part[3] (P.x P.y), summed in that order
                if (j == 0) sampled = r;
                else sampled = qm31_add(sampled, qm31_mul(r, j == 1 ? P.y : j == 2 ? P.x : qm31_mul(P.x, P.y)));
            }
            if (!qm31_eq(cp_eval, sampled)) FAIL(stwo_code(2, 0, 0, 3));
            phase = kDrawDeep;
            break;
        }
        case kDrawDeep: park(kParkDeepAlpha, drawn); phase = kFriRoot; break;
        // ---- stage III: fri_commit (fri/commit.simf:70-85)
        case kFriRoot: phase = kFriDraw; break;
        case kFriDraw:
            CQ(lay.c_fold + 4 * fri_l, drawn);
            fri_l++;
            phase = fri_l <= lay.K ? kFriRoot : kLast;
            break;
        case kLast: phase = kNonce; break;
        // ---- stage IV: check_proof_of_work (pow.simf:22-36)
        case kNonce:
            if (!(Hasher<HF>::pow_value(dig) < lay.pow_target)) FAIL(stwo_code(4, 0, 0, 0));
            phase = kQueries;
            fri_l = 0;
            break;
        // ---- stage V (first half): fri_generate_queries (fri/queries.simf:29-43)
        default:  // kQueries
#pragma unroll
            for (int j = 0; j < 8; j++)  // (Qd = Q except behind minimal records, whose expansion kernel pads the list)
                if (fri_l + j < lay.Qd) CW(lay.c_queries + fri_l + j, Hasher<HF>::native(st.v[j]) & qmask);
            fri_l += 8;
            if (fri_l >= lay.Qd) phase = kEnd;
            break;
        }
    }

    // ---- stage VI, query-independent part (fri/answers.simf:44-64,97-130)
    // deep_quotient_interpolant_coefficients (deep/quotients.simf:25-36) gives, for column k with
    // sample `value` at point sp and alpha_k = deep_alpha^(k+1):
    //     a_k = alpha_k a0_k,  b_k = alpha_k b0,  c_k = alpha_k (b0 value - a0_k sp.y)
    // with a0_k = (0, -2 im(value)) and b0 = (0, -2 im(sp.y)) the same for every column.  The
    // query kernel needs sum_k b_k v_k - y sum_k a_k - sum_k c_k, so this kernel keeps
    //     alpha_k (table),  b0,  A = sum alpha_k a0_k,  C = b0 V - A sp.y,  V = sum alpha_k value_k
    // and the query kernel computes b0 (sum_k alpha_k v_k).  Every operand is a field element in
    // [0, P] except the raw `value`, which the reference doubles with wrapping adds (kept) and
    // otherwise only multiplies (reduced first), so the sums are the reference's words.
    {
        const QM31 deep_alpha = parked(kParkDeepAlpha);
        const QM31Point P{parked(kParkPx), parked(kParkPy)};
        QM31Point P2 = qm31_point_add(P, P);
        CQ(lay.c_p, P.x);  CQ(lay.c_p + 4, P.y);
        CQ(lay.c_p2, P2.x); CQ(lay.c_p2 + 4, P2.y);
        uint4 *alpha_tab = reinterpret_cast<uint4 *>(ws + lay.ws_alpha) + (size_t)p * lay.n_pow;
        auto b0_of = [](const QM31Point &sp) {
            const CM31 im_py = q_im(sp.y);
            return cm31_neg(cm31_add(im_py, im_py));
        };
        QM31 alpha_i = deep_alpha, alpha_last = deep_alpha;
        auto batch = [&](const QM31Point &sp, CM31 b0, uint32_t h_base, uint32_t count, uint32_t tab0, QM31 &A,
                         QM31 &C) {
            QM31 V = qm31_zero();
            A = qm31_zero();
            for (uint32_t k = 0; k < count; k++) {
                const QM31 value = HQ(h_base + 4 * k);
                const CM31 im_v = q_im(value);
                const CM31 a0 = cm31_neg(cm31_add(im_v, im_v));
                alpha_tab[tab0 + k] = make_uint4(alpha_i.a, alpha_i.b, alpha_i.c, alpha_i.d);
                A = qm31_add(A, qm31_mul_im_c(alpha_i, a0));
                V = qm31_add(V, qm31_mul_c(alpha_i, qm31_red(value)));
                alpha_last = alpha_i;
                alpha_i = qm31_mul_c(alpha_i, deep_alpha);
            }
            C = qm31_sub(qm31_mul_im_c(V, b0), qm31_mul_c(A, sp.y));
        };
        const CM31 b01 = b0_of(P), b02 = b0_of(P2);
        CQ(lay.c_b01, q_make(CM31{0, 0}, b01));
        CQ(lay.c_b02, q_make(CM31{0, 0}, b02));
        QM31 A1, C1, A2, C2;
        batch(P, b01, lay.h_oods_trace, lay.N, 0, A1, C1);
        if (lay.mode == 1) {  // FIXTURE: second batch at 2P, alpha restarts (same powers, rewritten)
            alpha_i = deep_alpha;
            batch(P2, b02, lay.h_oods_cp, kCp, 0, A2, C2);
            CQ(lay.c_a1, A1); CQ(lay.c_c1, C1);
            CQ(lay.c_a2, A2); CQ(lay.c_c2, C2);
            CQ(lay.c_m1, alpha_last);  // alpha^16
        } else {  // LITERAL: one batch over all N + 16 columns at P
            batch(P, b01, lay.h_oods_cp, kCp, lay.N, A2, C2);
            CQ(lay.c_a1, qm31_add(A1, A2)); CQ(lay.c_c1, qm31_add(C1, C2));
            CQ(lay.c_a2, qm31_zero()); CQ(lay.c_c2, qm31_zero());
            CQ(lay.c_m1, alpha_i);  // alpha^(N+17), fri/answers.simf:126
        }
    }
    // `reset`: this kernel is the first of the pass that touches the status words, so it also RESETS them -- status[p] =
    // "no assert failed yet" or this lane's code, the accept count = 0 -- which saves the pass two memset dispatches (worth
    // 6 % each where a pass is 60 us: stark101 x 4 096 on 16 streams, profiles/r06_reset_in_kernel_ab.txt).  Behind minimal
    // records an earlier kernel has already written verdicts (reset = 0: the caller has reset, codes go through atomicMin).
    if (reset) {
        status[p] = fail;
        if (p == 0 && accept_count) *accept_count = 0;
    } else if (fail != 0xffffffffu) {
        atomicMin(&status[p], fail);
    }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
stwo_transcript_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                           uint32_t *__restrict__ status, uint32_t *__restrict__ accept_count, uint32_t reset)
{
    stwo_transcript_body<0>(lay, batch, ws, status, accept_count, reset);
}
__global__ void __launch_bounds__(64)
stwo_transcript_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                           uint32_t *__restrict__ status, uint32_t *__restrict__ accept_count, uint32_t reset)
{
    stwo_transcript_body<1>(lay, batch, ws, status, accept_count, reset);
}

// =============================================================================== query
// Coordinates the fold chain divides by (fri/folding.simf:15-41), without a point
// multiplication per layer.  With position_l = (query >> l) & ~1 the reference computes
//   layer 0:  y of circle_domain(L).at(bitrev(position_0, L))                       = y0
//   layer l:  x of line_domain(L - l).at(bitrev(position_l, L - l))
// and on the canonic cosets (groups/circle_domain.simf:17-25, line_domain.simf:18-31) these are
//   x_1 = s_1 x0,   x_l = s_l pi^(l-1)(x0),   pi(x) = 2x^2 - 1,   s_l = -1 iff bit l of query
// where (x0, y0) is the domain point of position_0: halving the position keeps the doubled
// point (line offset/step double with it), and clearing the position's low bit moves the
// circle index by 2^30, i.e. negates the point.  The point of the query itself is (x0, +-y0).
// All values are canonical field elements, so this re-association is exact.  The K+1 fold
// inverses and the (one or two) DEEP denominator norms share ONE m31 inversion (Montgomery's
// trick); a raw zero still reports the reference's abort code for exactly that inverse.

__global__ void __launch_bounds__(64)
stwo_query_kernel(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                  uint32_t *__restrict__ status)
{
    // batched-inversion scratch, sized by the launch for THIS config: 2 x (K + 3) rows of 64 words
    // (a fixed [K_max + 4] array held the kernel at 2 waves per SIMD: 17.9 KB per wave)
    extern __shared__ uint32_t sh_inv[];
    uint32_t (*sh_u)[64] = reinterpret_cast<uint32_t (*)[64]>(sh_inv);
    uint32_t (*sh_p)[64] = reinterpret_cast<uint32_t (*)[64]>(sh_inv + (lay.K + 3) * 64);
    const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= lay.ni) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t p = inst / lay.Q, q = inst - p * lay.Q;
    const uint32_t np = lay.np, nip = lay.nip;
    const uint32_t *ctx = ws + lay.ws_ctx;
    auto CG = [&](uint32_t w) { return ctx[(size_t)w * np + p]; };
    auto CGQ = [&](uint32_t w) { return QM31{CG(w), CG(w + 1), CG(w + 2), CG(w + 3)}; };
    uint32_t fail = 0xffffffffu;
    auto FAIL = [&](uint32_t code) { fail = code < fail ? code : fail; };

    const uint32_t query = CG(lay.c_queries + q);
    const uint32_t L = lay.L, K = lay.K;
    const bool two = lay.mode == 1;

    // Plan of the pair memoisation for the merkle kernel (lay.mchk, see stwo_top_kernel): at depth dd (root = 0) this
    // chain's position in EVERY tree of the proof is query >> (L - dd).  Which query of the proof leads that position
    // (the lowest one there, possibly this one) and which leads the sibling position (the other child of the parent).
    if (lay.mchk) {
        uint32_t lead[kTopMaxT], sibl[kTopMaxT];
#pragma unroll
        for (uint32_t j = 0; j < kTopMaxT; j++) { lead[j] = q; sibl[j] = 0xff; }
        for (uint32_t o = lay.Q; o-- > 0;) {  // downwards: the lowest match is written last
            const uint32_t x = CG(lay.c_queries + o) ^ query;
#pragma unroll
            for (uint32_t j = 0; j < kTopMaxT; j++) {
                const uint32_t dd = j + 1;
                if (dd > lay.T) continue;  // T <= L
                const uint32_t v = x >> (L - dd);
                if (v == 0 && o < q) lead[j] = o;
                if (v == 1) sibl[j] = o;
            }
        }
        uint4 pl;
        pl.x = lead[0] | lead[1] << 8 | lead[2] << 16 | lead[3] << 24;
        pl.y = lead[4] | lead[5] << 8 | lead[6] << 16 | lead[7] << 24;
        pl.z = sibl[0] | sibl[1] << 8 | sibl[2] << 16 | sibl[3] << 24;
        pl.w = sibl[4] | sibl[5] << 8 | sibl[6] << 16 | sibl[7] << 24;
        reinterpret_cast<uint4 *>(ws + lay.ws_plan)[inst] = pl;
    }

    // domain point of the even member of the query's leaf pair, and of the query itself
    const uint32_t pos0 = query & ~1u;
    const M31Point p0 = circle_point(circle_position_to_index(L, bit_reverse_position(pos0, L)));
    const M31Point dp = {p0.x, (query & 1) ? m31_sub(0, p0.y) : p0.y};

    // deep_quotient_denominator_inverse (deep/quotients.simf:15-22): d = dx * piy - dy * pix
    auto denominator = [&](uint32_t cw) {
        CM31 prx = {CG(cw + 0), CG(cw + 1)}, pix = {CG(cw + 2), CG(cw + 3)};
        CM31 pry = {CG(cw + 4), CG(cw + 5)}, piy = {CG(cw + 6), CG(cw + 7)};
        CM31 dx = cm31_sub_m31(prx, dp.x), dy = cm31_sub_m31(pry, dp.y);
        return cm31_sub(cm31_mul(dx, piy), cm31_mul(dy, pix));
    };
    const CM31 d1 = denominator(lay.c_p);
    const CM31 d2 = two ? denominator(lay.c_p2) : CM31{1, 0};

    // ---- one inversion for: y0, pi^(l-1)(x0) (l = 1..K), |d1|^2, |d2|^2
    const uint32_t n_inv = K + 3;
    {
        uint32_t t = p0.x, run = 1;
        for (uint32_t i = 0; i < n_inv; i++) {
            uint32_t u;
            if (i == 0) u = p0.y;
            else if (i <= K) { u = t; t = m31_dbl_x(t); }
            else if (i == K + 1) u = m31_add(m31_sqr(d1.a), m31_sqr(d1.b));
            else u = m31_add(m31_sqr(d2.a), m31_sqr(d2.b));
            if (u == 0) {  // m31_inv aborts on a zero word (fields/m31.simf:118-122)
                FAIL(i <= K ? stwo_code(7, i, q, 2) : stwo_code(6, 0, q, i - K - 1));
                u = 1;
            }
            run = m31_mul(run, u);
            sh_u[i][lane] = u;
            sh_p[i][lane] = run;
        }
        uint32_t r;
        m31_inv(run, r);
        for (uint32_t i = n_inv; i-- > 0;) {
            const uint32_t before = i ? sh_p[i - 1][lane] : 1u;
            sh_p[i][lane] = m31_mul(r, before);  // = 1 / u_i
            r = m31_mul(r, sh_u[i][lane]);
        }
    }

    // ---- stage VI: fri_answer at the query point (fri/answers.simf:97-130)
    QM31 eval;
    {
        const CM31 di1 = cm31_mul_m31(CM31{d1.a, m31_neg(d1.b)}, sh_p[K + 1][lane]);
        const CM31 di2 = cm31_mul_m31(CM31{d2.a, m31_neg(d2.b)}, sh_p[K + 2][lane]);
        const uint32_t *tv = batch + lay.off_trace_vals, *cv = batch + lay.off_cp_vals;
        const uint4 *alpha_tab = reinterpret_cast<const uint4 *>(ws + lay.ws_alpha) + (size_t)p * lay.n_pow;
        // sum_k alpha_k v_k with the four words kept as open 64-bit sums (three products per fold);
        // a queried value only meets multiplications, so it is reduced on load
        auto dot = [&](const uint32_t *vals, uint32_t count, uint32_t tab0) {
            uint64_t sa = 0, sb = 0, sc = 0, sd = 0;
            uint32_t open = 0;
            for (uint32_t k = 0; k < count; k++) {
                const uint4 al = alpha_tab[tab0 + k];
                const uint32_t v = m31_red(vals[(size_t)k * nip + inst]);
                sa = m31_mac(sa, al.x, v); sb = m31_mac(sb, al.y, v);
                sc = m31_mac(sc, al.z, v); sd = m31_mac(sd, al.w, v);
                if (++open == 3) {
                    sa = m31_fold62(sa); sb = m31_fold62(sb); sc = m31_fold62(sc); sd = m31_fold62(sd);
                    open = 0;
                }
            }
            return QM31{m31_red64(sa), m31_red64(sb), m31_red64(sc), m31_red64(sd)};
        };
        const QM31 S1 = dot(tv, lay.N, 0);
        const QM31 S2 = dot(cv, kCp, two ? 0 : lay.N);
        const CM31 b01 = q_im(CGQ(lay.c_b01));
        if (two) {
            const QM31 s = qm31_mul_im_c(S1, b01), s2 = qm31_mul_im_c(S2, q_im(CGQ(lay.c_b02)));
            QM31 n1 = qm31_sub(s, qm31_add(qm31_mul_m31(CGQ(lay.c_a1), dp.y), CGQ(lay.c_c1)));
            QM31 n2 = qm31_sub(s2, qm31_add(qm31_mul_m31(CGQ(lay.c_a2), dp.y), CGQ(lay.c_c2)));
            QM31 b1 = qm31_mul_cm31(n1, di1), b2 = qm31_mul_cm31(n2, di2);
            eval = qm31_add(qm31_mul_c(b1, CGQ(lay.c_m1)), b2);
        } else {
            const QM31 s = qm31_mul_im_c(qm31_add(S1, S2), b01);
            QM31 nn = qm31_sub(s, qm31_add(qm31_mul_m31(CGQ(lay.c_a1), dp.y), CGQ(lay.c_c1)));
            eval = qm31_mul_c(qm31_mul_cm31(nn, di1), CGQ(lay.c_m1));
        }
    }

    // ---- stage VII: fold chain (fri/layers.simf:48-70, fri/folding.simf:15-41)
    uint32_t *leaf = ws + lay.ws_leaf;
    const uint32_t *wit = batch + lay.off_fri_wit;
    uint32_t cur = query;
    // minimal records (ss_minimal.h): the other member of a fold pair is not in the proof when another query of the
    // proof sits there -- it is that chain's value entering this layer, one lane away (a proof's chains are lanes of this
    // wavefront: lay.Q divides 64, every lane of a live proof is live)
    const uint8_t *sibs = reinterpret_cast<const uint8_t *>(ws + lay.ws_sib) + (size_t)inst * 32;
    for (uint32_t l = 0; l <= K; l++) {
        QM31 w = {wit[((size_t)l * 4 + 0) * nip + inst], wit[((size_t)l * 4 + 1) * nip + inst],
                  wit[((size_t)l * 4 + 2) * nip + inst], wit[((size_t)l * 4 + 3) * nip + inst]};
        if (lay.minimal) {
            const uint32_t sb = sibs[l];
            const int src = (int)(lane - q + (sb == 0xff ? q : sb));
            const QM31 o = {(uint32_t)__shfl((int)eval.a, src), (uint32_t)__shfl((int)eval.b, src),
                            (uint32_t)__shfl((int)eval.c, src), (uint32_t)__shfl((int)eval.d, src)};
            if (sb != 0xff) w = o;
        }
        const bool odd = cur & 1;  // adjacent_leaves (fri/layers.simf:29-37)
        const uint32_t position = cur & ~1u;
        QM31 e0 = odd ? w : eval, e1 = odd ? eval : w;
        uint32_t *lf = leaf + (size_t)l * 8 * nip + inst;
        lf[0] = e0.a; lf[(size_t)nip] = e0.b; lf[(size_t)2 * nip] = e0.c; lf[(size_t)3 * nip] = e0.d;
        lf[(size_t)4 * nip] = e1.a; lf[(size_t)5 * nip] = e1.b; lf[(size_t)6 * nip] = e1.c;
        lf[(size_t)7 * nip] = e1.d;
        uint32_t cinv = sh_p[l][lane];
        if (l > 0 && ((query >> l) & 1)) cinv = m31_sub(0, cinv);
        QM31 f0 = qm31_add(e0, e1);
        QM31 f1 = qm31_mul_m31(qm31_sub(e0, e1), cinv);
        eval = qm31_add(f0, qm31_mul_c(CGQ(lay.c_fold + 4 * l), f1));
        cur = position >> 1;
    }

    // ---- last layer (fri/verify.simf:124-128, fri/layers.simf:73-78)
    {
        const uint32_t *head = batch + lay.off_head;
        QM31 last = {head[(size_t)(lay.h_last + 0) * np + p], head[(size_t)(lay.h_last + 1) * np + p],
                     head[(size_t)(lay.h_last + 2) * np + p], head[(size_t)(lay.h_last + 3) * np + p]};
        const uint32_t c = stwo_last_layer_code(lay.mode, L, K, q, cur, eval, last);
        if (c != 0xffffffffu) FAIL(c);
    }
    if (fail != 0xffffffffu) atomicMin(&status[p], fail);
}

// ============================================================================== merkle
// One wavefront = 64 chains of one kind.  Tile order: trace, cp, FRI layer 0..K (longest
// chains first, so the tail of the grid is made of the shortest ones).
// MIN: the batch was filled from minimal records (ss_minimal.hip) -- its own instantiation, so that the per-query
// kernels are the code they were.
template <int HF, bool MIN = false>
__device__ __forceinline__ void stwo_merkle_body(const StwoLayout &lay, const uint32_t *__restrict__ batch,
                                                 uint32_t *__restrict__ ws, uint32_t *__restrict__ status)
{
    const uint32_t tiles_per_type = lay.nip >> 6;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t type = tile / tiles_per_type;  // wave-uniform
    if (type >= lay.K + 3) return;
    const uint32_t g = tile - type * tiles_per_type;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t inst = g * 64 + lane;
    const bool live = inst < lay.ni;
    const uint32_t p = live ? inst / lay.Q : 0, q = live ? inst - p * lay.Q : 0;
    const uint32_t np = lay.np, nip = lay.nip;
    const uint32_t *ctx = ws + lay.ws_ctx;
    const uint32_t *head = batch + lay.off_head;
    const uint32_t query = ctx[(size_t)(lay.c_queries + q) * np + p];

    uint32_t node[8];  // native form
    uint32_t auth, len, root_w, code_base;
    const uint32_t *path;
    if (type < 2) {
        // verify_trace_evals / verify_cp_evals (evals/verify.simf:47-69); the leaf is
        // hash_node_m31_trace / hash_node_m31_cp (hasher.simf:85-97): H of ncol value words
        const uint32_t ncol = type == 0 ? lay.N : kCp;
        const uint32_t *vals = batch + (type == 0 ? lay.off_trace_vals : lay.off_cp_vals) + inst;
        Hasher<HF>::template stream<true, 0>(nullptr, [&](uint32_t i) { return vals[(size_t)i * nip]; }, ncol, node);
        auth = query + shl32(lay.L, 1);
        len = lay.L;
        path = batch + (type == 0 ? lay.off_trace_path : lay.off_cp_path);
        root_w = lay.h_roots + 8 * (type + 1);
        code_base = stwo_code(5, 0, q, 2 * type);
    } else {
        // verify_decommitment (fri/layers.simf:40-48)
        const uint32_t l = type - 2;
        const uint32_t *lf = ws + lay.ws_leaf + (size_t)l * 8 * nip + inst;
        uint32_t e0[4], e1[4], l0[8], l1[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { e0[j] = lf[(size_t)j * nip]; e1[j] = lf[(size_t)(4 + j) * nip]; }
        Hasher<HF>::template block<true, 0, 4>(nullptr, e0, l0);  // hash_node_qm31 (hasher.simf:100-104)
        Hasher<HF>::template block<true, 0, 4>(nullptr, e1, l1);
        Hasher<HF>::template pair<true>(l0, l1, node);
        const uint32_t logl = lay.L - l;
        const uint32_t position = (query >> l) & ~1u;
        auth = (position + shl32(logl, 1)) >> 1;
        len = logl - 1;
        path = batch + lay.off_fri_path[l];
        root_w = lay.h_fri_roots + 8 * l;
        code_base = stwo_code(7, l, q, 0);
    }

    // merkle_verify_32 (merkle.simf:22-44): fold the siblings leaf -> root.  With pair memoisation
    // on (lay.T) this kernel stops `top` levels below the root and hands the node to stwo_top_kernel.
    const uint32_t top = lay.T < len ? lay.T : len;
    const uint32_t n_lvl = len - top;
    // (the tiles hold only the n_lvl levels hashed here; the top ones live in the `top` section)
    const uint4 *tp = reinterpret_cast<const uint4 *>(path) + ((size_t)g * n_lvl * 2) * 64 + lane;
    const uint8_t *sibs = reinterpret_cast<const uint8_t *>(ws + lay.ws_sib) + (size_t)inst * 32;  // (minimal records only)
    const uint32_t min_shift = type < 2 ? 0 : type - 1;  // absolute level of the tree's first sibling (ss_minimal.h)
    uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0;
    if (n_lvl) { s0 = tp[0]; s1 = tp[64]; }
    for (uint32_t lvl = 0; lvl < n_lvl; lvl++) {
        uint4 n0 = s0, n1 = s1;
        if (lvl + 1 < n_lvl) {  // prefetch the next level while this one is hashed
            n0 = tp[(size_t)(lvl + 1) * 128];
            n1 = tp[(size_t)(lvl + 1) * 128 + 64];
        }
        uint32_t sib[8] = {Hasher<HF>::native(s0.x), Hasher<HF>::native(s0.y), Hasher<HF>::native(s0.z),
                           Hasher<HF>::native(s0.w), Hasher<HF>::native(s1.x), Hasher<HF>::native(s1.y),
                           Hasher<HF>::native(s1.z), Hasher<HF>::native(s1.w)};
        if (MIN) {
            // minimal records: a sibling that another query of the proof computes is not in the proof; that chain is a
            // lane of this wavefront and at the same level (same tree, lockstep), so its node is the sibling
            const uint32_t sb = live ? sibs[min_shift + lvl] : 0xffu;
            if (__any(sb != 0xff)) {
                const int src = (int)(lane - q + (sb == 0xff ? q : sb));
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t o = (uint32_t)__shfl((int)node[j], src);
                    if (sb != 0xff) sib[j] = o;
                }
            }
        }
        const bool right = auth & 1;  // node is the right child: H(sibling || node)
        uint32_t lft[8], rgt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lft[j] = right ? sib[j] : node[j];
            rgt[j] = right ? node[j] : sib[j];
        }
        Hasher<HF>::template pair<true>(lft, rgt, node);
        auth >>= 1;
        s0 = n0; s1 = n1;
    }
    if (!live) return;
    uint32_t fail = 0xffffffffu;
    if (top == 0) {
        bool same = true;
#pragma unroll
        for (int j = 0; j < 8; j++) same &= node[j] == Hasher<HF>::native(head[(size_t)(root_w + j) * np + p]);
        if (!same) fail = code_base + 1;   // assert!(eq_256(computed_root, root))  merkle.simf:43
    } else {
        uint4 *out = reinterpret_cast<uint4 *>(ws + lay.ws_top) + ((size_t)type * nip + inst) * 2;
        out[0] = make_uint4(node[0], node[1], node[2], node[3]);
        out[1] = make_uint4(node[4], node[5], node[6], node[7]);
        if (lay.mchk && !MIN) {  // (minimal records hold every sibling once: nothing to compare)
            // The byte compares of the pair memoisation (stwo_top_kernel: "same", "edge", "cross at the edge"), lane
            // against lane: 64 % Q == 0, so the Q chains of this tree of proof p are lanes lane - q .. lane - q + Q - 1
            // of this wavefront, all live (dead lanes pad whole proofs).  Every sibling of the top levels is read once,
            // by its own chain; what the other chain presents comes through the crossbar.  A difference flags the tree
            // (ws_flag, cleared before this kernel): stwo_top_cold_kernel then hashes its chains one by one.
            const uint4 pl = reinterpret_cast<const uint4 *>(ws + lay.ws_plan)[inst];
            const uint32_t base = lane - q;
            auto plan = [](uint32_t lo, uint32_t hi, uint32_t dd) {
                return ((dd <= 4 ? lo : hi) >> (8 * ((dd - 1) & 3))) & 0xff;
            };
            const uint4 *sp = reinterpret_cast<const uint4 *>(batch + lay.off_top + (size_t)p * lay.top_words + lay.top_off[type]) +
                              (size_t)q * 2;
            uint32_t diff = 0;
            {   // edge: chains at one position of depth `top` enter with the same node
                const int src = (int)(base + plan(pl.x, pl.y, top));
#pragma unroll
                for (int j = 0; j < 8; j++) diff |= node[j] ^ (uint32_t)__shfl((int)node[j], src);
            }
            const uint32_t o = plan(pl.z, pl.w, top);  // leader of the sibling position at depth `top`
            for (uint32_t j = 0; j < top; j++) {        // level n_lvl + j = the step from depth dd = top - j to dd - 1
                const uint4 a = sp[(size_t)j * lay.Q * 2], b = sp[(size_t)j * lay.Q * 2 + 1];
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                // same: a chain presents the sibling the leader of its position presents
                const int src = (int)(base + plan(pl.x, pl.y, top - j));
#pragma unroll
                for (int i = 0; i < 8; i++) diff |= w[i] ^ (uint32_t)__shfl((int)w[i], src);
                if (j == 0) {
                    // cross at the edge: this chain's entering node is what the sibling position's leader presents
                    const int so = (int)(base + (o == 0xff ? q : o));
                    uint32_t x = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) x |= node[i] ^ Hasher<HF>::native((uint32_t)__shfl((int)w[i], so));
                    if (o != 0xff) diff |= x;
                }
            }
            if (diff) ws[lay.ws_flag + (size_t)type * np + p] = 1;
        }
    }
    // assert!(eq_32(path, 1)), merkle.simf:42, evaluated first by the reference.  The index starts
    // in [2^len, 2^(len+1)) and loses one bit per sibling of the List<u256, 32>, so it ends at 1 iff
    // the proof's path holds exactly `len` siblings -- whatever they contain.
    if (batch[lay.off_plen + (size_t)type * nip + inst] != len) fail = code_base;
    if (fail != 0xffffffffu) atomicMin(&status[p], fail);
}

// ================================================================================= top
// Merkle pair memoisation (the reference notes at fri/queries.simf:41 that it
// does not deduplicate).  The Q queries of a proof index random leaves, so near the root their
// authentication paths run through the same nodes: at depth d (root = 0) the Q chains of a tree meet
// in at most min(2^d, Q) nodes, and an honest proof presents the same (left, right) pair for a node
// in every chain that passes through it.  This kernel hashes each DISTINCT node once.
//
// Exactness.  Chain c's node at depth d is H(pair_c), pair_c = its node at depth d+1 and its own
// sibling, ordered by its index bit.  Chains of one proof at the same position of depth d elect the
// lowest one as that node's leader.  Invariant I(d): every chain's true node at depth d equals the
// node stored for its depth-d leader.  Where a tree enters this kernel (depth `top`, nodes from
// stwo_merkle_kernel) I(top) is checked directly: chains at one position must carry equal nodes
// ("edge").  Step d+1 -> d, for a node P with chains S_L through its left and S_R through its right
// child, led at depth d+1 by cL and cR:
//   (i)  every chain of S_L presents the sibling bytes cL presents, likewise S_R / cR   ("same");
//   (ii) if both are non-empty, cL's sibling is cR's node and cR's sibling is cL's node ("cross";
//        each half is checked by the lane that has just produced that node, or from the stored
//        nodes where the tree enters).
// With I(d+1) these say that all chains through P present one pair, so one hash -- by P's leader, which
// is cL or cR -- gives every chain's node: I(d).  If every check of a (proof, tree) passes, comparing
// its single depth-0 node with the root is the reference's verdict for all Q chains (all fail or none;
// the reference stops at the first, query 0).  If any check fails the tree is flagged and its Q
// chains are re-hashed one by one from depth `top` (merkle.simf:22-44 as written), so a proof in which
// two queries disagree about a node still gets the reference's status word.  Positions depend on the
// queries only, and FRI layer l's leaf index is query >> (l+1) in a tree of depth L-1-l, so depth d
// of EVERY tree of a proof has position query >> (L-d): one plan per proof serves all its trees.
//
// One block = top_G proofs (top_G * Q <= 256 chains), persistent over groups.  The plan of all depths
// (leaders, slots, followers, sibling-position leaders) is made once per group in LDS.  The checks
// are 64 bytes of loads and a compare each and depend on the proof bytes only, so they form one queue
// per group, cut by depth, that the hash loop of the matching depth drains two per iteration: their
// loads are issued before a SHA-256 pair hash and compared after it, which hides their latency behind
// ALU work of the same wave (blocks sharing a CU run in lockstep, so nothing else would); what a
// shallow depth has beyond two per hash iteration is drained four in flight before its barrier.  Nodes of two consecutive depths live in the
// block's slice of ws_vals.
//
// Who makes the byte compares.  "same", "edge" and "cross at the edge" read proof bytes and nodes that exist before
// this kernel starts.  When Q divides 64 (lay.mchk) a proof's Q chains of a tree are lanes of ONE wavefront of
// stwo_merkle_kernel, which then makes them itself, lane against lane through the crossbar, from the per-query plan
// the query kernel leaves in ws_plan, and raises ws_flag; this kernel (LIGHTS = false: stwo_top_hash_kernel_*) starts
// from those flags, reads the same plan instead of searching for leaders, and only hashes -- with the cross check of
// every node it produces.  The set of compares is the same, so is the exactness argument.  Other query counts
// (and SS_FLAG_TOP_CHECKS) run everything here (LIGHTS = true).  Measured on the 2^20 config: 3.92 -> 3.46 ms, the
// merkle kernel unchanged at 14.7 ms within the run-to-run spread (+0.8 % instructions).
#ifndef SS_TOP_LIGHTS
#define SS_TOP_LIGHTS 2
#endif
#ifndef SS_TOP_WAVES
#define SS_TOP_WAVES 3
#endif
#ifndef SS_TOP_HASH_WAVES
#define SS_TOP_HASH_WAVES 3
#endif
constexpr uint32_t kTopLights = SS_TOP_LIGHTS;  // light checks that ride along one pair hash

template <int HF, bool LIGHTS, bool MIN = false>
__device__ __forceinline__ void stwo_top_body(const StwoLayout &lay, const uint32_t *__restrict__ batch,
                                              uint32_t *__restrict__ ws, uint32_t *__restrict__ status)
{
    constexpr uint32_t NT = kMaxList + 3;  // tree kinds: trace, cp, FRI layer 0..K (K <= 30)
    constexpr uint32_t D = kTopMaxT + 1;   // depths 0..T
    constexpr uint16_t kNone = 0xffff;
    constexpr uint32_t kMaxSeg = 3 * kTopMaxT;
    __shared__ uint32_t s_query[kTopChains];
    __shared__ uint16_t s_lead[D][kTopChains];   // [depth][chain]: the chain's leader
    __shared__ uint16_t s_slot[D][kTopChains];   // [depth][chain]: slot of the chain's leader
    __shared__ uint16_t s_item[D][kTopChains];   // [depth][slot]: the leader chain
    __shared__ uint16_t s_fol[D][kTopChains];    // [depth][j]: chains that are not leaders
    __shared__ uint16_t s_sibl[D][kTopChains];   // [depth][slot]: leader of the sibling position, or kNone
    __shared__ uint32_t s_cnt[D][kTopChains / 64];
    __shared__ uint32_t s_nlead[D];
    // light-check queue: segment s covers items [s_seg_start[s], s_seg_start[s+1]) = entries x trees
    __shared__ uint32_t s_seg_start[kMaxSeg + 1], s_seg_entries[kMaxSeg], s_seg_t0[kMaxSeg], s_seg_nt[kMaxSeg];
    __shared__ uint32_t s_seg_kind_dd[kMaxSeg];  // kind << 8 | depth ; kind 0 same, 1 cross at the edge, 2 edge
    __shared__ uint32_t s_seg_magic[kMaxSeg];    // floor(2^32 / entries) + 1: exact quotients below 2^16
    __shared__ uint32_t s_grp, s_take;
    __shared__ uint8_t s_g[kTopChains];            // chain -> proof of the group
    __shared__ uint8_t s_bad[NT][kTopChains / 2];  // [tree][proof of the group] (Q >= 2: <= 128 proofs)
    __shared__ uint32_t s_topoff[NT];              // word offset of the tree inside a proof's part of `top`
    __shared__ uint32_t s_len[NT], s_rootw[NT], s_code[NT];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t Q = lay.Q, G = lay.top_G, L = lay.L, K = lay.K, np = lay.np, nip = lay.nip;
    const uint32_t n_types = K + 3;
    const uint32_t Tmax = lay.T < L ? lay.T : L;   // <= kTopMaxT
    const uint32_t *head = batch + lay.off_head;
    const uint4 *topn = reinterpret_cast<const uint4 *>(ws + lay.ws_top);
    uint4 *vals = reinterpret_cast<uint4 *>(ws + lay.ws_vals) + (size_t)blockIdx.x * 2 * n_types * kTopChains * 2;

    if (tid < n_types) {
        const uint32_t l = tid < 2 ? 0 : tid - 2;
        s_len[tid] = tid < 2 ? L : L - 1 - l;
        s_topoff[tid] = lay.top_off[tid];
        s_rootw[tid] = tid < 2 ? lay.h_roots + 8 * (tid + 1) : lay.h_fri_roots + 8 * l;
        s_code[tid] = tid < 2 ? stwo_code(5, 0, 0, 2 * tid) : stwo_code(7, l, 0, 0);
    }

    struct H8 { uint4 a, b; };
    auto differ = [](const H8 &x, const H8 &y) {
        return ((x.a.x ^ y.a.x) | (x.a.y ^ y.a.y) | (x.a.z ^ y.a.z) | (x.a.w ^ y.a.w) | (x.b.x ^ y.b.x) |
                (x.b.y ^ y.b.y) | (x.b.z ^ y.b.z) | (x.b.w ^ y.b.w)) != 0;
    };
    // sibling of chain c of the group (first proof p0) at `lvl` levels above its leaf, as native words: the top
    // min(T, len) levels of a tree are stored as top[proof][type][level][query][8], 32 contiguous bytes each
    uint32_t p0 = 0;
    auto sibling = [&](uint32_t ti, uint32_t c, uint32_t lvl) {
        const uint32_t len = s_len[ti], top = lay.T < len ? lay.T : len, g = s_g[c];
        const uint4 *tp = reinterpret_cast<const uint4 *>(batch + lay.off_top + (size_t)(p0 + g) * lay.top_words + s_topoff[ti]) +
                          ((size_t)(lvl - (len - top)) * Q + (c - g * Q)) * 2;
        H8 h = {tp[0], tp[1]};
        h.a.x = Hasher<HF>::native(h.a.x); h.a.y = Hasher<HF>::native(h.a.y);
        h.a.z = Hasher<HF>::native(h.a.z); h.a.w = Hasher<HF>::native(h.a.w);
        h.b.x = Hasher<HF>::native(h.b.x); h.b.y = Hasher<HF>::native(h.b.y);
        h.b.z = Hasher<HF>::native(h.b.z); h.b.w = Hasher<HF>::native(h.b.w);
        return h;
    };
    auto unpack = [](const H8 &h, uint32_t (&v)[8]) {
        v[0] = h.a.x; v[1] = h.a.y; v[2] = h.a.z; v[3] = h.a.w; v[4] = h.b.x; v[5] = h.b.y; v[6] = h.b.z; v[7] = h.b.w;
    };
    const H8 zero8 = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};

    // Groups are handed out by a counter (zeroed before the launch), not by a fixed stride: the SIMD
    // favours its oldest wave, so co-resident blocks do not advance at the same pace, and with a fixed
    // share the fast ones would leave their CU half empty while the slow ones finish (measured: 2.8
    // of 4 waves per SIMD on average).  Every block keeps fetching until the counter runs out.
    // The counter counts quarter groups.  A block takes a whole group (4 quarters) while at least two
    // quarters per block are left and fewer after that (guided self-scheduling): the blocks then finish within a
    // quarter group's time of each other instead of a whole group's.  Small groups hash less efficiently,
    // so the switch is late: measured 4.19 ms without, 3.99 ms with, 4.08 / 4.05 ms switching at 4 / 1
    // quarters per block left (65 536 proofs, 768 resident blocks, 5.3 groups per block).
    uint32_t *counter = ws + lay.ws_counter;
    const uint32_t unit = (G & 3) == 0 ? G / 4 : G, upg = G / unit;
    const uint32_t n_units = (lay.n + unit - 1) / unit;
    while (true) {
        __syncthreads();  // the previous group's LDS is no longer read
        if (tid == 0) {
            uint32_t take = upg;
            if (upg > 1) {
                const uint32_t seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t guided = 2 * (seen < n_units ? n_units - seen : 0) / gridDim.x;
                take = guided < 1 ? 1 : guided < upg ? guided : upg;
            }
            s_grp = atomicAdd(counter, take);
            s_take = take;
        }
        __syncthreads();
        if (s_grp >= n_units) break;
        p0 = s_grp * unit;
        const uint32_t gp = lay.n - p0 < s_take * unit ? lay.n - p0 : s_take * unit;  // proofs of this group
        const uint32_t nch = gp * Q, inst0 = p0 * Q;
        if (tid < nch) {
            const uint32_t g = tid / Q, q = tid - g * Q;
            s_query[tid] = ws[lay.ws_ctx + (size_t)(lay.c_queries + q) * np + p0 + g];
            s_g[tid] = (uint8_t)g;
        }
        if (LIGHTS) {
            for (uint32_t i = tid; i < NT * (kTopChains / 2); i += kTopChains) (&s_bad[0][0])[i] = 0;
        } else {  // what stwo_merkle_kernel's byte compares found (lay.mchk)
            for (uint32_t i = tid; i < gp * n_types; i += kTopChains) {
                const uint32_t ti = i / gp, g = i - ti * gp;
                s_bad[ti][g] = (uint8_t)ws[lay.ws_flag + (size_t)ti * np + p0 + g];
            }
        }
        __syncthreads();

        // ---- plan, all depths at once: leaders, followers, slots
        const uint32_t first = tid - tid % Q;  // first chain of this chain's proof
        // (hash only: the query kernel has worked out who leads this chain's position and its sibling position at
        // every depth -- byte dd - 1 of pl.xy / pl.zw, as query numbers of the proof, ss_layout.h ws_plan)
        uint4 pl = make_uint4(0, 0, 0, 0);
        if (!LIGHTS && tid < nch) pl = reinterpret_cast<const uint4 *>(ws + lay.ws_plan)[inst0 + tid];
        auto plan_byte = [](uint32_t lo, uint32_t hi, uint32_t dd) { return ((dd <= 4 ? lo : hi) >> (8 * ((dd - 1) & 3))) & 0xff; };
        uint64_t votes[D];
        uint32_t lead_mask = 0;
#pragma unroll
        for (uint32_t dd = 0; dd < D; dd++) {
            bool lead = false;
            if (dd <= Tmax && tid < nch) {
                uint32_t c0 = tid;
                if (LIGHTS) {
                    const uint32_t pos = s_query[tid] >> (L - dd);
                    for (uint32_t c2 = first; c2 < tid; c2++)
                        if ((s_query[c2] >> (L - dd)) == pos) { c0 = c2; break; }
                } else {
                    c0 = first + (dd ? plan_byte(pl.x, pl.y, dd) : 0);
                }
                lead = c0 == tid;
                s_lead[dd][tid] = (uint16_t)c0;
            }
            votes[dd] = __ballot(lead);
            lead_mask |= (uint32_t)lead << dd;
            if (lane == 0) s_cnt[dd][wave] = (uint32_t)__popcll(votes[dd]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t dd = 0; dd < D; dd++) {
            if (dd > Tmax) continue;
            uint32_t before = (uint32_t)__popcll(votes[dd] & ((1ull << lane) - 1)), nlead = 0;
            for (uint32_t w = 0; w < kTopChains / 64; w++) {
                if (w < wave) before += s_cnt[dd][w];
                nlead += s_cnt[dd][w];
            }
            if (tid == 0) s_nlead[dd] = nlead;
            if (tid < nch) {
                if ((lead_mask >> dd) & 1) {
                    s_item[dd][before] = (uint16_t)tid;
                    s_slot[dd][tid] = (uint16_t)before;
                    // the chain that leads the sibling position of this depth (the other child of the parent)
                    uint16_t other = kNone;
                    if (dd && LIGHTS) {
                        const uint32_t want = (s_query[tid] >> (L - dd)) ^ 1;
                        for (uint32_t c2 = first, e = first + Q; c2 < e; c2++)
                            if ((s_query[c2] >> (L - dd)) == want) { other = (uint16_t)c2; break; }
                    } else if (dd) {
                        const uint32_t o = plan_byte(pl.z, pl.w, dd);
                        if (o != 0xff) other = (uint16_t)(first + o);
                    }
                    s_sibl[dd][before] = other;
                } else if (LIGHTS) {
                    s_fol[dd][tid - before] = (uint16_t)tid;
                }
            }
        }
        __syncthreads();
        if (tid < nch) {
#pragma unroll
            for (uint32_t dd = 0; dd < D; dd++)
                if (dd <= Tmax && !((lead_mask >> dd) & 1)) s_slot[dd][tid] = s_slot[dd][s_lead[dd][tid]];
        }
        // ---- the light-check queue: for dd = Tmax..1, "same" over the trees of depth >= dd, then "cross
        // at the edge" and "edge" (equal entering nodes) over the trees that enter this kernel at depth dd
        if (LIGHTS && tid == 0) {
            uint32_t ns = 0, at = 0;
            for (uint32_t dd = Tmax; dd >= 1; dd--) {
                const uint32_t fri = L - dd < K + 1 ? L - dd : K + 1;  // FRI trees with len >= dd
                auto magic = [](uint32_t v) { return v ? 0xffffffffu / v + 1 : 0; };  // v = 1: 0 (caught below)
                s_seg_start[ns] = at; s_seg_entries[ns] = nch - s_nlead[dd]; s_seg_t0[ns] = 0; s_seg_nt[ns] = 2 + fri;
                s_seg_kind_dd[ns] = dd; s_seg_magic[ns] = magic(s_seg_entries[ns]);
                at += s_seg_entries[ns] * s_seg_nt[ns];
                ns++;
                // trees with top == dd: every tree of length >= T when dd == Tmax, else the FRI tree of length dd
                uint32_t t0, nt;
                if (dd == Tmax) { t0 = 0; nt = 2 + fri; }
                else { const uint32_t l = L - 1 - dd; t0 = 2 + l; nt = l <= K ? 1 : 0; }
                s_seg_start[ns] = at; s_seg_entries[ns] = s_nlead[dd]; s_seg_t0[ns] = t0; s_seg_nt[ns] = nt;
                s_seg_kind_dd[ns] = 0x100 | dd; s_seg_magic[ns] = magic(s_seg_entries[ns]);
                at += s_seg_entries[ns] * nt;
                ns++;
                s_seg_start[ns] = at; s_seg_entries[ns] = nch - s_nlead[dd]; s_seg_t0[ns] = t0; s_seg_nt[ns] = nt;
                s_seg_kind_dd[ns] = 0x200 | dd; s_seg_magic[ns] = magic(s_seg_entries[ns]);
                at += s_seg_entries[ns] * nt;
                ns++;
            }
            s_seg_start[ns] = at;
        }
        __syncthreads();

        // node of chain c at depth dd: from stwo_merkle_kernel where the tree enters, else the stored
        // node of its leader (written at depth dd's step, parity dd & 1)
        auto node_at = [&](uint32_t ti, uint32_t c, uint32_t dd) {
            const uint32_t len = s_len[ti];
            const uint32_t top = lay.T < len ? lay.T : len;
            const uint4 *p = dd == top ? topn + ((size_t)ti * nip + inst0 + c) * 2
                                       : vals + ((size_t)((dd & 1) * n_types + ti) * kTopChains + s_slot[dd][c]) * 2;
            return H8{p[0], p[1]};
        };

        // one light check: the loads now, the compare later (after the hash they hide behind)
        struct Light { H8 a, b; uint32_t ti, g; };
        uint32_t seg = 0;  // each lane walks the queue in increasing order
        uint32_t light_end = 0;  // end of the part of the queue that belongs to the current depth step
        auto light_issue = [&](uint32_t i, Light &x) {
            x.ti = NT;
            x.a = x.b = zero8;
            if (i >= light_end) return;
            while (i >= s_seg_start[seg + 1]) seg++;
            const uint32_t ent = s_seg_entries[seg], kd = s_seg_kind_dd[seg], dd = kd & 0xff, kind = kd >> 8;
            const uint32_t r = i - s_seg_start[seg];
            const uint32_t qt = ent == 1 ? r : __umulhi(r, s_seg_magic[seg]);  // r / ent: r < 2^16, ent <= 256
            const uint32_t ti = s_seg_t0[seg] + qt, e = r - qt * ent;
            const uint32_t lvl = s_len[ti] - dd;  // siblings of the step dd -> dd-1
            if (kind == 1) {  // cross at the edge: leader c's entering node is what its sibling-position leader presents
                const uint32_t c = s_item[dd][e], o = s_sibl[dd][e];
                if (o == kNone) return;
                x.a = node_at(ti, c, dd);
                x.b = sibling(ti, o, lvl);
                x.g = s_g[c];
            } else {
                const uint32_t c = s_fol[dd][e], c1 = s_lead[dd][c];
                if (kind == 0) {  // same: a follower presents the sibling its leader presents
                    x.a = sibling(ti, c, lvl);
                    x.b = sibling(ti, c1, lvl);
                } else {          // edge: it enters with its leader's node
                    x.a = node_at(ti, c, dd);
                    x.b = node_at(ti, c1, dd);
                }
                x.g = s_g[c];
            }
            x.ti = ti;
        };
        auto light_settle = [&](const Light &x) {
            if (x.ti < NT && differ(x.a, x.b)) s_bad[x.ti][x.g] = 1;
        };
        uint32_t li = tid;  // next light item of this lane

        for (uint32_t d = Tmax; d-- > 0;) {
            const uint32_t par = d & 1;
            // The checks of depth d + 1 read the sibling level this step's hashes read, so they run in this
            // step: the users of a 128-byte line of the proof stay close in time.  (The resident blocks'
            // working sets exceed the 4 MB L2 of an XCD either way: 13.1 GB of L2 misses per 65 536-proof
            // launch against 13.4 GB with one queue per group drained at its own pace; same run time.)
            if (LIGHTS) {
                li = s_seg_start[3 * (Tmax - d - 1)] + tid;
                light_end = s_seg_start[3 * (Tmax - d)];
            }
            const uint32_t nlead = s_nlead[d];
            const uint32_t fri = L - 1 - d < K + 1 ? L - 1 - d : K + 1;  // FRI trees deeper than d
            const uint32_t total = nlead * (2 + fri);
            const uint32_t nlead_magic = 0xffffffffu / nlead + 1;  // nlead >= 1: every proof has a chain
            // ---- one hash per distinct node of depth d; the next item's bytes are fetched while this one
            // is hashed, and two light checks ride along
            H8 nd = zero8, sb = zero8, ys = zero8;
            uint32_t ti = 0, k = 0, flags = 0;  // flags: 1 = the chain's node is the right child, 2 = has a sibling leader
            auto fetch = [&](uint32_t i, H8 &nd_, H8 &sb_, H8 &ys_, uint32_t &ti_, uint32_t &k_, uint32_t &fl_) {
                ti_ = nlead == 1 ? i : __umulhi(i, nlead_magic);  // i / nlead: i < 2^16, nlead <= 256
                k_ = i - ti_ * nlead;
                const uint32_t c = s_item[d][k_], len = s_len[ti_];
                nd_ = node_at(ti_, c, d + 1);
                sb_ = sibling(ti_, c, len - 1 - d);
                fl_ = (s_query[c] >> (L - d - 1)) & 1;
                if (MIN) {
                    // minimal records: the sibling is the node another query's leader has just produced wherever there is
                    // one (c leads its depth-(d+1) node too: it is the lowest chain below the node it leads here)
                    const uint32_t o = s_sibl[d + 1][s_slot[d + 1][c]];
                    if (o != kNone) sb_ = node_at(ti_, o, d + 1);
                    return;  // ... and no proof bytes exist that the produced node could be compared with
                }
                const uint32_t y = d ? s_sibl[d][k_] : kNone;
                if (y != kNone) { ys_ = sibling(ti_, y, len - d); fl_ |= 2; }
            };
            if (tid < total) fetch(tid, nd, sb, ys, ti, k, flags);
            for (uint32_t i = tid; i < total; i += kTopChains) {
                H8 nd2 = zero8, sb2 = zero8, ys2 = zero8;
                uint32_t ti2 = 0, k2 = 0, flags2 = 0;
                if (i + kTopChains < total) fetch(i + kTopChains, nd2, sb2, ys2, ti2, k2, flags2);
                Light x[kTopLights];
                if (LIGHTS) {
#pragma unroll
                    for (uint32_t u = 0; u < kTopLights; u++) light_issue(li + u * kTopChains, x[u]);
                    li += kTopLights * kTopChains;
                }
                uint32_t a[8], b[8], lft[8], rgt[8], out[8];
                unpack(nd, a);
                unpack(sb, b);
                const bool right = flags & 1;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    lft[j] = right ? b[j] : a[j];
                    rgt[j] = right ? a[j] : b[j];
                }
                Hasher<HF>::template pair<true>(lft, rgt, out);
                const H8 o8 = {make_uint4(out[0], out[1], out[2], out[3]), make_uint4(out[4], out[5], out[6], out[7])};
                uint4 *o = vals + ((size_t)(par * n_types + ti) * kTopChains + k) * 2;
                o[0] = o8.a;
                o[1] = o8.b;
                // (ii): the leader of the sibling position presents this node as its sibling
                if ((flags & 2) && differ(o8, ys)) s_bad[ti][s_g[s_item[d][k]]] = 1;
                if (LIGHTS) {
#pragma unroll
                    for (uint32_t u = 0; u < kTopLights; u++) light_settle(x[u]);
                }
                nd = nd2; sb = sb2; ys = ys2; ti = ti2; k = k2; flags = flags2;
            }
            // what this step's hash iterations did not carry (the shallow depths have more checks than
            // hashes), four in flight
            while (LIGHTS && li - tid < light_end) {  // uniform over the block: li - tid is the same in every lane
                Light x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) light_issue(li + u * kTopChains, x[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) light_settle(x[u]);
                li += 4 * kTopChains;
            }
            __syncthreads();
        }

        // ---- roots: at depth 0 proof g's only leader is its first chain, slot g.  A tree whose checks failed is
        // flagged for stwo_top_cold_kernel, which re-hashes its Q chains one by one (merkle.simf:22-44 as written).
        for (uint32_t i = tid; i < gp * n_types; i += kTopChains) {
            const uint32_t ti = i / gp, g = i - ti * gp;
            const uint32_t badf = s_bad[ti][g];
            ws[lay.ws_flag + (size_t)ti * np + p0 + g] = badf;
            if (badf) { atomicAdd(counter + 1, 1u); continue; }
            const uint4 *v = vals + ((size_t)ti * kTopChains + g) * 2;  // parity 0
            uint32_t nd[8];
            unpack(H8{v[0], v[1]}, nd);
            bool same = true;
#pragma unroll
            for (int j = 0; j < 8; j++)
                same &= nd[j] == Hasher<HF>::native(head[(size_t)(s_rootw[ti] + j) * np + p0 + g]);
            if (!same) atomicMin(&status[p0 + g], s_code[ti] + 1);  // all Q chains fail: query 0 is first
        }
    }
}

// The cold path of the pair memoisation: trees in which two queries disagree about a node (flagged by
// stwo_top_kernel) get every chain hashed on its own from where the tree entered the top kernel, exactly as
// merkle_verify_32 is written.  An honest batch flags nothing: the kernel reads one counter and returns.
template <int HF>
__device__ __forceinline__ void stwo_top_cold_body(const StwoLayout &lay, const uint32_t *__restrict__ batch,
                                                   const uint32_t *__restrict__ ws, uint32_t *__restrict__ status)
{
    if (ws[lay.ws_counter + 1] == 0) return;  // nothing flagged: the whole (small) grid leaves here
    const uint32_t n_types = lay.K + 3, Q = lay.Q, L = lay.L, np = lay.np, nip = lay.nip;
    const uint64_t total = (uint64_t)n_types * lay.ni, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {  // (type, proof, query)
    const uint32_t ti = (uint32_t)(i / lay.ni), inst = (uint32_t)(i - (uint64_t)ti * lay.ni);
    const uint32_t p = inst / Q, q = inst - p * Q;
    if (!ws[lay.ws_flag + (size_t)ti * np + p]) continue;
    const uint32_t l = ti < 2 ? 0 : ti - 2;
    const uint32_t len = ti < 2 ? L : L - 1 - l, top = lay.T < len ? lay.T : len;
    const uint32_t root_w = ti < 2 ? lay.h_roots + 8 * (ti + 1) : lay.h_fri_roots + 8 * l;
    const uint32_t code = ti < 2 ? stwo_code(5, 0, q, 2 * ti) : stwo_code(7, l, q, 0);
    const uint32_t query = ws[lay.ws_ctx + (size_t)(lay.c_queries + q) * np + p];
    const uint4 *v = reinterpret_cast<const uint4 *>(ws + lay.ws_top) + ((size_t)ti * nip + inst) * 2;
    uint32_t nd[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
    const uint4 *tp = reinterpret_cast<const uint4 *>(batch + lay.off_top + (size_t)p * lay.top_words + lay.top_off[ti]);
    for (uint32_t d = top; d-- > 0;) {
        const uint4 *sp = tp + ((size_t)(top - 1 - d) * Q + q) * 2;  // level len - 1 - d, counted from the first top level
        const uint4 s0 = sp[0], s1 = sp[1];
        const uint32_t sib[8] = {Hasher<HF>::native(s0.x), Hasher<HF>::native(s0.y), Hasher<HF>::native(s0.z),
                                 Hasher<HF>::native(s0.w), Hasher<HF>::native(s1.x), Hasher<HF>::native(s1.y),
                                 Hasher<HF>::native(s1.z), Hasher<HF>::native(s1.w)};
        const bool right = (query >> (L - d - 1)) & 1;
        uint32_t lft[8], rgt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lft[j] = right ? sib[j] : nd[j];
            rgt[j] = right ? nd[j] : sib[j];
        }
        Hasher<HF>::template pair<false>(lft, rgt, nd);
    }
    const uint32_t *head = batch + lay.off_head;
    bool same = true;
#pragma unroll
    for (int j = 0; j < 8; j++) same &= nd[j] == Hasher<HF>::native(head[(size_t)(root_w + j) * np + p]);
    if (!same) atomicMin(&status[p], code + 1);
    }
}

__global__ void __launch_bounds__(256)
stwo_merkle_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                       uint32_t *__restrict__ status)
{
    stwo_merkle_body<0>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(256)
stwo_merkle_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                       uint32_t *__restrict__ status)
{
    stwo_merkle_body<1>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(256)
stwo_merkle_min_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                           uint32_t *__restrict__ status)
{
    stwo_merkle_body<0, true>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(256)
stwo_merkle_min_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                           uint32_t *__restrict__ status)
{
    stwo_merkle_body<1, true>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(kTopChains, SS_TOP_WAVES)
stwo_top_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                    uint32_t *__restrict__ status)
{
    stwo_top_body<0, true>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(kTopChains, SS_TOP_WAVES)
stwo_top_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                    uint32_t *__restrict__ status)
{
    stwo_top_body<1, true>(lay, batch, ws, status);
}
// lay.mchk: the byte compares were made by stwo_merkle_kernel, this one only hashes (and cross-checks what it hashes)
__global__ void __launch_bounds__(kTopChains, SS_TOP_HASH_WAVES)
stwo_top_hash_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                         uint32_t *__restrict__ status)
{
    stwo_top_body<0, false>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(kTopChains, SS_TOP_HASH_WAVES)
stwo_top_hash_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                         uint32_t *__restrict__ status)
{
    stwo_top_body<1, false>(lay, batch, ws, status);
}

// ... and behind minimal records: a sibling is the stored node of its position's leader wherever one exists
__global__ void __launch_bounds__(kTopChains, SS_TOP_HASH_WAVES)
stwo_top_min_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                        uint32_t *__restrict__ status)
{
    stwo_top_body<0, false, true>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(kTopChains, SS_TOP_HASH_WAVES)
stwo_top_min_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                        uint32_t *__restrict__ status)
{
    stwo_top_body<1, false, true>(lay, batch, ws, status);
}

__global__ void __launch_bounds__(256)
stwo_top_cold_kernel_sha(StwoLayout lay, const uint32_t *__restrict__ batch, const uint32_t *__restrict__ ws,
                         uint32_t *__restrict__ status)
{
    stwo_top_cold_body<0>(lay, batch, ws, status);
}
__global__ void __launch_bounds__(256)
stwo_top_cold_kernel_b2s(StwoLayout lay, const uint32_t *__restrict__ batch, const uint32_t *__restrict__ ws,
                         uint32_t *__restrict__ status)
{
    stwo_top_cold_body<1>(lay, batch, ws, status);
}

// ============================================================================ finalize
__global__ void stwo_finalize_kernel(uint32_t n, uint32_t *__restrict__ status,
                                     uint32_t *__restrict__ accept_count)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t s = status[p];
    s = s == 0xffffffffu ? 0u : s;
    status[p] = s;
    if (accept_count && s == 0) atomicAdd(accept_count, 1u);
}

}  // namespace ss
