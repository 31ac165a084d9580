// Device replay of the reference's known-answer tests ("tests of its CPU restatement
// and its HIP kernels").  ss_kat runs one reference function per item ON THE GPU, through the device functions the
// kernels are built from (ss_fields.h, ss_hash.h, ss_channel.h, ss_s101.h); tests/test_gpu_kats.py feeds it the vectors of
// tests/golden/kats.json -- the literals of the reference's `fn test_*` bodies -- and compares with the expected
// literals directly, not through the CPU checker.  Where a kernel evaluates a reference function in a re-associated form
// (hoisted DEEP coefficients, fold coordinates from a doubling chain: ss_stwo.hip) the op states the function as the
// .simf text does, over the same primitives; the kernels' own forms are held to the reference through the end-to-end
// proofs and the intermediates (tests/test_gpu_intermediates.py, which also compares them with KAT constants).
// Tests only.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "ss_abi.h"
#include "ss_channel.h"
#include "ss_ctx.h"
#include "ss_fields.h"
#include "ss_hash.h"
#include "ss_s101.h"

namespace ss {

struct KatOp { int in_w, out_w; };
// (in, out) words per item of every op (include/ss_verify.h documents them)
__host__ __device__ constexpr KatOp kat_op(int op)
{
    return op == 0 ? KatOp{97, 8} : op == 1 ? KatOp{267, 9} : op == 2 ? KatOp{34, 17} : op == 3 ? KatOp{4, 10} :
           op == 4 ? KatOp{8, 16} : op == 5 ? KatOp{4, 4} : op == 6 ? KatOp{3, 9} : op == 7 ? KatOp{18, 16} :
           op == 8 ? KatOp{93, 18} : op == 9 ? KatOp{19, 19} : op == 10 ? KatOp{15, 5} : op == 11 ? KatOp{12, 10} : KatOp{0, 0};
}
constexpr int kKatOps = 12;

__device__ inline QM31 ld_q(const uint32_t *p) { return QM31{p[0], p[1], p[2], p[3]}; }
__device__ inline void st_q(uint32_t *p, QM31 v) { p[0] = v.a; p[1] = v.b; p[2] = v.c; p[3] = v.d; }

// evals/composition_poly.simf:38-44
__device__ inline QM31 kat_from_partitions(QM31 p0, QM31 p1, QM31 p2, QM31 p3)
{
    QM31 r = qm31_add(p0, qm31_mul(p1, QM31{0, 1, 0, 0}));
    r = qm31_add(r, qm31_mul(p2, QM31{0, 0, 1, 0}));
    return qm31_add(r, qm31_mul(p3, QM31{0, 0, 0, 1}));
}

__global__ void kat_kernel(int op, uint32_t n, const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KatOp w = kat_op(op);
    const uint32_t *x = in + (size_t)i * w.in_w;
    uint32_t *o = out + (size_t)i * w.out_w;
    switch (op) {
    case 0: {  // SHA-256 of x[0] <= 96 big-endian words: sha256 / sha256_32 (sha256.simf:11-28, hasher.simf:34-52), the leaf
               // hashers hash_node_m31_trace / _cp / hash_node_qm31 (hasher.simf:85-104), sha256_pair, channel_mix_256 and
               // channel_mix_oods_evals (digest || values), through the merkle kernel's stream
        uint32_t d[8];
        Hasher<0>::template stream<false, 0>(nullptr, [&](uint32_t k) { return x[1 + k]; }, x[0] > 96 ? 96 : x[0], d);
        for (int j = 0; j < 8; j++) o[j] = d[j];
        break;
    }
    case 1: {  // merkle_verify_32: x = family (0 stark101 merkle.simf:22-43, 1 stwo merkle.simf:22-44 with `path == 1`), auth,
               // len <= 31, leaf[8], root[8], path[31][8] -> rc (0 ok, 1 path != 1, 2 root differs), computed root[8]
        uint32_t node[8], path = x[1];
        const uint32_t len = x[2] > 31 ? 31 : x[2];
        for (int j = 0; j < 8; j++) node[j] = x[3 + j];
        for (uint32_t l = 0; l < len; l++) {
            uint32_t sib[8], lft[8], rgt[8];
            for (int j = 0; j < 8; j++) sib[j] = x[19 + 8 * l + j];
            const bool right = path & 1;
            for (int j = 0; j < 8; j++) { lft[j] = right ? sib[j] : node[j]; rgt[j] = right ? node[j] : sib[j]; }
            Hasher<0>::template pair<false>(lft, rgt, node);
            path >>= 1;
        }
        bool same = true;
        for (int j = 0; j < 8; j++) { same &= node[j] == x[11 + j]; o[1 + j] = node[j]; }
        o[0] = (x[0] == 1 && path != 1) ? 1 : same ? 0 : 2;
        break;
    }
    case 2: {  // the stwo channel (channel.simf:31-172): x = digest[8], counter, k, payload[24] -> digest'[8], counter', result[8]
        Channel<0> c;
        for (int j = 0; j < 8; j++) c.dig.v[j] = x[j];
        c.ctr = x[8];
        const uint32_t *pl = x + 10;
        uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        switch (x[9]) {
        case 0: {  // two channel_draw_qm31 (:115-140)
            QM31 a, b;
            const bool ok = c.draw_qm31(a) && c.draw_qm31(b);
            st_q(r, a); st_q(r + 4, b);
            if (!ok) r[0] = 0xffffffffu;
            break;
        }
        case 1: {  // channel_draw_qm31_point (:143-151)
            QM31 t, inv;
            c.draw_qm31(t);
            const QM31 t2 = qm31_mul(t, t);
            if (!qm31_inv(qm31_add(qm31_one(), t2), inv)) { r[0] = 0xffffffffu; break; }
            st_q(r, qm31_mul(qm31_sub(qm31_one(), t2), inv));
            st_q(r + 4, qm31_mul(qm31_add(t, t), inv));
            break;
        }
        case 2: {  // channel_mix_u256 (:154-161)
            uint32_t v[8];
            for (int j = 0; j < 8; j++) v[j] = pl[j];
            c.mix(v);
            break;
        }
        case 3: {  // check_proof_of_work (pow.simf:22-36): mix_u64(nonce) then LE64(last 8 digest bytes) < target
            uint32_t v[2] = {pl[0], pl[1]};
            c.template mix_values<2>(v);
            r[0] = Hasher<0>::pow_value(c.dig.v) < (((uint64_t)pl[2] << 32) | pl[3]) ? 1 : 0;
            r[1] = __builtin_bswap32(pl[4]);  // reverse_bytes_32 (pow.simf:12-19)
            break;
        }
        case 4: {  // channel_draw_queries_8 (fri/queries.simf:14-26): mask = payload[0]
            uint32_t wds[8];
            c.draw_words(wds);
            for (int j = 0; j < 8; j++) r[j] = wds[j] & pl[0];
            break;
        }
        case 5: {  // evals_commit (evals/commit.simf:20-35): three roots -> cp_alpha
            uint32_t v[8];
            QM31 a;
            for (int j = 0; j < 8; j++) v[j] = pl[j];
            c.mix(v);
            for (int j = 0; j < 8; j++) v[j] = pl[8 + j];
            c.mix(v);
            c.draw_qm31(a);
            for (int j = 0; j < 8; j++) v[j] = pl[16 + j];
            c.mix(v);
            st_q(r, a);
            break;
        }
        case 6: {  // channel_mix_u256 then channel_draw_qm31: one step of fri_commit (fri/commit.simf:70-85)
            uint32_t v[8];
            QM31 a;
            for (int j = 0; j < 8; j++) v[j] = pl[j];
            c.mix(v);
            c.draw_qm31(a);
            st_q(r, a);
            break;
        }
        case 7: {  // channel_mix_line_poly (fri/commit.simf:48-57): four value words
            uint32_t v[4] = {pl[0], pl[1], pl[2], pl[3]};
            c.template mix_values<4>(v);
            break;
        }
        default: break;
        }
        for (int j = 0; j < 8; j++) { o[j] = c.dig.v[j]; o[9 + j] = r[j]; }
        o[8] = c.ctr;
        break;
    }
    case 3: {  // cm31 (fields/cm31.simf): a, b -> add, sub, mul, a / b (all-ones on abort), inv(a)
        const CM31 a = {x[0], x[1]}, b = {x[2], x[3]};
        const CM31 s = cm31_add(a, b), d = cm31_sub(a, b), m = cm31_mul(a, b);
        CM31 bi, ai;
        const bool okb = cm31_inv(b, bi), oka = cm31_inv(a, ai);
        const CM31 q = cm31_mul(a, bi);
        o[0] = s.a; o[1] = s.b; o[2] = d.a; o[3] = d.b; o[4] = m.a; o[5] = m.b;
        o[6] = okb ? q.a : 0xffffffffu; o[7] = okb ? q.b : 0xffffffffu;
        o[8] = oka ? ai.a : 0xffffffffu; o[9] = oka ? ai.b : 0xffffffffu;
        break;
    }
    case 4: {  // qm31 (fields/qm31.simf:30-80): a, b -> add, sub, a * m31(b.a), a * cm31(b.a, b.b)
        const QM31 a = ld_q(x), b = ld_q(x + 4);
        st_q(o, qm31_add(a, b)); st_q(o + 4, qm31_sub(a, b));
        st_q(o + 8, qm31_mul_m31(a, b.a)); st_q(o + 12, qm31_mul_cm31(a, CM31{b.a, b.b}));
        break;
    }
    case 5: {  // m31_point_add / m31_point_dbl (groups/m31_point.simf:40-55)
        const M31Point p = {x[0], x[1]}, q = {x[2], x[3]};
        const M31Point s = m31_point_add(p, q), d = m31_point_add(p, p);
        o[0] = s.x; o[1] = s.y; o[2] = d.x; o[3] = d.y;
        break;
    }
    case 6: {  // groups/coset.simf:20-52, circle_domain.simf:17-37, line_domain.simf:18-31: a, b, log ->
               // bit_reverse_position(a, log), index add / mul / neg, circle_domain(log) = (half, offset, step),
               // circle position a -> point index, line position a -> x coordinate
        const uint32_t a = x[0], b = x[1], lg = x[2];
        o[0] = bit_reverse_position(a, lg); o[1] = idx_add(a, b); o[2] = idx_mul(a, b); o[3] = idx_neg(a);
        o[4] = shl32((lg - 1) & 0xff, 1); o[5] = subgroup_gen((lg + 1) & 0xff); o[6] = subgroup_gen((lg - 1) & 0xff);
        o[7] = circle_position_to_index(lg, a);
        o[8] = circle_point(line_position_to_index(lg, a)).x;
        break;
    }
    case 7: {  // qm31_point_add and qm31_point_add_m31_point (groups/qm31_point.simf:37-43,68-74): P, Q, m
        const QM31Point P = {ld_q(x), ld_q(x + 4)}, Q = {ld_q(x + 8), ld_q(x + 12)};
        const QM31Point R = qm31_point_add(P, Q);
        // (the reference multiplies by the M31 coordinates directly; its test compares with the embedded point's sum)
        const QM31Point S = {qm31_sub(qm31_mul_m31(P.x, x[16]), qm31_mul_m31(P.y, x[17])),
                             qm31_add(qm31_mul_m31(P.x, x[17]), qm31_mul_m31(P.y, x[16]))};
        st_q(o, R.x); st_q(o + 4, R.y); st_q(o + 8, S.x); st_q(o + 12, S.y);
        break;
    }
    case 8: {  // x = log_size, P[8], four trace columns[16], alpha[4], sixteen cp parts[64] ->
               // vanishing_poly_eval (evals/composition_poly.simf:27-35,66-71), eval_composition_poly
               // (constraints/wide_fibonacci.simf:24-62; flag), composition_poly_eval_from_decomposed (:47-59),
               // composition_poly_eval_from_partitions of the first four parts (:38-44)
        const QM31Point P = {ld_q(x + 1), ld_q(x + 5)};
        const QM31 alpha = ld_q(x + 25);
        QM31 van = P.x;
        {
            const uint32_t n_iter = (x[0] - 1) & 0xff;
            for (uint32_t c = 0; c < 256; c++) {
                if (c == n_iter) break;
                van = qm31_dbl_x(van);
            }
        }
        QM31 acc = qm31_zero(), a = qm31_zero(), b = qm31_zero();
        uint32_t skip = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const QM31 c = ld_q(x + 9 + 4 * k);
            if (skip == 2) acc = qm31_add(qm31_mul(acc, alpha), qm31_sub(c, qm31_add(qm31_mul(b, b), qm31_mul(a, a))));
            else skip++;
            a = b;
            b = c;
        }
        QM31 vi;
        const bool ok = qm31_inv(van, vi);
        st_q(o, van);
        st_q(o + 4, qm31_mul(acc, vi));
        o[16] = ok ? 0 : 1;
        const uint32_t *d = x + 29;
        auto part = [&](int j) { return kat_from_partitions(ld_q(d + 4 * (0 + j)), ld_q(d + 4 * (4 + j)), ld_q(d + 4 * (8 + j)), ld_q(d + 4 * (12 + j))); };
        QM31 r = qm31_add(part(0), qm31_mul(part(1), P.y));
        r = qm31_add(r, qm31_mul(part(2), P.x));
        r = qm31_add(r, qm31_mul(part(3), qm31_mul(P.x, P.y)));
        st_q(o + 8, r);
        st_q(o + 12, kat_from_partitions(ld_q(d), ld_q(d + 4), ld_q(d + 8), ld_q(d + 12)));
        o[17] = 0;
        break;
    }
    case 9: {  // deep/quotients.simf:15-44: x = sample point[8], value[4], alpha_i[4], domain point[2], queried value ->
               // denominator inverse[2] (all-ones on abort), coefficients a, b, c[12], nominator[4], flag
        const QM31Point sp = {ld_q(x), ld_q(x + 4)};
        const QM31 value = ld_q(x + 8), alpha_i = ld_q(x + 12);
        const M31Point q = {x[16], x[17]};
        const CM31 dx = cm31_sub_m31(q_re(sp.x), q.x), dy = cm31_sub_m31(q_re(sp.y), q.y);
        const CM31 d = cm31_sub(cm31_mul(dx, q_im(sp.y)), cm31_mul(dy, q_im(sp.x)));
        CM31 di;
        const bool ok = cm31_inv(d, di);
        o[0] = ok ? di.a : 0xffffffffu; o[1] = ok ? di.b : 0xffffffffu;
        const CM31 iv = q_im(value), ipy = q_im(sp.y);
        const QM31 a = q_make(CM31{0, 0}, cm31_neg(cm31_add(iv, iv))), b = q_make(CM31{0, 0}, cm31_neg(cm31_add(ipy, ipy)));
        const QM31 c = qm31_sub(qm31_mul(b, value), qm31_mul(a, sp.y));
        const QM31 ca = qm31_mul(alpha_i, a), cb = qm31_mul(alpha_i, b), cc = qm31_mul(alpha_i, c);
        st_q(o + 2, ca); st_q(o + 6, cb); st_q(o + 10, cc);
        st_q(o + 14, qm31_sub(qm31_mul_m31(cb, x[18]), qm31_add(qm31_mul_m31(ca, q.y), cc)));
        o[18] = ok ? 0 : 1;
        break;
    }
    case 10: {  // circle_fold / line_fold (fri/folding.simf:15-41): x = kind (0 circle, 1 line), position, f_p[4], f_neg_p[4],
                // log_size, alpha[4] -> flag (1 = inverse abort), folded[4]
        const uint32_t pos = bit_reverse_position(x[1], x[10]);
        const uint32_t coord = x[0] == 0 ? circle_point(circle_position_to_index(x[10], pos)).y
                                         : circle_point(line_position_to_index(x[10], pos)).x;
        uint32_t inv;
        const bool ok = m31_inv(coord, inv);
        const QM31 fp = ld_q(x + 2), fn = ld_q(x + 6), alpha = ld_q(x + 11);
        const QM31 f0 = qm31_add(fp, fn), f1 = qm31_mul_m31(qm31_sub(fp, fn), ok ? inv : 0);
        o[0] = ok ? 0 : 1;
        st_q(o + 1, qm31_add(f0, qm31_mul(alpha, f1)));
        break;
    }
    case 11: {  // stark101: x = k, args[11] -> out[10]
        const uint32_t *a = x + 1;
        for (int j = 0; j < 10; j++) o[j] = 0;
        switch (x[0]) {
        case 0: {  // field.simf:24-94: a, b -> add, sub, mul, div (all-ones on abort), exp
            uint32_t d;
            o[0] = f101_add(a[0], a[1]); o[1] = f101_sub(a[0], a[1]); o[2] = f101_mul(a[0], a[1]);
            o[3] = f101_div(a[0], a[1], d) ? d : 0xffffffffu;
            o[4] = f101_pow(a[0], a[1]);
            break;
        }
        case 1: {  // channel_draw_32 (channel.simf:66-105): state[8], max -> value, state'[8]
            Dig101 st;
            for (int j = 0; j < 8; j++) st.v[j] = a[j];
            o[0] = s101_draw_mod(st, a[8]);
            for (int j = 0; j < 8; j++) o[1 + j] = st.v[j];
            break;
        }
        case 2: {  // fibsquare_read_coefficients (air.simf:30-35): state[8] -> three draws mod p, state'
            Dig101 st;
            for (int j = 0; j < 8; j++) st.v[j] = a[j];
            o[0] = s101_draw<S101_P>(st); o[1] = s101_draw<S101_P>(st); o[2] = s101_draw<S101_P>(st);
            for (int j = 0; j < 7; j++) o[3 + j] = st.v[j];  // (seven state words fit; the eighth is not compared)
            break;
        }
        case 3: {  // calc_x (air.simf:47-55), eval_p0 (:63-66): idx, x, f_x -> x(idx), p0 (all-ones on abort)
            uint32_t d;
            o[0] = f101_mul(5u, f101_pow(1734477367u, a[0]));
            o[1] = f101_div(f101_sub(a[2], 1), f101_sub(a[1], 1), d) ? d : 0xffffffffu;
            break;
        }
        case 4: {  // eval_cp (air.simf:58-101): a0, a1, a2, f_x, f_gx, f_ggx, x -> cp (all-ones on abort)
            const uint32_t xx = a[6], f_x = a[3], f_gx = a[4], f_ggx = a[5];
            uint32_t p0, p1, p2;
            bool ok = f101_div(f101_sub(f_x, 1), f101_sub(xx, 1), p0);
            ok &= f101_div(f101_sub(f_x, 2338775057u), f101_sub(xx, 2450347685u), p1);
            const uint32_t num0 = f101_sub(f_ggx, f101_add(f101_mul(f_x, f_x), f101_mul(f_gx, f_gx)));
            const uint32_t num1 = f101_mul(f101_mul(f101_sub(xx, 2342081930u), f101_sub(xx, 2450347685u)), f101_sub(xx, 532203874u));
            ok &= f101_div(f101_mul(num0, num1), f101_sub(f101_pow(xx, 1024), 1), p2);
            o[0] = ok ? f101_add(f101_add(f101_mul(p0, a[0]), f101_mul(p1, a[1])), f101_mul(p2, a[2])) : 0xffffffffu;
            break;
        }
        case 5: {  // fri_eval_cp_next (fri.simf:58-62): cpa, cpb, x, beta -> next (all-ones on abort)
            uint32_t op0, op1;
            bool ok = f101_div(f101_add(a[0], a[1]), 2, op0);
            ok &= f101_div(f101_sub(a[0], a[1]), f101_mul(a[2], 2), op1);
            o[0] = ok ? f101_add(op0, f101_mul(op1, a[3])) : 0xffffffffu;
            break;
        }
        case 6: {  // compute_auth_path (fri.simf:66-71): idx, domain_size -> cpa path, cpb path
            const uint32_t idx = a[0], dom = a[1];
            auto dv = [](uint32_t p, uint32_t q) { return q ? p / q : 0; };
            auto md = [](uint32_t p, uint32_t q) { return q ? p % q : p; };
            o[0] = md(idx, dom) + dom;
            o[1] = md(idx + dv(dom, 2), dom) + dom;
            break;
        }
        case 7: {  // channel_mix_32 (channel.simf:22-27): state <- sha256(state || be4(m))
            Dig101 st;
            for (int j = 0; j < 8; j++) st.v[j] = a[j];
            st = s101_hash_state<1>(st, a[8]);
            for (int j = 0; j < 8; j++) o[j] = st.v[j];
            break;
        }
        default: break;
        }
        break;
    }
    default: break;
    }
}

}  // namespace ss

using namespace ss;

extern "C" int ss_kat(ss_ctx *ctx, int op, size_t n, const uint32_t *in_host, size_t in_words, uint32_t *out_host, size_t out_words)
{
    if (!ctx || !in_host || !out_host || op < 0 || op >= kKatOps || !n || n > (1u << 20)) return set_err(SS_ERR_ARG, "bad argument");
    const KatOp w = kat_op(op);
    if (in_words != n * (size_t)w.in_w || out_words != n * (size_t)w.out_w)
        return set_err(SS_ERR_ARG, "op %d takes %d words and returns %d per item", op, w.in_w, w.out_w);
    std::lock_guard<std::mutex> lock(ctx->mu);
    SS_DEVICE_GUARD(ctx);
    uint32_t *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, in_words * 4));
    if (hipMalloc(&dout, out_words * 4) != hipSuccess) { (void)hipFree(din); return set_err(SS_ERR_HIP, "hipMalloc failed"); }
    int rc = SS_OK;
    auto run = [&]() -> int {
        HIP_TRY(hipMemcpy(din, in_host, in_words * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(kat_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (uint32_t)n, din, dout);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out_host, dout, out_words * 4, hipMemcpyDeviceToHost));
        return SS_OK;
    };
    rc = run();
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}
