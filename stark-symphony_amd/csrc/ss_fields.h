// Field and circle-group arithmetic for the device kernels.
//
// Bit-exactness rule.  The reference never range-checks the u32 words of a witness, and
// its M31 helpers discard the carry/borrow of add_32/subtract_32
// (stwo-verifier/src/fields/m31.simf:22-32).  The three primitives below therefore
// reproduce the *wrapping* semantics for ANY u32 input:
//      m31_add(a,b) = ((a + b) mod 2^32) mod P      m31_neg(a) = (P - a) mod 2^32
//      m31_mul(a,b) = (a * b as u64) mod P
// Everything else is composed from them in the reference's expression order wherever a raw
// witness word can reach an addition; where every operand is provably canonical (< P) the
// kernels are free to re-associate (DESIGN.md "exactness").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

constexpr uint32_t M31_P = 2147483647u;

// v mod P for any u32 v (fields/m31.simf:17-19)
__device__ __forceinline__ uint32_t m31_red(uint32_t v)
{
    uint32_t t = (v & M31_P) + (v >> 31);
    return t >= M31_P ? t - M31_P : t;
}
__device__ __forceinline__ uint32_t m31_add(uint32_t a, uint32_t b) { return m31_red(a + b); }
__device__ __forceinline__ uint32_t m31_neg(uint32_t a) { return M31_P - a; }
__device__ __forceinline__ uint32_t m31_sub(uint32_t a, uint32_t b) { return m31_add(a, m31_neg(b)); }
__device__ __forceinline__ uint32_t m31_mul(uint32_t a, uint32_t b)
{
    // x = a*b < 2^64 = x0 + x1 2^31 + x2 2^62, and 2^31 == 1 (mod P)
    uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t x0 = lo & M31_P;
    uint32_t x1 = ((lo >> 31) | (hi << 1)) & M31_P;
    uint32_t x2 = hi >> 30;
    uint32_t s = x0 + x1;                       // < 2^32
    s = (s & M31_P) + (s >> 31) + x2;           // <= P + 4
    return s >= M31_P ? s - M31_P : s;
}
__device__ __forceinline__ uint32_t m31_sqr(uint32_t a) { return m31_mul(a, a); }

// a^(P-2) by the reference's addition chain (m31.simf:117-133); aborts only on a raw 0 word.
__device__ inline bool m31_inv(uint32_t a, uint32_t &out)
{
    if (a == 0) { out = 0; return false; }
    auto pw = [](uint32_t v, int n) { for (int i = 0; i < n; i++) v = m31_sqr(v); return v; };
    uint32_t t0 = m31_mul(pw(a, 2), a);
    uint32_t t1 = m31_mul(pw(t0, 1), t0);
    uint32_t t2 = m31_mul(pw(t1, 3), t0);
    uint32_t t3 = m31_mul(pw(t2, 1), t0);
    uint32_t t4 = m31_mul(pw(t3, 8), t3);
    uint32_t t5 = m31_mul(pw(t4, 8), t3);
    out = m31_mul(pw(t5, 7), t2);
    return true;
}

struct CM31 { uint32_t a, b; };
struct QM31 { uint32_t a, b, c, d; };
struct M31Point { uint32_t x, y; };
struct QM31Point { QM31 x, y; };

// ------------------------------------------------------------ fields/cm31.simf:30-113
__device__ __forceinline__ CM31 cm31_add(CM31 x, CM31 y) { return {m31_add(x.a, y.a), m31_add(x.b, y.b)}; }
__device__ __forceinline__ CM31 cm31_neg(CM31 x) { return {m31_neg(x.a), m31_neg(x.b)}; }
__device__ __forceinline__ CM31 cm31_sub(CM31 x, CM31 y) { return {m31_sub(x.a, y.a), m31_sub(x.b, y.b)}; }
__device__ __forceinline__ CM31 cm31_sub_m31(CM31 x, uint32_t y) { return {m31_sub(x.a, y), x.b}; }
__device__ __forceinline__ CM31 cm31_mul_m31(CM31 x, uint32_t y) { return {m31_mul(x.a, y), m31_mul(x.b, y)}; }
__device__ __forceinline__ CM31 cm31_mul(CM31 x, CM31 y)
{
    return {m31_sub(m31_mul(x.a, y.a), m31_mul(x.b, y.b)), m31_add(m31_mul(x.a, y.b), m31_mul(x.b, y.a))};
}
__device__ inline bool cm31_inv(CM31 x, CM31 &out)
{
    CM31 conj = {x.a, m31_neg(x.b)};
    uint32_t norm = m31_add(m31_sqr(x.a), m31_sqr(x.b)), ninv;
    bool ok = m31_inv(norm, ninv);
    out = cm31_mul_m31(conj, ninv);
    return ok;
}

// ------------------------------------------------------------ fields/qm31.simf:20-132
__device__ __forceinline__ CM31 q_re(QM31 q) { return {q.a, q.b}; }
__device__ __forceinline__ CM31 q_im(QM31 q) { return {q.c, q.d}; }
__device__ __forceinline__ QM31 q_make(CM31 re, CM31 im) { return {re.a, re.b, im.a, im.b}; }
__device__ __forceinline__ QM31 qm31_zero() { return {0, 0, 0, 0}; }
__device__ __forceinline__ QM31 qm31_one() { return {1, 0, 0, 0}; }
__device__ __forceinline__ QM31 qm31_add(QM31 x, QM31 y)
{
    return {m31_add(x.a, y.a), m31_add(x.b, y.b), m31_add(x.c, y.c), m31_add(x.d, y.d)};
}
__device__ __forceinline__ QM31 qm31_sub(QM31 x, QM31 y)
{
    return {m31_sub(x.a, y.a), m31_sub(x.b, y.b), m31_sub(x.c, y.c), m31_sub(x.d, y.d)};
}
__device__ __forceinline__ QM31 qm31_mul_m31(QM31 x, uint32_t y)
{
    return {m31_mul(x.a, y), m31_mul(x.b, y), m31_mul(x.c, y), m31_mul(x.d, y)};
}
__device__ __forceinline__ QM31 qm31_mul_cm31(QM31 x, CM31 y)
{
    return q_make(cm31_mul(q_re(x), y), cm31_mul(q_im(x), y));
}
__device__ inline QM31 qm31_mul(QM31 x, QM31 y)
{
    CM31 ar = q_re(x), ai = q_im(x), br = q_re(y), bi = q_im(y);
    // (x + yi)(2 + i) = (2x - y) + (x + 2y)i; the product ai*bi is canonical, so adds suffice
    CM31 t = cm31_mul(ai, bi);
    CM31 tr = {m31_sub(m31_add(t.a, t.a), t.b), m31_add(t.a, m31_add(t.b, t.b))};
    CM31 re = cm31_add(cm31_mul(ar, br), tr);
    CM31 im = cm31_add(cm31_mul(ar, bi), cm31_mul(ai, br));
    return q_make(re, im);
}
__device__ inline bool qm31_inv(QM31 x, QM31 &out)
{
    CM31 ar = q_re(x), ai = q_im(x);
    CM31 ar_sq = cm31_mul(ar, ar), ai_sq = cm31_mul(ai, ai);
    CM31 ai_sq_dbl = cm31_add(ai_sq, ai_sq);
    CM31 ai_sq_rev = {m31_neg(ai_sq.b), ai_sq.a};
    CM31 den = cm31_add(ar_sq, cm31_neg(cm31_add(ai_sq_dbl, ai_sq_rev)));
    CM31 den_inv;
    bool ok = cm31_inv(den, den_inv);
    out = q_make(cm31_mul(ar, den_inv), cm31_mul(cm31_neg(ai), den_inv));
    return ok;
}
__device__ __forceinline__ bool qm31_eq(QM31 x, QM31 y)
{
    return x.a == y.a && x.b == y.b && x.c == y.c && x.d == y.d;
}

// ------------------------------------------------------------ lazily reduced arithmetic
// For operands that are field elements in [0, P] (canonical words, or the word P that m31_neg(0)
// returns) every reference primitive computes the exact field operation and returns the canonical
// word in [0, P - 1], so any algebraically equal evaluation gives the same bits.  The helpers
// below accumulate 32x32 products in 64 bits (one v_mad_u64_u32 each, 4.4 cycles/wave on gfx950,
// tools/probes/valu_mad64.hip) and reduce once per output word: a * b <= P^2 < 2^62, so four
// products fit.  A subtraction is a product with P - b.  A raw witness word may only enter
// through m31_red first, and only where the reference feeds it to multiplications alone.
__device__ __forceinline__ uint64_t m31_mac(uint64_t acc, uint32_t a, uint32_t b)
{
    return acc + (uint64_t)a * (uint64_t)b;
}
__device__ __forceinline__ uint32_t m31_red64(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t s = (lo & M31_P) + (((lo >> 31) | (hi << 1)) & M31_P);  // x0 + x1 < 2^32
    s = (s & M31_P) + (s >> 31) + (hi >> 30);                          // + x2, <= P + 4
    return s >= M31_P ? s - M31_P : s;
}
// x < 2^64 -> a congruent value < 2^62 + 4 (2^62 == 1 mod P): keeps a running sum of products open
__device__ __forceinline__ uint64_t m31_fold62(uint64_t x) { return (x & 0x3fffffffffffffffull) + (x >> 62); }

__device__ __forceinline__ CM31 cm31_mul_c(CM31 x, CM31 y)
{
    return {m31_red64(m31_mac(m31_mac(0, x.a, y.a), x.b, M31_P - y.b)),
            m31_red64(m31_mac(m31_mac(0, x.a, y.b), x.b, y.a))};
}
// (re, im) += (2 + i) t for a canonical t, as small terms of the open accumulators
__device__ __forceinline__ void cm31_acc_r(uint64_t &re, uint64_t &im, CM31 t)
{
    re += (uint64_t)t.a + t.a + (M31_P - t.b);
    im += (uint64_t)t.a + t.b + t.b;
}
__device__ inline QM31 qm31_mul_c(QM31 x, QM31 y)
{
    const CM31 t = cm31_mul_c(q_im(x), q_im(y));
    uint64_t ra = m31_mac(m31_mac(0, x.a, y.a), x.b, M31_P - y.b);
    uint64_t rb = m31_mac(m31_mac(0, x.a, y.b), x.b, y.a);
    cm31_acc_r(ra, rb, t);
    const uint64_t ia = m31_mac(m31_mac(m31_mac(m31_mac(0, x.a, y.c), x.b, M31_P - y.d), x.c, y.a), x.d, M31_P - y.b);
    const uint64_t ib = m31_mac(m31_mac(m31_mac(m31_mac(0, x.a, y.d), x.b, y.c), x.c, y.b), x.d, y.a);
    return {m31_red64(ra), m31_red64(rb), m31_red64(ia), m31_red64(ib)};
}
__device__ inline QM31 qm31_sqr_c(QM31 x)
{
    const CM31 t = {m31_red64(m31_mac(m31_mac(0, x.c, x.c), x.d, M31_P - x.d)), m31_red64(m31_mac(0, x.c + x.c, x.d))};
    uint64_t ra = m31_mac(m31_mac(0, x.a, x.a), x.b, M31_P - x.b);
    uint64_t rb = m31_mac(0, x.a + x.a, x.b);
    cm31_acc_r(ra, rb, t);
    const uint64_t ia = m31_mac(m31_mac(0, x.a + x.a, x.c), x.b + x.b, M31_P - x.d);
    const uint64_t ib = m31_mac(m31_mac(0, x.a + x.a, x.d), x.b + x.b, x.c);
    return {m31_red64(ra), m31_red64(rb), m31_red64(ia), m31_red64(ib)};
}
// x * (0 + y u): the product with an element whose real half is zero (DEEP a0 / b0)
__device__ inline QM31 qm31_mul_im_c(QM31 x, CM31 y)
{
    const CM31 t = cm31_mul_c(q_im(x), y);
    uint64_t ra = 0, rb = 0;
    cm31_acc_r(ra, rb, t);
    const CM31 im = cm31_mul_c(q_re(x), y);
    return {m31_red64(ra), m31_red64(rb), im.a, im.b};
}
// Canonical-operand forms (a, b < P) for code that never sees a raw word (the provers): the
// conditional subtraction is one v_min_u32 on the wrapped difference.
__device__ __forceinline__ uint32_t m31_add_c(uint32_t a, uint32_t b)
{
    const uint32_t s = a + b;
    return min(s, s - M31_P);
}
__device__ __forceinline__ uint32_t m31_sub_c(uint32_t a, uint32_t b)
{
    const uint32_t d = a - b;
    return min(d, d + M31_P);
}
__device__ __forceinline__ uint32_t m31_mul_c(uint32_t a, uint32_t b)
{
    const uint32_t lo = a * b, hi = __umulhi(a, b);                       // a b < 2^62
    const uint32_t s = (lo & M31_P) + __funnelshift_r(lo, hi, 31);        // x0 + (x >> 31) <= 2P
    return min(s, s - M31_P);
}
__device__ __forceinline__ QM31 qm31_red(QM31 x) { return {m31_red(x.a), m31_red(x.b), m31_red(x.c), m31_red(x.d)}; }

// ------------------------------------------------------ groups/m31_point.simf:33-97
__device__ __forceinline__ M31Point m31_point_add(M31Point l, M31Point r)
{
    return {m31_sub(m31_mul(l.x, r.x), m31_mul(l.y, r.y)), m31_add(m31_mul(l.x, r.y), m31_mul(l.y, r.x))};
}
__device__ __forceinline__ uint32_t m31_dbl_x(uint32_t x)
{
    uint32_t s = m31_sqr(x);
    return m31_sub(m31_add(s, s), 1);
}

// G^(2^k), k = 0..30, G = (2, 1268011823) (m31_point.simf:13).  Built at compile time by
// repeated doubling, so index -> point needs only the additions of the set bits; the result
// is the canonical pair the reference's 32-step double-and-add (m31_point.simf:59-97) returns.
struct PointTable { uint32_t x[31], y[31]; };
constexpr uint32_t cm31_mulmod(uint64_t a, uint64_t b) { return (uint32_t)((a * b) % M31_P); }
__host__ __device__ constexpr PointTable make_point_table()
{
    PointTable t{};
    uint32_t x = 2, y = 1268011823u;
    for (int k = 0; k < 31; k++) {
        t.x[k] = x; t.y[k] = y;
        uint32_t xx = cm31_mulmod(x, x);
        uint32_t nx = (uint32_t)(((uint64_t)2 * xx + M31_P - 1) % M31_P);
        uint32_t ny = (uint32_t)(((uint64_t)2 * cm31_mulmod(x, y)) % M31_P);
        x = nx; y = ny;
    }
    return t;
}
static __constant__ const PointTable kPointTable = make_point_table();

// index * G for a 31-bit circle point index (bit 31 of the raw word selects G^(2^31) = identity
// in the reference loop, so it is ignored here as well).
__device__ inline M31Point circle_point(uint32_t index)
{
    M31Point r = {1, 0};
    for (int k = 0; k < 31; k++) {
        if ((index >> k) & 1) r = m31_point_add(r, M31Point{kPointTable.x[k], kPointTable.y[k]});
    }
    return r;
}

// ------------------------------------------------------------ groups/coset.simf:14-52
__device__ __forceinline__ uint32_t shl32(uint32_t s, uint32_t v) { return (s & 0xff) >= 32 ? 0 : v << (s & 31); }
__device__ __forceinline__ uint32_t shr32(uint32_t s, uint32_t v) { return (s & 0xff) >= 32 ? 0 : v >> (s & 31); }
__device__ __forceinline__ uint32_t bit_reverse_position(uint32_t pos, uint32_t log_size)
{
    return shr32((32 - log_size) & 0xff, __brev(pos));
}
__device__ __forceinline__ uint32_t subgroup_gen(uint32_t log_size) { return shl32((31 - log_size) & 0xff, 1); }
__device__ __forceinline__ uint32_t idx_add(uint32_t a, uint32_t b) { return (a + b) & 0x7fffffffu; }
__device__ __forceinline__ uint32_t idx_mul(uint32_t a, uint32_t b) { return (a * b) & 0x7fffffffu; }
__device__ __forceinline__ uint32_t idx_neg(uint32_t a) { return (0x80000000u - a) & 0x7fffffffu; }

// groups/circle_domain.simf:17-37
__device__ inline uint32_t circle_position_to_index(uint32_t log_size, uint32_t position)
{
    uint32_t half = shl32((log_size - 1) & 0xff, 1);
    uint32_t offset = subgroup_gen((log_size + 1) & 0xff);
    uint32_t step = subgroup_gen((log_size - 1) & 0xff);
    if (position < half) return idx_add(offset, idx_mul(step, position));
    return idx_neg(idx_add(offset, idx_mul(step, position - half)));
}
// groups/line_domain.simf:18-31
__device__ inline uint32_t line_position_to_index(uint32_t log_size, uint32_t position)
{
    uint32_t offset = subgroup_gen((log_size + 2) & 0xff);
    uint32_t step = subgroup_gen(log_size & 0xff);
    return idx_add(offset, idx_mul(step, position));
}

// ------------------------------------------------ groups/qm31_point.simf:30-43
__device__ inline QM31 qm31_dbl_x(QM31 x)
{
    QM31 s = qm31_mul(x, x);
    return qm31_sub(qm31_add(s, s), qm31_one());
}
__device__ inline QM31Point qm31_point_add(QM31Point l, QM31Point r)
{
    return {qm31_sub(qm31_mul(l.x, r.x), qm31_mul(l.y, r.y)),
            qm31_add(qm31_mul(l.x, r.y), qm31_mul(l.y, r.x))};
}

// ------------------------------------------------------------------ stark101 field
// stark101/src/field.simf:14-94, p = 3 * 2^30 + 1.  add/sub/mul for any u32 inputs.
constexpr uint32_t S101_P = 3221225473u;
__device__ __forceinline__ uint32_t f101_add(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a + (uint64_t)b) % S101_P);
}
__device__ __forceinline__ uint32_t f101_sub(uint32_t a, uint32_t b) { return f101_add(a, S101_P - b); }
__device__ __forceinline__ uint32_t f101_mul(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) % S101_P);
}
__device__ inline uint32_t f101_pow(uint32_t a, uint32_t e)
{
    uint32_t res = 1, base = a;
    while (e) {
        if (e & 1) res = f101_mul(res, base);
        base = f101_mul(base, base);
        e >>= 1;
    }
    return res;
}
// div_mod (field.simf:42-63).  The reference's extended Euclid runs its updates in the field;
// it returns a * b^-1 exactly when 0 < b < p and aborts (gcd != 1) when b == 0 or b >= p
// (with r = p == 0 the first quotient step leaves new_r = 0, r = b != 1).  For 0 < b < p the
// inverse is unique, so Fermat's b^(p-2) gives the same word.
__device__ inline bool f101_div(uint32_t a, uint32_t b, uint32_t &out)
{
    if (b == 0 || b >= S101_P) { out = 0; return false; }
    out = f101_mul(a, f101_pow(b, S101_P - 2));
    return true;
}

}  // namespace ss
