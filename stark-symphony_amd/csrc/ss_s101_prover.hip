// ss_s101_prover.hip -- GPU building blocks of the stark101 (FibonacciSq) prover.
//
// SURVEY.md 8(f) row 2.  The reference's prover is Python (stark101/scripts/fibsquare/prover.py:25-171):
// Lagrange interpolation in O(n^2) polynomial arithmetic (polynomial.py:290-335), Horner evaluation on
// the 8x coset, quotient polynomials built in coefficient form, recursive Merkle trees.  A proof is a
// function of field VALUES only, so this file computes the same values the direct way:
//   * the trace polynomial by an inverse DFT over the size-1024 subgroup whose 1024th sample is chosen
//     so that the X^1023 coefficient vanishes (the unique interpolant of degree < 1023 through the 1023
//     trace points, prover.py:111),
//   * its evaluations on the coset by Horner (prover.py:112),
//   * the composition polynomial pointwise on the coset from f(x), f(gx), f(g^2 x) -- the identity the
//     verifier checks (stark101/src/air.simf:42-100) -- instead of dividing polynomials (prover.py:42-65),
//   * every FRI layer from the previous layer's evaluations (prover.py:68-88 folds coefficients and
//     re-evaluates; same values, fri.simf:44-62 is the verifier's form).
// Merkle trees reuse ss_p_hash_rows (leaf = SHA-256(be4 value), merkle.py:60-66) and ss_p_merkle.
#include <hip/hip_runtime.h>

#include "../../include/ss_prover.h"
#include "ss_ctx.h"
#include "ss_fields.h"

extern "C" int ss_internal_set_err(int code, const char *msg);

namespace ss {

constexpr uint32_t kGen101 = 5;                  // field.py:31 generator_val
constexpr uint32_t kTraceLog = 10, kLdeLog = 13;  // prover.py:100 domain_size 1024, x8
constexpr uint32_t kOrderOdd = 3;                 // p - 1 = 3 * 2^30

__device__ __forceinline__ uint32_t f101_subgroup_gen(uint32_t log_size)  // prover.py:33-36
{
    return f101_pow(kGen101, kOrderOdd << (30 - log_size));
}

// One block of 1024 threads: trace (prover.py:25-30) -> coefficients of the interpolant.
__global__ __launch_bounds__(1024) void p101_trace_poly_kernel(uint32_t seed, uint32_t *__restrict__ trace_out,
                                                               uint32_t *__restrict__ coef_out)
{
    __shared__ uint32_t t[1024];
    __shared__ uint32_t red[1024];
    const uint32_t k = threadIdx.x;
    if (k == 0) {
        uint32_t a = 1, b = seed % S101_P;
        t[0] = a;
        t[1] = b;
        for (uint32_t i = 2; i < 1023; i++) {
            const uint32_t c = f101_add(f101_mul(a, a), f101_mul(b, b));
            t[i] = c;
            a = b;
            b = c;
        }
    }
    __syncthreads();
    const uint32_t g = f101_subgroup_gen(kTraceLog);
    // q_1023 = (1/1024) sum_i y_i g^(-1023 i) = (1/1024) sum_i y_i g^i = 0  =>  y_1023 = -g * sum_{i<1023} y_i g^i
    red[k] = k < 1023 ? f101_mul(t[k], f101_pow(g, k)) : 0;
    __syncthreads();
    for (uint32_t s = 512; s > 0; s >>= 1) {
        if (k < s) red[k] = f101_add(red[k], red[k + s]);
        __syncthreads();
    }
    if (k == 0) t[1023] = f101_mul(f101_sub(0, red[0]), g);
    __syncthreads();
    const uint32_t gk = f101_pow(g, (1024 - k) & 1023);  // g^-k
    uint32_t acc = 0, w = 1;
    for (uint32_t i = 0; i < 1024; i++) {
        acc = f101_add(acc, f101_mul(t[i], w));
        w = f101_mul(w, gk);
    }
    coef_out[k] = f101_mul(acc, f101_pow(1024, S101_P - 2));
    if (k < 1023) trace_out[k] = t[k];
}

// out[j] = p(5 h^j), j < 8192 (prover.py:101,112; natural order of generate_left_coset).
__global__ __launch_bounds__(256) void p101_lde_kernel(const uint32_t *__restrict__ coef, uint32_t *__restrict__ out)
{
    __shared__ uint32_t c[1024];
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) c[i] = coef[i];
    __syncthreads();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t x = f101_mul(kGen101, f101_pow(f101_subgroup_gen(kLdeLog), j));
    uint32_t acc = 0;
    for (int i = 1023; i >= 0; i--) acc = f101_add(f101_mul(acc, x), c[i]);
    out[j] = acc;
}

// cp(x) = a0 (f(x) - 1)/(x - 1) + a1 (f(x) - claim)/(x - g^1022)
//       + a2 (f(g^2 x) - f(gx)^2 - f(x)^2) (x - g^1021)(x - g^1022)(x - g^1023)/(x^1024 - 1)
// (prover.py:42-65; air.simf:42-100 is the same expression on the verifier side).  g = h^8.
__global__ __launch_bounds__(256) void p101_composition_kernel(const uint32_t *__restrict__ p_ev, uint32_t a0,
                                                               uint32_t a1, uint32_t a2, uint32_t claim,
                                                               uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = 1u << kLdeLog;
    const uint32_t x = f101_mul(kGen101, f101_pow(f101_subgroup_gen(kLdeLog), j));
    const uint32_t g = f101_subgroup_gen(kTraceLog);
    const uint32_t g1021 = f101_pow(g, 1021), g1022 = f101_mul(g1021, g), g1023 = f101_mul(g1022, g);
    const uint32_t fx = p_ev[j], fgx = p_ev[(j + 8) & (n - 1)], fggx = p_ev[(j + 16) & (n - 1)];
    uint32_t p0, p1, p2;
    f101_div(f101_sub(fx, 1), f101_sub(x, 1), p0);
    f101_div(f101_sub(fx, claim), f101_sub(x, g1022), p1);
    uint32_t num = f101_sub(f101_sub(fggx, f101_mul(fgx, fgx)), f101_mul(fx, fx));
    num = f101_mul(num, f101_mul(f101_mul(f101_sub(x, g1021), f101_sub(x, g1022)), f101_sub(x, g1023)));
    f101_div(num, f101_sub(f101_pow(x, 1024), 1), p2);
    out[j] = f101_add(f101_add(f101_mul(a0, p0), f101_mul(a1, p1)), f101_mul(a2, p2));
}

// Layer i (length len) -> layer i+1 (len/2): next[j] = (a + b)/2 + beta (a - b)/(2x), a = in[j],
// b = in[j + len/2], x = (5 h^j)^(2^i) (prover.py:68-88).
__global__ __launch_bounds__(256) void p101_fold_kernel(uint32_t layer, uint32_t half, uint32_t beta,
                                                        const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    uint32_t x = f101_mul(kGen101, f101_pow(f101_subgroup_gen(kLdeLog), j));
    for (uint32_t s = 0; s < layer; s++) x = f101_mul(x, x);
    const uint32_t a = in[j], b = in[j + half];
    const uint32_t inv2 = (S101_P + 1) / 2;
    uint32_t odd;
    f101_div(f101_sub(a, b), x, odd);
    out[j] = f101_mul(f101_add(f101_add(a, b), f101_mul(beta, odd)), inv2);
}

}  // namespace ss

using namespace ss;

#define P101_TRY(expr)                                                           \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) return ss_internal_set_err(SS_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

extern "C" int ss_p101_trace_poly(ss_ctx *ctx, uint32_t seed, uint32_t *trace_out, uint32_t *coef_out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!trace_out || !coef_out) return ss_internal_set_err(SS_ERR_ARG, "ss_p101_trace_poly: null pointer");
    hipLaunchKernelGGL(p101_trace_poly_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seed, trace_out, coef_out);
    P101_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p101_lde(ss_ctx *ctx, const uint32_t *coef, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!coef || !out) return ss_internal_set_err(SS_ERR_ARG, "ss_p101_lde: null pointer");
    hipLaunchKernelGGL(p101_lde_kernel, dim3((1u << kLdeLog) / 256), dim3(256), 0, (hipStream_t)stream, coef, out);
    P101_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p101_composition(ss_ctx *ctx, const uint32_t *p_ev, const uint32_t alphas[3], uint32_t claim,
                                   uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!p_ev || !alphas || !out) return ss_internal_set_err(SS_ERR_ARG, "ss_p101_composition: null pointer");
    hipLaunchKernelGGL(p101_composition_kernel, dim3((1u << kLdeLog) / 256), dim3(256), 0, (hipStream_t)stream, p_ev,
                       alphas[0], alphas[1], alphas[2], claim, out);
    P101_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p101_fold(ss_ctx *ctx, uint32_t layer, uint32_t len, uint32_t beta, const uint32_t *in,
                            uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!in || !out || len < 2 || (len & (len - 1)) || len != (1u << kLdeLog) >> layer)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p101_fold: bad argument");
    const uint32_t half = len / 2;
    hipLaunchKernelGGL(p101_fold_kernel, dim3((half + 255) / 256), dim3(256), 0, (hipStream_t)stream, layer, half,
                       beta, in, out);
    P101_TRY(hipGetLastError());
    return SS_OK;
}
