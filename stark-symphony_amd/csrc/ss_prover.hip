// GPU building blocks of the wide-Fibonacci circle-STARK prover (include/ss_prover.h).
//
// Each kernel is one data-parallel step of tools/stwo_prover.py (the numpy prover that
// reproduces the reference's proofs byte for byte); stark-symphony_amd/prover.py chains them.
// All arithmetic is exact M31 / QM31 on canonical words, so any evaluation order gives the same
// bits as the numpy code.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/ss_prover.h"
#include "ss_channel.h"
#include "ss_ctx.h"
#include "ss_fields.h"
#include "ss_hash.h"
#include "ss_layout.h"

using namespace ss;

extern "C" int ss_internal_set_err(int code, const char *msg);

#define P_TRY(expr)                                                                  \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            char buf_[256];                                                          \
            snprintf(buf_, sizeof buf_, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            return ss_internal_set_err(SS_ERR_HIP, buf_);                            \
        }                                                                            \
    } while (0)

static inline unsigned blocks_for(uint64_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

namespace ss {

__device__ __forceinline__ QM31 ldq(const uint32_t *p) { return QM31{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void stq(uint32_t *p, QM31 v) { p[0] = v.a; p[1] = v.b; p[2] = v.c; p[3] = v.d; }

// ---------------------------------------------------------------------------------- trace
__global__ void p_trace_kernel(uint32_t n, uint32_t n_cols, uint32_t seed_term, uint32_t *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t a = 1, b = m31_add(m31_red(r), seed_term);
    out[r] = a;
    if (n_cols > 1) out[(size_t)n + r] = b;
    for (uint32_t k = 2; k < n_cols; k++) {
        uint32_t c = m31_add(m31_sqr(b), m31_sqr(a));
        out[(size_t)k * n + r] = c;
        a = b;
        b = c;
    }
}

// ------------------------------------------------------------------------------- twiddles
// Half coset of the canonic coset of log size m: index 2^(30-m) + j 2^(32-m); storage pair h
// is the natural position bitrev(h, m-1) (groups/circle_domain.simf:17-37).
__global__ void p_twiddles_kernel(uint32_t m, uint32_t *__restrict__ tw, uint32_t *__restrict__ itw,
                                  uint32_t *__restrict__ hx_out)
{
    const uint32_t half = 1u << (m - 1);
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= half) return;
    const uint32_t j = m > 1 ? (__brev(h) >> (32 - (m - 1))) : 0;
    const uint32_t idx = ((1u << (30 - m)) + j * (1u << (32 - m))) & 0x7fffffffu;
    const M31Point pt = circle_point(idx);
    uint32_t inv;
    tw[h] = pt.y;
    m31_inv(pt.y, inv);
    itw[h] = inv;
    if (hx_out) hx_out[h] = pt.x;
    uint32_t cur = pt.x;
    for (uint32_t i = 1; i < m; i++) {
        if (h & ((1u << i) - 1)) break;
        const size_t off = ((size_t)1 << m) - ((size_t)1 << (m - i));
        tw[off + (h >> i)] = cur;
        m31_inv(cur, inv);
        itw[off + (h >> i)] = inv;
        cur = m31_dbl_x(cur);
    }
}

// ------------------------------------------------------------------------------------ fft
// One butterfly layer over `ncols` columns; t indexes the 2^(m-1) butterflies of a column.
__global__ void p_fft_layer_kernel(uint32_t m, uint32_t layer, uint32_t *__restrict__ data,
                                   const uint32_t *__restrict__ tw, int inverse, uint32_t scale)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (m - 1))) return;
    uint32_t *col = data + ((size_t)blockIdx.y << m);
    const uint32_t h = t >> layer, l = t & ((1u << layer) - 1);
    const size_t i0 = ((size_t)h << (layer + 1)) + l, i1 = i0 + ((size_t)1 << layer);
    const size_t off = ((size_t)1 << m) - ((size_t)1 << (m - layer));
    const uint32_t w = tw[off + h];
    uint32_t v0 = col[i0], v1 = col[i1];
    if (inverse) {
        uint32_t s = m31_add(v0, v1), d = m31_mul(m31_sub(v0, v1), w);
        if (scale != 1) { s = m31_mul(s, scale); d = m31_mul(d, scale); }
        col[i0] = s;
        col[i1] = d;
    } else {
        const uint32_t x = m31_mul(v1, w);
        col[i0] = m31_add(v0, x);
        col[i1] = m31_sub(v0, x);
    }
}

// Several consecutive butterfly layers [lo, lo + nb) in ONE pass over HBM: a block stages 2^nb rows
// x 32 "lanes" of the layers' index space in LDS (rows = the 2^nb values of index bits
// [lo, lo + nb)), runs the nb layers there and writes the tile back, so m layers cost ceil(m / 8)
// read+write sweeps instead of m.
//   strided pass (lo >= 5): lanes = 32 neighbours in the bits below lo of one column; the tile's
//     2^nb - 1 twiddles do not depend on the lane and are staged in LDS.
//   contiguous pass (lo = 0): lanes = G groups of 2^nb contiguous words x CPB columns (G CPB = 32).
//     Here every butterfly of a column has its own twiddle (the table is as large as the column), so
//     the columns of a block re-use each twiddle line from L1/L2 instead of fetching it per column.
// Row stride 33 words keeps the row-wise (strided) and column-wise (contiguous) LDS accesses
// conflict free.  Operands are canonical, so the butterflies use the min-based m31 forms.
constexpr uint32_t kFftT = 32, kFftTp = 33, kFftMaxNb = 8;

// Low-degree extension (ss_p_lde): a forward transform whose input is 2^k coefficients followed by zeros.  In the top
// pass (lo + nb = m) only the rows below 2^(k - lo) hold anything, and the top z = m - k layers pair every such row
// with a zero row: v0 + 0 w = v0 - 0 w, i.e. they copy it.  So that pass reads the 2^k coefficients from their compact
// array (`src`, ncols x 2^k), writes each to its 2^z rows of the tile, runs the nb - z layers that do arithmetic and
// writes the whole tile to `data`: no zero-filled buffer, a sixteenth of the reads at blow-up 16, half the butterflies.
template <bool CONTIG>
__global__ void __launch_bounds__(256)
p_fft_pass_kernel(uint32_t m, uint32_t lo, uint32_t nb, uint32_t cpb_log, uint32_t *__restrict__ data,
                  const uint32_t *__restrict__ tw, int inverse, uint32_t scale, const uint32_t *__restrict__ src = nullptr,
                  uint32_t z = 0)
{
    __shared__ uint32_t tile[(1u << kFftMaxNb) * kFftTp];
    __shared__ uint32_t twl[1u << kFftMaxNb];
    const uint32_t tid = threadIdx.x;
    const uint32_t rows = 1u << nb, elems = rows * kFftT;
    const uint32_t glog = 5 - cpb_log, gmask = (1u << glog) - 1;  // contiguous: lane = column * G + group
    uint32_t hi;
    uint32_t *col;
    size_t base = 0;
    if (CONTIG) {
        hi = blockIdx.x << glog;
        col = data + ((size_t)(blockIdx.y << cpb_log) << m);
    } else {
        const uint32_t chunks = (1u << lo) / kFftT;
        hi = blockIdx.x / chunks;
        col = data + ((size_t)blockIdx.y << m);
        base = ((size_t)hi << (lo + nb)) + (size_t)(blockIdx.x % chunks) * kFftT;
    }
    auto gaddr = [&](uint32_t e) -> size_t {  // tile element e -> word offset from col
        if (CONTIG) {
            const uint32_t c = e >> nb;
            return ((size_t)(c >> glog) << m) + ((size_t)(hi + (c & gmask)) << nb) + (e & (rows - 1));
        }
        return base + ((size_t)(e >> 5) << lo) + (e & 31);
    };
    auto laddr = [&](uint32_t e) -> uint32_t {
        return CONTIG ? (e & (rows - 1)) * kFftTp + (e >> nb) : (e >> 5) * kFftTp + (e & 31);
    };
    if (!CONTIG && src) {  // (forward, top pass, hi = 0)
        const uint32_t *scol = src + ((size_t)blockIdx.y << (m - z));
        const uint32_t live = rows >> z;  // rows that hold coefficients
        for (uint32_t e = tid; e < live * kFftT; e += 256) {
            const uint32_t v = scol[gaddr(e)], r = e >> 5, c = e & 31;
            for (uint32_t j = 0; j < rows; j += live) tile[(r + j) * kFftTp + c] = v;
        }
    } else {
        for (uint32_t e = tid; e < elems; e += 256) tile[laddr(e)] = col[gaddr(e)];
    }
    if (!CONTIG) {  // layer ip of the tile at twl[rows - (rows >> ip) ..]
        for (uint32_t ip = 0; ip < nb; ip++) {
            const size_t goff = ((size_t)1 << m) - ((size_t)1 << (m - lo - ip));
            const uint32_t cnt = rows >> (ip + 1);
            for (uint32_t q = tid; q < cnt; q += 256)
                twl[rows - (rows >> ip) + q] = tw[goff + ((size_t)hi << (nb - 1 - ip)) + q];
        }
    }
    __syncthreads();
    // twiddle of local layer ip, butterfly group hl (= row >> (ip + 1)), lane c
    auto twid = [&](uint32_t ip, uint32_t hl, uint32_t c) -> uint32_t {
        if (CONTIG) {
            const size_t goff = ((size_t)1 << m) - ((size_t)1 << (m - ip));
            return tw[goff + ((size_t)(hi + (c & gmask)) << (nb - 1 - ip)) + hl];
        }
        return twl[rows - (rows >> ip) + hl];
    };
    auto bfly = [&](uint32_t &v0, uint32_t &v1, uint32_t w) {
        if (inverse) {
            const uint32_t sum = m31_add_c(v0, v1);
            v1 = m31_mul_c(m31_sub_c(v0, v1), w);
            v0 = sum;
        } else {
            const uint32_t x = m31_mul_c(v1, w);
            v1 = m31_sub_c(v0, x);
            v0 = m31_add_c(v0, x);
        }
    };
    // Two layers per LDS round trip (four rows per thread, three twiddles), one when nb is odd.
    for (uint32_t done = (!CONTIG && src) ? z : 0; done < nb;) {
        if (nb - done >= 2) {
            const uint32_t a = inverse ? done : nb - 2 - done;  // local layers a and a + 1
            const bool last = inverse && lo + a + 2 == m && scale != 1;
            for (uint32_t q = tid; q < elems / 4; q += 256) {
                // lanes of a wave differ in c: consecutive LDS words (no bank conflicts) and, in the
                // contiguous pass, only 32 / CPB distinct twiddle addresses per 32 lanes
                const uint32_t c = q & 31, pq = q >> 5;
                const uint32_t hq = pq >> a;
                const uint32_t j0 = (hq << (a + 2)) | (pq & ((1u << a) - 1));
                uint32_t *p0 = &tile[j0 * kFftTp + c], *p1 = p0 + (kFftTp << a), *p2 = p0 + (kFftTp << (a + 1)),
                         *p3 = p1 + (kFftTp << (a + 1));
                uint32_t v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
                const uint32_t wa0 = twid(a, 2 * hq, c), wa1 = twid(a, 2 * hq + 1, c), wb = twid(a + 1, hq, c);
                if (inverse) {
                    bfly(v0, v1, wa0); bfly(v2, v3, wa1);
                    bfly(v0, v2, wb);  bfly(v1, v3, wb);
                    if (last) {
                        v0 = m31_mul_c(v0, scale); v1 = m31_mul_c(v1, scale);
                        v2 = m31_mul_c(v2, scale); v3 = m31_mul_c(v3, scale);
                    }
                } else {
                    bfly(v0, v2, wb);  bfly(v1, v3, wb);
                    bfly(v0, v1, wa0); bfly(v2, v3, wa1);
                }
                *p0 = v0; *p1 = v1; *p2 = v2; *p3 = v3;
            }
            done += 2;
        } else {
            const uint32_t ip = inverse ? done : 0;  // the odd layer: last (inverse) or lowest (forward)
            const bool last = inverse && lo + ip + 1 == m && scale != 1;
            for (uint32_t b = tid; b < elems / 2; b += 256) {
                const uint32_t c = b & 31, pr = b >> 5;
                const uint32_t hl = pr >> ip;
                const uint32_t j0 = (hl << (ip + 1)) | (pr & ((1u << ip) - 1));
                uint32_t *p0 = &tile[j0 * kFftTp + c], *p1 = p0 + (kFftTp << ip);
                uint32_t v0 = *p0, v1 = *p1;
                bfly(v0, v1, twid(ip, hl, c));
                if (last) { v0 = m31_mul_c(v0, scale); v1 = m31_mul_c(v1, scale); }
                *p0 = v0; *p1 = v1;
            }
            done += 1;
        }
        __syncthreads();
    }
    for (uint32_t e = tid; e < elems; e += 256) col[gaddr(e)] = tile[laddr(e)];
}

// The strided pass again, with the layers in REGISTERS: a tile of 2^(A+4) rows x 32 lanes is a 2^A x 16 array of rows, and
// its A + 4 layers split into a radix-2^A stage over the upper row bits (a thread holds the 2^A rows k * 16 + r_lo of one
// lane: their twiddles depend on k only) and a radix-16 stage over the lower four (the 16 rows r_hi * 16 + k: twiddles
// from r_hi and k), with ONE exchange through LDS in between.  Against p_fft_pass_kernel<false> -- a tile staged in LDS,
// two layers per LDS round trip, every butterfly's operands addressed anew -- an element costs one LDS write and one read
// instead of ten LDS accesses, its global load and store go straight from and to registers, and the index arithmetic is
// per 16 rows instead of per 4.  Same butterflies on canonical values, so the same words come out.
//   forward: stage A (layers A+3 .. 4), exchange, stage B (layers 3 .. 0);  inverse: B (0 .. 3), exchange, A (4 .. A+3), scale.
//   LDE top pass (src): rows at or above `live` are zero and the top z layers copy (v0 + 0 w = v0 - 0 w = v0).
template <int A>
__global__ void __launch_bounds__(256)
p_fft_pass16_kernel(uint32_t m, uint32_t lo, uint32_t *__restrict__ data, const uint32_t *__restrict__ tw, int inverse,
                    uint32_t scale, const uint32_t *__restrict__ src, uint32_t z)
{
    constexpr uint32_t NB = A + 4, ROWS = 1u << NB, RA = 1u << A;
    __shared__ uint32_t tile[ROWS * kFftTp];
    __shared__ uint32_t twl[ROWS];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunks = (1u << lo) / kFftT;
    const uint32_t hi = blockIdx.x / chunks;
    uint32_t *col = data + ((size_t)blockIdx.y << m);
    const size_t base = ((size_t)hi << (lo + NB)) + (size_t)(blockIdx.x % chunks) * kFftT;
    for (uint32_t ip = 0; ip < NB; ip++) {  // layer ip of the tile at twl[ROWS - (ROWS >> ip) ..]
        const size_t goff = ((size_t)1 << m) - ((size_t)1 << (m - lo - ip));
        const uint32_t cnt = ROWS >> (ip + 1);
        for (uint32_t q = tid; q < cnt; q += 256) twl[ROWS - (ROWS >> ip) + q] = tw[goff + ((size_t)hi << (NB - 1 - ip)) + q];
    }
    __syncthreads();
    auto TW = [&](uint32_t ip, uint32_t hl) { return twl[ROWS - (ROWS >> ip) + hl]; };
    const uint32_t first_copy = NB - z;  // layers at or above this one pair a row with a zero row (LDE top pass only)
    auto fwd = [&](uint32_t &v0, uint32_t &v1, uint32_t ip, uint32_t hl) {
        if (ip >= first_copy) { v1 = v0; return; }
        const uint32_t x = m31_mul_c(v1, TW(ip, hl));
        v1 = m31_sub_c(v0, x);
        v0 = m31_add_c(v0, x);
    };
    auto inv = [&](uint32_t &v0, uint32_t &v1, uint32_t ip, uint32_t hl) {
        const uint32_t sum = m31_add_c(v0, v1);
        v1 = m31_mul_c(m31_sub_c(v0, v1), TW(ip, hl));
        v0 = sum;
    };
    const uint32_t c = tid & 31, rx = tid >> 5;
    const uint32_t *scol = src ? src + ((size_t)blockIdx.y << (m - z)) : nullptr;
    const uint32_t live = ROWS >> z;
    const bool last = inverse && lo + NB == m && scale != 1;
    if (!inverse) {
        // ---- stage A: rows k * 16 + r_lo
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t r_lo = rx + 8 * j;
            uint32_t v[RA];
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) {
                const uint32_t row = k * 16 + r_lo;
                const size_t g = base + ((size_t)row << lo) + c;
                v[k] = scol ? (row < live ? scol[g] : 0u) : col[g];
            }
#pragma unroll
            for (int s = A - 1; s >= 0; s--) {
#pragma unroll
                for (uint32_t k = 0; k < RA; k++)
                    if (!(k & (1u << s))) fwd(v[k], v[k + (1u << s)], 4 + s, k >> (s + 1));
            }
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) tile[(k * 16 + r_lo) * kFftTp + c] = v[k];
        }
        __syncthreads();
        // ---- stage B: rows r_hi * 16 + k
#pragma unroll
        for (uint32_t j = 0; j < (RA * 32 + 255) / 256; j++) {
            const uint32_t item = tid + 256 * j;
            if (item < RA * 32) {
                const uint32_t r_hi = item >> 5;
                uint32_t u[16];
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) u[k] = tile[(r_hi * 16 + k) * kFftTp + c];
#pragma unroll
                for (int s = 3; s >= 0; s--) {
#pragma unroll
                    for (uint32_t k = 0; k < 16; k++)
                        if (!(k & (1u << s))) fwd(u[k], u[k + (1u << s)], s, (r_hi << (3 - s)) + (k >> (s + 1)));
                }
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) col[base + ((size_t)(r_hi * 16 + k) << lo) + c] = u[k];
            }
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < (RA * 32 + 255) / 256; j++) {
            const uint32_t item = tid + 256 * j;
            if (item < RA * 32) {
                const uint32_t r_hi = item >> 5;
                uint32_t u[16];
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) u[k] = col[base + ((size_t)(r_hi * 16 + k) << lo) + c];
#pragma unroll
                for (int s = 0; s < 4; s++) {
#pragma unroll
                    for (uint32_t k = 0; k < 16; k++)
                        if (!(k & (1u << s))) inv(u[k], u[k + (1u << s)], s, (r_hi << (3 - s)) + (k >> (s + 1)));
                }
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) tile[(r_hi * 16 + k) * kFftTp + c] = u[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t r_lo = rx + 8 * j;
            uint32_t v[RA];
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) v[k] = tile[(k * 16 + r_lo) * kFftTp + c];
#pragma unroll
            for (int s = 0; s < A; s++) {
#pragma unroll
                for (uint32_t k = 0; k < RA; k++)
                    if (!(k & (1u << s))) inv(v[k], v[k + (1u << s)], 4 + s, k >> (s + 1));
            }
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) {
                const uint32_t out = last ? m31_mul_c(v[k], scale) : v[k];
                col[base + ((size_t)(k * 16 + r_lo) << lo) + c] = out;
            }
        }
    }
}

// The contiguous pass (the lowest A + 4 layers: lo = 0) with the layers in registers.  Here a lane is a (column, group of
// 2^(A+4) contiguous words) and every butterfly has its own twiddle, so the tile is still loaded and stored through LDS --
// rows are what is contiguous in memory -- but between those two coalesced sweeps the layers run as in
// p_fft_pass16_kernel: radix-2^A over the upper row bits, one exchange, radix-16 over the lower four, twiddles straight
// from the table (15 loads per 16 rows and stage, shared by the CPB columns of the block through L1).
template <int A>
__global__ void __launch_bounds__(256)
p_fft_pass16c_kernel(uint32_t m, uint32_t cpb_log, uint32_t *__restrict__ data, const uint32_t *__restrict__ tw, int inverse,
                     uint32_t scale)
{
    constexpr uint32_t NB = A + 4, ROWS = 1u << NB, RA = 1u << A, ELEMS = ROWS * kFftT;
    __shared__ uint32_t tile[ROWS * kFftTp];
    const uint32_t tid = threadIdx.x;
    const uint32_t glog = 5 - cpb_log, gmask = (1u << glog) - 1;  // lane = column * G + group
    const uint32_t hi = blockIdx.x << glog;
    uint32_t *col = data + ((size_t)(blockIdx.y << cpb_log) << m);
    auto gaddr = [&](uint32_t e) -> size_t {  // e = lane * ROWS + row
        const uint32_t l = e >> NB;
        return ((size_t)(l >> glog) << m) + ((size_t)(hi + (l & gmask)) << NB) + (e & (ROWS - 1));
    };
    for (uint32_t e = tid; e < ELEMS; e += 256) tile[(e & (ROWS - 1)) * kFftTp + (e >> NB)] = col[gaddr(e)];
    __syncthreads();
    const uint32_t c = tid & 31, rx = tid >> 5;
    const uint32_t g = hi + (c & gmask);
    auto TW = [&](uint32_t ip, uint32_t hl) {  // layer ip of the whole transform (lo = 0), butterfly group hl of this lane's group
        return tw[(((size_t)1 << m) - ((size_t)1 << (m - ip))) + ((size_t)g << (NB - 1 - ip)) + hl];
    };
    auto fwd = [&](uint32_t &v0, uint32_t &v1, uint32_t w) {
        const uint32_t x = m31_mul_c(v1, w);
        v1 = m31_sub_c(v0, x);
        v0 = m31_add_c(v0, x);
    };
    auto inv = [&](uint32_t &v0, uint32_t &v1, uint32_t w) {
        const uint32_t sum = m31_add_c(v0, v1);
        v1 = m31_mul_c(m31_sub_c(v0, v1), w);
        v0 = sum;
    };
    const bool last = inverse && NB == m && scale != 1;
    auto stage_a = [&]() {
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t r_lo = rx + 8 * j;
            uint32_t v[RA];
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) v[k] = tile[(k * 16 + r_lo) * kFftTp + c];
            if (!inverse) {
#pragma unroll
                for (int s = A - 1; s >= 0; s--) {
#pragma unroll
                    for (uint32_t k = 0; k < RA; k++)
                        if (!(k & (1u << s))) fwd(v[k], v[k + (1u << s)], TW(4 + s, k >> (s + 1)));
                }
            } else {
#pragma unroll
                for (int s = 0; s < A; s++) {
#pragma unroll
                    for (uint32_t k = 0; k < RA; k++)
                        if (!(k & (1u << s))) inv(v[k], v[k + (1u << s)], TW(4 + s, k >> (s + 1)));
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < RA; k++) tile[(k * 16 + r_lo) * kFftTp + c] = last ? m31_mul_c(v[k], scale) : v[k];
        }
    };
    auto stage_b = [&]() {
#pragma unroll
        for (uint32_t j = 0; j < (RA * 32 + 255) / 256; j++) {
            const uint32_t item = tid + 256 * j;
            if (item < RA * 32) {
                const uint32_t r_hi = item >> 5;
                uint32_t u[16];
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) u[k] = tile[(r_hi * 16 + k) * kFftTp + c];
                if (!inverse) {
#pragma unroll
                    for (int s = 3; s >= 0; s--) {
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++)
                            if (!(k & (1u << s))) fwd(u[k], u[k + (1u << s)], TW(s, (r_hi << (3 - s)) + (k >> (s + 1))));
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++)
                            if (!(k & (1u << s))) inv(u[k], u[k + (1u << s)], TW(s, (r_hi << (3 - s)) + (k >> (s + 1))));
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) tile[(r_hi * 16 + k) * kFftTp + c] = u[k];
            }
        }
    };
    if (!inverse) { stage_a(); __syncthreads(); stage_b(); }
    else { stage_b(); __syncthreads(); stage_a(); }
    __syncthreads();
    for (uint32_t e = tid; e < ELEMS; e += 256) col[gaddr(e)] = tile[(e & (ROWS - 1)) * kFftTp + (e >> NB)];
}

static void launch_contig_pass(uint32_t m, uint32_t nb, uint32_t cpb_log, uint32_t ncols, uint32_t *data, const uint32_t *tw, int inverse,
                               uint32_t scale, hipStream_t stream)
{
    const dim3 grid((1u << (m - nb)) >> (5 - cpb_log), ncols >> cpb_log);
#ifndef SS_FFT_LDS
    switch (nb) {
    case 5: hipLaunchKernelGGL(p_fft_pass16c_kernel<1>, grid, dim3(256), 0, stream, m, cpb_log, data, tw, inverse, scale); return;
    case 6: hipLaunchKernelGGL(p_fft_pass16c_kernel<2>, grid, dim3(256), 0, stream, m, cpb_log, data, tw, inverse, scale); return;
    case 7: hipLaunchKernelGGL(p_fft_pass16c_kernel<3>, grid, dim3(256), 0, stream, m, cpb_log, data, tw, inverse, scale); return;
    case 8: hipLaunchKernelGGL(p_fft_pass16c_kernel<4>, grid, dim3(256), 0, stream, m, cpb_log, data, tw, inverse, scale); return;
    default: break;
    }
#endif
    hipLaunchKernelGGL(p_fft_pass_kernel<true>, grid, dim3(256), 0, stream, m, 0u, nb, cpb_log, data, tw, inverse, scale, nullptr, 0u);
}

// a strided pass of nb layers: the register kernel for 5..8 layers, the LDS kernel otherwise (and with -DSS_FFT_LDS, for A/B runs)
static void launch_strided_pass(uint32_t m, uint32_t lo, uint32_t nb, uint32_t ncols, uint32_t *data, const uint32_t *tw, int inverse,
                                uint32_t scale, const uint32_t *src, uint32_t z, hipStream_t stream)
{
    const dim3 grid((1u << (m - nb)) / kFftT, ncols);
#ifndef SS_FFT_LDS
    switch (nb) {
    case 5: hipLaunchKernelGGL(p_fft_pass16_kernel<1>, grid, dim3(256), 0, stream, m, lo, data, tw, inverse, scale, src, z); return;
    case 6: hipLaunchKernelGGL(p_fft_pass16_kernel<2>, grid, dim3(256), 0, stream, m, lo, data, tw, inverse, scale, src, z); return;
    case 7: hipLaunchKernelGGL(p_fft_pass16_kernel<3>, grid, dim3(256), 0, stream, m, lo, data, tw, inverse, scale, src, z); return;
    case 8: hipLaunchKernelGGL(p_fft_pass16_kernel<4>, grid, dim3(256), 0, stream, m, lo, data, tw, inverse, scale, src, z); return;
    default: break;
    }
#endif
    hipLaunchKernelGGL(p_fft_pass_kernel<false>, grid, dim3(256), 0, stream, m, lo, nb, 0u, data, tw, inverse, scale, src, z);
}

// --------------------------------------------------------------------------------- hashing
// WC = the row width when it is one of the two the stwo prover hashes 2^24 times (4 trace columns, 16 composition
// columns): the padding words and, for 16, the whole second block's message schedule are then compile-time constants
// (the same digest; fewer instructions).  WC = 0: any width, from the argument.
template <int HF, uint32_t WC>
__global__ void p_hash_rows_kernel(size_t n, uint32_t w, const uint32_t *__restrict__ cols, size_t stride,
                                   uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d[8];
    if (WC == 16) {  // 64 bytes: the message of a pair hash (second SHA-256 block = constant padding)
        uint32_t a[8], b[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            a[k] = Hasher<HF>::native(cols[(size_t)k * stride + i]);
            b[k] = Hasher<HF>::native(cols[(size_t)(8 + k) * stride + i]);
        }
        Hasher<HF>::template pair<true>(a, b, d);
    } else {
        Hasher<HF>::template stream<true, 0>(nullptr, [&](uint32_t k) { return cols[(size_t)k * stride + i]; }, WC ? WC : w, d);
    }
    uint4 *o = reinterpret_cast<uint4 *>(out + i * 8);
    o[0] = make_uint4(Hasher<HF>::native(d[0]), Hasher<HF>::native(d[1]), Hasher<HF>::native(d[2]), Hasher<HF>::native(d[3]));
    o[1] = make_uint4(Hasher<HF>::native(d[4]), Hasher<HF>::native(d[5]), Hasher<HF>::native(d[6]), Hasher<HF>::native(d[7]));
}

template <int HF>
__global__ void p_hash_qm31_kernel(size_t n, const uint32_t *__restrict__ vals, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = reinterpret_cast<const uint4 *>(vals)[i];
    const uint32_t e[4] = {v.x, v.y, v.z, v.w};
    uint32_t d[8];
    Hasher<HF>::template block<true, 0, 4>(nullptr, e, d);
    uint4 *o = reinterpret_cast<uint4 *>(out + i * 8);
    o[0] = make_uint4(Hasher<HF>::native(d[0]), Hasher<HF>::native(d[1]), Hasher<HF>::native(d[2]), Hasher<HF>::native(d[3]));
    o[1] = make_uint4(Hasher<HF>::native(d[4]), Hasher<HF>::native(d[5]), Hasher<HF>::native(d[6]), Hasher<HF>::native(d[7]));
}

template <int HF>
__device__ __forceinline__ void merkle_parent(size_t i, const uint32_t *children, uint32_t *parents);

template <int HF>
__global__ void p_merkle_level_kernel(size_t n_parents, const uint32_t *__restrict__ children,
                                      uint32_t *__restrict__ parents)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    merkle_parent<HF>(i, children, parents);
}

// The last levels of a tree (n <= 512 nodes -> the root) by ONE block: a level with at most 256 parents is one wavefront per
// SIMD of a CU, i.e. one pair hash of latency, which is what a launch of its own takes too -- minus the launch.  Levels above
// stay launches: a single CU hashes them slower than the chip does.  levels: the n nodes, then their parents...
template <int HF>
__global__ void __launch_bounds__(256) p_merkle_tail_kernel(uint32_t n, uint32_t *levels)
{
    uint32_t off = 0;
    for (uint32_t cur = n; cur > 1; cur >>= 1) {
        if (threadIdx.x < (cur >> 1)) merkle_parent<HF>(threadIdx.x, levels + (size_t)off * 8, levels + (size_t)(off + cur) * 8);
        off += cur;
        __threadfence_block();
        __syncthreads();  // (the level just written is read by other wavefronts of this block next)
    }
}

template <int HF>
__device__ __forceinline__ void merkle_parent(size_t i, const uint32_t *children, uint32_t *parents)
{
    const uint4 *c = reinterpret_cast<const uint4 *>(children + i * 16);
    const uint4 a0 = c[0], a1 = c[1], b0 = c[2], b1 = c[3];
    const uint32_t l[8] = {Hasher<HF>::native(a0.x), Hasher<HF>::native(a0.y), Hasher<HF>::native(a0.z),
                           Hasher<HF>::native(a0.w), Hasher<HF>::native(a1.x), Hasher<HF>::native(a1.y),
                           Hasher<HF>::native(a1.z), Hasher<HF>::native(a1.w)};
    const uint32_t r[8] = {Hasher<HF>::native(b0.x), Hasher<HF>::native(b0.y), Hasher<HF>::native(b0.z),
                           Hasher<HF>::native(b0.w), Hasher<HF>::native(b1.x), Hasher<HF>::native(b1.y),
                           Hasher<HF>::native(b1.z), Hasher<HF>::native(b1.w)};
    uint32_t d[8];
    Hasher<HF>::template pair<true>(l, r, d);
    uint4 *o = reinterpret_cast<uint4 *>(parents + i * 8);
    o[0] = make_uint4(Hasher<HF>::native(d[0]), Hasher<HF>::native(d[1]), Hasher<HF>::native(d[2]), Hasher<HF>::native(d[3]));
    o[1] = make_uint4(Hasher<HF>::native(d[4]), Hasher<HF>::native(d[5]), Hasher<HF>::native(d[6]), Hasher<HF>::native(d[7]));
}

// parent[i] = H(leaf[i] || leaf[i]): a tree whose leaves come in equal pairs
template <int HF>
__global__ void p_merkle_dup_kernel(size_t n, const uint32_t *__restrict__ leaf, uint32_t *__restrict__ parents)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 *c = reinterpret_cast<const uint4 *>(leaf + i * 8);
    const uint4 a0 = c[0], a1 = c[1];
    const uint32_t l[8] = {Hasher<HF>::native(a0.x), Hasher<HF>::native(a0.y), Hasher<HF>::native(a0.z),
                           Hasher<HF>::native(a0.w), Hasher<HF>::native(a1.x), Hasher<HF>::native(a1.y),
                           Hasher<HF>::native(a1.z), Hasher<HF>::native(a1.w)};
    uint32_t d[8];
    Hasher<HF>::template pair<true>(l, l, d);
    uint4 *o = reinterpret_cast<uint4 *>(parents + i * 8);
    o[0] = make_uint4(Hasher<HF>::native(d[0]), Hasher<HF>::native(d[1]), Hasher<HF>::native(d[2]), Hasher<HF>::native(d[3]));
    o[1] = make_uint4(Hasher<HF>::native(d[4]), Hasher<HF>::native(d[5]), Hasher<HF>::native(d[6]), Hasher<HF>::native(d[7]));
}

// ----------------------------------------------------------------------------- composition
__global__ void p_composition_kernel(uint32_t n_log, uint32_t n_cols, const uint32_t *__restrict__ ev,
                                     const uint32_t *__restrict__ hx, QM31 alpha, uint32_t *__restrict__ out)
{
    const uint32_t size = 1u << (n_log + 1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    uint32_t van = hx[i >> 1];
    for (uint32_t k = 1; k < n_log; k++) van = m31_dbl_x(van);
    uint32_t van_inv;
    m31_inv(van, van_inv);
    QM31 acc = qm31_zero();
    uint32_t a = ev[i], b = n_cols > 1 ? ev[(size_t)size + i] : 0;
    for (uint32_t k = 2; k < n_cols; k++) {
        const uint32_t c = ev[(size_t)k * size + i];
        const uint32_t cons = m31_sub(c, m31_add(m31_sqr(b), m31_sqr(a)));
        acc = qm31_mul(acc, alpha);
        acc.a = m31_add(acc.a, cons);
        a = b;
        b = c;
    }
    out[i] = m31_mul(acc.a, van_inv);
    out[(size_t)size + i] = m31_mul(acc.b, van_inv);
    out[(size_t)2 * size + i] = m31_mul(acc.c, van_inv);
    out[(size_t)3 * size + i] = m31_mul(acc.d, van_inv);
}

// -------------------------------------------------------------------------- eval at point
__global__ void p_fold_m31_kernel(size_t n_out, const uint32_t *__restrict__ in, QM31 f, uint32_t *__restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    QM31 r = qm31_mul_m31(f, in[2 * j + 1]);
    r.a = m31_add(r.a, in[2 * j]);
    stq(out + 4 * j, r);
}
__global__ void p_fold_qm31_kernel(size_t n_out, const uint32_t *__restrict__ in, QM31 f, uint32_t *__restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    stq(out + 4 * j, qm31_add(ldq(in + 8 * j), qm31_mul(ldq(in + 8 * j + 4), f)));
}

// the same two folds over `ncols` columns at once (blockIdx.y = column): the OODS samples of all
// columns of a commitment share the point, so they share the factors and the launches
__global__ void p_fold_m31_batch_kernel(size_t n_out, const uint32_t *__restrict__ in, size_t in_stride, QM31 f,
                                        uint32_t *__restrict__ out, size_t out_stride)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    const uint32_t *src = in + (size_t)blockIdx.y * in_stride;
    QM31 r = qm31_mul_m31(f, src[2 * j + 1]);
    r.a = m31_add(r.a, src[2 * j]);
    stq(out + (size_t)blockIdx.y * out_stride + 4 * j, r);
}
__global__ void p_fold_qm31_batch_kernel(size_t n_out, const uint32_t *__restrict__ in, size_t in_stride, QM31 f,
                                         uint32_t *__restrict__ out, size_t out_stride)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    const uint32_t *src = in + (size_t)blockIdx.y * in_stride;
    stq(out + (size_t)blockIdx.y * out_stride + 4 * j, qm31_add(ldq(src + 8 * j), qm31_mul(ldq(src + 8 * j + 4), f)));
}

// ------------------------------------------------------------- device-side FRI commit channel
// fri_layer_commit (fri/commit.simf:34-46) on the device: digest <- H(digest || root), alpha <-
// draw_qm31.  state = 8 STORED digest words + the draw counter.  One lane: the channel is a
// dependent chain of two to three compressions per layer; running it here removes the host round
// trip (root download, hashlib, alpha upload) between a layer's Merkle tree and its fold.
template <int HF>
__global__ void p_channel_fri_layer_kernel(uint32_t *__restrict__ state, const uint32_t *__restrict__ root,
                                           uint32_t *__restrict__ alpha_out, uint32_t *__restrict__ root_out)
{
    if (blockIdx.x || threadIdx.x) return;
    Channel<HF> ch;
    for (int i = 0; i < 8; i++) ch.dig.v[i] = Hasher<HF>::native(state[i]);
    ch.ctr = state[8];
    uint32_t r[8];
    for (int i = 0; i < 8; i++) { r[i] = root[i]; root_out[i] = r[i]; }
    ch.mix(r);
    QM31 a;
    ch.draw_qm31(a);
    stq(alpha_out, a);
    for (int i = 0; i < 8; i++) state[i] = Hasher<HF>::native(ch.dig.v[i]);
    state[8] = ch.ctr;
}

// ------------------------------------------------------------------------------ quotients
struct QuotArgs {
    QM31 px, py, p2x, p2y, a1, c1, a2, c2, alpha16;
};

__device__ __forceinline__ CM31 deep_den_inv(QM31 sx, QM31 sy, uint32_t x, uint32_t y)
{
    // deep/quotients.simf:15-22
    CM31 dx = cm31_sub_m31(q_re(sx), x), dy = cm31_sub_m31(q_re(sy), y);
    CM31 d = cm31_sub(cm31_mul(dx, q_im(sy)), cm31_mul(dy, q_im(sx)));
    CM31 inv;
    cm31_inv(d, inv);
    return inv;
}

// One thread = the storage pair (2h, 2h+1): the two positions share x and have y, -y.  The column
// sums are open 64-bit multiply-accumulates (one v_mad_u64_u32 per product, folded every third
// product, reduced once per word); the four DEEP denominators (two sample points x two positions)
// share ONE M31 inversion.  All operands are canonical field elements, so every re-association gives
// the bits of the straightforward evaluation (which tools/stwo_prover.py performs).
__global__ void __launch_bounds__(256)
p_quotients_kernel(uint32_t lde_log, uint32_t n_cols, const uint32_t *__restrict__ trace_lde,
                   const uint32_t *__restrict__ cp_lde, uint32_t cp_log, const uint32_t *__restrict__ hx,
                   const uint32_t *__restrict__ hy, const uint32_t *__restrict__ bcoef, QuotArgs q,
                   uint32_t *__restrict__ out)
{
    const size_t size = (size_t)1 << lde_log;
    const size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (size >> 1)) return;
    const uint32_t x = hx[h], y0 = hy[h];
    const uint32_t ys[2] = {y0, m31_sub_c(0, y0)};
    // s[batch][position] = sum_k b_k v_k
    // (pairs = false: a column that depends on x only holds ONE value per storage pair, stride size / 2)
    auto dot = [&](const uint32_t *lde, const uint32_t *coef, uint32_t count, bool pairs, QM31 (&res)[2]) {
        uint64_t acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        uint32_t open = 0;
        for (uint32_t k = 0; k < count; k++) {
            uint2 v;
            if (pairs) v = *reinterpret_cast<const uint2 *>(lde + (size_t)k * size + 2 * h);
            else v.x = v.y = lde[(size_t)k * (size >> 1) + h];
            const uint4 c = *reinterpret_cast<const uint4 *>(coef + 4 * k);
            acc[0][0] = m31_mac(acc[0][0], c.x, v.x); acc[0][1] = m31_mac(acc[0][1], c.y, v.x);
            acc[0][2] = m31_mac(acc[0][2], c.z, v.x); acc[0][3] = m31_mac(acc[0][3], c.w, v.x);
            if (pairs) {  // (one value per pair: the second sum is the first)
                acc[1][0] = m31_mac(acc[1][0], c.x, v.y); acc[1][1] = m31_mac(acc[1][1], c.y, v.y);
                acc[1][2] = m31_mac(acc[1][2], c.z, v.y); acc[1][3] = m31_mac(acc[1][3], c.w, v.y);
            }
            if (++open == 3) {
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int w = 0; w < 4; w++) acc[p][w] = m31_fold62(acc[p][w]);
                open = 0;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++)
            res[p] = QM31{m31_red64(acc[p][0]), m31_red64(acc[p][1]), m31_red64(acc[p][2]), m31_red64(acc[p][3])};
        if (!pairs) res[1] = res[0];
    };
    QM31 s1[2], s2[2];
    dot(trace_lde, bcoef, n_cols, true, s1);
    dot(cp_lde, bcoef + 4 * n_cols, kCp, cp_log == lde_log, s2);
    // deep_quotient_denominator_inverse (deep/quotients.simf:15-22) for (P, 2P) x (y, -y)
    CM31 d[4];
    uint32_t nrm[4], pre[4];
    uint32_t run = 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const QM31 sx = j < 2 ? q.px : q.p2x, sy = j < 2 ? q.py : q.p2y;
        const CM31 dx = cm31_sub_m31(q_re(sx), x), dy = cm31_sub_m31(q_re(sy), ys[j & 1]);
        d[j] = cm31_sub(cm31_mul(dx, q_im(sy)), cm31_mul(dy, q_im(sx)));
        nrm[j] = m31_add(m31_sqr(d[j].a), m31_sqr(d[j].b));
        pre[j] = run;
        run = m31_mul(run, nrm[j] ? nrm[j] : 1u);  // a zero norm inverts to 0, as cm31_inv gives
    }
    uint32_t inv;
    m31_inv(run, inv);
    CM31 dinv[4];
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const uint32_t ni = nrm[j] ? m31_mul(inv, pre[j]) : 0u;
        inv = m31_mul(inv, nrm[j] ? nrm[j] : 1u);
        dinv[j] = cm31_mul_m31(CM31{d[j].a, m31_neg(d[j].b)}, ni);
    }
    const QM31 a1y = qm31_mul_m31(q.a1, y0), a2y = qm31_mul_m31(q.a2, y0);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        // a * (-y) = -(a * y)
        const QM31 t1 = p ? qm31_sub(q.c1, a1y) : qm31_add(a1y, q.c1);
        const QM31 t2 = p ? qm31_sub(q.c2, a2y) : qm31_add(a2y, q.c2);
        const QM31 b1 = qm31_mul_cm31(qm31_sub(s1[p], t1), dinv[p]);
        const QM31 b2 = qm31_mul_cm31(qm31_sub(s2[p], t2), dinv[2 + p]);
        stq(out + 4 * (2 * h + p), qm31_add(qm31_mul(b1, q.alpha16), b2));
    }
}

// ------------------------------------------------------------------------------- fri fold
__global__ void p_fri_fold_kernel(size_t n_out, const uint32_t *__restrict__ in, const uint32_t *__restrict__ cinv,
                                  QM31 alpha, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const QM31 v0 = ldq(in + 8 * i), v1 = ldq(in + 8 * i + 4);
    const QM31 f0 = qm31_add(v0, v1);
    const QM31 f1 = qm31_mul_m31(qm31_sub(v0, v1), cinv[i]);
    stq(out + 4 * i, qm31_add(f0, qm31_mul(f1, alpha)));
}

// the same fold with alpha read from device memory (written by p_channel_fri_layer_kernel)
__global__ void p_fri_fold_dev_kernel(size_t n_out, const uint32_t *__restrict__ in, const uint32_t *__restrict__ cinv,
                                      const uint32_t *__restrict__ alpha_dev, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const QM31 alpha = ldq(alpha_dev);
    const QM31 v0 = ldq(in + 8 * i), v1 = ldq(in + 8 * i + 4);
    const QM31 f0 = qm31_add(v0, v1);
    const QM31 f1 = qm31_mul_m31(qm31_sub(v0, v1), cinv[i]);
    stq(out + 4 * i, qm31_add(f0, qm31_mul(f1, alpha)));
}

// ------------------------------------------------------------------------------------ pow
template <int HF>
__global__ void p_pow_kernel(Dig digest_native, uint64_t target, uint64_t start, uint64_t count,
                             unsigned long long *__restrict__ best)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t nonce = start + t;
    const uint32_t m[2] = {(uint32_t)(nonce >> 32), (uint32_t)nonce};
    uint32_t d[8];
    Hasher<HF>::template block<true, 8, 2>(digest_native.v, m, d);
    if (Hasher<HF>::pow_value(d) < target) atomicMin(best, (unsigned long long)nonce);
}

}  // namespace ss

// =============================================================================== C ABI
static QM31 q4(const uint32_t v[4]) { return QM31{v[0], v[1], v[2], v[3]}; }

extern "C" int ss_p_trace(ss_ctx *ctx, uint32_t n_log, uint32_t n_cols, uint32_t seed_term, uint32_t *cols_out,
                          void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!cols_out || n_log > 26 || !n_cols) return ss_internal_set_err(SS_ERR_ARG, "ss_p_trace: bad argument");
    const uint32_t n = 1u << n_log;
    hipLaunchKernelGGL(p_trace_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, n_cols,
                       seed_term % M31_P, cols_out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_twiddles(ss_ctx *ctx, uint32_t m, uint32_t *tw_out, uint32_t *itw_out, uint32_t *hx_out,
                             void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!tw_out || !itw_out || m < 1 || m > 28) return ss_internal_set_err(SS_ERR_ARG, "ss_p_twiddles: bad argument");
    hipLaunchKernelGGL(p_twiddles_kernel, dim3(blocks_for(1u << (m - 1))), dim3(256), 0, (hipStream_t)stream, m,
                       tw_out, itw_out, hx_out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_fft(ss_ctx *ctx, uint32_t m, uint32_t ncols, uint32_t *data, const uint32_t *tw, int inverse,
                        void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!data || !tw || m < 1 || m > 28 || !ncols) return ss_internal_set_err(SS_ERR_ARG, "ss_p_fft: bad argument");
    // 2^-m mod P = 2^(31-m) because 2^31 == 1
    const uint32_t scale = (1u << ((31 - (m % 31)) % 31)) % M31_P;
    if (m >= 13) {
        // ceil(m / 8) LDS-staged passes of 5..8 layers each; the lowest pass is the contiguous one
        const uint32_t k = (m + 7) / 8, base = m / k, extra = m % k;
        uint32_t lo[4], nb[4];
        for (uint32_t i = 0, at = 0; i < k; i++) { lo[i] = at; nb[i] = base + (i < extra ? 1 : 0); at += nb[i]; }
        uint32_t cpb_log = 0;  // columns per block of the contiguous pass: 8, 4, 2 or 1
        while (cpb_log < 3 && ncols % (2u << cpb_log) == 0) cpb_log++;
        for (uint32_t s = 0; s < k; s++) {
            const uint32_t i = inverse ? s : k - 1 - s;
            if (i == 0)
                launch_contig_pass(m, nb[i], cpb_log, ncols, data, tw, inverse, scale, (hipStream_t)stream);
            else
                launch_strided_pass(m, lo[i], nb[i], ncols, data, tw, inverse, scale, nullptr, 0u, (hipStream_t)stream);
        }
        P_TRY(hipGetLastError());
        return SS_OK;
    }
    const dim3 grid(blocks_for(1u << (m - 1)), ncols);
    if (inverse) {
        for (uint32_t i = 0; i < m; i++)
            hipLaunchKernelGGL(p_fft_layer_kernel, grid, dim3(256), 0, (hipStream_t)stream, m, i, data, tw, 1,
                               i + 1 == m ? scale : 1u);
    } else {
        for (uint32_t i = m; i-- > 0;)
            hipLaunchKernelGGL(p_fft_layer_kernel, grid, dim3(256), 0, (hipStream_t)stream, m, i, data, tw, 0, 1u);
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

// Evaluations on the canonic coset of log size m of ncols polynomials given by 2^k coefficients each (k <= m):
// ss_p_fft of the zero-extended array, without materialising the zeros (see p_fft_pass_kernel).
extern "C" int ss_p_lde(ss_ctx *ctx, uint32_t k, uint32_t m, uint32_t ncols, const uint32_t *coefs, uint32_t *out,
                        const uint32_t *tw, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!coefs || !out || !tw || m < 1 || m > 28 || k > m || !ncols) return ss_internal_set_err(SS_ERR_ARG, "ss_p_lde: bad argument");
    const uint32_t z = m - k;
    const uint32_t np = (m + 7) / 8, base = m / np, extra = m % np;
    uint32_t lo[4], nb[4];
    for (uint32_t i = 0, at = 0; i < np; i++) { lo[i] = at; nb[i] = base + (i < extra ? 1 : 0); at += nb[i]; }
    // the shortcut needs a strided top pass whose rows are whole runs of coefficients: z <= nb and k >= lo of that pass
    if (m < 13 || np < 2 || z == 0 || z > nb[np - 1]) {
        for (uint32_t c = 0; c < ncols; c++) {
            P_TRY(hipMemcpyAsync(out + ((size_t)c << m), coefs + ((size_t)c << k), (size_t)4 << k, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
            if (z) P_TRY(hipMemsetAsync(out + ((size_t)c << m) + ((size_t)1 << k), 0, ((size_t)4 << m) - ((size_t)4 << k), (hipStream_t)stream));
        }
        return ss_p_fft(ctx, m, ncols, out, tw, 0, stream);
    }
    uint32_t cpb_log = 0;
    while (cpb_log < 3 && ncols % (2u << cpb_log) == 0) cpb_log++;
    for (uint32_t s = 0; s < np; s++) {
        const uint32_t i = np - 1 - s;
        if (i == 0)
            launch_contig_pass(m, nb[i], cpb_log, ncols, out, tw, 0, 1u, (hipStream_t)stream);
        else
            launch_strided_pass(m, lo[i], nb[i], ncols, out, tw, 0, 1u, i == np - 1 ? coefs : nullptr, i == np - 1 ? z : 0u,
                                (hipStream_t)stream);
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_hash_rows(ss_ctx *ctx, uint32_t hash, size_t n, uint32_t w, const uint32_t *cols,
                              size_t col_stride, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!cols || !out || !n || !w || hash > 1) return ss_internal_set_err(SS_ERR_ARG, "ss_p_hash_rows: bad argument");
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, w, cols, col_stride, out);
    };
    if (hash) {
        if (w == 4) launch(p_hash_rows_kernel<1, 4>);
        else if (w == 16) launch(p_hash_rows_kernel<1, 16>);
        else launch(p_hash_rows_kernel<1, 0>);
    } else {
        if (w == 4) launch(p_hash_rows_kernel<0, 4>);
        else if (w == 16) launch(p_hash_rows_kernel<0, 16>);
        else launch(p_hash_rows_kernel<0, 0>);
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_hash_qm31(ss_ctx *ctx, uint32_t hash, size_t n, const uint32_t *vals, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!vals || !out || !n || hash > 1) return ss_internal_set_err(SS_ERR_ARG, "ss_p_hash_qm31: bad argument");
    if (hash)
        hipLaunchKernelGGL(p_hash_qm31_kernel<1>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, vals, out);
    else
        hipLaunchKernelGGL(p_hash_qm31_kernel<0>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, vals, out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_merkle(ss_ctx *ctx, uint32_t hash, size_t n_leaves, uint32_t *levels, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!levels || !n_leaves || (n_leaves & (n_leaves - 1)) || hash > 1)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_merkle: bad argument");
    size_t off = 0;
    for (size_t n = n_leaves; n > 1; n >>= 1) {
        if (n <= 512) {  // the rest of the tree in one launch
            if (hash) hipLaunchKernelGGL(p_merkle_tail_kernel<1>, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint32_t)n, levels + off * 8);
            else hipLaunchKernelGGL(p_merkle_tail_kernel<0>, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint32_t)n, levels + off * 8);
            break;
        }
        const size_t parents = n >> 1;
        uint32_t *children = levels + off * 8, *out = levels + (off + n) * 8;
        if (hash)
            hipLaunchKernelGGL(p_merkle_level_kernel<1>, dim3(blocks_for(parents)), dim3(256), 0, (hipStream_t)stream,
                               parents, children, out);
        else
            hipLaunchKernelGGL(p_merkle_level_kernel<0>, dim3(blocks_for(parents)), dim3(256), 0, (hipStream_t)stream,
                               parents, children, out);
        off += n;
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_merkle_dup(ss_ctx *ctx, uint32_t hash, size_t n, const uint32_t *leaf, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!leaf || !out || !n || hash > 1) return ss_internal_set_err(SS_ERR_ARG, "ss_p_merkle_dup: bad argument");
    if (hash)
        hipLaunchKernelGGL(p_merkle_dup_kernel<1>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, leaf, out);
    else
        hipLaunchKernelGGL(p_merkle_dup_kernel<0>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, leaf, out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_composition(ss_ctx *ctx, uint32_t n_log, uint32_t n_cols, const uint32_t *ev, const uint32_t *hx_c,
                                const uint32_t alpha[4], uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!ev || !hx_c || !alpha || !out || n_log < 1 || n_log > 26)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_composition: bad argument");
    hipLaunchKernelGGL(p_composition_kernel, dim3(blocks_for(1u << (n_log + 1))), dim3(256), 0, (hipStream_t)stream,
                       n_log, n_cols, ev, hx_c, q4(alpha), out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

// factors: m QM31 values on the HOST (y, x, pi(x), pi^2(x), ...), bit 0 first
extern "C" int ss_p_eval_at_point(ss_ctx *ctx, uint32_t m, const uint32_t *coeffs, const uint32_t *factors_host,
                                  uint32_t *scratch, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!coeffs || !factors_host || !scratch || !out || m < 1 || m > 28)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_eval_at_point: bad argument");
    hipStream_t s = (hipStream_t)stream;
    size_t n = (size_t)1 << (m - 1);
    uint32_t *a = scratch, *b = scratch + 4 * n;  // ping-pong: a holds n QM31, b n/2
    hipLaunchKernelGGL(p_fold_m31_kernel, dim3(blocks_for(n)), dim3(256), 0, s, n, coeffs, q4(factors_host),
                       m == 1 ? out : a);
    uint32_t *src = a, *dst = b;
    for (uint32_t lvl = 1; lvl < m; lvl++) {
        n >>= 1;
        hipLaunchKernelGGL(p_fold_qm31_kernel, dim3(blocks_for(n)), dim3(256), 0, s, n, src,
                           q4(factors_host + 4 * lvl), lvl + 1 == m ? out : dst);
        uint32_t *t = src; src = dst; dst = t;
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

// all `ncols` columns (stride col_stride words) at one point: m launches instead of m per column;
// scratch: ncols * 3 * 2^m words; out[ncols][4]
extern "C" int ss_p_eval_at_point_batch(ss_ctx *ctx, uint32_t m, uint32_t ncols, const uint32_t *coeffs,
                                        size_t col_stride, const uint32_t *factors_host, uint32_t *scratch,
                                        uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!coeffs || !factors_host || !scratch || !out || m < 1 || m > 28 || !ncols || ncols > 65535)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_eval_at_point_batch: bad argument");
    hipStream_t s = (hipStream_t)stream;
    size_t n = (size_t)1 << (m - 1);
    const size_t sstride = (size_t)3 << m;  // per column: a holds n QM31 (4n words), b n/2 (2n words)
    uint32_t *a = scratch, *b = scratch + 4 * n;
    hipLaunchKernelGGL(p_fold_m31_batch_kernel, dim3(blocks_for(n), ncols), dim3(256), 0, s, n, coeffs, col_stride,
                       q4(factors_host), m == 1 ? out : a, m == 1 ? (size_t)4 : sstride);
    uint32_t *src = a, *dst = b;
    for (uint32_t lvl = 1; lvl < m; lvl++) {
        n >>= 1;
        const bool last = lvl + 1 == m;
        hipLaunchKernelGGL(p_fold_qm31_batch_kernel, dim3(blocks_for(n), ncols), dim3(256), 0, s, n, src, sstride,
                           q4(factors_host + 4 * lvl), last ? out : dst, last ? (size_t)4 : sstride);
        uint32_t *t = src; src = dst; dst = t;
    }
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_channel_fri_layer(ss_ctx *ctx, uint32_t hash, uint32_t *state_dev, const uint32_t *root_dev,
                                      uint32_t *alpha_out_dev, uint32_t *root_out_dev, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!state_dev || !root_dev || !alpha_out_dev || !root_out_dev || hash > 1)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_channel_fri_layer: bad argument");
    if (hash)
        hipLaunchKernelGGL(p_channel_fri_layer_kernel<1>, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, root_dev,
                           alpha_out_dev, root_out_dev);
    else
        hipLaunchKernelGGL(p_channel_fri_layer_kernel<0>, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, root_dev,
                           alpha_out_dev, root_out_dev);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_fri_fold_dev(ss_ctx *ctx, size_t n_out, const uint32_t *in, const uint32_t *coord_inv,
                                 const uint32_t *alpha_dev, uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!in || !coord_inv || !alpha_dev || !out || !n_out)
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_fri_fold_dev: bad argument");
    hipLaunchKernelGGL(p_fri_fold_dev_kernel, dim3(blocks_for(n_out)), dim3(256), 0, (hipStream_t)stream, n_out, in,
                       coord_inv, alpha_dev, out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_quotients(ss_ctx *ctx, uint32_t lde_log, uint32_t n_cols, const uint32_t *trace_lde,
                              const uint32_t *cp_lde, uint32_t cp_log, const uint32_t *hx_hy, const uint32_t *bcoef,
                              const uint32_t p[8], const uint32_t p2[8], const uint32_t sums_alpha16[20],
                              uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!trace_lde || !cp_lde || !hx_hy || !bcoef || !p || !p2 || !sums_alpha16 || !out || lde_log < 2 || lde_log > 28 ||
        (cp_log != lde_log && cp_log + 1 != lde_log))
        return ss_internal_set_err(SS_ERR_ARG, "ss_p_quotients: bad argument");
    QuotArgs q;
    q.px = q4(p); q.py = q4(p + 4); q.p2x = q4(p2); q.p2y = q4(p2 + 4);
    q.a1 = q4(sums_alpha16); q.c1 = q4(sums_alpha16 + 4); q.a2 = q4(sums_alpha16 + 8); q.c2 = q4(sums_alpha16 + 12);
    q.alpha16 = q4(sums_alpha16 + 16);
    const size_t half = (size_t)1 << (lde_log - 1);
    hipLaunchKernelGGL(p_quotients_kernel, dim3(blocks_for(half)), dim3(256), 0, (hipStream_t)stream,
                       lde_log, n_cols, trace_lde, cp_lde, cp_log, hx_hy, hx_hy + half, bcoef, q, out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_fri_fold(ss_ctx *ctx, size_t n_out, const uint32_t *in, const uint32_t *coord_inv,
                             const uint32_t alpha[4], uint32_t *out, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!in || !coord_inv || !alpha || !out || !n_out) return ss_internal_set_err(SS_ERR_ARG, "ss_p_fri_fold: bad argument");
    hipLaunchKernelGGL(p_fri_fold_kernel, dim3(blocks_for(n_out)), dim3(256), 0, (hipStream_t)stream, n_out, in,
                       coord_inv, q4(alpha), out);
    P_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_p_pow(ss_ctx *ctx, uint32_t hash, const uint32_t digest[8], uint64_t target, uint64_t start,
                        uint64_t count, uint64_t *nonce_out_dev, void *stream)
{
    SS_DEVICE_GUARD(ctx);
    if (!digest || !nonce_out_dev || !count || hash > 1) return ss_internal_set_err(SS_ERR_ARG, "ss_p_pow: bad argument");
    hipStream_t s = (hipStream_t)stream;
    P_TRY(hipMemsetAsync(nonce_out_dev, 0xff, 8, s));
    Dig d;
    for (int i = 0; i < 8; i++) d.v[i] = hash ? __builtin_bswap32(digest[i]) : digest[i];
    if (hash)
        hipLaunchKernelGGL(p_pow_kernel<1>, dim3(blocks_for(count)), dim3(256), 0, s, d, target, start, count,
                           (unsigned long long *)nonce_out_dev);
    else
        hipLaunchKernelGGL(p_pow_kernel<0>, dim3(blocks_for(count)), dim3(256), 0, s, d, target, start, count,
                           (unsigned long long *)nonce_out_dev);
    P_TRY(hipGetLastError());
    return SS_OK;
}
