// Is the SHA-256 "roof" of DESIGN.md the roof of ONE formulation?  (tools only; not part of the library.)
//
// The library's pair hash (csrc/ss_sha256.h) is 2 301 VALU instructions per sibling level: rotates as
// v_alignbit_b32 and the 4-term sums as v_add3_u32 -- both issue at half rate on gfx950 -- plus v_bitop3_b32
// for xor3 / Ch / Maj.  profiles/r01_valu_*.txt says a stream that contains any half-rate instruction costs
// ~4 cycles per instruction whatever the mix, so its cost is its instruction count.  This tool pins other
// formulations with inline assembly (so the compiler cannot fold them back) and measures the same register-only
// chain as tools/sha_bench.hip:
//   lib        the library's formulation (reference point)
//   lshl_or    every rotate as v_lshrrev_b32 + v_lshl_or_b32, every v_add3_u32 as two v_add_u32
//   fullrate   only instructions measured at full rate in isolation: rotates as two shifts whose OR is folded
//              into the following xor (v_bitop3_b32 over the six shifted words of a Sigma), two-input adds only
//   lib_x2     the library's formulation, two independent chains interleaved per lane (is the half-rate
//              penalty a dependency artefact that more ILP would hide?)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/sha_formulations.hip -o /tmp/sha_form && /tmp/sha_form
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../stark-symphony_amd/csrc/ss_sha256.h"

using namespace ss;

template <int N> __device__ __forceinline__ uint32_t a_lshr(uint32_t x)
{
    uint32_t r;
    asm("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "n"(N), "v"(x));
    return r;
}
template <int N> __device__ __forceinline__ uint32_t a_lshl(uint32_t x)
{
    uint32_t r;
    asm("v_lshlrev_b32 %0, %1, %2" : "=v"(r) : "n"(N), "v"(x));
    return r;
}
template <int N> __device__ __forceinline__ uint32_t a_lshl_or(uint32_t x, uint32_t o)  // (x << N) | o
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(N), "v"(o));
    return r;
}
__device__ __forceinline__ uint32_t a_add(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t a_addk(uint32_t x, uint32_t k)  // literal / SGPR operand allowed
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(x));
    return r;
}

struct Lib {
    static __device__ __forceinline__ uint32_t S0(uint32_t a) { return sha_S0(a); }
    static __device__ __forceinline__ uint32_t S1(uint32_t e) { return sha_S1(e); }
    static __device__ __forceinline__ uint32_t s0(uint32_t x) { return sha_s0(x); }
    static __device__ __forceinline__ uint32_t s1(uint32_t x) { return sha_s1(x); }
    static __device__ __forceinline__ uint32_t sum4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a + b + c + d; }
    static __device__ __forceinline__ uint32_t sum2(uint32_t a, uint32_t b) { return a + b; }
    static __device__ __forceinline__ uint32_t sumk(uint32_t a, uint32_t k) { return a + k; }
};

struct LshlOr {
    template <int N> static __device__ __forceinline__ uint32_t rot(uint32_t x) { return a_lshl_or<32 - N>(x, a_lshr<N>(x)); }
    static __device__ __forceinline__ uint32_t S0(uint32_t a) { return xor3(rot<2>(a), rot<13>(a), rot<22>(a)); }
    static __device__ __forceinline__ uint32_t S1(uint32_t e) { return xor3(rot<6>(e), rot<11>(e), rot<25>(e)); }
    static __device__ __forceinline__ uint32_t s0(uint32_t x) { return xor3(rot<7>(x), rot<18>(x), a_lshr<3>(x)); }
    static __device__ __forceinline__ uint32_t s1(uint32_t x) { return xor3(rot<17>(x), rot<19>(x), a_lshr<10>(x)); }
    static __device__ __forceinline__ uint32_t sum4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a_add(a_add(a, b), a_add(c, d)); }
    static __device__ __forceinline__ uint32_t sum2(uint32_t a, uint32_t b) { return a_add(a, b); }
    static __device__ __forceinline__ uint32_t sumk(uint32_t a, uint32_t k) { return a_addk(a, k); }
};

struct FullRate {
    // a rotate is two shifts with disjoint bits, so their OR is an XOR and folds into the Sigma's xor
    template <int A, int B, int C> static __device__ __forceinline__ uint32_t sig3(uint32_t x)
    {
        const uint32_t u = xor3(a_lshr<A>(x), a_lshl<32 - A>(x), a_lshr<B>(x));
        return xor3(u, a_lshl<32 - B>(x), xor3(a_lshr<C>(x), a_lshl<32 - C>(x), 0u));
    }
    template <int A, int B, int C> static __device__ __forceinline__ uint32_t sig2(uint32_t x)  // two rotates and a shift
    {
        const uint32_t u = xor3(a_lshr<A>(x), a_lshl<32 - A>(x), a_lshr<B>(x));
        return xor3(u, a_lshl<32 - B>(x), a_lshr<C>(x));
    }
    static __device__ __forceinline__ uint32_t S0(uint32_t a) { return sig3<2, 13, 22>(a); }
    static __device__ __forceinline__ uint32_t S1(uint32_t e) { return sig3<6, 11, 25>(e); }
    static __device__ __forceinline__ uint32_t s0(uint32_t x) { return sig2<7, 18, 3>(x); }
    static __device__ __forceinline__ uint32_t s1(uint32_t x) { return sig2<17, 19, 10>(x); }
    static __device__ __forceinline__ uint32_t sum4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a_add(a_add(a, b), a_add(c, d)); }
    static __device__ __forceinline__ uint32_t sum2(uint32_t a, uint32_t b) { return a_add(a, b); }
    static __device__ __forceinline__ uint32_t sumk(uint32_t a, uint32_t k) { return a_addk(a, k); }
};

template <class F> __device__ __forceinline__ void round_(uint32_t a, uint32_t b, uint32_t c, uint32_t &d, uint32_t e, uint32_t f,
                                                          uint32_t g, uint32_t &h, uint32_t wk, bool with_wk)
{
    const uint32_t s1 = F::S1(e), ch = sha_ch(e, f, g);
    const uint32_t t1 = with_wk ? F::sum4(h, s1, ch, wk) : F::sum2(F::sum2(h, s1), ch);
    const uint32_t t2 = F::sum2(F::S0(a), sha_maj(a, b, c));
    d = F::sum2(d, t1);
    h = F::sum2(t1, t2);
}

template <class F> __device__ __forceinline__ void compress(uint32_t (&st)[8], uint32_t (&w)[16], bool pad)
{
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = st[i];
#pragma unroll
    for (int r = 0; r < 64; r++) {
        uint32_t wk;
        // rotating register names instead of moving values: position p of the state at round r is v[(p - r) & 7]
        uint32_t &a = v[(0 - r) & 7], &b = v[(1 - r) & 7], &c = v[(2 - r) & 7], &d = v[(3 - r) & 7], &e = v[(4 - r) & 7],
                 &f = v[(5 - r) & 7], &g = v[(6 - r) & 7], &h = v[(7 - r) & 7];
        if (pad) {  // constant schedule: h + K + W is one add with a literal
            h = F::sumk(h, kPad64WK.k[r]);
            round_<F>(a, b, c, d, e, f, g, h, 0u, false);
        } else {
            if (r >= 16) w[r & 15] = F::sum4(w[r & 15], F::s0(w[(r + 1) & 15]), w[(r + 9) & 15], F::s1(w[(r + 14) & 15]));
            round_<F>(a, b, c, d, e, f, g, h, F::sumk(w[r & 15], kK.k[r]), true);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] += v[i];
}

template <class F, int CHAINS> __global__ void __launch_bounds__(256) chain_kernel(uint32_t iters, uint32_t *out)
{
    uint32_t node[CHAINS][8], sib[CHAINS][8];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) { node[c][j] = (t + c * 77777u) * 0x9E3779B1u + j; sib[c][j] = (t + c) ^ (0x85EBCA6Bu * (j + 1)); }
    uint32_t auth = t;
    for (uint32_t it = 0; it < iters; it++) {
        const bool right = auth & 1;
        uint32_t w[CHAINS][16];
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                w[c][j] = right ? sib[c][j] : node[c][j];
                w[c][8 + j] = right ? node[c][j] : sib[c][j];
            }
            sha_iv(node[c]);
        }
#pragma unroll
        for (int c = 0; c < CHAINS; c++) compress<F>(node[c], w[c], false);  // (independent: the scheduler interleaves them)
#pragma unroll
        for (int c = 0; c < CHAINS; c++) compress<F>(node[c], w[c], true);
        auth = (auth >> 1) | (auth << 31);
#pragma unroll
        for (int c = 0; c < CHAINS; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) sib[c][j] += node[c][(j + 3) & 7];
    }
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) x ^= node[c][j];
    out[t] = x;
}

template <class F, int CHAINS> static double run(const char *name, uint32_t iters, int blocks_per_cu, int cus, uint32_t *check)
{
    const int grid = cus * blocks_per_cu;
    uint32_t *out;
    (void)hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    chain_kernel<F, CHAINS><<<grid, 256>>>(iters / 8 + 1, out);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(a);
        chain_kernel<F, CHAINS><<<grid, 256>>>(iters, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    (void)hipMemcpy(check, out, 4, hipMemcpyDeviceToHost);
    const double pairs = (double)grid * 256 * iters * CHAINS;
    printf("%-10s chains/lane %d  blocks/CU %2d  %8.3f ms  %7.2f G compressions/s  (lane 0 digest word %08x)\n", name, CHAINS,
           blocks_per_cu, best, 2 * pairs / best / 1e6, *check);
    (void)hipFree(out);
    return 2 * pairs / best / 1e6;
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? atoi(argv[1]) : 256;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; register-only chain of SHA-256 pair hashes, %u levels per lane\n", prop.gcnArchName, cus, iters);
    uint32_t c0, c1, c2, c3;
    for (int bpc : {4, 8}) {
        run<Lib, 1>("lib", iters, bpc, cus, &c0);
        run<LshlOr, 1>("lshl_or", iters, bpc, cus, &c1);
        run<FullRate, 1>("fullrate", iters, bpc, cus, &c2);
        if (c0 != c1 || c0 != c2) { printf("MISMATCH between formulations\n"); return 1; }
    }
    for (int bpc : {2, 4}) run<Lib, 2>("lib_x2", iters, bpc, cus, &c3);
    return 0;
}
