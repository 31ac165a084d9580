// SHA-256 calibration micro-benchmark for gfx950 (tools only; not part of the library).
//
// Measures the integer-ALU roof the Merkle kernels are bound by: every lane folds `iters`
// register-resident siblings into its node (sha256_pair = 1 full + 1 constant-schedule
// compression, exactly the inner step of stwo_merkle_kernel), with no memory traffic.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/sha_bench.hip -o /tmp/sha_bench
//   /tmp/sha_bench [iters] [blocks_per_cu] [threads]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../stark-symphony_amd/csrc/ss_hash.h"

using namespace ss;

template <int WAVES_PER_EU>
__global__ void __launch_bounds__(256, WAVES_PER_EU) chain_kernel(uint32_t iters, uint32_t *out)
{
    uint32_t node[8], sib[8];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) { node[j] = t * 0x9E3779B1u + j; sib[j] = t ^ (0x85EBCA6Bu * (j + 1)); }
    uint32_t auth = t;
    for (uint32_t it = 0; it < iters; it++) {
        const bool right = auth & 1;
        uint32_t w[16];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            w[j] = right ? sib[j] : node[j];
            w[8 + j] = right ? node[j] : sib[j];
        }
        sha_iv(node);
        sha256_compress(node, w);
        sha256_compress_pad64(node);
        auth = (auth >> 1) | (auth << 31);
#pragma unroll
        for (int j = 0; j < 8; j++) sib[j] += node[(j + 3) & 7];
    }
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) x ^= node[j];
    out[t] = x;
}

// A/B for the north-star's "message schedule staged in LDS": the same chain with the 16-word
// rolling schedule window of the first compression kept in LDS (column per lane, conflict free)
// instead of VGPRs.  Every round then costs one ds_read_b32, every schedule update three reads
// and one write, on top of the same VALU instructions -- the loop's cost is its instruction count,
// so this can only lose; measured here so the claim has a number (profiles/r02_sha_lds_ab.txt).
__global__ void __launch_bounds__(256) chain_kernel_lds(uint32_t iters, uint32_t *out)
{
    // volatile: without it the compiler forwards every store to the later loads of the same lane
    // (no barrier, so it may) and only the ds_write traffic remains -- measured: no difference
    __shared__ volatile uint32_t sw[16][256];
    uint32_t node[8], sib[8];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, l = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) { node[j] = t * 0x9E3779B1u + j; sib[j] = t ^ (0x85EBCA6Bu * (j + 1)); }
    uint32_t auth = t;
    for (uint32_t it = 0; it < iters; it++) {
        const bool right = auth & 1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            sw[j][l] = right ? sib[j] : node[j];
            sw[8 + j][l] = right ? node[j] : sib[j];
        }
        sha_iv(node);
        uint32_t a = node[0], b = node[1], c = node[2], d = node[3], e = node[4], f = node[5], g = node[6],
                 h = node[7];
#pragma unroll
        for (int r = 0; r < 64; r++) {
            if (r >= 16)
                sw[r & 15][l] = sw[r & 15][l] + sha_s0(sw[(r + 1) & 15][l]) + sw[(r + 9) & 15][l] + sha_s1(sw[(r + 14) & 15][l]);
            const uint32_t t1 = h + sha_S1(e) + sha_ch(e, f, g) + (kK.k[r] + sw[r & 15][l]);
            const uint32_t t2 = sha_S0(a) + sha_maj(a, b, c);
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        node[0] += a; node[1] += b; node[2] += c; node[3] += d; node[4] += e; node[5] += f; node[6] += g;
        node[7] += h;
        sha256_compress_pad64(node);
        auth = (auth >> 1) | (auth << 31);
#pragma unroll
        for (int j = 0; j < 8; j++) sib[j] += node[(j + 3) & 7];
    }
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) x ^= node[j];
    out[t] = x;
}

static void run_lds(uint32_t iters, int blocks_per_cu, int cus)
{
    const int grid = cus * blocks_per_cu;
    uint32_t *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    chain_kernel_lds<<<grid, 256>>>(iters / 8 + 1, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a);
        chain_kernel_lds<<<grid, 256>>>(iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double pairs = (double)grid * 256 * iters;
    printf("%-28s blocks/CU %2d  iters %5u  %8.3f ms  %7.2f G pair-hashes/s  %7.2f G compressions/s\n",
           "schedule window in LDS", blocks_per_cu, iters, best, pairs / best / 1e6, 2 * pairs / best / 1e6);
    hipFree(out);
}

// Blake2s-256 variant of the same chain: one compression per 64-byte node.
__global__ void __launch_bounds__(256) chain_kernel_b2s(uint32_t iters, uint32_t *out)
{
    uint32_t node[8], sib[8];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) { node[j] = t * 0x9E3779B1u + j; sib[j] = t ^ (0x85EBCA6Bu * (j + 1)); }
    uint32_t auth = t;
    for (uint32_t it = 0; it < iters; it++) {
        const bool right = auth & 1;
        uint32_t l[8], r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t s = Hasher<1>::native(sib[j]);
            l[j] = right ? s : node[j];
            r[j] = right ? node[j] : s;
        }
        Hasher<1>::pair<true>(l, r, node);
        auth = (auth >> 1) | (auth << 31);
#pragma unroll
        for (int j = 0; j < 8; j++) sib[j] += node[(j + 3) & 7];
    }
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) x ^= node[j];
    out[t] = x;
}

static void run_b2s(uint32_t iters, int blocks_per_cu, int cus)
{
    const int grid = cus * blocks_per_cu;
    uint32_t *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    chain_kernel_b2s<<<grid, 256>>>(iters / 8 + 1, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a);
        chain_kernel_b2s<<<grid, 256>>>(iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double pairs = (double)grid * 256 * iters;
    printf("%-28s blocks/CU %2d  iters %5u  %8.3f ms  %7.2f G pair-hashes/s = G compressions/s\n", "blake2s",
           blocks_per_cu, iters, best, pairs / best / 1e6);
    hipFree(out);
}

template <int W>
static void run(const char *name, uint32_t iters, int blocks_per_cu, int cus)
{
    const int grid = cus * blocks_per_cu;
    uint32_t *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    chain_kernel<W><<<grid, 256>>>(iters / 8 + 1, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(a);
        chain_kernel<W><<<grid, 256>>>(iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double pairs = (double)grid * 256 * iters;
    printf("%-28s blocks/CU %2d  iters %5u  %8.3f ms  %7.2f G pair-hashes/s  %7.2f G compressions/s\n", name,
           blocks_per_cu, iters, best, pairs / best / 1e6, 2 * pairs / best / 1e6);
    hipFree(out);
}

int main(int argc, char **argv)
{
    uint32_t iters = argc > 1 ? atoi(argv[1]) : 512;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    for (int bpc : {1, 2, 4, 8}) {
        run<1>("launch_bounds(256,1)", iters, bpc, cus);
    }
    for (int bpc : {4, 8}) {
        run<4>("launch_bounds(256,4)", iters, bpc, cus);
        run<6>("launch_bounds(256,6)", iters, bpc, cus);
        run<8>("launch_bounds(256,8)", iters, bpc, cus);
    }
    for (int bpc : {2, 4, 8}) run_lds(iters, bpc, cus);
    for (int bpc : {1, 2, 4, 8}) run_b2s(iters * 2, bpc, cus);
    return 0;
}
