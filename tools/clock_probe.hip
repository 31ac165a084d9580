// Effective shader clock and block residency while the SHA-256 chain runs (tools only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/clock_probe.hip -o build/clock_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../stark-symphony_amd/csrc/ss_sha256.h"
using namespace ss;

__global__ void __launch_bounds__(256) probe(uint32_t iters, uint32_t *out, unsigned long long *stamps)
{
    uint32_t node[8], sib[8];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < 8; j++) { node[j] = t * 0x9E3779B1u + j; sib[j] = t ^ (0x85EBCA6Bu * (j + 1)); }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t w[16];
        for (int j = 0; j < 8; j++) { w[j] = node[j]; w[8 + j] = sib[j]; }
        sha_iv(node);
        sha256_compress(node, w);
        sha256_compress_pad64(node);
        for (int j = 0; j < 8; j++) sib[j] += node[(j + 3) & 7];
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
    for (int j = 0; j < 8; j++) x ^= node[j];
    out[t] = x;
    if (threadIdx.x == 0) {
        stamps[4 * blockIdx.x] = c1 - c0;
        stamps[4 * blockIdx.x + 1] = r0;
        stamps[4 * blockIdx.x + 2] = r1;
        stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const uint32_t iters = 1024;
    for (int bpc : {1, 2, 4, 8}) {
        int grid = prop.multiProcessorCount * bpc;
        uint32_t *out;
        unsigned long long *st;
        hipMalloc(&out, (size_t)grid * 256 * 4);
        hipMalloc(&st, (size_t)grid * 32);
        for (int rep = 0; rep < 2; rep++) probe<<<grid, 256>>>(iters, out, st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(4 * grid);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<std::pair<unsigned long long, int>> ev;
        for (int b = 0; b < grid; b++) {
            mhz.push_back((double)h[4 * b] / (double)(h[4 * b + 2] - h[4 * b + 1]) * 100.0);
            t0 = std::min(t0, h[4 * b + 1]);
            t1 = std::max(t1, h[4 * b + 2]);
            ev.push_back({h[4 * b + 1], +1});
            ev.push_back({h[4 * b + 2], -1});
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0, peak = 0;
        for (auto &e : ev) { cur += e.second; peak = std::max(peak, cur); }
        std::sort(mhz.begin(), mhz.end());
        printf("blocks/CU %d: clock median %.0f MHz; %.0f cycles per pair-hash per wave; span %.3f ms, one block %.3f ms, "
               "peak concurrent blocks %d (= %.2f per CU)\n",
               bpc, mhz[grid / 2], (double)h[0] / iters, (t1 - t0) / 1e5, (h[2] - h[1]) / 1e5, peak,
               (double)peak / prop.multiProcessorCount);
        hipFree(out);
        hipFree(st);
    }
    return 0;
}
