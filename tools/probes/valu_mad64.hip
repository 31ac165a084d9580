// Issue rate of the 64-bit multiply-accumulate ops a lazily-reduced M31 dot product would use
// (tools only).   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/valu_mad64.hip -o build/valu_mad64
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x

#define DEF64(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t iters, uint64_t *out)                    \
    {                                                                                             \
        uint64_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,           \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                        \
        uint32_t b = blockIdx.x | 1, c = threadIdx.x ^ 0x55;                                      \
        for (uint32_t i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),  \
                         "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");                            \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }

#define MAD8                                                                                       \
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"                      \
    "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"                      \
    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"                      \
    "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
#define SHR8                                                                                       \
    "v_lshrrev_b64 %0, 31, %0\n v_lshrrev_b64 %1, 31, %1\n v_lshrrev_b64 %2, 31, %2\n"             \
    "v_lshrrev_b64 %3, 31, %3\n v_lshrrev_b64 %4, 31, %4\n v_lshrrev_b64 %5, 31, %5\n"             \
    "v_lshrrev_b64 %6, 31, %6\n v_lshrrev_b64 %7, 31, %7\n"

DEF64(k_mad64, MAD8)
DEF64(k_shr64, SHR8)

#define DEF32(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t iters, uint64_t *out)                    \
    {                                                                                             \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,           \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = blockIdx.x | 1;                    \
        for (uint32_t i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),  \
                         "+v"(a6), "+v"(a7) : "v"(b));                                            \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }
#define OP8(op)                                                                                    \
    op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n"                    \
    op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
DEF32(k_mul_hi, OP8("v_mul_hi_u32"))
DEF32(k_mul_lo, OP8("v_mul_lo_u32"))
// the pair a 32x32 -> 64 product costs today
DEF32(k_mul_lohi, "v_mul_lo_u32 %0, %1, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_lo_u32 %2, %3, %8\n"
                  "v_mul_hi_u32 %3, %3, %8\n v_mul_lo_u32 %4, %5, %8\n v_mul_hi_u32 %5, %5, %8\n"
                  "v_mul_lo_u32 %6, %7, %8\n v_mul_hi_u32 %7, %7, %8\n")

typedef void (*kern_t)(uint32_t, uint64_t *);

static void run(const char *name, kern_t k, int cus, uint64_t *out)
{
    const uint32_t iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("%-16s", name);
    for (int bpc : {1, 2, 4, 8}) {
        const int grid = cus * bpc;
        k<<<grid, 256>>>(10, out);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(a);
            k<<<grid, 256>>>(iters, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("  %dw/SIMD %5.2f cyc/inst", bpc, best * 1e-3 * 2.4e9 / (bpc * (double)iters * 64));
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint64_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    printf("%s: cycles per wave64 instruction per SIMD assuming 2.4 GHz\n", prop.gcnArchName);
    run("v_mad_u64_u32", k_mad64, cus, out);
    run("v_lshrrev_b64", k_shr64, cus, out);
    run("v_mul_hi_u32", k_mul_hi, cus, out);
    run("v_mul_lo_u32", k_mul_lo, cus, out);
    run("mul_lo+mul_hi", k_mul_lohi, cus, out);
    return 0;
}
