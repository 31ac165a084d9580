// VALU issue-rate micro-benchmark for gfx950 (tools only): cycles per wave64 instruction for
// the integer ops SHA-256 is made of, at 1, 2 and 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/valu_bench.hip -o build/valu_bench
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define DEF_KERNEL(NAME, ASM)                                                                  \
    __global__ void __launch_bounds__(256) NAME(uint32_t iters, uint32_t *out)                 \
    {                                                                                          \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,        \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = blockIdx.x | 1, c = a0 ^ 0x55;  \
        for (uint32_t i = 0; i < iters; i++) {                                                 \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4),         \
                         "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));                       \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;    \
    }

#define OP8(op, tail)                                                                          \
    op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n"                 \
    op " %3, %3, " tail "\n" op " %4, %4, " tail "\n" op " %5, %5, " tail "\n"                 \
    op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"

DEF_KERNEL(k_add_u32, OP8("v_add_u32_e32", "%8"))
DEF_KERNEL(k_xor_b32, OP8("v_xor_b32_e32", "%8"))
DEF_KERNEL(k_lshrrev, OP8("v_lshrrev_b32_e32", "%8"))
DEF_KERNEL(k_alignbit, OP8("v_alignbit_b32", "%8, 7"))
DEF_KERNEL(k_alignbit_same, "v_alignbit_b32 %0, %0, %0, 7\nv_alignbit_b32 %1, %1, %1, 7\nv_alignbit_b32 %2, %2, %2, 7\nv_alignbit_b32 %3, %3, %3, 7\nv_alignbit_b32 %4, %4, %4, 7\nv_alignbit_b32 %5, %5, %5, 7\nv_alignbit_b32 %6, %6, %6, 7\nv_alignbit_b32 %7, %7, %7, 7\n")
DEF_KERNEL(k_add3, OP8("v_add3_u32", "%8, %9"))
DEF_KERNEL(k_bitop3, OP8("v_bitop3_b32", "%8, %9 bitop3:0x96"))
DEF_KERNEL(k_xad, OP8("v_xad_u32", "%8, %9"))
DEF_KERNEL(k_and_or, OP8("v_and_or_b32", "%8, %9"))
DEF_KERNEL(k_perm, OP8("v_perm_b32", "%8, %9"))
DEF_KERNEL(k_fma_f32, OP8("v_fma_f32", "%8, %9"))
DEF_KERNEL(k_add_f32, OP8("v_add_f32_e32", "%8"))
DEF_KERNEL(k_mul_lo, OP8("v_mul_lo_u32", "%8"))
DEF_KERNEL(k_mad_u32_u24, OP8("v_mad_u32_u24", "%8, %9"))
DEF_KERNEL(k_cndmask, OP8("v_cndmask_b32_e32", "%8, vcc"))

typedef void (*kern_t)(uint32_t, uint32_t *);

static void run(const char *name, kern_t k, int cus, uint32_t *out, double ops_per_iter)
{
    const uint32_t iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("%-16s", name);
    for (int bpc : {1, 2, 4, 8}) {  // 256-thread blocks per CU = waves per SIMD
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
        // wave-instructions per SIMD = bpc * iters * ops_per_iter ; cycles at 2.4 GHz
        const double cyc = best * 1e-3 * 2.4e9 / (bpc * (double)iters * ops_per_iter);
        printf("  %dw/SIMD %5.2f cyc/inst", bpc, cyc);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    printf("%s: cycles per wave64 instruction per SIMD assuming 2.4 GHz (lower = faster)\n", prop.gcnArchName);
    run("v_add_u32", k_add_u32, cus, out, 64);
    run("v_xor_b32", k_xor_b32, cus, out, 64);
    run("v_lshrrev_b32", k_lshrrev, cus, out, 64);
    run("v_alignbit_b32", k_alignbit, cus, out, 64);
    run("v_alignbit(x,x)", k_alignbit_same, cus, out, 64);
    run("v_add3_u32", k_add3, cus, out, 64);
    run("v_bitop3_b32", k_bitop3, cus, out, 64);
    run("v_xad_u32", k_xad, cus, out, 64);
    run("v_and_or_b32", k_and_or, cus, out, 64);
    run("v_perm_b32", k_perm, cus, out, 64);
    run("v_fma_f32", k_fma_f32, cus, out, 64);
    run("v_add_f32", k_add_f32, cus, out, 64);
    run("v_mul_lo_u32", k_mul_lo, cus, out, 64);
    run("v_mad_u32_u24", k_mad_u32_u24, cus, out, 64);
    run("v_cndmask_b32", k_cndmask, cus, out, 64);
    return 0;
}
