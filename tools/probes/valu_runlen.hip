// Does the VALU pay for switching between a half-rate (v_alignbit_b32) and a full-rate (v_bitop3_b32) instruction?
// Kernel kN issues N of the one, then N of the other, per loop iteration (eight rotating destination registers, no
// dependencies); cycles per instruction against the run length N.  hipcc --offload-arch=gfx950 valu_runlen.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7"
#define A(r) "v_alignbit_b32 v" #r ", v8, v8, 7\n"
#define B(r) "v_bitop3_b32 v" #r ", v9, v14, v19 bitop3:0x96\n"
#define A4 A(0) A(1) A(2) A(3)
#define B4 B(0) B(1) B(2) B(3)
#define A8 A4 A(4) A(5) A(6) A(7)
#define B8 B4 B(4) B(5) B(6) B(7)
#define X2(x) x x
#define X4(x) X2(x) X2(x)
#define X8(x) X4(x) X4(x)
#define X16(x) X8(x) X8(x)
#define KERNEL(name, body)                                                                        \
    __global__ void __launch_bounds__(256) name(uint32_t iters, uint32_t *out)                    \
    {                                                                                             \
        for (uint32_t i = 0; i < iters; i++) asm volatile(body ::: CLOB);                         \
        if (iters == 0xffffffff) out[0] = 1;                                                      \
    }
KERNEL(k4, A4 B4)
KERNEL(k8, A8 B8)
KERNEL(k16, X2(A8) X2(B8))
KERNEL(k32, X4(A8) X4(B8))
KERNEL(k64, X8(A8) X8(B8))
KERNEL(k128, X16(A8) X16(B8))

typedef void (*kern_t)(uint32_t, uint32_t*);
static void run(int n, kern_t k, int cus, uint32_t* out){
  const uint32_t iters=40000/n; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); printf("run length %3d: cycles per instr:", n);
  for (int bpc : {1,2,4,8}) { int grid=cus*bpc; k<<<grid,256>>>(10,out); hipDeviceSynchronize(); float best=1e30f;
    for (int r=0;r<3;r++){ hipEventRecord(a); k<<<grid,256>>>(iters,out); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); best = ms<best?ms:best; }
    printf("  %dw %5.2f", bpc, best*1e-3*2.39e9/(bpc*(double)iters*2*n)); }
  printf("   (ideal 3.27)\n"); }
int main(){ hipDeviceProp_t p; hipGetDeviceProperties(&p,0); uint32_t* out; hipMalloc(&out, 4096); int c=p.multiProcessorCount;
  run(4,k4,c,out); run(8,k8,c,out); run(16,k16,c,out); run(32,k32,c,out); run(64,k64,c,out); run(128,k128,c,out); return 0; }
