#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20"
#define K(NAME, BODY) __global__ void __launch_bounds__(256) NAME(uint32_t iters, uint32_t* out){ \
  for (uint32_t i=0;i<iters;i++){ asm volatile(REP8(BODY) ::: CLOB); } if (iters==0xffffffff) out[0]=1; }
#define H(d) "v_alignbit_b32 v" #d ", v8, v8, 7\n"
#define F(d) "v_bitop3_b32 v" #d ", v9, v14, v19 bitop3:0x96\n"
#define A(d) "v_add_u32 v" #d ", v9, v14\n"
#define T(d) "v_add3_u32 v" #d ", v9, v14, v19\n"
K(p_hfhf, H(0) F(1) H(2) F(3) H(4) F(5) H(6) F(7))
K(p_hhff, H(0) H(1) F(2) F(3) H(4) H(5) F(6) F(7))
K(p_h4f4, H(0) H(1) H(2) H(3) F(4) F(5) F(6) F(7))
K(p_hfff, H(0) F(1) F(2) F(3) H(4) F(5) F(6) F(7))
K(p_h6f2, H(0) H(1) H(2) H(3) H(4) H(5) F(6) F(7))
K(p_h2f6, H(0) H(1) F(2) F(3) F(4) F(5) F(6) F(7))
K(p_h1f7, H(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7))
K(p_f8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7))
K(p_h8, H(0) H(1) H(2) H(3) H(4) H(5) H(6) F(7))
K(p_hahA, H(0) A(1) H(2) A(3) H(4) A(5) H(6) A(7))
K(p_tata, T(0) A(1) T(2) A(3) T(4) A(5) T(6) A(7))
K(p_fafa, F(0) A(1) F(2) A(3) F(4) A(5) F(6) A(7))
typedef void (*kern_t)(uint32_t, uint32_t*);
static void run(const char* name, kern_t k, int cus, uint32_t* out){
  const uint32_t iters=2000; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); printf("%-10s cycles per 8 instrs:", name);
  for (int bpc : {1,2,4,8}) { int grid=cus*bpc; k<<<grid,256>>>(10,out); hipDeviceSynchronize(); float best=1e30f;
    for (int r=0;r<3;r++){ hipEventRecord(a); k<<<grid,256>>>(iters,out); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); best = ms<best?ms:best; }
    printf("  %dw %6.2f", bpc, best*1e-3*2.39e9/(bpc*(double)iters*8)); }
  printf("\n"); }
int main(){ hipDeviceProp_t p; hipGetDeviceProperties(&p,0); uint32_t* out; hipMalloc(&out, 4096); int c=p.multiProcessorCount;
  run("HFHFHFHF", p_hfhf,c,out); run("HHFFHHFF", p_hhff,c,out); run("HHHHFFFF", p_h4f4,c,out); run("HFFFHFFF", p_hfff,c,out);
  run("HHHHHHFF", p_h6f2,c,out); run("HHFFFFFF", p_h2f6,c,out); run("HFFFFFFF", p_h1f7,c,out); run("FFFFFFFF", p_f8,c,out); run("HHHHHHHF", p_h8,c,out);
  run("HAHAHAHA", p_hahA,c,out); run("TATATATA", p_tata,c,out); run("FAFAFAFA", p_fafa,c,out);
  return 0; }
