#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define K(NAME, BODY) __global__ void __launch_bounds__(256) NAME(uint32_t iters, uint32_t* out){ \
  for (uint32_t i=0;i<iters;i++){ asm volatile(REP8(BODY) ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40"); } \
  if (iters==0xffffffff) out[0]=1; }
// same bank: sources v8,v12,v16 (all index%4==0); dst rotating
K(bitop3_samebank, "v_bitop3_b32 v0, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v1, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v2, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v3, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v4, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v5, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v6, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v7, v8, v12, v16 bitop3:0x96\n")
K(bitop3_diffbank, "v_bitop3_b32 v0, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v1, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v2, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v3, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v4, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v5, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v6, v9, v14, v19 bitop3:0x96\nv_bitop3_b32 v7, v9, v14, v19 bitop3:0x96\n")
K(bitop3_twobank, "v_bitop3_b32 v0, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v1, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v2, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v3, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v4, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v5, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v6, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v7, v8, v12, v17 bitop3:0x96\n")
K(add3_samebank, "v_add3_u32 v0, v8, v12, v16\nv_add3_u32 v1, v8, v12, v16\nv_add3_u32 v2, v8, v12, v16\nv_add3_u32 v3, v8, v12, v16\nv_add3_u32 v4, v8, v12, v16\nv_add3_u32 v5, v8, v12, v16\nv_add3_u32 v6, v8, v12, v16\nv_add3_u32 v7, v8, v12, v16\n")
K(add3_diffbank, "v_add3_u32 v0, v9, v14, v19\nv_add3_u32 v1, v9, v14, v19\nv_add3_u32 v2, v9, v14, v19\nv_add3_u32 v3, v9, v14, v19\nv_add3_u32 v4, v9, v14, v19\nv_add3_u32 v5, v9, v14, v19\nv_add3_u32 v6, v9, v14, v19\nv_add3_u32 v7, v9, v14, v19\n")
K(add_samebank, "v_add_u32 v0, v8, v12\nv_add_u32 v1, v8, v12\nv_add_u32 v2, v8, v12\nv_add_u32 v3, v8, v12\nv_add_u32 v4, v8, v12\nv_add_u32 v5, v8, v12\nv_add_u32 v6, v8, v12\nv_add_u32 v7, v8, v12\n")
K(add_diffbank, "v_add_u32 v0, v9, v14\nv_add_u32 v1, v9, v14\nv_add_u32 v2, v9, v14\nv_add_u32 v3, v9, v14\nv_add_u32 v4, v9, v14\nv_add_u32 v5, v9, v14\nv_add_u32 v6, v9, v14\nv_add_u32 v7, v9, v14\n")
// dependent chain vs alternating half/full rate
K(mix_align_bitop, "v_alignbit_b32 v0, v8, v8, 7\nv_bitop3_b32 v1, v9, v14, v19 bitop3:0x96\nv_alignbit_b32 v2, v8, v8, 7\nv_bitop3_b32 v3, v9, v14, v19 bitop3:0x96\nv_alignbit_b32 v4, v8, v8, 7\nv_bitop3_b32 v5, v9, v14, v19 bitop3:0x96\nv_alignbit_b32 v6, v8, v8, 7\nv_bitop3_b32 v7, v9, v14, v19 bitop3:0x96\n")
typedef void (*kern_t)(uint32_t, uint32_t*);
static void run(const char* name, kern_t k, int cus, uint32_t* out){
  const uint32_t iters=2000; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); printf("%-18s", name);
  for (int bpc : {1,2,4,8}) { int grid=cus*bpc; k<<<grid,256>>>(10,out); hipDeviceSynchronize(); float best=1e30f;
    for (int r=0;r<3;r++){ hipEventRecord(a); k<<<grid,256>>>(iters,out); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); best = ms<best?ms:best; }
    printf("  %dw %5.2f", bpc, best*1e-3*2.39e9/(bpc*(double)iters*64)); }
  printf("\n"); }
int main(){ hipDeviceProp_t p; hipGetDeviceProperties(&p,0); uint32_t* out; hipMalloc(&out, 4096);
  run("bitop3 samebank", bitop3_samebank, p.multiProcessorCount, out);
  run("bitop3 twobank", bitop3_twobank, p.multiProcessorCount, out);
  run("bitop3 diffbank", bitop3_diffbank, p.multiProcessorCount, out);
  run("add3 samebank", add3_samebank, p.multiProcessorCount, out);
  run("add3 diffbank", add3_diffbank, p.multiProcessorCount, out);
  run("add samebank", add_samebank, p.multiProcessorCount, out);
  run("add diffbank", add_diffbank, p.multiProcessorCount, out);
  run("mix align/bitop", mix_align_bitop, p.multiProcessorCount, out);
  return 0; }
