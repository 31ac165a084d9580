// Kernel entry points shared between the translation units of libss_verify.so.
#pragma once
#include <hip/hip_runtime.h>

#include "ss_layout.h"

namespace ss {

__global__ void stwo_transcript_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status, uint32_t *accept_count, uint32_t reset);
__global__ void stwo_transcript_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status, uint32_t *accept_count, uint32_t reset);
__global__ void stwo_query_kernel(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_merkle_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_merkle_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_hash_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_hash_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_merkle_min_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_merkle_min_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_min_kernel_sha(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_min_kernel_b2s(StwoLayout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status);
__global__ void stwo_top_cold_kernel_sha(StwoLayout lay, const uint32_t *batch, const uint32_t *ws, uint32_t *status);
__global__ void stwo_top_cold_kernel_b2s(StwoLayout lay, const uint32_t *batch, const uint32_t *ws, uint32_t *status);
__global__ void stwo_finalize_kernel(uint32_t n, uint32_t *status, uint32_t *accept_count);

__global__ void s101_transcript_kernel(S101Layout lay, const uint32_t *batch, uint32_t *ws, uint32_t *status, uint32_t *accept_count);
__global__ void s101_merkle_kernel(S101Layout lay, const uint32_t *batch, const uint32_t *ws, uint32_t *status);

}  // namespace ss
