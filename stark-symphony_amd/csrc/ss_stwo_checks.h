// Status codes of the stwo verifier and the asserts behind the FRI layer loop, shared by the query kernel
// (ss_stwo.hip) and the device self-test (ss_api.hip, op 6).
#pragma once
#include <hip/hip_runtime.h>

#include "ss_fields.h"

namespace ss {

__device__ __forceinline__ uint32_t stwo_code(uint32_t stage, uint32_t layer, uint32_t query, uint32_t sub)
{
    return (stage << 24) | (layer << 16) | (query << 4) | sub;
}

// What the reference asserts about one query once the layer loop is through (fri/verify.simf:124-128,
// fri/layers.simf:73-78): `cur` = its position after the K + 1 halvings, `eval` its folded value.  Returns the smallest
// failing code (0xffffffff: none) -- codes are ordered like the reference's evaluation order.  LITERAL only:
// log_size_ex, a u8 that lost 1 per layer (fri/verify.simf:73-74), must be 0 (:127), and the position must be 0
// (fri/layers.simf:75); the repository's own proofs violate both.
__device__ __forceinline__ uint32_t stwo_last_layer_code(uint32_t mode, uint32_t L, uint32_t K, uint32_t q, uint32_t cur,
                                                         const QM31 &eval, const QM31 &last)
{
    uint32_t fail = 0xffffffffu;
    if (mode == 0) {
        if (((L - (K + 1)) & 0xff) != 0) fail = min(fail, stwo_code(8, 0, 0, 0));
        if (cur != 0) fail = min(fail, stwo_code(9, 0, q, 0));
    }
    if (!qm31_eq(eval, last)) fail = min(fail, stwo_code(9, 0, q, 1));
    return fail;
}

}  // namespace ss
