// Device helpers of the stark101 verifier (stark101/src/channel.simf, sha256.simf), shared by the kernels
// (ss_stark101.hip) and the device replay of the reference's known-answer tests (ss_kat.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ss_fields.h"
#include "ss_sha256.h"

namespace ss {

struct Dig101 { uint32_t v[8]; };
struct W16101 { uint32_t v[16]; };

__device__ __noinline__ Dig101 s101_compress_call(Dig101 st, W16101 w)
{
    sha256_compress(st.v, w.v);
    return st;
}
__device__ __noinline__ Dig101 s101_compress_pad64_call(Dig101 st)
{
    sha256_compress_pad64(st.v);
    return st;
}

// H(state || m[0..NW)), NW <= 1  (sha256 :11, channel_mix_32 channel.simf:22-27)
template <int NW>
__device__ inline Dig101 s101_hash_state(const Dig101 &st, uint32_t m)
{
    W16101 w;
#pragma unroll
    for (int i = 0; i < 16; i++) w.v[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) w.v[i] = st.v[i];
    if (NW) w.v[8] = m;
    w.v[8 + NW] = 0x80000000u;
    w.v[15] = 32u * (8 + NW);
    Dig101 iv;
    sha_iv(iv.v);
    return s101_compress_call(iv, w);
}

// channel_draw_32 (channel.simf:66-105): value = state mod MAX from the PRE-hash state
// (big-endian limbs), then state <- sha256(state).
__device__ __forceinline__ uint32_t s101_draw_mod(Dig101 &st, uint32_t max)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r = (uint32_t)((((uint64_t)r << 32) + st.v[i]) % max);
    st = s101_hash_state<0>(st, 0);
    return r;
}
template <uint32_t MAX>
__device__ inline uint32_t s101_draw(Dig101 &st) { return s101_draw_mod(st, MAX); }

__device__ __forceinline__ uint32_t s101_code(uint32_t stage, uint32_t sub) { return (stage << 8) | sub; }

}  // namespace ss
