// The Fiat-Shamir channel of stwo-verifier/src/channel.simf:31-172 as device code, shared by the
// verifier's transcript kernels (ss_stwo.hip) and the prover's device-side FRI commit (ss_prover.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ss_fields.h"
#include "ss_hash.h"

namespace ss {

// ------------------------------------------------------------------ channel (channel.simf)
// The digest is kept in the hash's native form; `ctr` is num_sent.
template <int HF>
struct Channel {
    Dig dig;
    uint32_t ctr;

    __device__ void init()  // channel_init, channel.simf:31
    {
#pragma unroll
        for (int i = 0; i < 8; i++) dig.v[i] = 0;
        ctr = 0;
    }
    // channel_mix_u256 (channel.simf:154-161): digest <- H(digest || in); `in` is stored words
    __device__ void mix(const uint32_t (&in)[8])
    {
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = Hasher<HF>::native(in[i]);
        Hasher<HF>::template pair<false>(dig.v, r, dig.v);
        ctr = 0;
    }
    // digest <- H(digest || NV value words): mix_u64, mix_line_poly (one block)
    template <int NV>
    __device__ void mix_values(const uint32_t (&vals)[NV])
    {
        Hasher<HF>::template block<false, 8, NV>(dig.v, vals, dig.v);
        ctr = 0;
    }
    // channel_draw_words (channel.simf:36-65): the 8 big-endian words of H(digest || be4(ctr))
    __device__ void draw_words(uint32_t (&w)[8])
    {
        uint32_t m[1] = {ctr}, d[8];
        Hasher<HF>::template block<false, 8, 1>(dig.v, m, d);
        ctr = ctr + 1;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = Hasher<HF>::native(d[i]);  // native -> stored is the same swap
    }
    // channel_draw_qm31 (channel.simf:115-140): retry while any of the first four words
    // >= 2^32 - 2; the for_while counter is a u8, so at most 256 attempts.
    __device__ bool draw_qm31(QM31 &out)
    {
        for (int it = 0; it < 256; it++) {
            uint32_t w[8];
            draw_words(w);
            if (w[0] < 4294967294u && w[1] < 4294967294u && w[2] < 4294967294u && w[3] < 4294967294u) {
                out = {m31_red(w[0]), m31_red(w[1]), m31_red(w[2]), m31_red(w[3])};
                return true;
            }
        }
        out = qm31_zero();
        return false;
    }
};

}  // namespace ss
