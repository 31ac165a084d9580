// SHA-256 for gfx950 wavefronts: one independent hash per lane, everything in VGPRs.
//
// Replaces the Simplicity jets sha_256_ctx_8_{init,add_*,finalize} that the reference calls
// through stark101/src/sha256.simf:11-30 and stwo-verifier/src/hasher.simf:13-104.
//
// Design notes (DESIGN.md "Merkle kernel"):
//  * the 64-round compression is fully unrolled over a 16-word rolling schedule window, so W
//    never leaves registers; rotates map to v_alignbit_b32, Ch/Maj to v_bfi_b32, the sigma
//    xors to v_xor3_b32 and the adds to v_add3_u32; round constants are SGPR/literal operands.
//  * a Merkle node is SHA-256 of exactly 64 bytes, so its second block is pure padding: the
//    whole message schedule of that block is a compile-time constant (kPad64WK), which removes
//    the 48 schedule updates -- a third of the work of that compression.
//  * words are the big-endian integers of the byte stream == SHA state words, so digests feed
//    the next hash with no byte swaps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

struct Sha256K {
    uint32_t k[64];
};

__host__ __device__ constexpr Sha256K make_k()
{
    return Sha256K{{0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1,
                    0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
                    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
                    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
                    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
                    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
                    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
                    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
                    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
                    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
                    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2}};
}

constexpr Sha256K kK = make_k();

constexpr uint32_t crotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// K[i] + W[i] of the block that pads a message of `bits` bits ending on a block boundary:
// W = {0x80000000, 0 x 14, bits}.
__host__ __device__ constexpr Sha256K make_pad_wk(uint32_t bits)
{
    uint32_t w[64] = {};
    w[0] = 0x80000000u;
    w[15] = bits;
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = crotr(w[i - 15], 7) ^ crotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = crotr(w[i - 2], 17) ^ crotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    Sha256K r{};
    for (int i = 0; i < 64; i++) r.k[i] = kK.k[i] + w[i];
    return r;
}

constexpr Sha256K kPad64WK = make_pad_wk(512);

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
// gfx950 has v_bitop3_b32: any 3-input bitwise function in one VALU op, selected by an 8-bit
// truth table (table = f(0xF0, 0xCC, 0xAA)).  xor3, Ch and Maj are one instruction each.
#ifndef SS_NO_BITOP3
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
#else
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return g ^ (e & (f ^ g)); }
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return b ^ ((a ^ b) & (c ^ b)); }
#endif
__device__ __forceinline__ uint32_t sha_S0(uint32_t a) { return xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)); }
__device__ __forceinline__ uint32_t sha_S1(uint32_t e) { return xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)); }
__device__ __forceinline__ uint32_t sha_s0(uint32_t x) { return xor3(rotr32(x, 7), rotr32(x, 18), x >> 3); }
__device__ __forceinline__ uint32_t sha_s1(uint32_t x) { return xor3(rotr32(x, 17), rotr32(x, 19), x >> 10); }

__device__ __forceinline__ void sha_iv(uint32_t (&h)[8])
{
    h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
    h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}

#define SS_SHA_ROUND(a, b, c, d, e, f, g, h, wk)                      \
    do {                                                              \
        uint32_t t1_ = (h) + sha_S1(e) + sha_ch(e, f, g) + (wk);      \
        uint32_t t2_ = sha_S0(a) + sha_maj(a, b, c);                  \
        (d) += t1_;                                                   \
        (h) = t1_ + t2_;                                              \
    } while (0)

// state <- compress(state, w); w is consumed (used as the rolling schedule window).
__device__ __forceinline__ void sha256_compress(uint32_t (&st)[8], uint32_t (&w)[16])
{
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = i + j;
            if (r >= 16) {
                w[r & 15] += sha_s0(w[(r + 1) & 15]) + w[(r + 9) & 15] + sha_s1(w[(r + 14) & 15]);
            }
        }
        SS_SHA_ROUND(a, b, c, d, e, f, g, h, kK.k[i + 0] + w[(i + 0) & 15]);
        SS_SHA_ROUND(h, a, b, c, d, e, f, g, kK.k[i + 1] + w[(i + 1) & 15]);
        SS_SHA_ROUND(g, h, a, b, c, d, e, f, kK.k[i + 2] + w[(i + 2) & 15]);
        SS_SHA_ROUND(f, g, h, a, b, c, d, e, kK.k[i + 3] + w[(i + 3) & 15]);
        SS_SHA_ROUND(e, f, g, h, a, b, c, d, kK.k[i + 4] + w[(i + 4) & 15]);
        SS_SHA_ROUND(d, e, f, g, h, a, b, c, kK.k[i + 5] + w[(i + 5) & 15]);
        SS_SHA_ROUND(c, d, e, f, g, h, a, b, kK.k[i + 6] + w[(i + 6) & 15]);
        SS_SHA_ROUND(b, c, d, e, f, g, h, a, kK.k[i + 7] + w[(i + 7) & 15]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// state <- compress(state, padding block of a 64-byte message): constant schedule.
__device__ __forceinline__ void sha256_compress_pad64(uint32_t (&st)[8])
{
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        SS_SHA_ROUND(a, b, c, d, e, f, g, h, kPad64WK.k[i + 0]);
        SS_SHA_ROUND(h, a, b, c, d, e, f, g, kPad64WK.k[i + 1]);
        SS_SHA_ROUND(g, h, a, b, c, d, e, f, kPad64WK.k[i + 2]);
        SS_SHA_ROUND(f, g, h, a, b, c, d, e, kPad64WK.k[i + 3]);
        SS_SHA_ROUND(e, f, g, h, a, b, c, d, kPad64WK.k[i + 4]);
        SS_SHA_ROUND(d, e, f, g, h, a, b, c, kPad64WK.k[i + 5]);
        SS_SHA_ROUND(c, d, e, f, g, h, a, b, kPad64WK.k[i + 6]);
        SS_SHA_ROUND(b, c, d, e, f, g, h, a, kPad64WK.k[i + 7]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// out = SHA-256(l || r): sha256_pair (sha256.simf:25, hasher.simf:27).  `out` may alias l or r.
__device__ __forceinline__ void sha256_pair(const uint32_t (&l)[8], const uint32_t (&r)[8],
                                            uint32_t (&out)[8])
{
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[i] = l[i]; w[8 + i] = r[i]; }
    uint32_t st[8];
    sha_iv(st);
    sha256_compress(st, w);
    sha256_compress_pad64(st);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// out = SHA-256 of `nwords` (<= 13) big-endian words: one block.  Zero words and the length
// are compile-time constants after unrolling, so the compiler folds the early schedule.
template <int NWORDS>
__device__ __forceinline__ void sha256_words(const uint32_t (&m)[NWORDS], uint32_t (&out)[8])
{
    static_assert(NWORDS <= 13, "single block only");
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) w[i] = m[i];
    w[NWORDS] = 0x80000000u;
    w[15] = 32u * NWORDS;
    sha_iv(out);
    sha256_compress(out, w);
}

}  // namespace ss
