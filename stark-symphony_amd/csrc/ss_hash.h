// Hash families of the stwo path behind one device interface.
//
//   Hasher<0>  SHA-256      -- the reference (stwo-verifier/src/hasher.simf:13-104, channel.simf)
//   Hasher<1>  Blake2s-256  -- the "Blake2s Merkle" variant BASELINE.json names for configs 3-5.
//                              The reference contains no Blake2s: this variant is
//                              the same protocol over the SAME byte strings with the hash function
//                              swapped, pinned only by RFC 7693 vectors (parity unpinned).
//
// Conventions.  Everything stored in a batch uses the record convention of include/ss_verify.h:
// a field element / counter is its value, a hash is 8 words whose j-th word is the big-endian
// integer of digest bytes 4j..4j+3 ("stored" words).  Inside a kernel a running digest is kept
// in the hash's own chaining form ("native" words): for SHA-256 native == stored; Blake2s works
// on little-endian words, so native == byte-swapped stored, and every message word taken from
// the batch is byte-swapped once (v_perm_b32) on its way in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ss_sha256.h"

namespace ss {

// ------------------------------------------------------------------------------ Blake2s
struct B2sSigma { uint8_t s[10][16]; };
__host__ __device__ constexpr B2sSigma make_sigma()
{
    return B2sSigma{{{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
                     {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                     {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
                     {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                     {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
                     {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                     {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
                     {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                     {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
                     {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}}};
}
constexpr B2sSigma kSigma = make_sigma();
constexpr uint32_t kB2sIV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

__device__ __forceinline__ void b2s_iv(uint32_t (&h)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = kB2sIV[i];
    h[0] ^= 0x01010020u;  // digest length 32, no key, fanout = depth = 1
}

#define SS_B2S_G(a, b, c, d, x, y)                                        \
    do {                                                                  \
        a = a + b + (x); d = rotr32(d ^ a, 16);                           \
        c = c + d;       b = rotr32(b ^ c, 12);                           \
        a = a + b + (y); d = rotr32(d ^ a, 8);                            \
        c = c + d;       b = rotr32(b ^ c, 7);                            \
    } while (0)

// h <- F(h, m, t, last); m are little-endian message words, t the byte counter (< 2^32 here).
__device__ __forceinline__ void blake2s_compress(uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t,
                                                 bool last)
{
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = kB2sIV[0], v9 = kB2sIV[1], v10 = kB2sIV[2], v11 = kB2sIV[3];
    uint32_t v12 = kB2sIV[4] ^ t, v13 = kB2sIV[5], v14 = last ? ~kB2sIV[6] : kB2sIV[6], v15 = kB2sIV[7];
#pragma unroll
    for (int r = 0; r < 10; r++) {
        SS_B2S_G(v0, v4, v8, v12, m[kSigma.s[r][0]], m[kSigma.s[r][1]]);
        SS_B2S_G(v1, v5, v9, v13, m[kSigma.s[r][2]], m[kSigma.s[r][3]]);
        SS_B2S_G(v2, v6, v10, v14, m[kSigma.s[r][4]], m[kSigma.s[r][5]]);
        SS_B2S_G(v3, v7, v11, v15, m[kSigma.s[r][6]], m[kSigma.s[r][7]]);
        SS_B2S_G(v0, v5, v10, v15, m[kSigma.s[r][8]], m[kSigma.s[r][9]]);
        SS_B2S_G(v1, v6, v11, v12, m[kSigma.s[r][10]], m[kSigma.s[r][11]]);
        SS_B2S_G(v2, v7, v8, v13, m[kSigma.s[r][12]], m[kSigma.s[r][13]]);
        SS_B2S_G(v3, v4, v9, v14, m[kSigma.s[r][14]], m[kSigma.s[r][15]]);
    }
    h[0] = xor3(h[0], v0, v8);   h[1] = xor3(h[1], v1, v9);
    h[2] = xor3(h[2], v2, v10);  h[3] = xor3(h[3], v3, v11);
    h[4] = xor3(h[4], v4, v12);  h[5] = xor3(h[5], v5, v13);
    h[6] = xor3(h[6], v6, v14);  h[7] = xor3(h[7], v7, v15);
}

struct Dig { uint32_t v[8]; };
struct W16 { uint32_t v[16]; };

// Out-of-line compressions for the sequential transcript kernels (small code footprint).
static __device__ __noinline__ Dig sha_compress_call(Dig st, W16 w)
{
    sha256_compress(st.v, w.v);
    return st;
}
static __device__ __noinline__ Dig sha_compress_pad64_call(Dig st)
{
    sha256_compress_pad64(st.v);
    return st;
}
static __device__ __noinline__ Dig b2s_compress_call(Dig st, W16 m, uint32_t t, bool last)
{
    blake2s_compress(st.v, m.v, t, last);
    return st;
}

template <int H>
struct Hasher;

// ================================================================================ SHA-256
template <>
struct Hasher<0> {
    static __device__ __forceinline__ uint32_t native(uint32_t stored) { return stored; }

    // out = H(l || r), 64 bytes
    template <bool INL>
    static __device__ __forceinline__ void pair(const uint32_t (&l)[8], const uint32_t (&r)[8], uint32_t (&out)[8])
    {
        if (INL) {
            sha256_pair(l, r, out);
        } else {
            Dig st;
            W16 w;
            sha_iv(st.v);
#pragma unroll
            for (int i = 0; i < 8; i++) { w.v[i] = l[i]; w.v[8 + i] = r[i]; }
            st = sha_compress_pad64_call(sha_compress_call(st, w));
#pragma unroll
            for (int i = 0; i < 8; i++) out[i] = st.v[i];
        }
    }

    // out = H(prefix[0..NP) || vals[0..NV)) with NP + NV <= 13 words: one block.  prefix is
    // native (a digest), vals are stored/value words.
    template <bool INL, int NP, int NV>
    static __device__ __forceinline__ void block(const uint32_t *prefix, const uint32_t *vals, uint32_t (&out)[8])
    {
        static_assert(NP + NV <= 13, "single block only");
        W16 w;
#pragma unroll
        for (int i = 0; i < 16; i++) w.v[i] = 0;
#pragma unroll
        for (int i = 0; i < NP; i++) w.v[i] = prefix[i];
#pragma unroll
        for (int i = 0; i < NV; i++) w.v[NP + i] = vals[i];
        w.v[NP + NV] = 0x80000000u;
        w.v[15] = 32u * (NP + NV);
        Dig st;
        sha_iv(st.v);
        if (INL) sha256_compress(st.v, w.v);
        else st = sha_compress_call(st, w);
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = st.v[i];
    }

    // out = H(prefix (NP native words, 0 or 8) || get(0..nvals)), any length
    template <bool INL, int NP, class G>
    static __device__ __forceinline__ void stream(const uint32_t *prefix, G get, uint32_t nvals, uint32_t (&out)[8])
    {
        const uint32_t total = NP + nvals;            // message words
        const uint32_t nblk = (total + 2) / 16 + 1;   // + 0x80 word + 64-bit length
        Dig st;
        sha_iv(st.v);
        for (uint32_t b = 0; b < nblk; b++) {
            W16 w;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t i = b * 16 + j;
                uint32_t v = 0;
                if (i < NP) v = 0;  // patched below with static indexing
                else if (i < total) v = get(i - NP);
                else if (i == total) v = 0x80000000u;
                else if (i == nblk * 16 - 1) v = 32u * total;
                w.v[j] = v;
            }
            if (NP && b == 0) {
#pragma unroll
                for (int j = 0; j < NP; j++) w.v[j] = prefix[j];
            }
            if (INL) sha256_compress(st.v, w.v);
            else st = sha_compress_call(st, w);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = st.v[i];
    }

    // pow.simf:27-30: the last 8 digest bytes read as a little-endian u64
    static __device__ __forceinline__ uint64_t pow_value(const uint32_t (&d)[8])
    {
        return ((uint64_t)__builtin_bswap32(d[7]) << 32) | __builtin_bswap32(d[6]);
    }

    // H(prefix (8 native words) || value words) block by block, for a caller that keeps ONE inlined compression in a
    // loop (the transcript kernels): `total` = message words, block b of n_blocks(total).  get(j, i) = value word i for
    // slot j (j is a compile-time index, so a caller may keep short messages in registers).
    static __device__ __forceinline__ void iv(uint32_t (&h)[8]) { sha_iv(h); }
    static __device__ __forceinline__ uint32_t n_blocks(uint32_t total) { return (total + 2) / 16 + 1; }  // + 0x80 word + 64-bit length
    template <class G>
    static __device__ __forceinline__ void fill(W16 &w, uint32_t b, uint32_t nblk, uint32_t total, const uint32_t (&prefix)[8], G get)
    {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t i = b * 16 + j;
            uint32_t v = 0;
            if (i < total) { if (i >= 8) v = get(j, i - 8); }
            else if (i == total) v = 0x80000000u;
            else if (i == nblk * 16 - 1) v = 32u * total;
            w.v[j] = v;
        }
        if (b == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) w.v[j] = prefix[j];
        }
    }
    static __device__ __forceinline__ void compress(Dig &st, W16 &w, uint32_t, uint32_t, uint32_t) { sha256_compress(st.v, w.v); }  // (w is the rolling schedule: consumed)
};

// ================================================================================ Blake2s
template <>
struct Hasher<1> {
    static __device__ __forceinline__ uint32_t native(uint32_t stored) { return __builtin_bswap32(stored); }

    template <bool INL>
    static __device__ __forceinline__ void finish(Dig &st, W16 &m, uint32_t t, uint32_t (&out)[8])
    {
        if (INL) blake2s_compress(st.v, m.v, t, true);
        else st = b2s_compress_call(st, m, t, true);
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = st.v[i];
    }

    template <bool INL>
    static __device__ __forceinline__ void pair(const uint32_t (&l)[8], const uint32_t (&r)[8], uint32_t (&out)[8])
    {
        Dig st;
        W16 m;
        b2s_iv(st.v);
#pragma unroll
        for (int i = 0; i < 8; i++) { m.v[i] = l[i]; m.v[8 + i] = r[i]; }
        finish<INL>(st, m, 64, out);
    }

    template <bool INL, int NP, int NV>
    static __device__ __forceinline__ void block(const uint32_t *prefix, const uint32_t *vals, uint32_t (&out)[8])
    {
        static_assert(NP + NV <= 16, "single block only");
        Dig st;
        W16 m;
        b2s_iv(st.v);
#pragma unroll
        for (int i = 0; i < 16; i++) m.v[i] = 0;
#pragma unroll
        for (int i = 0; i < NP; i++) m.v[i] = prefix[i];
#pragma unroll
        for (int i = 0; i < NV; i++) m.v[NP + i] = native(vals[i]);
        finish<INL>(st, m, 4u * (NP + NV), out);
    }

    template <bool INL, int NP, class G>
    static __device__ __forceinline__ void stream(const uint32_t *prefix, G get, uint32_t nvals, uint32_t (&out)[8])
    {
        const uint32_t total = NP + nvals;
        const uint32_t nblk = total ? (total + 15) / 16 : 1;
        Dig st;
        b2s_iv(st.v);
        for (uint32_t b = 0; b < nblk; b++) {
            W16 m;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t i = b * 16 + j;
                m.v[j] = (i >= NP && i < total) ? native(get(i - NP)) : 0u;
            }
            if (NP && b == 0) {
#pragma unroll
                for (int j = 0; j < NP; j++) m.v[j] = prefix[j];
            }
            const bool last = b + 1 == nblk;
            const uint32_t t = last ? 4u * total : 64u * (b + 1);
            if (INL) blake2s_compress(st.v, m.v, t, last);
            else st = b2s_compress_call(st, m, t, last);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = st.v[i];
    }

    static __device__ __forceinline__ uint64_t pow_value(const uint32_t (&d)[8])
    {
        return ((uint64_t)d[7] << 32) | d[6];
    }

    // (see Hasher<0>) Blake2s pads with zeros and carries the byte count in the compression itself
    static __device__ __forceinline__ void iv(uint32_t (&h)[8]) { b2s_iv(h); }
    static __device__ __forceinline__ uint32_t n_blocks(uint32_t total) { return total ? (total + 15) / 16 : 1; }
    template <class G>
    static __device__ __forceinline__ void fill(W16 &w, uint32_t b, uint32_t, uint32_t total, const uint32_t (&prefix)[8], G get)
    {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t i = b * 16 + j;
            w.v[j] = (i >= 8 && i < total) ? native(get(j, i - 8)) : 0u;
        }
        if (b == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) w.v[j] = prefix[j];
        }
    }
    static __device__ __forceinline__ void compress(Dig &st, W16 &w, uint32_t b, uint32_t nblk, uint32_t total)
    {
        const bool last = b + 1 == nblk;
        blake2s_compress(st.v, w.v, last ? 4u * total : 64u * (b + 1), last);
    }
};

}  // namespace ss
