/*
 * ss_oracle.c -- CPU restatement of the stark-symphony verifier hot path.
 * TEST INFRASTRUCTURE ONLY (see ss_oracle.h).  Plain sequential C; every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference).  Unsigned wrap-around is used exactly where the reference
 * discards a jet's carry/borrow bit.
 */
#include "ss_oracle.h"

#include <string.h>

/* ===================================================================== SHA-256
 * FIPS 180-4.  Stands in for the jets sha_256_ctx_8_{init,add_*,finalize}
 * (stark101/src/sha256.simf:11-30, stwo-verifier/src/hasher.simf:13-32); pinned
 * by test_sha256 / test_sha256_32 (sha256.simf:32-42).                        */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4,
    0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
    0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f,
    0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
    0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc,
    0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116,
    0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

static __thread uint64_t g_blocks;
uint64_t so_sha256_blocks(void) { return g_blocks; }
void so_sha256_blocks_reset(void) { g_blocks = 0; }

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

/* The CPU baseline of bench.py times this file, so the compression uses the x86 SHA extensions when
 * the host has them (same function, FIPS 180-4; pinned by the same known-answer tests; set
 * SS_ORACLE_NO_SHANI=1 to force the portable code below). */
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#include <stdlib.h>
static int g_shani = -1;
static int shani_available(void)
{
    if (g_shani < 0) {
        unsigned a, b, c, d;
        int ok = __get_cpuid_count(7, 0, &a, &b, &c, &d) && (b & (1u << 29));      /* SHA */
        ok = ok && __get_cpuid(1, &a, &b, &c, &d) && (c & (1u << 19)) && (c & (1u << 9)); /* SSE4.1, SSSE3 */
        const char *off = getenv("SS_ORACLE_NO_SHANI");
        g_shani = ok && !(off && off[0] == '1');
    }
    return g_shani;
}
__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_block_ni(uint32_t h[8], const uint8_t blk[64])
{
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&h[0]), 0xB1);  /* CDAB */
    __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&h[4]), 0x1B);   /* EFGH */
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                                        /* ABEF */
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                                             /* CDGH */
    const __m128i save0 = s0, save1 = s1;
    __m128i w[4];
    for (int i = 0; i < 4; i++)
        w[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(blk + 16 * i)), bswap);
    for (int i = 0; i < 16; i++) { /* four rounds per step */
        __m128i t = _mm_add_epi32(w[i & 3], _mm_loadu_si128((const __m128i *)&K256[4 * i]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, t);
        s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(t, 0x0E));
        if (i < 12) { /* W[4(i+4) .. 4(i+4)+3] replaces W[4i ..] */
            const __m128i w1 = w[(i + 1) & 3], w2 = w[(i + 2) & 3], w3 = w[(i + 3) & 3];
            __m128i x = _mm_add_epi32(_mm_sha256msg1_epu32(w[i & 3], w1), _mm_alignr_epi8(w3, w2, 4));
            w[i & 3] = _mm_sha256msg2_epu32(x, w3);
        }
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                 /* FEBA */
    s1 = _mm_shuffle_epi32(s1, 0xB1);                  /* DCHG */
    _mm_storeu_si128((__m128i *)&h[0], _mm_blend_epi16(tmp, s1, 0xF0)); /* DCBA */
    _mm_storeu_si128((__m128i *)&h[4], _mm_alignr_epi8(s1, tmp, 8));    /* HGFE */
}
#endif

static void sha256_block(uint32_t h[8], const uint8_t blk[64])
{
#if defined(__x86_64__)
    if (shani_available()) {
        sha256_block_ni(h, blk);
        g_blocks++;
        return;
    }
#endif
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    g_blocks++;
}

typedef struct {
    uint32_t h[8];
    uint8_t buf[64];
    size_t fill;
    uint64_t total;
    int kind; /* 0 SHA-256 (the reference), 1 Blake2s-256 (extension, parity unpinned) */
} sha_ctx;

/* Hash family of the running stwo verification (so_stwo_cfg.hash).  The reference only has
 * SHA-256; Blake2s is the BASELINE.json variant, pinned by RFC 7693 vectors only.  The byte
 * strings that get hashed are identical in both variants. */
static __thread int g_hash_kind;

/* ---- Blake2s-256 (RFC 7693), unkeyed */
static const uint32_t B2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                   0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static void b2s_compress(uint32_t h[8], const uint8_t blk[64], uint64_t t, int last)
{
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) |
               ((uint32_t)blk[4 * i + 3] << 24);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define B2S_G(a, b, c, d, x, y)                                   \
    do {                                                          \
        v[a] = v[a] + v[b] + (x); v[d] = rotr(v[d] ^ v[a], 16);   \
        v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 12);   \
        v[a] = v[a] + v[b] + (y); v[d] = rotr(v[d] ^ v[a], 8);    \
        v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 7);    \
    } while (0)
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2S_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    g_blocks++;
}

static void sha_init(sha_ctx *c)
{
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->h, iv, sizeof iv);
    c->fill = 0;
    c->total = 0;
    c->kind = g_hash_kind;
    if (c->kind == 1) c->h[0] ^= 0x01010020u; /* digest length 32, no key, fanout = depth = 1 */
}

static void sha_add(sha_ctx *c, const uint8_t *p, size_t n)
{
    if (c->kind == 1) {
        /* the last block must be compressed with the final flag: keep it buffered */
        while (n) {
            if (c->fill == 64) {
                c->total += 64;
                b2s_compress(c->h, c->buf, c->total, 0);
                c->fill = 0;
            }
            size_t k = 64 - c->fill;
            if (k > n) k = n;
            memcpy(c->buf + c->fill, p, k);
            c->fill += k; p += k; n -= k;
        }
        return;
    }
    c->total += n;
    while (n) {
        size_t k = 64 - c->fill;
        if (k > n) k = n;
        memcpy(c->buf + c->fill, p, k);
        c->fill += k; p += k; n -= k;
        if (c->fill == 64) { sha256_block(c->h, c->buf); c->fill = 0; }
    }
}

/* jet sha_256_ctx_8_add_4 / add_8: big-endian integer bytes */
static void sha_add_u32(sha_ctx *c, uint32_t v)
{
    uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    sha_add(c, b, 4);
}

static void sha_add_u64(sha_ctx *c, uint64_t v)
{
    sha_add_u32(c, (uint32_t)(v >> 32));
    sha_add_u32(c, (uint32_t)v);
}

static void sha_final(sha_ctx *c, uint8_t out[32])
{
    if (c->kind == 1) {
        c->total += c->fill;
        memset(c->buf + c->fill, 0, 64 - c->fill);
        b2s_compress(c->h, c->buf, c->total, 1);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)c->h[i];
            out[4 * i + 1] = (uint8_t)(c->h[i] >> 8);
            out[4 * i + 2] = (uint8_t)(c->h[i] >> 16);
            out[4 * i + 3] = (uint8_t)(c->h[i] >> 24);
        }
        return;
    }
    uint64_t bits = c->total * 8;
    uint8_t pad = 0x80;
    sha_add(c, &pad, 1);
    uint8_t z = 0;
    while (c->fill != 56) sha_add(c, &z, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_add(c, len, 8);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(c->h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(c->h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(c->h[i] >> 8);
        out[4 * i + 3] = (uint8_t)c->h[i];
    }
}

void so_sha256(const uint8_t *msg, size_t len, uint8_t out[32])
{
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, msg, len);
    sha_final(&c, out);
}

/* hash family used by the stwo leaf functions when called one by one (tests); so_stwo_verify
 * sets it from cfg->hash itself */
void so_set_hash(int kind) { g_hash_kind = kind == 1 ? 1 : 0; }

void so_blake2s(const uint8_t *msg, size_t len, uint8_t out[32])
{
    int saved = g_hash_kind;
    g_hash_kind = 1;
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, msg, len);
    sha_final(&c, out);
    g_hash_kind = saved;
}

/* sha256.simf:11 / hasher.simf:13 */
static void sha256_u256(const uint8_t in[32], uint8_t out[32]) { so_sha256(in, 32, out); }

/* sha256.simf:18 / hasher.simf:20 */
static void sha256_32(uint32_t v, uint8_t out[32])
{
    sha_ctx c;
    sha_init(&c);
    sha_add_u32(&c, v);
    sha_final(&c, out);
}

/* sha256.simf:25 / hasher.simf:27 */
static void sha256_pair(const uint8_t l[32], const uint8_t r[32], uint8_t out[32])
{
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, l, 32);
    sha_add(&c, r, 32);
    sha_final(&c, out);
}

/* Simplicity jet conventions for a zero divisor (not reachable by well-formed
 * proofs and not pinned by any reference test; see ss_oracle.h header):
 * divide_32(a,0)=0, modulo_32(a,0)=a. */
static inline uint32_t jet_divide_32(uint32_t a, uint32_t b) { return b ? a / b : 0; }
static inline uint32_t jet_modulo_32(uint32_t a, uint32_t b) { return b ? a % b : a; }

/* ==================================================================== stark101 */

/* field.simf:14-21 */
uint32_t so_s101_add_mod(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a + (uint64_t)b) % SO_S101_P);
}

/* field.simf:24-27 -- the borrow of subtract_32 is discarded */
uint32_t so_s101_sub_mod(uint32_t a, uint32_t b)
{
    uint32_t b_neg = SO_S101_P - b;
    return so_s101_add_mod(a, b_neg);
}

/* field.simf:30-35 */
uint32_t so_s101_mul_mod(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) % SO_S101_P);
}

/* field.simf:42-63: extended Euclid with *field* updates; for_while has a u16 counter */
int so_s101_div_mod(uint32_t a, uint32_t b, uint32_t *out)
{
    uint32_t t = 0, r = SO_S101_P, new_t = 1, new_r = b;
    for (uint32_t it = 0; it < 65536; it++) {
        if (new_r == 0) {
            if (r != 1) return 1; /* assert!(eq_32(r, 1)) field.simf:46 */
            *out = so_s101_mul_mod(a, t);
            return 0;
        }
        uint32_t q = jet_divide_32(r, new_r);
        uint32_t t2 = so_s101_sub_mod(t, so_s101_mul_mod(q, new_t));
        uint32_t r2 = so_s101_sub_mod(r, so_s101_mul_mod(q, new_r));
        t = new_t; new_t = t2;
        r = new_r; new_r = r2;
    }
    return 1; /* unwrap_left on Right */
}

/* field.simf:74-94 */
uint32_t so_s101_exp_mod(uint32_t a, uint32_t b)
{
    uint32_t res = 1, base = a, e = b;
    while (e != 0) {
        if (e & 1) res = so_s101_mul_mod(res, base);
        base = so_s101_mul_mod(base, base);
        e >>= 1;
    }
    return res;
}

/* channel.simf:66-93 */
uint32_t so_s101_reduce_256_mod_32(const uint8_t v[32], uint32_t modulo)
{
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t limb = ((uint32_t)v[4 * i] << 24) | ((uint32_t)v[4 * i + 1] << 16) |
                        ((uint32_t)v[4 * i + 2] << 8) | v[4 * i + 3];
        uint64_t x = ((uint64_t)r << 32) + limb;
        r = (uint32_t)(modulo ? x % modulo : x); /* modulo_64 */
    }
    return r;
}

/* channel.simf:102-105: value from the PRE-hash state, then state = sha256(state) */
uint32_t so_s101_channel_draw_32(uint8_t state[32], uint32_t max)
{
    uint32_t v = so_s101_reduce_256_mod_32(state, max);
    uint8_t n[32];
    sha256_u256(state, n);
    memcpy(state, n, 32);
    return v;
}

/* channel.simf:22-27 */
void so_s101_channel_mix_32(uint8_t state[32], uint32_t input)
{
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, state, 32);
    sha_add_u32(&c, input);
    sha_final(&c, state);
}

/* channel.simf:35-40 */
void so_s101_channel_mix_256(uint8_t state[32], const uint8_t input[32])
{
    uint8_t n[32];
    sha256_pair(state, input, n);
    memcpy(state, n, 32);
}

/* merkle.simf:22-43 (stark101: no final path==1 assert).  0 ok, 1 root mismatch */
int so_s101_merkle_verify(const uint8_t leaf[32], uint32_t auth_path, const uint8_t *proof,
                          uint32_t len, const uint8_t root[32])
{
    uint8_t cur[32], nxt[32];
    uint32_t path = auth_path;
    memcpy(cur, leaf, 32);
    for (uint32_t i = 0; i < len; i++) {
        const uint8_t *sib = proof + 32 * (size_t)i;
        if (path & 1) sha256_pair(sib, cur, nxt); /* divides_32(2,path)==false */
        else sha256_pair(cur, sib, nxt);
        memcpy(cur, nxt, 32);
        path = path / 2;
    }
    return memcmp(cur, root, 32) != 0;
}

/* air.simf:16-18 */
#define S101_IDX_OFFSET 8u
#define S101_DOMAIN_EX 8192u
#define S101_COSET_GEN 1734477367u
#define S101_FIELD_GEN 5u

/* air.simf:58-60 */
uint32_t so_s101_calc_x(uint32_t idx)
{
    return so_s101_mul_mod(S101_FIELD_GEN, so_s101_exp_mod(S101_COSET_GEN, idx));
}

/* air.simf:63-66 */
int so_s101_eval_p0(uint32_t x, uint32_t f_x, uint32_t *out)
{
    return so_s101_div_mod(so_s101_sub_mod(f_x, 1), so_s101_sub_mod(x, 1), out);
}

/* air.simf:69-72 */
static int s101_eval_p1(uint32_t x, uint32_t f_x, uint32_t *out)
{
    return so_s101_div_mod(so_s101_sub_mod(f_x, 2338775057u), so_s101_sub_mod(x, 2450347685u), out);
}

/* air.simf:75-83 */
static int s101_eval_p2(uint32_t x, uint32_t f_x, uint32_t f_gx, uint32_t f_ggx, uint32_t *out)
{
    uint32_t num0 = so_s101_sub_mod(
        f_ggx, so_s101_add_mod(so_s101_mul_mod(f_x, f_x), so_s101_mul_mod(f_gx, f_gx)));
    uint32_t num1 = so_s101_mul_mod(
        so_s101_mul_mod(so_s101_sub_mod(x, 2342081930u), so_s101_sub_mod(x, 2450347685u)),
        so_s101_sub_mod(x, 532203874u));
    uint32_t den = so_s101_sub_mod(so_s101_exp_mod(x, 1024), 1);
    return so_s101_div_mod(so_s101_mul_mod(num0, num1), den, out);
}

/* air.simf:86-91; returns 0 ok or 1+j when the j-th division aborts */
int so_s101_eval_cp(uint32_t x, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t f_x,
                    uint32_t f_gx, uint32_t f_ggx, uint32_t *out)
{
    uint32_t p0, p1, p2;
    if (so_s101_eval_p0(x, f_x, &p0)) return 1;
    if (s101_eval_p1(x, f_x, &p1)) return 2;
    if (s101_eval_p2(x, f_x, f_gx, f_ggx, &p2)) return 3;
    *out = so_s101_add_mod(so_s101_add_mod(so_s101_mul_mod(p0, a0), so_s101_mul_mod(p1, a1)),
                           so_s101_mul_mod(p2, a2));
    return 0;
}

/* fri.simf:58-62 */
int so_s101_fri_eval_cp_next(uint32_t cpa, uint32_t cpb, uint32_t x, uint32_t beta, uint32_t *out)
{
    uint32_t op0, op1;
    if (so_s101_div_mod(so_s101_add_mod(cpa, cpb), 2, &op0)) return 1;
    if (so_s101_div_mod(so_s101_sub_mod(cpa, cpb), so_s101_mul_mod(x, 2), &op1)) return 1;
    *out = so_s101_add_mod(op0, so_s101_mul_mod(op1, beta));
    return 0;
}

/* fri.simf:66-71 */
void so_s101_compute_auth_path(uint32_t idx, uint32_t domain_size, uint32_t *cpa_path,
                               uint32_t *cpb_path)
{
    *cpa_path = jet_modulo_32(idx, domain_size) + domain_size;
    uint32_t cpb_idx = idx + jet_divide_32(domain_size, 2);
    *cpb_path = jet_modulo_32(cpb_idx, domain_size) + domain_size;
}

#define S101_FAIL(stage, sub)                                                      \
    do {                                                                           \
        if (!status) status = ((uint32_t)(stage) << 8) | (uint32_t)(sub);          \
    } while (0)

/* verifier.simf:24-42 */
uint32_t so_s101_verify(const so_s101_proof *p, so_s101_trace *tr)
{
    uint32_t status = 0;
    uint8_t state[32];
    uint32_t n = p->n_layers > SO_MAX_LIST ? SO_MAX_LIST : p->n_layers;

    /* :27 read trace root */
    sha256_u256(p->root, state);
    /* :29 air.simf:30-35 */
    uint32_t a0 = so_s101_channel_draw_32(state, SO_S101_P);
    uint32_t a1 = so_s101_channel_draw_32(state, SO_S101_P);
    uint32_t a2 = so_s101_channel_draw_32(state, SO_S101_P);
    /* :31 fri.simf:37-54 */
    for (uint32_t i = 0; i < n; i++) {
        so_s101_channel_mix_256(state, p->layers[i].root);
        uint32_t random = so_s101_channel_draw_32(state, SO_S101_P);
        if (random != p->layers[i].beta) S101_FAIL(1, i);
    }
    so_s101_channel_mix_32(state, p->last);
    if (tr) memcpy(tr->state_after_commit, state, 32);
    /* :33 */
    uint32_t idx = so_s101_channel_draw_32(state, S101_DOMAIN_EX);
    /* :35 air.simf:38-55 */
    {
        uint32_t id = idx;
        for (int k = 0; k < 3; k++) {
            uint8_t leaf[32];
            uint32_t auth = id + S101_DOMAIN_EX;
            sha256_32(p->evals[k].ev, leaf);
            if (so_s101_merkle_verify(leaf, auth, &p->evals[k].path[0][0], p->evals[k].len, p->root))
                S101_FAIL(2, k);
            so_s101_channel_mix_32(state, p->evals[k].ev);
            id = id + S101_IDX_OFFSET;
        }
    }
    /* :37 */
    uint32_t x = so_s101_calc_x(idx);
    /* :39 air.simf:94-101 */
    uint32_t cp = 0;
    int rc = so_s101_eval_cp(x, a0, a1, a2, p->evals[0].ev, p->evals[1].ev, p->evals[2].ev, &cp);
    if (rc) S101_FAIL(3, rc - 1);
    if (tr) {
        tr->alpha[0] = a0; tr->alpha[1] = a1; tr->alpha[2] = a2;
        tr->idx = idx; tr->x = x; tr->cp = cp;
    }
    /* :41 fri.simf:74-91 */
    uint32_t xx = x, dom = S101_DOMAIN_EX, cur = cp;
    for (uint32_t i = 0; i < n; i++) {
        const so_s101_layer *l = &p->layers[i];
        uint8_t leaf[32];
        uint32_t pa, pb, nxt = 0;
        if (tr) tr->fold[i] = cur;
        if (cur != l->cpa.ev) S101_FAIL(4, 4 * i + 0);
        so_s101_compute_auth_path(idx, dom, &pa, &pb);
        sha256_32(l->cpa.ev, leaf);
        if (so_s101_merkle_verify(leaf, pa, &l->cpa.path[0][0], l->cpa.len, l->root))
            S101_FAIL(4, 4 * i + 1);
        sha256_32(l->cpb.ev, leaf);
        if (so_s101_merkle_verify(leaf, pb, &l->cpb.path[0][0], l->cpb.len, l->root))
            S101_FAIL(4, 4 * i + 2);
        if (so_s101_fri_eval_cp_next(l->cpa.ev, l->cpb.ev, xx, l->beta, &nxt))
            S101_FAIL(4, 4 * i + 3);
        cur = nxt;
        xx = so_s101_mul_mod(xx, xx);
        dom = jet_divide_32(dom, 2);
    }
    if (tr) tr->fold[n] = cur;
    if (cur != p->last) S101_FAIL(5, 0);
    return status;
}

/* ======================================================================== stwo */

/* ------------------------------------------------------ fields/m31.simf:17-138 */
static inline uint32_t m31_reduce(uint32_t v) { return v % SO_M31_P; } /* :17-19 */
uint32_t so_m31_add(uint32_t a, uint32_t b) { return m31_reduce(a + b); } /* :22-26 wrap */
uint32_t so_m31_neg(uint32_t a) { return SO_M31_P - a; }                  /* :29-32 wrap */
uint32_t so_m31_sub(uint32_t a, uint32_t b) { return so_m31_add(a, so_m31_neg(b)); } /* :35 */
uint32_t so_m31_mul(uint32_t a, uint32_t b)                               /* :40-45 */
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) % SO_M31_P);
}

/* :57-77 */
uint32_t so_m31_exp(uint32_t a, uint32_t b)
{
    uint32_t res = 1, base = a, e = b;
    while (e != 0) {
        if (e & 1) res = so_m31_mul(res, base);
        base = so_m31_mul(base, base);
        e >>= 1;
    }
    return res;
}

static uint32_t m31_pow2(uint32_t a) { return so_m31_mul(a, a); }
static uint32_t m31_pow4(uint32_t a) { return m31_pow2(m31_pow2(a)); }
static uint32_t m31_pow8(uint32_t a) { return m31_pow2(m31_pow4(a)); }
static uint32_t m31_pow16(uint32_t a) { return m31_pow4(m31_pow4(a)); }

/* :117-133 -- aborts only when the raw word is 0 */
int so_m31_inv(uint32_t a, uint32_t *out)
{
    if (a == 0) return 1;
    uint32_t t0 = so_m31_mul(m31_pow4(a), a);
    uint32_t t1 = so_m31_mul(m31_pow2(t0), t0);
    uint32_t t2 = so_m31_mul(m31_pow8(t1), t0);
    uint32_t t3 = so_m31_mul(m31_pow2(t2), t0);
    uint32_t t4 = so_m31_mul(m31_pow16(m31_pow16(t3)), t3);
    uint32_t t5 = so_m31_mul(m31_pow16(m31_pow16(t4)), t3);
    *out = so_m31_mul(m31_pow16(m31_pow8(t5)), t2);
    return 0;
}

/* ------------------------------------------------------ fields/cm31.simf:30-113 */
so_cm31 so_cm31_add(so_cm31 a, so_cm31 b)
{
    so_cm31 r = {so_m31_add(a.a, b.a), so_m31_add(a.b, b.b)};
    return r;
}
static so_cm31 cm31_neg(so_cm31 a)
{
    so_cm31 r = {so_m31_neg(a.a), so_m31_neg(a.b)};
    return r;
}
so_cm31 so_cm31_sub(so_cm31 a, so_cm31 b)
{
    so_cm31 r = {so_m31_sub(a.a, b.a), so_m31_sub(a.b, b.b)};
    return r;
}
static so_cm31 cm31_sub_m31(so_cm31 a, uint32_t b) /* :51-54 */
{
    so_cm31 r = {so_m31_sub(a.a, b), a.b};
    return r;
}
static so_cm31 cm31_mul_m31(so_cm31 a, uint32_t b) /* :57-60 */
{
    so_cm31 r = {so_m31_mul(a.a, b), so_m31_mul(a.b, b)};
    return r;
}
static so_cm31 cm31_conj(so_cm31 a) /* :74-77 */
{
    so_cm31 r = {a.a, so_m31_neg(a.b)};
    return r;
}
so_cm31 so_cm31_mul(so_cm31 a, so_cm31 b) /* :80-86 */
{
    so_cm31 r;
    r.a = so_m31_sub(so_m31_mul(a.a, b.a), so_m31_mul(a.b, b.b));
    r.b = so_m31_add(so_m31_mul(a.a, b.b), so_m31_mul(a.b, b.a));
    return r;
}
int so_cm31_inv(so_cm31 a, so_cm31 *out) /* :89-94, cm31_div_m31 :63-66 */
{
    so_cm31 conj = cm31_conj(a);
    uint32_t norm = so_m31_add(m31_pow2(a.a), m31_pow2(a.b));
    uint32_t ninv;
    if (so_m31_inv(norm, &ninv)) return 1;
    *out = cm31_mul_m31(conj, ninv);
    return 0;
}
int so_cm31_div(so_cm31 a, so_cm31 b, so_cm31 *out) /* :97-100 */
{
    so_cm31 bi;
    if (so_cm31_inv(b, &bi)) return 1;
    *out = so_cm31_mul(a, bi);
    return 0;
}
static so_cm31 cm31_dbl(so_cm31 a) { return so_cm31_add(a, a); } /* :103-105 */

/* ------------------------------------------------------ fields/qm31.simf:20-132 */
static inline so_cm31 q_re(so_qm31 q) { so_cm31 r = {q.a, q.b}; return r; }
static inline so_cm31 q_im(so_qm31 q) { so_cm31 r = {q.c, q.d}; return r; }
static inline so_qm31 q_make(so_cm31 re, so_cm31 im)
{
    so_qm31 r = {re.a, re.b, im.a, im.b};
    return r;
}
static const so_qm31 QM31_ZERO = {0, 0, 0, 0};
static const so_qm31 QM31_ONE = {1, 0, 0, 0};

so_qm31 so_qm31_add(so_qm31 a, so_qm31 b)
{
    return q_make(so_cm31_add(q_re(a), q_re(b)), so_cm31_add(q_im(a), q_im(b)));
}
so_qm31 so_qm31_sub(so_qm31 a, so_qm31 b)
{
    return q_make(so_cm31_sub(q_re(a), q_re(b)), so_cm31_sub(q_im(a), q_im(b)));
}
so_qm31 so_qm31_mul_m31(so_qm31 a, uint32_t b) /* :55-58 */
{
    return q_make(cm31_mul_m31(q_re(a), b), cm31_mul_m31(q_im(a), b));
}
so_qm31 so_qm31_mul_cm31(so_qm31 a, so_cm31 b) /* :61-64 */
{
    return q_make(so_cm31_mul(q_re(a), b), so_cm31_mul(q_im(a), b));
}
so_qm31 so_qm31_mul(so_qm31 a, so_qm31 b) /* :73-80, R = 2 + i */
{
    so_cm31 ar = q_re(a), ai = q_im(a), br = q_re(b), bi = q_im(b);
    so_cm31 two_i = {2, 1};
    so_cm31 re = so_cm31_add(so_cm31_mul(ar, br), so_cm31_mul(so_cm31_mul(ai, bi), two_i));
    so_cm31 im = so_cm31_add(so_cm31_mul(ar, bi), so_cm31_mul(ai, br));
    return q_make(re, im);
}
static so_qm31 qm31_pow2(so_qm31 a) { return so_qm31_mul(a, a); }
int so_qm31_inv(so_qm31 a, so_qm31 *out) /* :87-98 */
{
    so_cm31 ar = q_re(a), ai = q_im(a);
    so_cm31 ar_sq = so_cm31_mul(ar, ar);
    so_cm31 ai_sq = so_cm31_mul(ai, ai);
    so_cm31 ai_sq_dbl = so_cm31_add(ai_sq, ai_sq);
    so_cm31 ai_sq_rev = {so_m31_neg(ai_sq.b), ai_sq.a};
    so_cm31 den = so_cm31_add(ar_sq, cm31_neg(so_cm31_add(ai_sq_dbl, ai_sq_rev)));
    so_cm31 den_inv;
    if (so_cm31_inv(den, &den_inv)) return 1;
    *out = q_make(so_cm31_mul(ar, den_inv), so_cm31_mul(cm31_neg(ai), den_inv));
    return 0;
}
static int qm31_div(so_qm31 a, so_qm31 b, so_qm31 *out) /* :101-104 */
{
    so_qm31 bi;
    if (so_qm31_inv(b, &bi)) return 1;
    *out = so_qm31_mul(a, bi);
    return 0;
}
static int qm31_eq(so_qm31 a, so_qm31 b) /* :117-124 raw word compare */
{
    return a.a == b.a && a.b == b.b && a.c == b.c && a.d == b.d;
}

/* ------------------------------------------------- groups/m31_point.simf:33-97 */
static uint32_t m31_point_dbl_x(uint32_t x) /* :33-37 */
{
    uint32_t x_sq = m31_pow2(x);
    return so_m31_sub(so_m31_add(x_sq, x_sq), 1);
}
so_m31_point so_m31_point_add(so_m31_point l, so_m31_point r) /* :40-46 */
{
    so_m31_point o;
    o.x = so_m31_sub(so_m31_mul(l.x, r.x), so_m31_mul(l.y, r.y));
    o.y = so_m31_add(so_m31_mul(l.x, r.y), so_m31_mul(l.y, r.x));
    return o;
}
so_m31_point so_m31_point_dbl(so_m31_point p) /* :49-55 */
{
    so_m31_point o;
    o.x = m31_point_dbl_x(p.x);
    uint32_t xy = so_m31_mul(p.x, p.y);
    o.y = so_m31_add(xy, xy);
    return o;
}
/* :59-97: double-and-add, least significant bit first, all 32 bits */
so_m31_point so_circle_point_index_to_m31_point(uint32_t index)
{
    so_m31_point res = {1, 0}, cur = {2, 1268011823u};
    for (int i = 0; i < 32; i++) {
        if ((index >> i) & 1) res = so_m31_point_add(res, cur);
        cur = so_m31_point_dbl(cur);
    }
    return res;
}

/* ------------------------------------------------ groups/qm31_point.simf:24-74 */
static so_qm31 qm31_point_dbl_x(so_qm31 x) /* :30-34 */
{
    so_qm31 x_sq = so_qm31_mul(x, x);
    return so_qm31_sub(so_qm31_add(x_sq, x_sq), QM31_ONE);
}
so_qm31_point so_qm31_point_add(so_qm31_point l, so_qm31_point r) /* :37-43 */
{
    so_qm31_point o;
    o.x = so_qm31_sub(so_qm31_mul(l.x, r.x), so_qm31_mul(l.y, r.y));
    o.y = so_qm31_add(so_qm31_mul(l.x, r.y), so_qm31_mul(l.y, r.x));
    return o;
}
so_qm31_point so_qm31_point_add_m31_point(so_qm31_point l, so_m31_point r) /* :68-74 */
{
    so_qm31_point o;
    o.x = so_qm31_sub(so_qm31_mul_m31(l.x, r.x), so_qm31_mul_m31(l.y, r.y));
    o.y = so_qm31_add(so_qm31_mul_m31(l.x, r.y), so_qm31_mul_m31(l.y, r.x));
    return o;
}

/* ------------------------------------------------------ groups/coset.simf:14-52 */
static inline uint32_t jet_shl32(uint8_t s, uint32_t v) { return s >= 32 ? 0 : v << s; }
static inline uint32_t jet_shr32(uint8_t s, uint32_t v) { return s >= 32 ? 0 : v >> s; }

uint32_t so_bit_reverse_position(uint32_t position, uint8_t log_size) /* :20-25 */
{
    uint32_t v = position, r = 0;
    for (int i = 0; i < 32; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    uint8_t shift = (uint8_t)(32 - log_size);
    return jet_shr32(shift, r);
}
static uint32_t circle_subgroup_gen(uint8_t log_size) /* :28-31 */
{
    uint8_t shift = (uint8_t)(31 - log_size);
    return jet_shl32(shift, 1);
}
uint32_t so_circle_point_index_add(uint32_t a, uint32_t b) { return (a + b) & 0x7fffffffu; }
uint32_t so_circle_point_index_mul(uint32_t a, uint32_t b) { return (a * b) & 0x7fffffffu; }
uint32_t so_circle_point_index_neg(uint32_t a) { return (0x80000000u - a) & 0x7fffffffu; }

/* ---------------------------------------------- groups/circle_domain.simf:17-43 */
void so_circle_domain(uint8_t log_size, uint32_t out[3])
{
    uint8_t lm1 = (uint8_t)(log_size - 1), lp1 = (uint8_t)(log_size + 1);
    out[0] = jet_shl32(lm1, 1);
    out[1] = circle_subgroup_gen(lp1);
    out[2] = circle_subgroup_gen(lm1);
}
uint32_t so_circle_position_to_point_index(uint8_t log_size, uint32_t position) /* :28-37 */
{
    uint32_t d[3];
    so_circle_domain(log_size, d);
    if (position < d[0])
        return so_circle_point_index_add(d[1], so_circle_point_index_mul(d[2], position));
    uint32_t pos = position - d[0];
    uint32_t idx = so_circle_point_index_add(d[1], so_circle_point_index_mul(d[2], pos));
    return so_circle_point_index_neg(idx);
}
static so_m31_point circle_position_to_m31_point(uint8_t log_size, uint32_t position) /* :40-43 */
{
    return so_circle_point_index_to_m31_point(so_circle_position_to_point_index(log_size, position));
}

/* ------------------------------------------------ groups/line_domain.simf:18-31 */
uint32_t so_line_position_to_x_coord(uint8_t log_size, uint32_t position)
{
    uint32_t offset = circle_subgroup_gen((uint8_t)(log_size + 2));
    uint32_t step = circle_subgroup_gen(log_size);
    uint32_t index = so_circle_point_index_add(offset, so_circle_point_index_mul(step, position));
    return so_circle_point_index_to_m31_point(index).x;
}

/* ------------------------------------------------------------ channel.simf:31-172 */
void so_channel_init(so_channel *s) { memset(s, 0, sizeof *s); } /* :31 */

static void channel_draw_words(so_channel *s, uint32_t w[8]) /* :36-65 */
{
    sha_ctx c;
    uint8_t out[32];
    sha_init(&c);
    sha_add(&c, s->digest, 32);
    sha_add_u32(&c, s->counter);
    sha_final(&c, out);
    s->counter = s->counter + 1;
    for (int i = 0; i < 8; i++)
        w[i] = ((uint32_t)out[4 * i] << 24) | ((uint32_t)out[4 * i + 1] << 16) |
               ((uint32_t)out[4 * i + 2] << 8) | out[4 * i + 3];
}

#define DBL_P 4294967294u /* :20 */

/* :115-135: for_while with a u8 counter = at most 256 attempts */
static int channel_draw_m31x4(so_channel *s, uint32_t out[4])
{
    for (int it = 0; it < 256; it++) {
        uint32_t w[8];
        channel_draw_words(s, w);
        if (w[0] < DBL_P && w[1] < DBL_P && w[2] < DBL_P && w[3] < DBL_P) {
            for (int i = 0; i < 4; i++) out[i] = m31_reduce(w[i]);
            return 0;
        }
    }
    return 1;
}

int so_channel_draw_qm31(so_channel *s, so_qm31 *out) /* :137-140 */
{
    uint32_t v[4];
    if (channel_draw_m31x4(s, v)) return 1;
    out->a = v[0]; out->b = v[1]; out->c = v[2]; out->d = v[3];
    return 0;
}

/* :143-151; returns 0 ok, 1 draw exhausted, 2 inverse abort */
int so_channel_draw_qm31_point(so_channel *s, so_qm31_point *out)
{
    so_qm31 t, inv;
    if (so_channel_draw_qm31(s, &t)) return 1;
    so_qm31 t_sq = qm31_pow2(t);
    if (so_qm31_inv(so_qm31_add(QM31_ONE, t_sq), &inv)) return 2;
    out->x = so_qm31_mul(so_qm31_sub(QM31_ONE, t_sq), inv);
    out->y = so_qm31_mul(so_qm31_add(t, t), inv);
    return 0;
}

void so_channel_mix_u256(so_channel *s, const uint8_t in[32]) /* :154-161 */
{
    uint8_t n[32];
    sha256_pair(s->digest, in, n);
    memcpy(s->digest, n, 32);
    s->counter = 0;
}

void so_channel_mix_u64(so_channel *s, uint64_t in) /* :164-172 */
{
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, s->digest, 32);
    sha_add_u64(&c, in);
    sha_final(&c, s->digest);
    s->counter = 0;
}

/* ---------------------------------------------------------------- pow.simf:12-36 */
uint32_t so_reverse_bytes_32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
}

int so_check_proof_of_work(so_channel *s, uint64_t nonce, uint64_t target)
{
    so_channel_mix_u64(s, nonce);
    const uint8_t *d = s->digest;
    uint32_t g = ((uint32_t)d[24] << 24) | ((uint32_t)d[25] << 16) | ((uint32_t)d[26] << 8) | d[27];
    uint32_t h = ((uint32_t)d[28] << 24) | ((uint32_t)d[29] << 16) | ((uint32_t)d[30] << 8) | d[31];
    uint64_t value = ((uint64_t)so_reverse_bytes_32(h) << 32) | so_reverse_bytes_32(g);
    return !(value < target); /* assert!(lt_64(value, POW_TARGET_64)) :33 */
}

/* ------------------------------------------------------------ hasher.simf:34-104 */
void so_hash_u32s(const uint32_t *vals, size_t n, uint8_t out[32])
{
    sha_ctx c;
    sha_init(&c);
    for (size_t i = 0; i < n; i++) sha_add_u32(&c, vals[i]);
    sha_final(&c, out);
}

static void sha_add_qm31(sha_ctx *c, so_qm31 v) /* :55-62 */
{
    sha_add_u32(c, v.a); sha_add_u32(c, v.b); sha_add_u32(c, v.c); sha_add_u32(c, v.d);
}

static void hash_node_qm31(so_qm31 v, uint8_t out[32]) /* :100-104 */
{
    sha_ctx c;
    sha_init(&c);
    sha_add_qm31(&c, v);
    sha_final(&c, out);
}

/* ------------------------------------------------------------ merkle.simf:22-44 */
int so_stwo_merkle_verify(const uint8_t leaf[32], uint32_t auth_path, const uint8_t *proof,
                          uint32_t len, const uint8_t root[32])
{
    uint8_t cur[32], nxt[32];
    uint32_t path = auth_path;
    memcpy(cur, leaf, 32);
    for (uint32_t i = 0; i < len; i++) {
        const uint8_t *sib = proof + 32 * (size_t)i;
        if (path & 1) sha256_pair(sib, cur, nxt);
        else sha256_pair(cur, sib, nxt);
        memcpy(cur, nxt, 32);
        path = path / 2;
    }
    if (path != 1) return 1;               /* :42 */
    if (memcmp(cur, root, 32)) return 2;   /* :43 */
    return 0;
}

/* -------------------------------------------------------- evals/commit.simf:20-35 */
int so_evals_commit(so_channel *s, const uint8_t roots[3][32], so_qm31 *cp_alpha)
{
    so_channel_mix_u256(s, roots[0]);
    so_channel_mix_u256(s, roots[1]);
    if (so_channel_draw_qm31(s, cp_alpha)) return 1;
    so_channel_mix_u256(s, roots[2]);
    return 0;
}

/* ----------------------------------------------- evals/composition_poly.simf:27-72 */
so_qm31 so_composition_poly_eval_from_partitions(const so_qm31 p[4]) /* :38-44 */
{
    static const so_qm31 U1 = {0, 1, 0, 0}, U2 = {0, 0, 1, 0}, U3 = {0, 0, 0, 1};
    so_qm31 res = so_qm31_add(p[0], so_qm31_mul(p[1], U1));
    res = so_qm31_add(res, so_qm31_mul(p[2], U2));
    res = so_qm31_add(res, so_qm31_mul(p[3], U3));
    return res;
}

so_qm31 so_composition_poly_eval_from_decomposed(const so_qm31 d[16], so_qm31_point pt) /* :47-59 */
{
    /* (a0,b0,c0,d0,a1,...) : index = 4*coord + part */
    so_qm31 pa[4] = {d[0], d[4], d[8], d[12]};
    so_qm31 pb[4] = {d[1], d[5], d[9], d[13]};
    so_qm31 pc[4] = {d[2], d[6], d[10], d[14]};
    so_qm31 pd[4] = {d[3], d[7], d[11], d[15]};
    so_qm31 cpa = so_composition_poly_eval_from_partitions(pa);
    so_qm31 cpb = so_composition_poly_eval_from_partitions(pb);
    so_qm31 cpc = so_composition_poly_eval_from_partitions(pc);
    so_qm31 cpd = so_composition_poly_eval_from_partitions(pd);
    so_qm31 res = so_qm31_add(cpa, so_qm31_mul(cpb, pt.y));
    res = so_qm31_add(res, so_qm31_mul(cpc, pt.x));
    return so_qm31_add(res, so_qm31_mul(cpd, so_qm31_mul(pt.x, pt.y)));
}

so_qm31 so_vanishing_poly_eval(uint8_t log_size, so_qm31_point pt) /* :66-71, pi_fn :27-35 */
{
    uint8_t n_iter = (uint8_t)(log_size - 1);
    so_qm31 acc = pt.x;
    for (unsigned counter = 0; counter < 256; counter++) {
        if ((uint8_t)counter == n_iter) return acc;
        acc = qm31_point_dbl_x(acc);
    }
    return acc; /* unreachable: counter hits every u8 value */
}

/* ------------------------------------------ constraints/wide_fibonacci.simf:24-62 */
int so_eval_composition_poly(uint8_t log_size, so_qm31_point pt, const so_qm31 *trace_evals,
                             uint32_t n_cols, so_qm31 alpha, so_qm31 *out)
{
    so_qm31 acc = QM31_ZERO, a = QM31_ZERO, b = QM31_ZERO;
    uint8_t skip_2 = 0;
    for (uint32_t k = 0; k < n_cols; k++) {
        so_qm31 c = trace_evals[k];
        if (skip_2 == 2) {
            so_qm31 constraint = so_qm31_sub(c, so_qm31_add(qm31_pow2(b), qm31_pow2(a)));
            acc = so_qm31_add(so_qm31_mul(acc, alpha), constraint);
        } else {
            skip_2 = (uint8_t)(skip_2 + 1);
        }
        a = b;
        b = c;
    }
    so_qm31 van = so_vanishing_poly_eval(log_size, pt);
    return qm31_div(acc, van, out);
}

/* ---------------------------------------------------------- deep/oods.simf:23-39 */
void so_channel_mix_oods_evals(so_channel *s, const so_qm31 *trace, uint32_t n_cols,
                               const so_qm31 cp[16])
{
    sha_ctx c;
    sha_init(&c);
    sha_add(&c, s->digest, 32);
    for (uint32_t k = 0; k < n_cols; k++) sha_add_qm31(&c, trace[k]);
    for (int k = 0; k < 16; k++) sha_add_qm31(&c, cp[k]);
    sha_final(&c, s->digest);
    s->counter = 0;
}

/* ---------------------------------------------------- deep/quotients.simf:15-44 */
int so_deep_quotient_denominator_inverse(so_qm31_point sp, so_m31_point q, so_cm31 *out)
{
    so_cm31 prx = q_re(sp.x), pix = q_im(sp.x), pry = q_re(sp.y), piy = q_im(sp.y);
    so_cm31 dx = cm31_sub_m31(prx, q.x);
    so_cm31 dy = cm31_sub_m31(pry, q.y);
    so_cm31 d = so_cm31_sub(so_cm31_mul(dx, piy), so_cm31_mul(dy, pix));
    return so_cm31_inv(d, out);
}

void so_deep_quotient_interpolant_coefficients(so_qm31_point sp, so_qm31 value, so_qm31 alpha_i,
                                               so_qm31 out[3])
{
    so_cm31 zero = {0, 0};
    so_qm31 a = q_make(zero, cm31_neg(cm31_dbl(q_im(value))));
    so_qm31 b = q_make(zero, cm31_neg(cm31_dbl(q_im(sp.y))));
    so_qm31 a_py = so_qm31_mul(a, sp.y);
    so_qm31 b_val = so_qm31_mul(b, value);
    so_qm31 c = so_qm31_sub(b_val, a_py);
    out[0] = so_qm31_mul(alpha_i, a);
    out[1] = so_qm31_mul(alpha_i, b);
    out[2] = so_qm31_mul(alpha_i, c);
}

so_qm31 so_deep_quotient_nominator(const so_qm31 co[3], so_m31_point q, uint32_t value)
{
    so_qm31 b_val = so_qm31_mul_m31(co[1], value);
    so_qm31 a_py = so_qm31_mul_m31(co[0], q.y);
    return so_qm31_sub(b_val, so_qm31_add(a_py, co[2]));
}

/* ------------------------------------------------------- fri/folding.simf:15-41 */
int so_circle_fold(uint32_t position, so_qm31 f_p, so_qm31 f_neg_p, uint8_t log_size_ex,
                   so_qm31 alpha, so_qm31 *out)
{
    so_m31_point pt =
        circle_position_to_m31_point(log_size_ex, so_bit_reverse_position(position, log_size_ex));
    uint32_t y_inv;
    if (so_m31_inv(pt.y, &y_inv)) return 1;
    so_qm31 f0 = so_qm31_add(f_p, f_neg_p);
    so_qm31 f1 = so_qm31_mul_m31(so_qm31_sub(f_p, f_neg_p), y_inv);
    *out = so_qm31_add(f0, so_qm31_mul(alpha, f1));
    return 0;
}

int so_line_fold(uint32_t position, so_qm31 f_p, so_qm31 f_neg_p, uint8_t log_size_ex,
                 so_qm31 alpha, so_qm31 *out)
{
    uint32_t x =
        so_line_position_to_x_coord(log_size_ex, so_bit_reverse_position(position, log_size_ex));
    uint32_t x_inv;
    if (so_m31_inv(x, &x_inv)) return 1;
    so_qm31 f0 = so_qm31_add(f_p, f_neg_p);
    so_qm31 f1 = so_qm31_mul_m31(so_qm31_sub(f_p, f_neg_p), x_inv);
    *out = so_qm31_add(f0, so_qm31_mul(alpha, f1));
    return 0;
}

/* ------------------------------------------------------- fri/queries.simf:14-43 */
void so_channel_draw_queries_8(so_channel *s, uint32_t mask, uint32_t out[8])
{
    uint32_t w[8];
    channel_draw_words(s, w);
    for (int i = 0; i < 8; i++) out[i] = w[i] & mask;
}

#define STWO_FAIL(stage, layer, query, sub)                                             \
    do {                                                                                \
        if (!status)                                                                    \
            status = ((uint32_t)(stage) << 24) | ((uint32_t)(layer) << 16) |            \
                     ((uint32_t)(query) << 4) | (uint32_t)(sub);                        \
    } while (0)

/* What is asserted once the layer loop is through, as seen from one query (the first failing code in the reference's
 * order, 0 = none): `assert!(jet::eq_8(log_size_ex, 0))` fri/verify.simf:127 -- evaluated once, before the last-layer
 * loop, so it precedes every query's asserts --, then fri_verify_last_layer fri/layers.simf:73-78:
 * `assert!(jet::eq_32(folded_query, 0))` :75 and `assert!(qm31_eq(folded_eval, last_layer_eval))` :76.  The first two
 * are what the repository's own proofs violate: LITERAL mode only. */
static uint32_t stwo_fri_tail(int mode, uint8_t log_size_ex, uint32_t q, uint32_t folded_query, so_qm31 eval, so_qm31 last)
{
    uint32_t status = 0;
    if (mode == SO_MODE_LITERAL && log_size_ex != 0) STWO_FAIL(8, 0, 0, 0);
    if (mode == SO_MODE_LITERAL && folded_query != 0) STWO_FAIL(9, 0, q, 0);
    if (!qm31_eq(eval, last)) STWO_FAIL(9, 0, q, 1);
    return status;
}

/* the same for a caller that has (lde_log, n_layers) instead of the running u8: fri/verify.simf:73-74 subtracts 1 per
 * layer in 8-bit arithmetic */
uint32_t so_stwo_fri_tail(int mode, uint32_t lde_log, uint32_t n_layers, uint32_t q, uint32_t folded_query, so_qm31 eval,
                          so_qm31 last)
{
    uint8_t log_size_ex = (uint8_t)lde_log;
    for (uint32_t l = 0; l <= n_layers; l++) log_size_ex = (uint8_t)(log_size_ex - 1);
    return stwo_fri_tail(mode, log_size_ex, q, folded_query, eval, last);
}

/* fri/answers.simf:97-130, literal: ONE batch over trace+CP columns at the OODS point,
 * alpha powers alpha^1.. running across both groups, result * alpha^(n+16). */
static int fri_answer_literal(const so_stwo_cfg *cfg, const so_stwo_proof *p, uint32_t qi,
                              uint32_t query, so_qm31 alpha, so_qm31_point oods, so_qm31 *out)
{
    uint8_t L = (uint8_t)cfg->lde_log;
    so_m31_point dp = circle_position_to_m31_point(L, so_bit_reverse_position(query, L));
    so_cm31 den_inv;
    if (so_deep_quotient_denominator_inverse(oods, dp, &den_inv)) return 1;
    so_qm31 acc = QM31_ZERO, alpha_i = alpha;
    for (uint32_t k = 0; k < cfg->n_cols; k++) {
        so_qm31 co[3];
        so_deep_quotient_interpolant_coefficients(oods, p->oods_trace[k], alpha_i, co);
        acc = so_qm31_add(acc, so_deep_quotient_nominator(co, dp, p->trace_vals[qi * cfg->n_cols + k]));
        alpha_i = so_qm31_mul(alpha_i, alpha);
    }
    for (uint32_t k = 0; k < 16; k++) {
        so_qm31 co[3];
        so_deep_quotient_interpolant_coefficients(oods, p->oods_cp[k], alpha_i, co);
        acc = so_qm31_add(acc, so_deep_quotient_nominator(co, dp, p->cp_vals[qi * 16 + k]));
        alpha_i = so_qm31_mul(alpha_i, alpha);
    }
    *out = so_qm31_mul(so_qm31_mul_cm31(acc, den_inv), alpha_i); /* :126 */
    return 0;
}

/* What the reference's fixtures satisfy (docs/batching_samples.md:62-70):
 * two sample batches -- trace columns sampled at P, the 16 CP partition columns at 2P --
 * alpha restarts at alpha^1 per batch; row = b1 * alpha^16 + b2, b_k = num_k * den_inv_k.
 * Built from the same leaf functions (deep/quotients.simf:15-44).  Returns 1+batch on abort. */
static int fri_answer_fixture(const so_stwo_cfg *cfg, const so_stwo_proof *p, uint32_t qi,
                              uint32_t query, so_qm31 alpha, so_qm31_point oods, so_qm31 *out)
{
    uint8_t L = (uint8_t)cfg->lde_log;
    so_m31_point dp = circle_position_to_m31_point(L, so_bit_reverse_position(query, L));
    so_qm31_point oods2 = so_qm31_point_add(oods, oods);
    so_cm31 di1, di2;
    if (so_deep_quotient_denominator_inverse(oods, dp, &di1)) return 1;
    if (so_deep_quotient_denominator_inverse(oods2, dp, &di2)) return 2;
    so_qm31 acc = QM31_ZERO, alpha_i = alpha;
    for (uint32_t k = 0; k < cfg->n_cols; k++) {
        so_qm31 co[3];
        so_deep_quotient_interpolant_coefficients(oods, p->oods_trace[k], alpha_i, co);
        acc = so_qm31_add(acc, so_deep_quotient_nominator(co, dp, p->trace_vals[qi * cfg->n_cols + k]));
        alpha_i = so_qm31_mul(alpha_i, alpha);
    }
    so_qm31 b1 = so_qm31_mul_cm31(acc, di1);
    acc = QM31_ZERO;
    alpha_i = alpha;
    so_qm31 alpha_pow = QM31_ONE;
    for (uint32_t k = 0; k < 16; k++) {
        so_qm31 co[3];
        so_deep_quotient_interpolant_coefficients(oods2, p->oods_cp[k], alpha_i, co);
        acc = so_qm31_add(acc, so_deep_quotient_nominator(co, dp, p->cp_vals[qi * 16 + k]));
        alpha_i = so_qm31_mul(alpha_i, alpha);
        alpha_pow = so_qm31_mul(alpha_pow, alpha);
    }
    so_qm31 b2 = so_qm31_mul_cm31(acc, di2);
    *out = so_qm31_add(so_qm31_mul(b1, alpha_pow), b2);
    return 0;
}

/* Stages I-IV and the query draw of stage V (verifier.simf:36-51): everything that reads only the per-proof part of
 * the witness.  Shared by so_stwo_verify and the minimal-decommitment walk below.  Returns the status so far. */
typedef struct {
    so_channel st;
    so_qm31 cp_alpha, deep_alpha;
    so_qm31_point oods;
    so_qm31 fold_alpha[SO_MAX_LIST + 1];
    uint32_t queries[64];
} stwo_head_out;

static uint32_t stwo_head(const so_stwo_cfg *cfg, const so_stwo_proof *p, stwo_head_out *h, so_stwo_trace *tr)
{
    uint32_t status = 0;
    const uint32_t Q = cfg->n_queries, K = cfg->n_layers, N = cfg->n_cols;
    so_channel st;
    so_qm31 cp_alpha = QM31_ZERO, deep_alpha = QM31_ZERO;
    so_qm31_point oods;
    so_qm31 *fold_alpha = h->fold_alpha;
    uint32_t *queries = h->queries;
    uint32_t draw_ord = 0;

    memset(&oods, 0, sizeof oods);
    memset(h->fold_alpha, 0, sizeof h->fold_alpha);

    /* :36 */
    so_channel_init(&st);
    /* :39 stage I */
    if (so_evals_commit(&st, p->roots, &cp_alpha)) STWO_FAIL(1, 0, 0, draw_ord);
    draw_ord++;
    if (tr) memcpy(tr->digest_after[0], st.digest, 32);

    /* :42 stage II, deep/oods.simf:44-64 */
    {
        int rc = so_channel_draw_qm31_point(&st, &oods);
        if (rc == 1) STWO_FAIL(1, 0, 0, draw_ord);
        if (rc == 2) STWO_FAIL(2, 0, 0, 1);
        draw_ord++;
        so_channel_mix_oods_evals(&st, p->oods_trace, N, p->oods_cp);
        so_qm31 cp_eval = QM31_ZERO;
        if (so_eval_composition_poly((uint8_t)cfg->trace_log, oods, p->oods_trace, N, cp_alpha,
                                     &cp_eval))
            STWO_FAIL(2, 0, 0, 2);
        so_qm31 sampled = so_composition_poly_eval_from_decomposed(p->oods_cp, oods);
        if (!qm31_eq(cp_eval, sampled)) STWO_FAIL(2, 0, 0, 3);
        if (so_channel_draw_qm31(&st, &deep_alpha)) STWO_FAIL(1, 0, 0, draw_ord);
        draw_ord++;
    }
    if (tr) memcpy(tr->digest_after[1], st.digest, 32);

    /* :45 stage III, fri/commit.simf:70-85 */
    for (uint32_t l = 0; l <= K; l++) {
        so_channel_mix_u256(&st, p->fri_roots + 32 * (size_t)l);
        if (so_channel_draw_qm31(&st, &fold_alpha[l])) STWO_FAIL(1, 0, 0, draw_ord);
        draw_ord++;
    }
    {
        sha_ctx c; /* channel_mix_line_poly fri/commit.simf:48-57 */
        sha_init(&c);
        sha_add(&c, st.digest, 32);
        sha_add_qm31(&c, p->last_layer);
        sha_final(&c, st.digest);
        st.counter = 0;
    }
    if (tr) memcpy(tr->digest_after[2], st.digest, 32);

    /* :48 stage IV */
    if (so_check_proof_of_work(&st, p->pow_nonce, cfg->pow_target)) STWO_FAIL(4, 0, 0, 0);
    if (tr) memcpy(tr->digest_after[3], st.digest, 32);

    /* :51 stage V, first half: fri/queries.simf:29-43 */
    {
        uint8_t L = (uint8_t)cfg->lde_log;
        uint32_t mask = jet_shl32(L, 1) - 1;
        for (uint32_t base = 0; base < Q; base += 8) {
            uint32_t q8[8];
            so_channel_draw_queries_8(&st, mask, q8);
            for (uint32_t j = 0; j < 8 && base + j < Q; j++) queries[base + j] = q8[j];
        }
    }
    h->st = st; h->cp_alpha = cp_alpha; h->deep_alpha = deep_alpha; h->oods = oods;
    return status;
}

/* verifier.simf:32-58 */
uint32_t so_stwo_verify(const so_stwo_cfg *cfg, const so_stwo_proof *p, int mode,
                        so_stwo_trace *tr)
{
    uint32_t status = 0;
    const uint32_t Q = cfg->n_queries, K = cfg->n_layers, N = cfg->n_cols;
    stwo_head_out H;
    so_qm31 evals[64];

    if (Q > 64 || K > SO_MAX_LIST) return 0xffffffffu;
    g_hash_kind = cfg->hash == 1 ? 1 : 0;
    status = stwo_head(cfg, p, &H, tr);
    so_channel st = H.st;
    const so_qm31 cp_alpha = H.cp_alpha, deep_alpha = H.deep_alpha;
    const so_qm31_point oods = H.oods;
    so_qm31 *fold_alpha = H.fold_alpha;
    uint32_t *queries = H.queries;

    /* :51 stage V, evals/verify.simf:108-123 */
    {
        uint8_t L = (uint8_t)cfg->lde_log;
        uint32_t domain_size = jet_shl32(L, 1);
        for (uint32_t q = 0; q < Q; q++) {
            uint8_t leaf[32];
            uint32_t auth = queries[q] + domain_size;
            so_hash_u32s(p->trace_vals + (size_t)q * N, N, leaf); /* hasher.simf:85-90 */
            int rc = so_stwo_merkle_verify(leaf, auth, p->trace_paths[q].nodes,
                                           p->trace_paths[q].len, p->roots[1]);
            if (rc) STWO_FAIL(5, 0, q, rc - 1);
            so_hash_u32s(p->cp_vals + (size_t)q * 16, 16, leaf); /* hasher.simf:93-97 */
            rc = so_stwo_merkle_verify(leaf, auth, p->cp_paths[q].nodes, p->cp_paths[q].len,
                                       p->roots[2]);
            if (rc) STWO_FAIL(5, 0, q, 2 + rc - 1);
        }
    }
    if (tr) {
        memcpy(tr->digest_after[4], st.digest, 32);
        tr->cp_alpha = cp_alpha; tr->deep_alpha = deep_alpha; tr->oods_point = oods;
        memcpy(tr->fold_alpha, fold_alpha, sizeof H.fold_alpha);
        memcpy(tr->queries, queries, sizeof(uint32_t) * Q);
    }

    /* :54 stage VI */
    for (uint32_t q = 0; q < Q; q++) {
        evals[q] = QM31_ZERO;
        int rc = mode == SO_MODE_LITERAL
                     ? fri_answer_literal(cfg, p, q, queries[q], deep_alpha, oods, &evals[q])
                     : fri_answer_fixture(cfg, p, q, queries[q], deep_alpha, oods, &evals[q]);
        if (rc) STWO_FAIL(6, 0, q, rc - 1);
        if (tr) tr->answers[q] = evals[q];
    }

    /* :57 stage VII, fri/verify.simf:112-128 */
    uint8_t log_size_ex = (uint8_t)cfg->lde_log;
    for (uint32_t l = 0; l <= K; l++) {
        const uint8_t *root = p->fri_roots + 32 * (size_t)l;
        for (uint32_t q = 0; q < Q; q++) { /* fri/layers.simf:48-70 */
            so_qm31 witness = p->fri_witness[(size_t)l * Q + q];
            const so_path *path = &p->fri_paths[(size_t)l * Q + q];
            uint32_t position;
            so_qm31 e0, e1;
            if ((queries[q] & 1) == 0) { /* adjacent_leaves :29-37 */
                position = queries[q]; e0 = evals[q]; e1 = witness;
            } else {
                position = queries[q] - 1; e0 = witness; e1 = evals[q];
            }
            /* verify_decommitment :40-48 */
            uint8_t l0[32], l1[32], node[32];
            uint32_t domain_size = jet_shl32(log_size_ex, 1);
            hash_node_qm31(e0, l0);
            hash_node_qm31(e1, l1);
            sha256_pair(l0, l1, node);
            uint32_t auth = jet_divide_32(position + domain_size, 2);
            int rc = so_stwo_merkle_verify(node, auth, path->nodes, path->len, root);
            if (rc) STWO_FAIL(7, l, q, rc - 1);
            so_qm31 folded = QM31_ZERO;
            rc = l == 0 ? so_circle_fold(position, e0, e1, log_size_ex, fold_alpha[l], &folded)
                        : so_line_fold(position, e0, e1, log_size_ex, fold_alpha[l], &folded);
            if (rc) STWO_FAIL(7, l, q, 2);
            evals[q] = folded;
            queries[q] = jet_divide_32(position, 2);
        }
        log_size_ex = (uint8_t)(log_size_ex - 1); /* fri/verify.simf:73-74 */
    }
    for (uint32_t q = 0; q < Q; q++) { /* fri/verify.simf:127 (once, before the queries), fri/layers.simf:73-78 */
        const uint32_t c = stwo_fri_tail(mode, log_size_ex, q, queries[q], evals[q], p->last_layer);
        if (c && !status) status = c;
        if (tr) { tr->folded[q] = evals[q]; tr->folded_query[q] = queries[q]; }
    }
    if (tr) tr->final_log_size = log_size_ex;
    g_hash_kind = 0;
    return status;
}

/* ===================================================================== minimal decommitment
 * SURVEY.md 8(f) row 4, second half: "query dedup / sorted multi-proof Merkle (real stwo format)".  The reference
 * presents one full authentication path per query (fri/queries.simf:41 "we do not sort and remove duplicates";
 * scripts/generate_wit.py:36-42 cuts the prover's witness lists per query; merkle.simf:22-44 folds one path).  The
 * external prover the reference's proofs come from -- starkware-libs/stwo, a dependency that is ABSENT from
 * /root/reference (the repository ships two proofs, not the prover) -- emits one decommitment per TREE instead:
 * queries sorted and deduplicated, queried values once per distinct position, and in `hash_witness` only the
 * siblings that cannot be computed from other queried nodes; likewise `fri_witness` holds only the evaluations of
 * the fold pairs' members that are not queried themselves.  PARITY UNPINNED: no bytes or behaviour of that form
 * exist in the reference.  What follows restates the published algorithms
 *   MerkleVerifier::verify            (stwo, crates/prover/src/core/vcs/verifier.rs: layer by layer from the largest
 *                                      layer to the root; a child that was not computed is read from hash_witness,
 *                                      left before right; leftover witnesses are an error; then the root compare)
 *   FriLayerVerifier::extract_evaluation / SparseEvaluation  (stwo, crates/prover/src/core/fri.rs: group the layer's
 *                                      queries by fold pair; a member that is not queried comes from the proof's
 *                                      evals, in order; leftover evals are an error)
 * for the case at hand (all columns of a tree have the leaf layer's size, fold step 1) and anchors the result on the
 * reference's own per-query path: a minimal input M corresponds to the per-query record R(M) in which every omitted
 * sibling / evaluation is the value the walk computes (so_stwo_minimal_expand); status(M) is DEFINED as what
 * so_stwo_verify returns for R(M), and tests/test_minimal.py holds this function against that definition.
 *
 * Minimal record (include/ss_verify.h "minimal record"): head words as in the per-query record, then
 *   n_vals[2]  n_fw[1+K]  n_hw[3+K]                    declared list lengths (distinct positions / evaluations / hashes)
 *   trace_vals[n_vals0][N]  cp_vals[n_vals1][16]       ascending position
 *   fri_wit[l][n_fw[l]][4]                             layer by layer, ascending pair
 *   hash_wit[t][n_hw[t]][8]                            tree by tree (trace, cp, FRI layer 0..K), level by level from the
 *                                                      leaves, ascending position inside a level
 * A tree's lists of the wrong length make every query's check of that tree fail the way a path of the wrong length
 * does (`path == 1`, merkle.simf:42): sub 0 of stage 5 / 7 with query 0 -- in R(M) that tree's path_len is 0.      */
typedef struct {
    uint32_t N, L, Q, K, head, nv, nfw, nhw, data;
} min_map;

static min_map min_map_of(const so_stwo_cfg *c)
{
    min_map m;
    m.N = c->n_cols; m.L = c->lde_log; m.Q = c->n_queries; m.K = c->n_layers;
    m.head = 24 + 4 * m.N + 64 + 8 * (m.K + 1) + 4 + 2;
    m.nv = m.head; m.nfw = m.nv + 2; m.nhw = m.nfw + m.K + 1; m.data = m.nhw + m.K + 3;
    return m;
}

static uint32_t min_tree_len(uint32_t L, uint32_t t) { return t < 2 ? L : L + 1 - t; }

typedef struct {
    const uint32_t *trace_vals, *cp_vals;
    const uint32_t *fri_wit[SO_MAX_LIST + 1];
    const uint32_t *hash_wit[SO_MAX_LIST + 3];
} min_lists;

/* 0, or 2 = no minimal record of this config (SS_STATUS_MALFORMED) */
static int min_parse(const so_stwo_cfg *c, const min_map *m, const uint32_t *rec, size_t words, min_lists *ls)
{
    if (words < m->data) return 2;
    size_t o = m->data;
    if (rec[m->nv] > m->Q || rec[m->nv + 1] > m->Q) return 2;
    ls->trace_vals = rec + o; o += (size_t)rec[m->nv] * m->N;
    ls->cp_vals = rec + o;    o += (size_t)rec[m->nv + 1] * 16;
    for (uint32_t l = 0; l <= m->K; l++) {
        if (rec[m->nfw + l] > m->Q) return 2;
        ls->fri_wit[l] = rec + o; o += (size_t)rec[m->nfw + l] * 4;
    }
    for (uint32_t t = 0; t < m->K + 3; t++) {
        if (rec[m->nhw + t] > m->Q * min_tree_len(m->L, t)) return 2;
        ls->hash_wit[t] = rec + o; o += (size_t)rec[m->nhw + t] * 8;
    }
    (void)c;
    return o == words ? 0 : 2;
}

static void words_to_bytes(const uint32_t *w, size_t n_words, uint8_t *out)
{
    for (size_t i = 0; i < n_words; i++) {
        out[4 * i] = (uint8_t)(w[i] >> 24); out[4 * i + 1] = (uint8_t)(w[i] >> 16);
        out[4 * i + 2] = (uint8_t)(w[i] >> 8); out[4 * i + 3] = (uint8_t)w[i];
    }
}

static void bytes_to_words(const uint8_t *b, size_t n_words, uint32_t *out)
{
    for (size_t i = 0; i < n_words; i++)
        out[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
}

#define MIN_MAX_NODES 128 /* both members of up to 64 fold pairs */
typedef struct {
    uint32_t n;
    uint32_t pos[MIN_MAX_NODES];
    uint8_t hash[MIN_MAX_NODES][32];
    uint8_t sib[MIN_MAX_NODES][32]; /* what the node was hashed with */
} min_level;

/* MerkleVerifier::verify for one tree: `cur` holds the queried nodes of the lowest layer (ascending, distinct).
 * Layer by layer: a parent takes each child from the layer below if it was computed there and from the witness
 * otherwise (left first).  keep (may be NULL) receives every layer's nodes and siblings, lowest layer first.
 * 0 ok, 1 witness too short, 2 witness too long, 3 root mismatch. */
static int merkle_multi_verify(min_level *cur, uint32_t levels, const uint32_t *wit_words, uint32_t n_wit,
                               const uint8_t root[32], min_level *keep)
{
    uint32_t used = 0;
    for (uint32_t lvl = 0; lvl < levels; lvl++) {
        min_level nxt;
        nxt.n = 0;
        for (uint32_t i = 0; i < cur->n;) {
            const uint32_t parent = cur->pos[i] >> 1;
            uint8_t w[32];
            const uint8_t *left, *right;
            uint32_t a = i, b = i; /* the computed children */
            if (!(cur->pos[i] & 1)) {
                left = cur->hash[i];
                if (i + 1 < cur->n && cur->pos[i + 1] == 2 * parent + 1) { b = i + 1; right = cur->hash[b]; }
                else {
                    if (used == n_wit) return 1;
                    words_to_bytes(wit_words + 8 * (size_t)used++, 8, w);
                    right = w;
                }
            } else {
                if (used == n_wit) return 1;
                words_to_bytes(wit_words + 8 * (size_t)used++, 8, w);
                left = w;
                right = cur->hash[i];
            }
            memcpy(cur->sib[a], a == b ? ((cur->pos[i] & 1) ? left : right) : cur->hash[b], 32);
            if (b != a) memcpy(cur->sib[b], cur->hash[a], 32);
            sha256_pair(left, right, nxt.hash[nxt.n]);
            nxt.pos[nxt.n++] = parent;
            i = b + 1;
        }
        if (keep) keep[lvl] = *cur;
        *cur = nxt;
    }
    if (used != n_wit) return 2;
    if (cur->n != 1 || memcmp(cur->hash[0], root, 32)) return 3;
    return 0;
}

static void sort_unique(uint32_t *v, uint32_t *n)
{
    for (uint32_t i = 1; i < *n; i++) {
        uint32_t x = v[i], j = i;
        while (j && v[j - 1] > x) { v[j] = v[j - 1]; j--; }
        v[j] = x;
    }
    uint32_t m = 0;
    for (uint32_t i = 0; i < *n; i++)
        if (!m || v[m - 1] != v[i]) v[m++] = v[i];
    *n = m;
}

static uint32_t find_pos(const uint32_t *v, uint32_t n, uint32_t x)
{
    for (uint32_t i = 0; i < n; i++)
        if (v[i] == x) return i;
    return 0xffffffffu;
}

static void min_head_proof(const min_map *m, const uint32_t *rec, so_stwo_proof *p, so_qm31 *oods_trace, uint8_t *fri_roots)
{
    memset(p, 0, sizeof *p);
    words_to_bytes(rec, 24, &p->roots[0][0]);
    for (uint32_t k = 0; k < m->N; k++) memcpy(&oods_trace[k], rec + 24 + 4 * k, 16);
    p->oods_trace = oods_trace;
    memcpy(p->oods_cp, rec + 24 + 4 * m->N, 256);
    words_to_bytes(rec + 24 + 4 * m->N + 64, 8 * (m->K + 1), fri_roots);
    p->fri_roots = fri_roots;
    memcpy(&p->last_layer, rec + m->head - 6, 16);
    p->pow_nonce = ((uint64_t)rec[m->head - 2] << 32) | rec[m->head - 1];
}

/* The walk.  rec_out == NULL: verify, return the status.  rec_out != NULL: also write R(M), the per-query record
 * (include/ss_verify.h) this minimal record corresponds to.  0xffffffff = unsupported config, 2 = malformed. */
static uint32_t min_walk(const so_stwo_cfg *cfg, const uint32_t *rec, size_t words, int mode, uint32_t *rec_out)
{
    uint32_t status = 0;
    const min_map m = min_map_of(cfg);
    const uint32_t Q = m.Q, K = m.K, N = m.N, L = m.L;
    if (Q > 64 || Q < 1 || K > SO_MAX_LIST || L > 31 || K + 1 >= L || N > 1024) return 0xffffffffu;
    /* per-query record offsets */
    const uint32_t qstride = N + 16 + 16 * L, fbase = m.head + Q * qstride;
    uint32_t foff[SO_MAX_LIST + 1], o = 0;
    for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
    const uint32_t tbase = fbase + o, rec_words = tbase + (K + 3) * Q;
    if (rec_out) memset(rec_out, 0, (size_t)rec_words * 4);
    min_lists ls;
    if (min_parse(cfg, &m, rec, words, &ls)) return 2;
    g_hash_kind = cfg->hash == 1 ? 1 : 0;

    so_stwo_proof hp;
    so_qm31 *oods_trace = malloc(sizeof(so_qm31) * N);
    uint8_t fri_roots[32 * (SO_MAX_LIST + 1)];
    min_head_proof(&m, rec, &hp, oods_trace, fri_roots);
    stwo_head_out H;
    status = stwo_head(cfg, &hp, &H, 0);
    uint32_t *queries = H.queries;
    if (rec_out) memcpy(rec_out, rec, (size_t)m.head * 4);

    /* ---- stage V: the trace and composition trees (MerkleVerifier::verify) */
    uint32_t uq[64], nu = Q;
    memcpy(uq, queries, sizeof(uint32_t) * Q);
    sort_unique(uq, &nu);
    uint32_t *tv = calloc((size_t)Q * N + 1, 4), *cv = calloc((size_t)Q * 16 + 1, 4); /* per-query values, as R(M) holds them */
    min_level *keep = malloc(sizeof(min_level) * 32);
    for (uint32_t t = 0; t < 2; t++) {
        const uint32_t ncol = t == 0 ? N : 16, *vals = t == 0 ? ls.trace_vals : ls.cp_vals;
        int bad = rec[m.nv + t] != nu; /* TooFewQueriedValues / TooManyQueriedValues */
        int rc = 0;
        if (!bad) {
            min_level cur;
            cur.n = nu;
            for (uint32_t i = 0; i < nu; i++) {
                cur.pos[i] = uq[i];
                so_hash_u32s(vals + (size_t)i * ncol, ncol, cur.hash[i]); /* hasher.simf:85-97 */
            }
            rc = merkle_multi_verify(&cur, L, ls.hash_wit[t], rec[m.nhw + t], hp.roots[1 + t], keep);
            bad = rc == 1 || rc == 2;
        }
        if (bad) STWO_FAIL(5, 0, 0, 2 * t);
        else if (rc == 3) STWO_FAIL(5, 0, 0, 2 * t + 1);
        for (uint32_t q = 0; q < Q && !bad; q++) {
            const uint32_t i = find_pos(uq, nu, queries[q]);
            memcpy((t == 0 ? tv : cv) + (size_t)q * ncol, vals + (size_t)i * ncol, (size_t)ncol * 4);
            if (rec_out) {
                uint32_t *dst = rec_out + m.head + q * qstride;
                memcpy(dst + (t == 0 ? 0 : N), vals + (size_t)i * ncol, (size_t)ncol * 4);
                for (uint32_t lvl = 0; lvl < L; lvl++) {
                    const uint32_t j = find_pos(keep[lvl].pos, keep[lvl].n, queries[q] >> lvl);
                    bytes_to_words(keep[lvl].sib[j], 8, dst + N + 16 + t * 8 * L + 8 * lvl);
                }
                rec_out[tbase + t * Q + q] = L;
            }
        }
    }

    /* ---- stage VI (fri/answers.simf:97-130) on the per-query values */
    so_stwo_proof vp = hp;
    vp.trace_vals = tv;
    vp.cp_vals = cv;
    so_qm31 evals[64];
    for (uint32_t q = 0; q < Q; q++) {
        evals[q] = QM31_ZERO;
        int rc = mode == SO_MODE_LITERAL ? fri_answer_literal(cfg, &vp, q, queries[q], H.deep_alpha, H.oods, &evals[q])
                                         : fri_answer_fixture(cfg, &vp, q, queries[q], H.deep_alpha, H.oods, &evals[q]);
        if (rc) STWO_FAIL(6, 0, q, rc - 1);
    }

    /* ---- stage VII: per layer the sparse evaluation, the layer's tree, the folds */
    uint8_t log_size_ex = (uint8_t)L;
    for (uint32_t l = 0; l <= K; l++) {
        /* the layer's queries (positions, ascending, distinct) with the value some chain carries there */
        uint32_t lp[64], nl = Q;
        memcpy(lp, queries, sizeof(uint32_t) * Q);
        sort_unique(lp, &nl);
        so_qm31 lv[64];
        for (uint32_t i = 0; i < nl; i++) lv[i] = evals[find_pos(queries, Q, lp[i])];
        /* extract_evaluation: both members of every fold pair, queried value or the next proof evaluation */
        min_level cur;
        so_qm31 member[MIN_MAX_NODES];
        cur.n = 0;
        uint32_t used = 0;
        int bad = 0;
        const uint32_t n_fw = rec[m.nfw + l];
        for (uint32_t i = 0; i < nl && !bad;) {
            const uint32_t pair = lp[i] >> 1;
            for (uint32_t pos = 2 * pair; pos <= 2 * pair + 1; pos++) {
                so_qm31 e;
                if (i < nl && lp[i] == pos) e = lv[i++];
                else if (used < n_fw) { memcpy(&e, ls.fri_wit[l] + 4 * (size_t)used++, 16); }
                else { bad = 1; break; } /* InsufficientWitness */
                member[cur.n] = e;
                cur.pos[cur.n] = pos;
                hash_node_qm31(e, cur.hash[cur.n]); /* hasher.simf:100-104 */
                cur.n++;
            }
        }
        if (used != n_fw) bad = 1;
        const min_level leaves = cur;
        int rc = 0;
        const uint32_t logl = L - l;
        if (!bad) {
            rc = merkle_multi_verify(&cur, logl, ls.hash_wit[2 + l], rec[m.nhw + 2 + l], fri_roots + 32 * (size_t)l, keep);
            bad = rc == 1 || rc == 2;
        }
        if (bad) { STWO_FAIL(7, l, 0, 0); break; } /* (what follows is ordered behind this code) */
        if (rc == 3) STWO_FAIL(7, l, 0, 1);
        for (uint32_t q = 0; q < Q; q++) { /* fri/layers.simf:48-70 */
            const uint32_t position = queries[q] & ~1u;
            const uint32_t i0 = find_pos(leaves.pos, leaves.n, position);
            const so_qm31 e0 = member[i0], e1 = member[i0 + 1];
            if (rec_out) {
                uint32_t *dst = rec_out + fbase + foff[l] + q * (4 + 8 * (logl - 1));
                const so_qm31 w = (queries[q] & 1) ? e0 : e1; /* adjacent_leaves :29-37 */
                memcpy(dst, &w, 16);
                for (uint32_t lvl = 0; lvl + 1 < logl; lvl++) {
                    const uint32_t j = find_pos(keep[lvl + 1].pos, keep[lvl + 1].n, queries[q] >> (lvl + 1));
                    bytes_to_words(keep[lvl + 1].sib[j], 8, dst + 4 + 8 * lvl);
                }
                rec_out[tbase + (2 + l) * Q + q] = logl - 1;
            }
            so_qm31 folded = QM31_ZERO;
            int frc = l == 0 ? so_circle_fold(position, e0, e1, log_size_ex, H.fold_alpha[l], &folded)
                             : so_line_fold(position, e0, e1, log_size_ex, H.fold_alpha[l], &folded);
            if (frc) STWO_FAIL(7, l, q, 2);
            evals[q] = folded;
            queries[q] = jet_divide_32(position, 2);
        }
        log_size_ex = (uint8_t)(log_size_ex - 1);
        if (l == K)
            for (uint32_t q = 0; q < Q; q++) {
                const uint32_t c = stwo_fri_tail(mode, log_size_ex, q, queries[q], evals[q], hp.last_layer);
                if (c && !status) status = c;
            }
    }
    free(oods_trace); free(tv); free(cv); free(keep);
    g_hash_kind = 0;
    return status;
}

uint32_t so_stwo_verify_minimal(const so_stwo_cfg *cfg, const uint32_t *rec, size_t words, int mode)
{
    return min_walk(cfg, rec, words, mode, 0);
}

/* R(M): returns what min_walk returns; record_out (ss_stwo_record_words words) is written unless the input is malformed */
uint32_t so_stwo_minimal_expand(const so_stwo_cfg *cfg, const uint32_t *rec, size_t words, int mode, uint32_t *record_out)
{
    return min_walk(cfg, rec, words, mode, record_out);
}
