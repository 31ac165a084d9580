// Host copy into pinned staging with streaming (non-temporal) stores.
//
// The destination is read next by the DMA engine, not by a core, so it should neither be fetched into the cache
// first (a read for ownership per line) nor push the source out of it.  With plain memcpy the staging copy and the
// DMA reads of the previous buffer compete for host memory and the uploads themselves slow down by a fifth
// (profiles/r03_text_staging_ab.txt: 84.6 k -> 65-71 k proofs/s from proof.json).
#pragma once
#include <emmintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace ss {

inline void copy_streaming(void *dst_, const void *src_, size_t n)
{
    uint8_t *dst = (uint8_t *)dst_;
    const uint8_t *src = (const uint8_t *)src_;
    const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;  // up to the first 16-byte boundary of the destination
    if (head >= n) { memcpy(dst, src, n); return; }
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
        _mm_stream_si128((__m128i *)(dst + i), a);
        _mm_stream_si128((__m128i *)(dst + i + 16), b);
        _mm_stream_si128((__m128i *)(dst + i + 32), c);
        _mm_stream_si128((__m128i *)(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}

}  // namespace ss
