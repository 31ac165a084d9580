// Launch interface of the GPU text reader (csrc/ss_textdev.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ss_text.h"

namespace ss {

// what a 1 KiB window of a text says on its own (text_summary_kernel) ...
struct WinSum {
    uint32_t flags;   // bit 0 bad byte, 1 quote parity, 2 decided (holds a non-alnum byte), 3-4 run state behind it,
                      // 5-6 class of its first byte (kRunNone: not alnum)
    uint32_t fixed;   // low 16: skeleton bytes that do not depend on the incoming states; high 16: numbers that start behind the leading run
    uint32_t ws;      // whitespace bytes at even (low 16) / odd (high 16) quote parity counted from the window start
    uint32_t lead;    // length of the leading alnum run (continues whatever run enters the window)
};
// ... and what the scan over a text's windows adds (text_scan_kernel)
struct WinIn {
    uint32_t skel_pos, tok_pos;  // skeleton bytes / numbers before the window
    uint32_t state;              // bit 0: inside a JSON string; bits 1-2: run state entering the window
};

// per text of format 3 (the minimal proof.json; ss_text.h): the landmarks its windows found, the totals the scan saw, the
// gaps the list lengths imply
struct MinHint {
    MinTextGaps g;
    uint32_t n_lm[3];                  // landmarks of each kind found (zeroed by text_index_kernel)
    uint32_t seen_skel, seen_tok;      // skeleton bytes / numbers of the whole text (text_scan_kernel)
    uint32_t lm[3][kMaxLandmarks];     // numbers in front of each landmark, in any order
};

// which template a text is read against (TextParseArgs::fmt / ::tmpl)
enum : uint32_t { kTextJson = 0, kTextWit = 1, kTextShared = 2, kTextMinimal = 3 };

struct TextParseArgs {
    const uint8_t *texts;      // the chunk's texts, each starting at a 16-byte aligned offset; >= kTextSlack readable bytes behind the last
    const uint64_t *offs;      // [n] byte offset of text i in `texts`
    const uint32_t *lens;      // [n] its length
    const uint32_t *win_base;  // [n + 1] windows (ceil(len / 1024)) of the texts before text i
    const uint8_t *fmt;        // [n] 0 = tmpl[0] (proof.json), 1 = tmpl[1] (proof.wit), 2 = tmpl[2] (shared-path proof.json),
                               //     3 = tmpl[3] (minimal proof.json: `records` are capacity-form minimal records then)
    TextTemplate tmpl[4];      // device pointers inside; skel == nullptr: no fast path for that format
    // format 2 (ss_text.h, "shared-path proof.json"): where the hash lists sit in tmpl[2]; per text the positions read
    // from its tail and the gaps they imply; the capacity-form shared records the place pass writes for such texts
    SharedTextInfo sinfo;
    TextHint *hints;           // [n] scratch
    uint32_t *shared_records;  // [n][tmpl[2].record_words]
    // format 3: where the 2 K + 6 lists sit in tmpl[3]; per text the landmarks and what they imply
    MinTextInfo minfo;
    MinHint *mhints;           // [n] scratch (nullptr: no text of format 3)
    uint32_t *win_text;        // [n_windows] scratch: the text a window belongs to
    WinSum *win_sum;           // [n_windows] scratch
    WinIn *win_in;             // [n_windows] scratch
    uint32_t *records;         // [n][record_words]
    uint32_t *outcome;         // [n] 0 = record written on the fast path, 1 = the host reader decides
    uint32_t record_words;
    uint32_t n, n_windows;
};

constexpr size_t kTextSlack = 4096;  // bytes readable behind the last text of a chunk (whole 1 KiB windows + the next)

void launch_text_parse(const TextParseArgs &a, hipStream_t s);
// records re-made by the host reader: src holds n of them back to back, record j goes to slot idx[j] of dst
void launch_text_scatter(size_t n, size_t record_words, const uint32_t *src, const uint32_t *idx, uint32_t *dst, hipStream_t s);

}  // namespace ss
