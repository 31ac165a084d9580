// Canonical proof texts: writers, and the "template" form of a config's text that lets the GPU turn
// proof.json / proof.wit bytes into records without a parse tree (csrc/ss_textdev.hip).
//
// The reference's callers hand the verifier text (stwo-verifier/scripts/generate_wit.py:106-245 reads the
// proof.json schema and prints the .wit of :218-243 that `simfony run --witness` consumes,
// simfony-cli/src/main.rs:163-209).  For one expected config every text an honest producer emits is the
// same byte string except for its NUMBERS (and, outside JSON strings, its whitespace): the k-th number
// of the text always lands in the same word of the record.  A template is that text with every number
// replaced by a marker (the "skeleton") plus, for each marker in order, where the number goes (the
// "slots").  A text is taken on the fast path only if its skeleton equals the template's byte for byte
// and every number is written in canonical form and in range; ANY deviation -- other key order, escapes,
// a float, a leading zero, a byte above 255, another shape -- sends that text to the tree parser of
// ss_ingest.cpp, which alone decides parsed / other config / malformed.  The fast path never rejects.
//
// Tokenizer (identical on host and device; `scan_byte` below is the single definition):
//   alnum = [0-9A-Za-z_];  ws = space \n \t \r;  anything else is punctuation.
//   A maximal alnum run that starts with a digit is a NUMBER token (so "0x1f..", "12" are tokens, "u32",
//   "sha256", "list" are not); one marker byte stands for it in the skeleton.  Every other byte is copied
//   to the skeleton, except whitespace outside JSON strings (json.loads skips exactly these four bytes
//   there; inside a string -- e.g. inside the SimplicityHL literal of a .wit value -- whitespace is kept).
//   A backslash, a control character or a byte >= 0x80 anywhere takes the text off the fast path.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/ss_verify.h"

#ifndef SS_HD
#ifdef __HIPCC__
#define SS_HD __host__ __device__
#else
#define SS_HD
#endif
#endif

namespace ss {

constexpr uint8_t kSkelMark = 0x01;  // stands for a number in a skeleton (control bytes never come from a text)

enum SlotKind : uint32_t {
    kSlotU32 = 0,    // decimal < 2^32            -> record word `dst`
    kSlotByte = 1,   // decimal <= 255            -> byte `dst` of the record (hash byte k of word w: 4w + 3 - k % 4)
    kSlotU64 = 2,    // decimal < 2^64            -> words dst (high), dst + 1 (low)
    kSlotHex256 = 3, // 0x + exactly 64 hex digits -> words dst .. dst + 7
    kSlotConst = 4,  // decimal that must equal `dst` (declared parameters, numbers inside type strings)
    kSlotDec256 = 5, // decimal < 2^256 (stark101 writes its hashes as big integers) -> words dst .. dst + 7, most significant first
};
constexpr uint32_t kMaxTokenBytes = 80;  // no number of the formats is longer (78 decimal digits of a u256, 0x + 64 hex digits)

struct TextSlot {
    uint32_t dst;
    uint32_t kind;
};

// byte classes
SS_HD inline bool txt_is_digit(uint32_t c) { return c - '0' < 10u; }
SS_HD inline bool txt_is_alnum(uint32_t c) { return c - '0' < 10u || (c | 0x20) - 'a' < 26u || c == '_'; }
SS_HD inline bool txt_is_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }
SS_HD inline bool txt_is_bad(uint32_t c) { return c == '\\' || c >= 0x80 || (c < 0x20 && !txt_is_ws(c)); }

// run state between bytes
enum : uint32_t { kRunNone = 0, kRunToken = 1, kRunIdent = 2 };

// What one byte does.  Returns a mask: bit 0 = the byte itself goes to the skeleton, bit 1 = a marker goes to the
// skeleton BEFORE it (a number starts here).  `run` / `in_str` are the states before the byte, updated.
SS_HD inline uint32_t scan_byte(uint32_t c, uint32_t &run, uint32_t &in_str)
{
    if (txt_is_alnum(c)) {
        uint32_t r = 0;
        if (run == kRunNone) {
            run = txt_is_digit(c) ? kRunToken : kRunIdent;
            if (run == kRunToken) r = 2;
        }
        return r | (run == kRunIdent ? 1u : 0u);
    }
    run = kRunNone;
    if (c == '"') { in_str ^= 1; return 1; }
    return (txt_is_ws(c) && !in_str) ? 0u : 1u;
}

// A config's template for one text format.  skel is padded with zeros to skel_pad bytes (the device
// stages 2 KiB windows of it without bounds checks).
struct TextTemplate {
    const uint8_t *skel = nullptr;
    uint32_t skel_len = 0;
    const TextSlot *slots = nullptr;
    uint32_t n_slots = 0;
    uint32_t record_words = 0;
    // words of the record that a canonical text implies without spelling them: stwo's path_len trailer is the
    // run [tbase, tbase + n_trailer) (values in `trailer`); stark101's n_layers / path lengths are scattered:
    // n_fixed (word index, value) pairs in `fixed`
    uint32_t tbase = 0, n_trailer = 0;
    const uint32_t *trailer = nullptr;
    uint32_t n_fixed = 0;
    const uint32_t *fixed = nullptr;
};

constexpr uint32_t kSkelSlack = 4096;  // zero bytes after the skeleton

// ---- shared-path proof.json (every distinct Merkle sibling of a tree once + a "queries" member: formats.stwo_to_json(shared=True))
// Such a text has no fixed skeleton: the K + 3 hash_witness lists hold count_t entries, and the counts follow from the
// query positions the text itself names.  But every entry of a list looks the same, so the text's skeleton is the
// skeleton of the FULL-LENGTH text (every list with its Q * len_t entries) with K + 3 runs of whole entries cut out,
// and the k-th number lands where the (k + numbers cut before it)-th number of the full-length text lands.  The
// template of format 3 is that full-length text, its slots pointing into a shared record in "capacity" form (tree t's
// nodes at a fixed base, room for Q * len_t of them: csrc/ss_shared.h); SharedTextInfo says where the lists sit in it.
constexpr uint32_t kMaxTrees = 34;  // kMaxList + 3 >= K + 3
struct SharedTextInfo {
    uint32_t n_trees, Q, L, K;
    uint32_t entry_skel;        // skeleton bytes from one entry of a hash list to the next ("[" 32 markers, 31 commas "]" ",")
    uint32_t entry_toks;        // numbers per entry (32)
    uint32_t S[kMaxTrees];      // skeleton position of the first entry of tree t's list in the full-length text
    uint32_t T[kMaxTrees];      // index of its first number
    uint32_t n[kMaxTrees];      // entries of the full-length list (Q * len_t)
};
// What the counts of one text cut out, in the text's own (shared) coordinates: a skeleton position p >= G[t] (and below
// the next gap) is position p + D[t] of the full-length skeleton; number k >= Gk[t] is number k + Dk[t].
struct TextGaps {
    uint32_t skel_len, n_slots;  // totals of this text
    uint32_t G[kMaxTrees], D[kMaxTrees], Gk[kMaxTrees], Dk[kMaxTrees];
};
struct TextHint {               // per text of format 3, left by text_hint_kernel
    TextGaps g;
    uint32_t pos[64];           // the positions read from the text's tail (compared with what the place kernel stores)
};

SS_HD inline void shared_text_gaps(const SharedTextInfo &I, const uint32_t *counts, uint32_t full_skel, uint32_t full_slots,
                                   TextGaps &g)
{
    uint32_t d = 0, dk = 0;
    for (uint32_t t = 0; t < I.n_trees; t++) {
        const uint32_t cut = I.n[t] - counts[t];
        g.G[t] = I.S[t] + counts[t] * I.entry_skel - 1 - d;
        g.Gk[t] = I.T[t] + counts[t] * I.entry_toks - dk;
        d += cut * I.entry_skel;
        dk += cut * I.entry_toks;
        g.D[t] = d;
        g.Dk[t] = dk;
    }
    g.skel_len = full_skel - d;
    g.n_slots = full_slots - dk;
}
SS_HD inline uint32_t gap_map(const uint32_t *G, const uint32_t *D, uint32_t n_trees, uint32_t p)
{
    uint32_t d = 0;
    for (uint32_t t = 0; t < n_trees && G[t] <= p; t++) d = D[t];
    return p + d;
}

// The positions a shared-path text names, read BACKWARDS from its end: `... [p0, p1, .., pQ-1] }` with JSON whitespace
// anywhere between the tokens.  tail[0 .. n) are the last n bytes of the text.  Only a first guess: the place pass
// compares the whole text with the template these positions imply and stores the numbers it finds, and the two
// readings must agree.  false = not of that form (the host reader decides).
SS_HD inline bool shared_text_hint(const uint8_t *tail, uint32_t n, uint32_t Q, uint32_t *pos)
{
    uint32_t i = n;
    auto skip = [&]() { while (i && txt_is_ws(tail[i - 1])) i--; };
    auto expect = [&](uint8_t c) { skip(); if (!i || tail[i - 1] != c) return false; i--; return true; };
    if (!expect('}') || !expect(']')) return false;
    for (uint32_t q = Q; q-- > 0;) {
        skip();
        uint64_t v = 0, mul = 1;
        uint32_t digits = 0;
        while (i && txt_is_digit(tail[i - 1]) && digits < 10) { v += mul * (tail[i - 1] - '0'); mul *= 10; i--; digits++; }
        if (!digits || v > 0xffffffffull) return false;
        pos[q] = (uint32_t)v;
        if (q && !expect(',')) return false;
    }
    return expect('[');
}

}  // namespace ss

#include <string>
#include <vector>

namespace ss {

enum TextStyle : int {
    kStyleCompact = 0,  // JSON: "," and ":" (what the external stwo prover / serde_json writes: tests/data/proof.json)
    kStylePython = 1,   // JSON: ", " and ": " (json.dumps default, what formats.stwo_to_json callers get)
};

// Record -> text, byte for byte what formats.py writes (json.dumps(stwo_to_json(p)) resp. stwo_to_wit(p)).
// Only records whose Merkle paths all have the config's lengths can be written; returns false otherwise
// (or for a pow_target that is no 2^(64-b) - 1: proof.json declares pow_bits).
bool stwo_write_json(const ss_stwo_cfg &cfg, const uint32_t *record, TextStyle style, std::string &out);
bool stwo_write_wit(const ss_stwo_cfg &cfg, const uint32_t *record, std::string &out);

// Host-side owner of a template.
struct TextTemplateHost {
    SharedTextInfo sinfo{};        // format 3 only
    std::vector<uint8_t> skel;     // skel_len bytes + kSkelSlack zeros
    uint32_t skel_len = 0;
    std::vector<TextSlot> slots;
    std::vector<uint32_t> trailer;
    std::vector<uint32_t> fixed;   // (word, value) pairs
    uint32_t record_words = 0, tbase = 0;
    bool ok = false;               // false: no canonical text exists for this config / format (fast path off)
    TextTemplate view() const;
};
// fmt: SS_TEXT_JSON, SS_TEXT_WIT or SS_TEXT_JSON_SHARED (the full-length text over a capacity-form shared record)
void stwo_build_template(const ss_stwo_cfg &cfg, int fmt, TextTemplateHost &out);
// shared record (include/ss_verify.h) -> the shared-path proof.json, byte for byte json.dumps(formats.stwo_to_json(p,
// shared=True)); false when `shared` is no shared record of the config
bool stwo_write_json_shared(const ss_stwo_cfg &cfg, const uint32_t *shared, size_t words, TextStyle style, std::string &out);
// minimal record -> the minimal proof.json (formats.stwo_minimal_to_json); false = no minimal record of the config
bool stwo_write_json_minimal(const ss_stwo_cfg &cfg, const uint32_t *minimal, size_t words, TextStyle style, std::string &out);
// Scalar statement of the fast path for format 3: hint from the tail, gaps, the scan of text_scan_reference through the
// gap maps into a capacity-form shared record, the stored positions against the hint, expansion (ss_stwo_unshare_record's
// rule) into `record` (the per-query record).  scratch: shared_capacity_words(cfg) words.
bool shared_text_scan_reference(const ss_stwo_cfg &cfg, const TextTemplateHost &t, const char *text, size_t len,
                                uint32_t *record);

// stark101 (stark101/scripts/fibsquare/prover.py:108,143-167 writes proof.json, stark101/scripts/generate_wit.py:13-30
// the .wit).  The protocol fixes the proof's shape: an LDE domain of 2^13 points, so Merkle paths of 13 siblings for the
// three trace evaluations and of 13 - i for the two openings of FRI layer i, 10 layers (prover.py:94-171;
// stark101/src/verifier.simf:44-388 is that proof).  The template is the text of THAT shape, written into records of
// shape {kS101Layers, kS101Path}; a proof of any other shape goes to the host reader.
constexpr uint32_t kS101Layers = 10, kS101Path = 13;
void s101_build_template(int fmt, TextTemplateHost &out);
// record (shape {kS101Layers, kS101Path}, canonical path lengths) -> text; false if the record's lengths are not canonical
bool s101_write_json(const uint32_t *record, TextStyle style, std::string &out);
bool s101_write_wit(const uint32_t *record, std::string &out);

// Scalar statement of the fast path (what the device kernel computes, byte by byte): true = `text` is a
// canonical text of the template and `record` (record_words words) holds its record; false = not on the fast
// path (record contents unspecified).  Used by the CPU tests and by nothing on the product path.
bool text_scan_reference(const TextTemplate &t, const char *text, size_t len, uint32_t *record);

}  // namespace ss
