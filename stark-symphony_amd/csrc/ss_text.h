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

#include "ss_abi.h"

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

// ---- minimal proof.json (one sorted, deduplicated decommitment per tree: csrc/ss_minimal.h, formats.stwo_minimal_to_json)
// The same idea with more lists and nothing in the text that says how long they are.  2 K + 6 lists have lengths that are
// data -- in text order: the hash_witness of the trace and of the composition tree, the two flat queried_values lists, then
// per FRI layer its fri_witness and its hash_witness -- and every entry of a list looks the same (a hash: 66 skeleton bytes
// and 32 numbers; a value: 2 and 1; a QM31 evaluation: 14 and 4).  The template is the FULL-LENGTH text (every list at the
// capacity the config allows: csrc/ss_minimal.h min_max_words) over a minimal record in CAPACITY form (every list at a fixed
// base with room for that many entries).  The lengths of one text are FOUND first: three member names are landmarks --
// "hash_witness", "column_witness", "proof_of_work" sit right behind / in front of the lists -- and the number of NUMBERS in
// front of each landmark (a by-product of the tokenizer's scan) gives every list length by subtraction.  That is only a
// guess: the place pass compares the whole text with the template those lengths imply, so a text whose landmarks lie (a
// member name inside a string value, another member order) simply fails the comparison and goes to the host reader.
constexpr uint32_t kMaxTextLists = 68;  // 2 K + 6, K <= 30
constexpr uint32_t kMaxLandmarks = 36;  // K + 4 of a kind
enum : uint32_t { kLmHash = 0, kLmColumn = 1, kLmPow = 2 };  // "hash_w.. , "colu.. , "proof_..
struct MinTextInfo {
    uint32_t n_lists, N, Q, L, K;
    uint32_t S[kMaxTextLists];   // skeleton position of the first entry of list j in the full-length text (text order)
    uint32_t T[kMaxTextLists];   // index of its first number
    uint32_t n[kMaxTextLists];   // entries of the full-length list
    uint32_t es[kMaxTextLists];  // skeleton bytes from one entry to the next
    uint32_t et[kMaxTextLists];  // numbers per entry
    uint32_t word[kMaxTextLists];  // the record word that holds this list's length, as the record counts it (rows / witnesses / hashes)
    uint32_t per[kMaxTextLists];   // text entries per counted unit (a trace row is N flat values, a composition row 16)
};
struct MinTextGaps {
    uint32_t skel_len, n_slots;
    uint32_t G[kMaxTextLists], D[kMaxTextLists], Gk[kMaxTextLists], Dk[kMaxTextLists];
};
// text-order list index: 0 / 1 = hash_witness of the trace / composition tree, 2 / 3 = their queried values,
// 4 + 2 l = fri_witness of layer l, 5 + 2 l = its hash_witness
SS_HD inline void min_text_gaps(const MinTextInfo &I, const uint32_t *counts, uint32_t full_skel, uint32_t full_slots,
                                MinTextGaps &g)
{
    uint32_t d = 0, dk = 0;
    for (uint32_t j = 0; j < I.n_lists; j++) {
        // c entries occupy c * es - 1 skeleton bytes (no comma behind the last one), an empty list none
        const uint32_t c = counts[j], kept = c ? c * I.es[j] - 1 : 0, full = I.n[j] * I.es[j] - 1;
        g.G[j] = I.S[j] + kept - d;
        g.Gk[j] = I.T[j] + c * I.et[j] - dk;
        d += full - kept;
        dk += (I.n[j] - c) * I.et[j];
        g.D[j] = d;
        g.Dk[j] = dk;
    }
    g.skel_len = full_skel - d;
    g.n_slots = full_slots - dk;
}
// Landmarks -> list lengths (entries, text order).  h / c / p: the numbers in front of every "hash_w.. / "colu.. / "proof_..
// of the text, ascending.  false = not the landmarks of a minimal proof.json of this config.
SS_HD inline bool min_text_counts(const MinTextInfo &I, const uint32_t *h, uint32_t nh, const uint32_t *c, uint32_t nc,
                                  const uint32_t *p, uint32_t np, uint32_t *counts)
{
    if (nh != I.K + 4 || nc != I.K + 4 || np != 1) return false;
    auto div = [](uint32_t hi, uint32_t lo, uint32_t by, uint32_t cap, uint32_t &out) {
        if (hi < lo || (hi - lo) % by || (hi - lo) / by > cap) return false;
        out = (hi - lo) / by;
        return true;
    };
    // the preprocessed tree's empty lists, then the two trees of the proof
    if (h[0] != I.T[0] || c[0] != I.T[0] || h[1] != I.T[0] || h[2] != c[1]) return false;
    if (!div(c[1], h[1], 32, I.n[0], counts[0]) || !div(c[2], h[2], 32, I.n[1], counts[1])) return false;
    uint32_t rows;  // the two flat value lists together: rows x (N + 16) numbers (they must describe the same positions)
    if (!div(p[0], c[2], I.N + 16, I.Q, rows)) return false;
    counts[2] = rows * I.N;
    counts[3] = rows * 16;
    uint32_t prev = p[0] + 1;  // the nonce
    for (uint32_t l = 0; l <= I.K; l++) {
        if (!div(h[3 + l], prev, 4, I.n[4 + 2 * l], counts[4 + 2 * l])) return false;
        if (!div(c[3 + l], h[3 + l], 32, I.n[5 + 2 * l], counts[5 + 2 * l])) return false;
        prev = c[3 + l] + 32;  // the layer's commitment
    }
    return true;
}
// is a landmark's name at text[i ..)?  (avail = bytes of the text from i on)
SS_HD inline int min_text_landmark(const uint8_t *t, uint32_t avail)
{
    if (avail < 7 || t[0] != '"') return -1;
    if (t[1] == 'h') return (t[2] == 'a' && t[3] == 's' && t[4] == 'h' && t[5] == '_' && t[6] == 'w') ? (int)kLmHash : -1;
    if (t[1] == 'c') return (t[2] == 'o' && t[3] == 'l' && t[4] == 'u') ? (int)kLmColumn : -1;
    if (t[1] == 'p') return (t[2] == 'r' && t[3] == 'o' && t[4] == 'o' && t[5] == 'f' && t[6] == '_') ? (int)kLmPow : -1;
    return -1;
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
    MinTextInfo minfo{};           // format 4 only
    std::vector<uint8_t> skel;     // skel_len bytes + kSkelSlack zeros
    uint32_t skel_len = 0;
    std::vector<TextSlot> slots;
    std::vector<uint32_t> trailer;
    std::vector<uint32_t> fixed;   // (word, value) pairs
    uint32_t record_words = 0, tbase = 0;
    bool ok = false;               // false: no canonical text exists for this config / format (fast path off)
    TextTemplate view() const;
};
// fmt: SS_TEXT_JSON, SS_TEXT_WIT, SS_TEXT_JSON_SHARED (the full-length text over a capacity-form shared record) or
// SS_TEXT_JSON_MINIMAL (the full-length text over a capacity-form minimal record)
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

// Scalar statement of the fast path for format 4 (the minimal proof.json): landmarks, list lengths, gaps, the scan of
// text_scan_reference through the gap maps into a capacity-form minimal record, which `capacity` receives
// (min_max_words words).  minimal_compact turns that into the minimal record of include/ss_verify.h.
bool minimal_text_scan_reference(const ss_stwo_cfg &cfg, const TextTemplateHost &t, const char *text, size_t len,
                                 uint32_t *capacity);
// minimal record <-> its capacity form (every list at a fixed base: the layout of a minimal record whose lists all have
// their largest length).  to_capacity: false when `minimal` is no minimal record of the config (sizes, counts).
bool minimal_to_capacity(const ss_stwo_cfg &cfg, const uint32_t *minimal, size_t words, uint32_t *capacity);
void minimal_compact(const ss_stwo_cfg &cfg, const uint32_t *capacity, std::vector<uint32_t> &minimal);

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
