// Text -> record: native readers of proof.json / proof.wit (see ss_ingest.h).
//
// One small tree parser serves both syntaxes: JSON (objects, arrays, unsigned integers of any
// size up to 2^256, strings) and the SimplicityHL value literals inside a .wit (`( .. )` tuples,
// `[ .. ]` arrays, `list![ .. ]`, decimal / 0x integers with `_` separators; "(x)" without a comma
// is x itself -- generate_wit.py:8 wraps every FriLayer in a redundant pair of parentheses).
// Every acceptance / rejection rule mirrors stark-symphony_amd/formats.py, which tests/test_ingest.py
// holds this file against on the reference's own files, on random proofs and on malformed inputs.
#include "ss_ingest.h"
#include "ss_minimal.h"
#include "ss_shared.h"

#include <emmintrin.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ss_layout.h"

namespace ss {
namespace {

// kBytes32: a JSON list of exactly 32 byte values, kept as its 8 stored hash words (a list to every
// accessor that asks for its length, a hash to get_hash)
enum Kind : uint8_t { kInt, kBig, kList, kObj, kStr, kAlias, kOther, kBytes32 };

struct Node {
    uint8_t kind = kOther, key_len = 0;
    uint32_t val = 0;      // kInt: value; kBig: index into bigs; kList / kObj: children; kStr: text offset
    uint32_t next = 0;     // next sibling (0 = last)
    uint32_t key_off = 0;  // members of an object: their key
    uint32_t len = 0;      // kStr: length
};

struct U256 { uint32_t w[8]; };  // w[0] most significant = word 0 of a stored hash

// JSON string body (already validated by Json::string) -> the bytes json.loads would give, UTF-8.
// Lone surrogates come out as their 3-byte form (they cannot equal any name or digit of the formats).
static void json_unescape(const char *s, size_t n, std::string &out)
{
    out.clear();
    auto hex4 = [](const char *p) {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = p[k];
            v = v * 16 + (uint32_t)(c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10);
        }
        return v;
    };
    auto put = [&](uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xc0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 63))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xe0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 63)));
            out.push_back((char)(0x80 | (cp & 63)));
        } else {
            out.push_back((char)(0xf0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 63)));
            out.push_back((char)(0x80 | ((cp >> 6) & 63))); out.push_back((char)(0x80 | (cp & 63)));
        }
    };
    for (size_t i = 0; i < n;) {
        if (s[i] != '\\') { out.push_back(s[i++]); continue; }
        const char e = s[i + 1];
        if (e == 'u') {
            uint32_t cp = hex4(s + i + 2);
            i += 6;
            if (cp >= 0xd800 && cp < 0xdc00 && i + 6 <= n && s[i] == '\\' && s[i + 1] == 'u') {
                const uint32_t lo = hex4(s + i + 2);
                if (lo >= 0xdc00 && lo < 0xe000) { cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00); i += 6; }
            }
            put(cp);
        } else {
            out.push_back(e == 'b' ? '\b' : e == 'f' ? '\f' : e == 'n' ? '\n' : e == 'r' ? '\r' : e == 't' ? '\t' : e);
            i += 2;
        }
    }
}

struct Tree {
    std::string owned;  // the text itself when it had to be un-escaped first (a .wit value with escapes)
    const char *text = nullptr;
    std::vector<Node> nodes;
    std::vector<U256> bigs;
    bool ok = true;
    uint32_t root = 0;  // accessors use index 0 for "absent": literal trees keep a dummy there

    uint32_t push(Kind k)
    {
        nodes.emplace_back();
        nodes.back().kind = k;
        return (uint32_t)nodes.size() - 1;
    }
    uint32_t resolve(uint32_t i) const
    {
        while (nodes[i].kind == kAlias) i++;
        return i;
    }
    bool is_list(uint32_t i) const { return nodes[i].kind == kList || nodes[i].kind == kBytes32; }
    uint32_t count(uint32_t i) const { return nodes[i].kind == kBytes32 ? 32 : nodes[i].val; }
    // k-th child of list / object i (resolved), 0 if absent (node 0 is the root, never a child)
    uint32_t child(uint32_t i, uint32_t k) const
    {
        if ((nodes[i].kind != kList && nodes[i].kind != kObj) || k >= nodes[i].val) return 0;
        uint32_t c = i + 1;
        while (k--) c = nodes[c].next;
        return resolve(c);
    }
    // first child of list / object i as a cursor for next_child (0 = none).  The schema walkers iterate
    // with cursors only: list lengths come from untrusted text, and child(i, k) inside a loop over k is
    // quadratic (210 s for a 2.9 MB text).
    uint32_t first_child(uint32_t i) const
    {
        return ((nodes[i].kind == kList || nodes[i].kind == kObj) && nodes[i].val) ? i + 1 : 0;
    }
    // the element under cursor c (resolved; 0 when the list is exhausted), cursor advanced
    uint32_t next_child(uint32_t &c) const
    {
        if (!c) return 0;
        const uint32_t r = resolve(c);
        c = nodes[c].next;
        return r;
    }
    uint32_t member(uint32_t obj, const char *key) const
    {
        if (nodes[obj].kind != kObj) return 0;
        const size_t kl = strlen(key);
        uint32_t c = nodes[obj].val ? obj + 1 : 0, found = 0;
        std::string tmp;
        while (c) {  // the LAST member of that name, as Python's json.loads keeps it
            const char *k = text + nodes[c].key_off;
            size_t n = nodes[c].key_len;
            if (memchr(k, '\\', n)) {  // an escaped spelling: compare what it decodes to
                json_unescape(k, n, tmp);
                k = tmp.data();
                n = tmp.size();
            }
            if (n == kl && memcmp(k, key, kl) == 0) found = resolve(c);
            c = nodes[c].next;
        }
        return found;
    }
};

// value of a hex digit (callers only pass characters they have classified as hex digits)
struct HexTable {
    uint8_t v[256];
    constexpr HexTable() : v()
    {
        for (int i = 0; i < 256; i++) v[i] = 0xff;
        for (int i = 0; i < 10; i++) v['0' + i] = (uint8_t)i;
        for (int i = 0; i < 6; i++) { v['a' + i] = (uint8_t)(10 + i); v['A' + i] = (uint8_t)(10 + i); }
    }
};
constexpr HexTable kHexTable;
#define kHexVal kHexTable.v

// ------------------------------------------------------------------------------ numbers
// digits[0..n) decimal -> node.  Values below 2^32 are kInt, others kBig; >= 2^256 is an error.
bool number_node(Tree &t, uint32_t idx, const char *digits, size_t n, int base)
{
    if (base == 10 && n <= 9) {  // bytes, field words below 10^9: most numbers of a proof.json
        uint32_t v = 0;
        for (size_t i = 0; i < n; i++) v = v * 10 + (uint32_t)(digits[i] - '0');
        t.nodes[idx].kind = kInt;
        t.nodes[idx].val = v;
        return true;
    }
    uint32_t limb[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // little endian
    if (base == 16) {
        while (n && *digits == '0') { digits++; n--; }
        if (n > 64) return false;
        // digit i of n sits at position 64 - n + i of the zero-extended 64-digit number; word w of the
        // big-endian 8-word form holds positions 8w .. 8w + 7
        uint32_t be[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t pos = 64 - n;
        for (size_t i = 0; i < n; i++, pos++) be[pos >> 3] = (be[pos >> 3] << 4) | kHexVal[(unsigned char)digits[i]];
        for (int l = 0; l < 8; l++) limb[l] = be[7 - l];
    } else {
        if (n > 78 + 16) {  // longer than any u256, allowing for leading zeros
            while (n && *digits == '0') { digits++; n--; }
            if (n > 78) return false;
        }
        size_t i = 0;
        while (i < n) {
            const size_t k = n - i < 9 ? n - i : 9;
            uint32_t chunk = 0, mul = 1;
            for (size_t j = 0; j < k; j++) { chunk = chunk * 10 + (uint32_t)(digits[i + j] - '0'); mul *= 10; }
            uint64_t carry = chunk;
            for (int l = 0; l < 8; l++) {
                const uint64_t v = (uint64_t)limb[l] * mul + carry;
                limb[l] = (uint32_t)v;
                carry = v >> 32;
            }
            if (carry) return false;  // >= 2^256
            i += k;
        }
    }
    bool small = true;
    for (int l = 1; l < 8; l++) small &= limb[l] == 0;
    Node &nd = t.nodes[idx];
    if (small) {
        nd.kind = kInt;
        nd.val = limb[0];
    } else {
        U256 b;
        for (int l = 0; l < 8; l++) b.w[l] = limb[7 - l];
        nd.kind = kBig;
        nd.val = (uint32_t)t.bigs.size();
        t.bigs.push_back(b);
    }
    return true;
}

constexpr int kMaxDepth = 48;

// ------------------------------------------------------------------------------- JSON
// A hash as both writers print it -- "[b,b,..]" or "[b, b, ..]", 32 byte values; three quarters of a proof.json (nearly
// all of a minimal one) are such lists.  A digit loop is serial -- where a value starts depends on how long the one before
// it was -- and mispredicts on most values, so: (1) the digit bytes of the next 64 text bytes as a bit mask (SSE2, the
// x86-64 baseline), whose rising edges are the 32 starts; (2) every value from one 4-byte read at its start, independent
// of its neighbours, with the separators checked against the starts.  false = not of that exact shape (p untouched): the
// caller's general loop decides.
static inline uint64_t digit_mask64(const char *p)
{
    uint64_t m = 0;
    const __m128i zero = _mm_set1_epi8('0'), nine = _mm_set1_epi8(9);
    for (int k = 0; k < 4; k++) {
        const __m128i t = _mm_sub_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * k)), zero);
        m |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(t, nine), t)) << (16 * k);
    }
    return m;
}

static inline bool bytes32_fast(const char *&pp, const char *end, uint32_t *w)
{
    const char *p = pp;
    if (end - p < 3 * 64 + 8 || *p != '[') return false;  // room for every read below (a hash is at most 1 + 32 * 5 bytes)
    uint32_t at[32], n_at = 0;
    uint64_t carry = 0;
    for (uint32_t c = 0; c < 3 && n_at < 32; c++) {
        const uint64_t m = digit_mask64(p + 64 * c);
        uint64_t st = m & ~((m << 1) | carry);
        carry = m >> 63;
        while (st && n_at < 32) {
            at[n_at++] = 64 * c + (uint32_t)__builtin_ctzll(st);
            st &= st - 1;
        }
    }
    if (n_at < 32 || at[0] != 1) return false;
    uint32_t bad = 0, acc = 0, e = 0;
    for (uint32_t k = 0; k < 32; k++) {
        uint32_t x;
        memcpy(&x, p + at[k], 4);
        const uint32_t t = x ^ 0x30303030u;                                   // digits -> 0..9
        const uint32_t other = ((t + 0x76767676u) | t) & 0x80808080u;         // bytes that are no digit
        const uint32_t n = (uint32_t)__builtin_ctz(other | 0x80000000u) >> 3;  // 1..3 digits (4: other == 0, flagged below)
        const uint32_t u = t << (8 * (3 - n));                                // right-aligned: hundreds, tens, units
        const uint32_t v = (u & 0xff) * 100 + ((u >> 8) & 0xff) * 10 + ((u >> 16) & 0xff);
        bad |= (other == 0) | (v > 255) | ((n > 1) & ((t & 0xff) == 0));      // (no leading zeros in JSON)
        acc = (acc << 8) | v;
        if ((k & 3) == 3) w[k >> 2] = acc;
        e = at[k] + n;
        if (k < 31) {
            const uint32_t gap = at[k + 1] - e;                               // "," or ", "
            bad |= (p[e] != ',') | (gap == 2 ? p[e + 1] != ' ' : gap != 1);
        }
    }
    if (bad || p[e] != ']') return false;
    pp = p + e + 1;
    return true;
}

struct Json {
    Tree &t;
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail() { t.ok = false; return false; }

    // A string body up to the closing quote, held to the JSON grammar (what json.loads enforces):
    // no raw control characters, only the defined escapes, well-formed UTF-8.  Escapes are validated,
    // never decoded: no key or literal of the formats contains one.
    bool string(uint32_t &off, uint32_t &len)
    {
        p++;  // opening quote
        const char *s = p;
        while (p < end && *p != '"') {
            const unsigned char c = (unsigned char)*p;
            if (c < 0x20) return fail();
            if (c == '\\') {
                if (p + 1 >= end) return fail();
                const char e = p[1];
                if (e == 'u') {
                    if (end - p < 6) return fail();
                    for (int k = 2; k < 6; k++)
                        if (kHexVal[(unsigned char)p[k]] == 0xff) return fail();
                    p += 6;
                } else if (e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't') {
                    p += 2;
                } else {
                    return fail();
                }
                continue;
            }
            if (c >= 0x80) {  // UTF-8: lead byte, continuation count, no overlongs / surrogates / > U+10FFFF
                int n = c >= 0xf0 ? 3 : c >= 0xe0 ? 2 : c >= 0xc2 ? 1 : -1;
                if (n < 0 || c > 0xf4 || end - p <= n) return fail();
                const unsigned char c1 = (unsigned char)p[1];
                if ((c1 & 0xc0) != 0x80) return fail();
                if ((c == 0xe0 && c1 < 0xa0) || (c == 0xed && c1 > 0x9f) || (c == 0xf0 && c1 < 0x90) || (c == 0xf4 && c1 > 0x8f))
                    return fail();
                for (int k = 2; k <= n; k++)
                    if (((unsigned char)p[k] & 0xc0) != 0x80) return fail();
                p += n + 1;
                continue;
            }
            p++;
        }
        if (p >= end) return fail();
        off = (uint32_t)(s - t.text);
        len = (uint32_t)(p - s);
        p++;
        return true;
    }

    // the rest of a number after its integer digits: optional fraction and exponent, each with at
    // least one digit (JSON grammar); returns false on a malformed tail
    bool number_tail(bool &is_float)
    {
        is_float = false;
        if (p < end && *p == '.') {
            is_float = true;
            p++;
            if (p >= end || *p < '0' || *p > '9') return false;
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            is_float = true;
            p++;
            if (p < end && (*p == '+' || *p == '-')) p++;
            if (p >= end || *p < '0' || *p > '9') return false;
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        return true;
    }
    bool word(const char *w)
    {
        const size_t n = strlen(w);
        if ((size_t)(end - p) < n || memcmp(p, w, n) != 0) return false;
        p += n;
        return true;
    }

    uint32_t value(int depth)
    {
        ws();
        if (p >= end || depth > kMaxDepth) { fail(); return 0; }
        const char ch = *p;
        if (ch == '[') {
            // Three quarters of a proof.json are hashes written as lists of 32 byte values: read such a
            // list straight into one node (kind kBig, the 8 stored words) instead of 33.  Anything
            // else -- other lengths, values above 255, nesting -- falls through to the general path.
            const char *q = p;
            uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int k = 0;
            bool ok = true;
            if (bytes32_fast(q, end, w)) { k = 32; q--; }  // (q on the closing bracket, as the loop below leaves it)
            else q = p + 1;
            while (ok && k < 32) {
                while (q < end && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) q++;
                uint32_t v = 0;
                int nd = 0;
                const char first = q < end ? *q : 0;
                while (q < end && *q >= '0' && *q <= '9' && nd < 4) { v = v * 10 + (uint32_t)(*q - '0'); q++; nd++; }
                if (nd == 0 || nd > 3 || v > 255 || (nd > 1 && first == '0')) { ok = false; break; }
                w[k >> 2] = (w[k >> 2] << 8) | v;
                k++;
                while (q < end && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) q++;
                if (k < 32) { if (q < end && *q == ',') q++; else ok = false; }
            }
            if (ok && q < end && *q == ']') {
                const uint32_t idx = t.push(kBytes32);
                U256 b;
                memcpy(b.w, w, 32);
                t.nodes[idx].val = (uint32_t)t.bigs.size();
                t.bigs.push_back(b);
                p = q + 1;
                return idx;
            }
        }
        if (ch == '[' || ch == '{') {
            const bool obj = ch == '{';
            const uint32_t idx = t.push(obj ? kObj : kList);
            p++;
            uint32_t prev = 0, n = 0;
            ws();
            if (p < end && *p == (obj ? '}' : ']')) { p++; return idx; }
            while (t.ok) {
                uint32_t ko = 0, kl = 0;
                if (obj) {
                    ws();
                    if (p >= end || *p != '"' || !string(ko, kl)) { fail(); break; }
                    ws();
                    if (p >= end || *p != ':') { fail(); break; }
                    p++;
                }
                const uint32_t c = value(depth + 1);
                if (!t.ok) break;
                if (obj) {
                    t.nodes[c].key_off = ko;
                    t.nodes[c].key_len = kl > 255 ? 0 : (uint8_t)kl;  // no member name is that long: matches nothing
                }
                if (prev) t.nodes[prev].next = c;
                prev = c;
                n++;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == (obj ? '}' : ']')) { p++; break; }
                fail();
            }
            t.nodes[idx].val = n;
            return idx;
        }
        if (ch == '"') {
            const uint32_t idx = t.push(kStr);
            uint32_t off = 0, len = 0;
            if (!string(off, len)) return idx;
            t.nodes[idx].val = off;
            t.nodes[idx].len = len;
            return idx;
        }
        if (ch >= '0' && ch <= '9') {
            const char *s = p;
            while (p < end && *p >= '0' && *p <= '9') p++;
            const size_t nd = (size_t)(p - s);
            const uint32_t idx = t.push(kOther);
            if (nd > 1 && *s == '0') { fail(); return idx; }  // JSON: no leading zeros
            bool is_float;
            if (!number_tail(is_float)) { fail(); return idx; }
            if (is_float) return idx;  // a float: never a witness word
            if (!number_node(t, idx, s, nd, 10)) t.nodes[idx].kind = kOther;
            return idx;
        }
        // true / false / null, negative numbers, and json.loads' NaN / Infinity: syntactically fine,
        // never a witness value
        const uint32_t idx = t.push(kOther);
        if (ch == '-') {
            p++;
            if (word("Infinity")) return idx;
            const char *s = p;
            while (p < end && *p >= '0' && *p <= '9') p++;
            const size_t nd = (size_t)(p - s);
            bool is_float;
            if (nd == 0 || (nd > 1 && *s == '0') || !number_tail(is_float)) { fail(); return idx; }
            if (!is_float && nd == 1 && *s == '0') { t.nodes[idx].kind = kInt; t.nodes[idx].val = 0; }  // json.loads("-0") == 0
            return idx;
        }
        if (word("true") || word("false") || word("null") || word("NaN") || word("Infinity")) return idx;
        fail();
        return idx;
    }
};

// Trees are per-thread scratch that keeps its capacity from one text to the next: a 2^20-row proof
// needs a few MB of nodes, and a fresh allocation per text means fresh page faults per text.
void tree_reset(Tree &t, const char *text)
{
    t.text = text;
    t.nodes.clear();
    t.bigs.clear();
    t.ok = true;
    t.root = 0;
}

bool parse_json(Tree &t, const char *text, size_t len)
{
    tree_reset(t, text);
    if (t.nodes.capacity() < len / 16 + 16) t.nodes.reserve(len / 16 + 16);
    Json j{t, text, text + len};
    j.value(0);
    j.ws();
    return t.ok && j.p == j.end;
}

// ------------------------------------------------------------------ SimplicityHL literal
struct Literal {
    Tree &t;
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }

    uint32_t seq(char close, int depth)
    {
        const uint32_t idx = t.push(kList);
        p++;  // opening bracket
        uint32_t prev = 0, n = 0, commas = 0;
        while (t.ok) {
            ws();
            if (p >= end) { t.ok = false; break; }
            if (*p == close) {
                p++;
                if (close == ')' && n == 1 && commas == 0) t.nodes[idx].kind = kAlias;  // "(x)" is x
                break;
            }
            const uint32_t c = value(depth + 1);
            if (!t.ok) break;
            if (prev) t.nodes[prev].next = c;
            prev = c;
            n++;
            ws();
            if (p < end && *p == ',') { p++; commas++; }
        }
        if (t.nodes[idx].kind == kList) t.nodes[idx].val = n;
        return idx;
    }

    uint32_t value(int depth)
    {
        ws();
        if (p >= end || depth > kMaxDepth) { t.ok = false; return 0; }
        if (end - p >= 5 && memcmp(p, "list!", 5) == 0) {
            p += 5;
            ws();
            if (p >= end || *p != '[') { t.ok = false; return 0; }
            return seq(']', depth);
        }
        if (end - p >= 4 && memcmp(p, "qm31", 4) == 0) {
            // qm31(a, b, c, d) of the .simf snippets (formats.parse_literal accepts it anywhere):
            // the value ((a, b), (c, d)), built in place
            p += 4;
            ws();
            if (p >= end || *p != '(') { t.ok = false; return 0; }
            p++;
            const uint32_t outer = t.push(kList), in1 = t.push(kList);
            uint32_t in2 = 0, prev = 0, n = 0;
            while (t.ok) {
                ws();
                if (p >= end) { t.ok = false; break; }
                if (*p == ')') { p++; break; }
                if (n == 4) { t.ok = false; break; }
                if (n == 2) { in2 = t.push(kList); prev = 0; }
                const uint32_t c = value(depth + 1);
                if (!t.ok) break;
                if (prev) t.nodes[prev].next = c;
                prev = c;
                n++;
                ws();
                if (p < end && *p == ',') p++;
            }
            if (n != 4) t.ok = false;
            if (!t.ok) return outer;
            t.nodes[outer].val = 2;
            t.nodes[in1].val = 2;
            t.nodes[in2].val = 2;
            t.nodes[in1].next = in2;
            return outer;
        }
        if (*p == '(') return seq(')', depth);
        if (*p == '[') return seq(']', depth);
        const uint32_t idx = t.push(kOther);
        int base = 10;
        if (end - p >= 2 && p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
            base = 16;
            p += 2;
        }
        const char *s0 = p;
        bool sep = false;
        if (base == 16)
            while (p < end && (kHexVal[(unsigned char)*p] != 0xff || *p == '_')) sep |= *p++ == '_';
        else
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '_')) sep |= *p++ == '_';
        size_t nd = (size_t)(p - s0);
        char buf[128];  // '_' separators removed (only then is a copy needed)
        const char *digits = s0;
        if (sep) {
            size_t k = 0;
            for (const char *c = s0; c < p; c++)
                if (*c != '_') { if (k == sizeof buf) { k = sizeof buf + 1; break; } buf[k++] = *c; }
            if (k > sizeof buf) return idx;  // longer than any u256 even with leading zeros: kOther
            digits = buf;
            nd = k;
        }
        if (nd == 0) { t.ok = false; return idx; }
        if (!number_node(t, idx, digits, nd, base)) t.nodes[idx].kind = kOther;
        return idx;
    }
};

// the literal inside {"NAME": {"value": "<literal>", ...}} of a .wit, parsed into its own tree
bool wit_member(const Tree &j, const char *name, Tree &out)
{
    const uint32_t m = j.member(0, name);
    if (!m) return false;
    const uint32_t v = j.member(m, "value");
    if (!v || j.nodes[v].kind != kStr) return false;
    const char *s = j.text + j.nodes[v].val;
    size_t n = j.nodes[v].len;
    if (memchr(s, '\\', n)) {  // escapes in the literal: parse what they decode to
        json_unescape(s, n, out.owned);
        s = out.owned.data();
        n = out.owned.size();
    }
    tree_reset(out, s);
    out.push(kOther);
    out.root = 1;
    Literal l{out, s, s + n};
    l.value(0);
    l.ws();
    return out.ok && l.p == l.end;
}

// --------------------------------------------------------------------------- accessors
bool get_u32(const Tree &t, uint32_t i, uint32_t &out)
{
    if (!i || t.nodes[i].kind != kInt) return false;
    out = t.nodes[i].val;
    return true;
}

bool get_u64(const Tree &t, uint32_t i, uint64_t &out)
{
    if (!i) return false;
    if (t.nodes[i].kind == kInt) { out = t.nodes[i].val; return true; }
    if (t.nodes[i].kind != kBig) return false;
    const U256 &b = t.bigs[t.nodes[i].val];
    for (int k = 0; k < 6; k++)
        if (b.w[k]) return false;
    out = ((uint64_t)b.w[6] << 32) | b.w[7];
    return true;
}

// a u256 integer, or a list of 32 byte values -> 8 stored hash words
bool get_hash(const Tree &t, uint32_t i, uint32_t *out)
{
    if (!i) return false;
    const Node &nd = t.nodes[i];
    if (nd.kind == kInt) {
        for (int k = 0; k < 7; k++) out[k] = 0;
        out[7] = nd.val;
        return true;
    }
    if (nd.kind == kBig || nd.kind == kBytes32) {
        memcpy(out, t.bigs[nd.val].w, 32);
        return true;
    }
    if (nd.kind != kList || nd.val != 32) return false;
    uint32_t c = i + 1;
    for (int k = 0; k < 8; k++) {
        uint32_t w = 0;
        for (int b = 0; b < 4; b++) {
            const Node &x = t.nodes[t.resolve(c)];
            if (x.kind != kInt || x.val > 255) return false;
            w = (w << 8) | x.val;
            c = t.nodes[c].next;
        }
        out[k] = w;
    }
    return true;
}

// formats._qm31: single-element lists around the value are peeled off, then ((a, b), (c, d))
bool get_qm31(const Tree &t, uint32_t i, uint32_t *out)
{
    if (!i) return false;
    while (t.is_list(i) && t.count(i) == 1 && t.is_list(t.child(i, 0))) i = t.child(i, 0);
    if (!t.is_list(i) || t.count(i) != 2) return false;
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t pr = t.child(i, h);
        if (!pr || !t.is_list(pr) || t.count(pr) != 2) return false;
        if (!get_u32(t, t.child(pr, 0), out[2 * h]) || !get_u32(t, t.child(pr, 1), out[2 * h + 1])) return false;
    }
    return true;
}

// nodes [first, first + count) of list `lst` as one Merkle path: at most 31 siblings (List<u256, 32>),
// the first `slot` of them stored, the real length reported
bool get_path(const Tree &t, uint32_t lst, uint32_t first, uint32_t count, uint32_t slot, uint32_t *dst,
              uint32_t &plen)
{
    if (count > kMaxList) return false;
    uint32_t c = lst + 1;
    for (uint32_t k = 0; k < first; k++) c = t.nodes[c].next;
    for (uint32_t k = 0; k < count; k++) {
        uint32_t h[8];
        if (!get_hash(t, t.resolve(c), h)) return false;
        if (k < slot) memcpy(dst + 8 * k, h, 32);
        c = t.nodes[c].next;
    }
    plen = count;
    return true;
}

// Walks the elements of a JSON list in order, whichever way the parser kept it (kList, or the
// packed kBytes32 whose elements are its 32 byte values).  The caller stays within count().
struct ListIter {
    const Tree &t;
    uint32_t lst, c = 0, k = 0;
    bool blob;
    ListIter(const Tree &t_, uint32_t lst_) : t(t_), lst(lst_), blob(t_.nodes[lst_].kind == kBytes32)
    {
        if (!blob && t.nodes[lst].val) c = lst + 1;
    }
    bool u32(uint32_t &v)
    {
        if (blob) {
            v = (t.bigs[t.nodes[lst].val].w[k >> 2] >> (8 * (3 - (k & 3)))) & 255;
            k++;
            return true;
        }
        const bool ok = get_u32(t, t.resolve(c), v);
        c = t.nodes[c].next;
        return ok;
    }
    bool hash(uint32_t *h)
    {
        if (blob) {  // a small integer read as a u256
            uint32_t v;
            u32(v);
            for (int j = 0; j < 7; j++) h[j] = 0;
            h[7] = v;
            return true;
        }
        const bool ok = get_hash(t, t.resolve(c), h);
        c = t.nodes[c].next;
        return ok;
    }
    bool qm31(uint32_t *o)
    {
        if (blob) return false;
        const bool ok = get_qm31(t, t.resolve(c), o);
        c = t.nodes[c].next;
        return ok;
    }
};

uint64_t pow_target_of_bits(uint32_t bits) { return bits == 0 ? ~(uint64_t)0 : (((uint64_t)1 << (64 - bits)) - 1); }

// word offsets of a record's sections (include/ss_verify.h)
struct RecordMap {
    uint32_t N, L, Q, K;
    uint32_t head, qstride, fbase, tbase, words;
    uint32_t foff[kMaxList + 1];
    explicit RecordMap(const ss_stwo_cfg &c) : N(c.n_cols), L(c.lde_log), Q(c.n_queries), K(c.n_layers)
    {
        head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
        qstride = N + kCp + 16 * L;
        fbase = head + Q * qstride;
        uint32_t o = 0;
        for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
        tbase = fbase + o;
        words = tbase + (K + 3) * Q;
    }
    uint32_t *trace_vals(uint32_t *r, uint32_t q) const { return r + head + q * qstride; }
    uint32_t *cp_vals(uint32_t *r, uint32_t q) const { return trace_vals(r, q) + N; }
    uint32_t *trace_path(uint32_t *r, uint32_t q) const { return cp_vals(r, q) + kCp; }
    uint32_t *cp_path(uint32_t *r, uint32_t q) const { return trace_path(r, q) + 8 * L; }
    uint32_t *fri_wit(uint32_t *r, uint32_t l, uint32_t q) const { return r + fbase + foff[l] + q * (4 + 8 * (L - 1 - l)); }
    uint32_t &plen(uint32_t *r, uint32_t kind, uint32_t q) const { return r[tbase + kind * Q + q]; }
};

// The "shared paths" variant of proof.json (formats.shared_path_order): every distinct sibling of a tree once, in
// the order a walk over query 0, 1, .. leaf -> root first needs it, plus a top-level "queries" member with the
// positions -- at most kMaxQueries of them (no verifier config has more; the bound keeps the plan small whatever the
// text claims).  The order in closed form is ss_shared.h's; expand_shared undoes the sharing for tree `tree`:
// the list must hold exactly the count the positions imply, out = Q x len hashes.
bool expand_shared(const Tree &t, uint32_t hw, const SharedPlan &p, uint32_t tree, uint32_t Q, uint32_t L,
                   std::vector<U256> &out)
{
    const uint32_t len = shared_tree_len(L, tree), sh = shared_tree_shift(tree), count = p.base[tree][Q];
    if (!hw || !t.is_list(hw) || t.count(hw) != count) return false;
    std::vector<U256> nodes(count);  // <= kMaxQueries * kMaxList
    ListIter it(t, hw);
    for (uint32_t i = 0; i < count; i++)
        if (!it.hash(nodes[i].w)) return false;
    out.resize((size_t)Q * len);
    for (uint32_t q = 0; q < Q; q++)
        for (uint32_t lvl = 0; lvl < len; lvl++) out[(size_t)q * len + lvl] = nodes[p.base[tree][p.lead[q][sh + lvl]] + lvl];
    return true;
}

// the hashes of a hash_witness in (query, level) order: the list itself, or what expand_shared made of it
struct HashSeq {
    ListIter it;
    const std::vector<U256> *vec;
    size_t pos = 0;
    HashSeq(const Tree &t, uint32_t hw, const std::vector<U256> *v) : it(t, v ? 0 : hw), vec(v) {}
    bool next(uint32_t *h)
    {
        if (!vec) return it.hash(h);
        memcpy(h, (*vec)[pos++].w, 32);
        return true;
    }
};

// ---------------------------------------------------------------- stwo, format C (proof.json)
// formats.stwo_from_json, with the expected config given (`expect=`): what the JSON does not
// declare is the verifier's; what it declares, and every array length, must agree with it.
ParseResult stwo_from_json(const ss_stwo_cfg &cfg, const Tree &t, uint32_t *rec)
{
    const RecordMap m(cfg);
    if (t.nodes.empty() || t.nodes[0].kind != kObj) return kMalformed;
    bool mismatch = false;
    // ---- declared parameters
    const uint32_t conf = t.member(0, "config");
    const uint32_t fconf = conf ? t.member(conf, "fri_config") : 0;
    uint32_t Q = cfg.n_queries, v;
    if (conf && t.nodes[conf].kind != kObj) return kMalformed;
    if (fconf && t.nodes[fconf].kind != kObj) return kMalformed;
    if (uint32_t x = fconf ? t.member(fconf, "n_queries") : 0) {
        if (!get_u32(t, x, Q)) return kMalformed;
    }
    if (uint32_t x = conf ? t.member(conf, "pow_bits") : 0) {
        if (!get_u32(t, x, v)) return kMalformed;
        mismatch |= v > 64 || pow_target_of_bits(v) != cfg.pow_target;  // > 64 bits: no verifier's config
    }
    if (uint32_t x = conf ? t.member(conf, "hash") : 0) {
        const Node &s = t.nodes[x];
        if (s.kind != kStr) return kMalformed;
        const bool sha = s.len == 6 && memcmp(t.text + s.val, "sha256", 6) == 0;
        const bool b2s = s.len == 7 && memcmp(t.text + s.val, "blake2s", 7) == 0;
        if (!sha && !b2s) return kMalformed;
        mismatch |= (b2s ? SS_HASH_BLAKE2S : SS_HASH_SHA256) != cfg.hash;
    }
    if (Q == 0) return kMalformed;  // formats._split: n <= 0
    mismatch |= Q != cfg.n_queries;
    // ---- commitments
    const uint32_t com = t.member(0, "commitments");
    if (!com || !t.is_list(com) || t.count(com) != 3) return kMalformed;
    uint32_t roots[24];
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t c = t.child(com, k);
        if (!t.is_list(c) || !get_hash(t, c, roots + 8 * k)) return kMalformed;
    }
    // ---- sampled values
    const uint32_t sv = t.member(0, "sampled_values");
    const uint32_t sv1 = sv ? t.child(sv, 1) : 0, sv2 = sv ? t.child(sv, 2) : 0;
    if (!sv1 || !sv2 || !t.is_list(sv1) || !t.is_list(sv2) || t.count(sv2) != kCp) return kMalformed;
    const uint32_t N = t.count(sv1);
    mismatch |= N != cfg.n_cols;
    // ---- decommitments and queried values: concatenated over the queries, split equally
    const uint32_t dec = t.member(0, "decommitments"), qv = t.member(0, "queried_values");
    const uint32_t d1 = dec ? t.child(dec, 1) : 0, d2 = dec ? t.child(dec, 2) : 0;
    const uint32_t hw1 = d1 ? t.member(d1, "hash_witness") : 0, hw2 = d2 ? t.member(d2, "hash_witness") : 0;
    const uint32_t qv1 = qv ? t.child(qv, 1) : 0, qv2 = qv ? t.child(qv, 2) : 0;
    if (!hw1 || !hw2 || !qv1 || !qv2 || !t.is_list(hw1) || !t.is_list(hw2) || !t.is_list(qv1) || !t.is_list(qv2))
        return kMalformed;
    // ---- the shared-path variant names its query positions; the LDE size is the verifier's (paths have no ends there)
    const uint32_t qn = t.member(0, "queries");
    const bool shared = qn != 0;
    uint32_t qpos[kMaxQueries];
    std::vector<U256> ex1, ex2;
    if (shared) {
        if (Q > kMaxQueries || !t.is_list(qn) || t.count(qn) != Q) return kMalformed;
        ListIter it(t, qn);
        for (uint32_t q = 0; q < Q; q++)
            if (!it.u32(qpos[q]) || (qpos[q] >> cfg.lde_log)) return kMalformed;
    }
    if (!shared && (t.count(hw1) % Q || t.count(hw2) % Q)) return kMalformed;
    if (t.count(qv1) % Q || t.count(qv2) % Q) return kMalformed;
    const uint32_t tlen = shared ? cfg.lde_log : t.count(hw1) / Q, clen = shared ? cfg.lde_log : t.count(hw2) / Q;
    if (tlen > kMaxList || clen > kMaxList) return kMalformed;
    if (t.count(qv1) != Q * N || t.count(qv2) != Q * kCp) return kMalformed;
    mismatch |= tlen != cfg.lde_log;  // LDE_LOG_SIZE is the length of the first trace path
    // ---- FRI
    const uint32_t fri = t.member(0, "fri_proof");
    const uint32_t first = fri ? t.member(fri, "first_layer") : 0;
    if (!fri || !first) return kMalformed;
    const uint32_t inner = t.member(fri, "inner_layers");
    if (inner && !t.is_list(inner)) return kMalformed;
    const uint32_t K = inner ? t.count(inner) : 0;
    if (K > kMaxList) return kMalformed;
    mismatch |= K != cfg.n_layers;
    const uint32_t llp = t.member(fri, "last_layer_poly");
    const uint32_t coeffs = llp ? t.member(llp, "coeffs") : 0;
    if (!coeffs || !t.is_list(coeffs) || t.count(coeffs) != 1) return kMalformed;
    // ---- trace_log = lde_log - declared blow-up
    if (uint32_t x = fconf ? t.member(fconf, "log_blowup_factor") : 0) {
        if (!get_u32(t, x, v)) return kMalformed;
        mismatch |= (int64_t)tlen - (int64_t)v != (int64_t)cfg.trace_log;
    }
    uint64_t nonce = 0;
    if (uint32_t x = t.member(0, "proof_of_work")) {
        if (!get_u64(t, x, nonce)) return kMalformed;
    }
    // Everything below is validated even when the shape already mismatches, so that a proof that is
    // malformed AND of another shape is reported as malformed, as the Python reader does (it parses
    // completely before the policy looks at the config).  The record is written only on a match.
    std::vector<uint32_t> scratch;
    uint32_t *r = rec;
    if (mismatch) {  // validate into a scratch record of the proof's own shape? no: validate without storing
        r = nullptr;
    }
    uint32_t tmp[8 * kMaxList];
    auto store = [&](uint32_t *dst, const uint32_t *src, size_t words) {
        if (r) memcpy(dst, src, words * 4);
    };
    if (r) {
        memset(r, 0, (size_t)m.words * 4);
        store(r, roots, 24);
    }
    {
        uint32_t c1 = t.first_child(sv1), c2 = t.first_child(sv2);
        for (uint32_t k = 0; k < N; k++) {
            if (!get_qm31(t, t.next_child(c1), tmp)) return kMalformed;
            if (r) store(r + 24 + 4 * k, tmp, 4);
        }
        for (uint32_t k = 0; k < kCp; k++) {
            if (!get_qm31(t, t.next_child(c2), tmp)) return kMalformed;
            if (r) store(r + 24 + 4 * m.N + 4 * k, tmp, 4);
        }
    }
    // queried values: walk the two flat lists once
    {
        ListIter i1(t, qv1), i2(t, qv2);
        for (uint32_t q = 0; q < Q; q++) {
            for (uint32_t k = 0; k < N; k++) {
                if (!i1.u32(v)) return kMalformed;
                if (r) m.trace_vals(r, q)[k] = v;
            }
            for (uint32_t k = 0; k < kCp; k++) {
                if (!i2.u32(v)) return kMalformed;
                if (r) m.cp_vals(r, q)[k] = v;
            }
        }
    }
    auto paths = [&](uint32_t hw, const std::vector<U256> *expanded, uint32_t len, uint32_t slot, auto dst_of,
                     uint32_t kind) -> bool {
        HashSeq it(t, hw, expanded);
        for (uint32_t q = 0; q < Q; q++) {
            for (uint32_t k = 0; k < len; k++) {
                uint32_t h[8];
                if (!it.next(h)) return false;
                if (r && k < slot) memcpy(dst_of(q) + 8 * k, h, 32);
            }
            if (r) m.plen(r, kind, q) = len;
        }
        return true;
    };
    SharedPlan plan;  // (K + 3 trees of the TEXT's shape over the verifier's LDE size: Q <= 64, K <= 31)
    if (shared) {
        if (K + 1 >= cfg.lde_log) return kMalformed;  // a FRI tree would have no levels left
        shared_plan(shared_map(0, cfg.lde_log, Q, K), qpos, plan);  // (positions were range-checked above)
        if (!expand_shared(t, hw1, plan, 0, Q, cfg.lde_log, ex1) || !expand_shared(t, hw2, plan, 1, Q, cfg.lde_log, ex2)) return kMalformed;
    }
    if (!paths(hw1, shared ? &ex1 : nullptr, tlen, m.L, [&](uint32_t q) { return m.trace_path(r, q); }, 0)) return kMalformed;
    if (!paths(hw2, shared ? &ex2 : nullptr, clen, m.L, [&](uint32_t q) { return m.cp_path(r, q); }, 1)) return kMalformed;
    uint32_t inner_c = inner ? t.first_child(inner) : 0;
    for (uint32_t l = 0; l <= K; l++) {
        const uint32_t layer = l == 0 ? first : t.next_child(inner_c);
        if (!layer || t.nodes[layer].kind != kObj) return kMalformed;
        const uint32_t w = t.member(layer, "fri_witness"), d = t.member(layer, "decommitment");
        const uint32_t hw = d ? t.member(d, "hash_witness") : 0, cm = t.member(layer, "commitment");
        if (!w || !hw || !cm || !t.is_list(w) || !t.is_list(hw) || t.count(w) != Q) return kMalformed;
        std::vector<U256> exl;
        if (shared) {  // FRI layer l is indexed by query >> (l + 1) in a tree of depth lde_log - 1 - l
            if (!expand_shared(t, hw, plan, 2 + l, Q, cfg.lde_log, exl)) return kMalformed;
        } else if (t.count(hw) % Q) {
            return kMalformed;
        }
        uint32_t h[8];
        if (!t.is_list(cm) || !get_hash(t, cm, h)) return kMalformed;
        const bool keep = r && l <= m.K;
        if (keep) memcpy(r + 24 + 4 * m.N + 64 + 8 * l, h, 32);
        const uint32_t len = shared ? cfg.lde_log - 1 - l : t.count(hw) / Q;
        if (len > kMaxList) return kMalformed;
        ListIter iw(t, w);
        HashSeq ih(t, hw, shared ? &exl : nullptr);
        for (uint32_t q = 0; q < Q; q++) {
            if (!iw.qm31(tmp)) return kMalformed;
            uint32_t *dst = keep ? m.fri_wit(r, l, q) : nullptr;
            if (dst) memcpy(dst, tmp, 16);
            const uint32_t slot = keep ? m.L - 1 - l : 0;
            for (uint32_t k = 0; k < len; k++) {
                if (!ih.next(h)) return kMalformed;
                if (dst && k < slot) memcpy(dst + 4 + 8 * k, h, 32);
            }
            if (keep) m.plen(r, 2 + l, q) = len;
        }
    }
    if (!get_qm31(t, t.child(coeffs, 0), tmp)) return kMalformed;
    if (mismatch) return kConfigMismatch;
    memcpy(r + 24 + 4 * m.N + 64 + 8 * (m.K + 1), tmp, 16);
    r[m.head - 2] = (uint32_t)(nonce >> 32);
    r[m.head - 1] = (uint32_t)nonce;
    return kParsed;
}

// ------------------------------------------------------ stwo, the minimal proof.json (ss_minimal.h)
// formats.stwo_minimal_from_json against the expected config: the schema of format C with one decommitment per tree,
// as upstream stwo's prover fills it (queried values once per distinct position, only the siblings / fold partners the
// verifier cannot compute).  List lengths are data; what the text declares must agree with the verifier's config.
ParseResult stwo_min_from_json(const ss_stwo_cfg &cfg, const Tree &t, std::vector<uint32_t> &out)
{
    const MinMap m = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    if (t.nodes.empty() || t.nodes[0].kind != kObj) return kMalformed;
    bool mismatch = false;
    const uint32_t conf = t.member(0, "config");
    const uint32_t fconf = conf ? t.member(conf, "fri_config") : 0;
    uint32_t v;
    if (conf && t.nodes[conf].kind != kObj) return kMalformed;
    if (fconf && t.nodes[fconf].kind != kObj) return kMalformed;
    if (uint32_t x = fconf ? t.member(fconf, "n_queries") : 0) {
        if (!get_u32(t, x, v)) return kMalformed;
        mismatch |= v != cfg.n_queries;
    }
    if (uint32_t x = conf ? t.member(conf, "pow_bits") : 0) {
        if (!get_u32(t, x, v)) return kMalformed;
        mismatch |= v > 64 || pow_target_of_bits(v) != cfg.pow_target;
    }
    if (uint32_t x = fconf ? t.member(fconf, "log_blowup_factor") : 0) {
        if (!get_u32(t, x, v)) return kMalformed;
        mismatch |= v != cfg.lde_log - cfg.trace_log;
    }
    if (uint32_t x = conf ? t.member(conf, "hash") : 0) {
        const Node &s = t.nodes[x];
        if (s.kind != kStr) return kMalformed;
        const bool sha = s.len == 6 && memcmp(t.text + s.val, "sha256", 6) == 0;
        const bool b2s = s.len == 7 && memcmp(t.text + s.val, "blake2s", 7) == 0;
        if (!sha && !b2s) return kMalformed;
        mismatch |= (b2s ? SS_HASH_BLAKE2S : SS_HASH_SHA256) != cfg.hash;
    }
    const uint32_t com = t.member(0, "commitments");
    if (!com || !t.is_list(com) || t.count(com) != 3) return kMalformed;
    uint32_t roots[24];
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t c = t.child(com, k);
        if (!t.is_list(c) || !get_hash(t, c, roots + 8 * k)) return kMalformed;
    }
    const uint32_t sv = t.member(0, "sampled_values");
    const uint32_t sv1 = sv ? t.child(sv, 1) : 0, sv2 = sv ? t.child(sv, 2) : 0;
    if (!sv1 || !sv2 || !t.is_list(sv1) || !t.is_list(sv2) || t.count(sv2) != kCp) return kMalformed;
    const uint32_t N = t.count(sv1);
    mismatch |= N != cfg.n_cols;
    const uint32_t dec = t.member(0, "decommitments"), qv = t.member(0, "queried_values");
    const uint32_t d1 = dec ? t.child(dec, 1) : 0, d2 = dec ? t.child(dec, 2) : 0;
    const uint32_t hw1 = d1 ? t.member(d1, "hash_witness") : 0, hw2 = d2 ? t.member(d2, "hash_witness") : 0;
    const uint32_t qv1 = qv ? t.child(qv, 1) : 0, qv2 = qv ? t.child(qv, 2) : 0;
    if (!hw1 || !hw2 || !qv1 || !qv2 || !t.is_list(hw1) || !t.is_list(hw2) || !t.is_list(qv1) || !t.is_list(qv2))
        return kMalformed;
    if (N == 0 || t.count(qv1) % N || t.count(qv2) % kCp) return kMalformed;
    const uint32_t fri = t.member(0, "fri_proof");
    const uint32_t first = fri ? t.member(fri, "first_layer") : 0;
    if (!fri || !first) return kMalformed;
    const uint32_t inner = t.member(fri, "inner_layers");
    if (inner && !t.is_list(inner)) return kMalformed;
    const uint32_t K = inner ? t.count(inner) : 0;
    if (K > kMaxList) return kMalformed;
    mismatch |= K != cfg.n_layers;
    const uint32_t llp = t.member(fri, "last_layer_poly");
    const uint32_t coeffs = llp ? t.member(llp, "coeffs") : 0;
    if (!coeffs || !t.is_list(coeffs) || t.count(coeffs) != 1) return kMalformed;
    uint64_t nonce = 0;
    if (uint32_t x = t.member(0, "proof_of_work")) {
        if (!get_u64(t, x, nonce)) return kMalformed;
    }
    // everything is validated even when the shape already mismatches (malformed wins, as in the Python reader);
    // the record is only kept on a match
    out.clear();
    out.resize(m.data, 0);
    uint32_t tmp[8];
    const bool keep = !mismatch;
    if (keep) memcpy(out.data(), roots, 96);
    {
        uint32_t c1 = t.first_child(sv1), c2 = t.first_child(sv2);
        for (uint32_t k = 0; k < N; k++) {
            if (!get_qm31(t, t.next_child(c1), tmp)) return kMalformed;
            if (keep) memcpy(out.data() + 24 + 4 * k, tmp, 16);
        }
        for (uint32_t k = 0; k < kCp; k++) {
            if (!get_qm31(t, t.next_child(c2), tmp)) return kMalformed;
            if (keep) memcpy(out.data() + 24 + 4 * m.N + 4 * k, tmp, 16);
        }
    }
    // list lengths beyond what any set of n_queries positions gives: no witness of this config (and no reason to
    // let a text size the record)
    const uint32_t n0 = t.count(qv1) / N, n1 = t.count(qv2) / kCp;
    if (n0 > cfg.n_queries || n1 > cfg.n_queries) return kMalformed;
    if (keep) { out[m.nv] = n0; out[m.nv + 1] = n1; }
    auto u32_list = [&](uint32_t lst) -> bool {
        ListIter it(t, lst);
        for (uint32_t i = 0, n = t.count(lst); i < n; i++) {
            if (!it.u32(v)) return false;
            if (keep) out.push_back(v);
        }
        return true;
    };
    if (!u32_list(qv1) || !u32_list(qv2)) return kMalformed;
    // the layers: fri_witness lists first (the record keeps them in front of the hashes), then every tree's hashes
    std::vector<uint32_t> layers(K + 1);
    {
        uint32_t inner_c = inner ? t.first_child(inner) : 0;
        for (uint32_t l = 0; l <= K; l++) {
            const uint32_t layer = l == 0 ? first : t.next_child(inner_c);
            if (!layer || t.nodes[layer].kind != kObj) return kMalformed;
            layers[l] = layer;
            const uint32_t w = t.member(layer, "fri_witness"), cm = t.member(layer, "commitment");
            if (!w || !cm || !t.is_list(w) || !t.is_list(cm)) return kMalformed;
            if (t.count(w) > cfg.n_queries) return kMalformed;
            if (!get_hash(t, cm, tmp)) return kMalformed;
            if (keep && l <= m.K) { memcpy(out.data() + 24 + 4 * m.N + 64 + 8 * l, tmp, 32); out[m.nfw + l] = t.count(w); }
            ListIter iw(t, w);
            for (uint32_t i = 0, n = t.count(w); i < n; i++) {
                uint32_t q4[4];
                if (!iw.qm31(q4)) return kMalformed;
                if (keep) out.insert(out.end(), q4, q4 + 4);
            }
        }
    }
    auto hashes = [&](uint32_t hw, uint32_t tree) -> bool {
        if (!hw || !t.is_list(hw)) return false;
        const uint32_t n = t.count(hw);
        if (tree < m.K + 3 && n > cfg.n_queries * min_tree_len(cfg.lde_log, tree)) return false;
        if (n > kMaxQueries * kMaxList) return false;
        if (keep) out[m.nhw + tree] = n;
        ListIter it(t, hw);
        for (uint32_t i = 0; i < n; i++) {
            if (!it.hash(tmp)) return false;
            if (keep) out.insert(out.end(), tmp, tmp + 8);
        }
        return true;
    };
    if (!hashes(hw1, 0) || !hashes(hw2, 1)) return kMalformed;
    for (uint32_t l = 0; l <= K; l++) {
        const uint32_t d = t.member(layers[l], "decommitment");
        if (!d || !hashes(t.member(d, "hash_witness"), 2 + l)) return kMalformed;
    }
    if (!get_qm31(t, t.child(coeffs, 0), tmp)) return kMalformed;
    if (mismatch) { out.clear(); return kConfigMismatch; }
    memcpy(out.data() + 24 + 4 * m.N + 64 + 8 * (m.K + 1), tmp, 16);
    out[m.head - 2] = (uint32_t)(nonce >> 32);
    out[m.head - 1] = (uint32_t)nonce;
    return kParsed;
}

// ------------------------------------------------------------------ stwo, format D (proof.wit)
// formats.stwo_from_wit / _stwo_from_parts: six literals; a .wit declares nothing, so TRACE_LOG_SIZE,
// the PoW target and the hash are the verifier's and only the shape is compared.
ParseResult stwo_from_wit(const ss_stwo_cfg &cfg, const Tree &j, uint32_t *rec)
{
    const RecordMap m(cfg);
    if (j.nodes.empty() || j.nodes[0].kind != kObj) return kMalformed;
    static thread_local Tree com, dec, oods, fric, frid, non;
    if (!wit_member(j, "COMMITMENTS", com) || !wit_member(j, "DECOMMITMENTS", dec) ||
        !wit_member(j, "OODS_EVALS", oods) || !wit_member(j, "FRI_COMMITMENTS", fric) ||
        !wit_member(j, "FRI_DECOMMITMENTS", frid) || !wit_member(j, "POW_NONCE", non))
        return kMalformed;
    const uint32_t c0 = com.resolve(com.root), d0 = dec.resolve(dec.root), o0 = oods.resolve(oods.root), fc0 = fric.resolve(fric.root),
                   fd0 = frid.resolve(frid.root);
    if (!com.is_list(c0) || com.count(c0) != 3 || !dec.is_list(d0)) return kMalformed;
    const uint32_t Q = dec.count(d0);
    const uint32_t ot = oods.child(o0, 0), oc = oods.child(o0, 1);
    if (!ot || !oc || !oods.is_list(ot) || !oods.is_list(oc) || oods.count(oc) != kCp) return kMalformed;
    const uint32_t N = oods.count(ot);
    const uint32_t fd_first = frid.child(fd0, 0), fd_inner = frid.child(fd0, 1);
    if (!fd_first || !fd_inner || !frid.is_list(fd_first) || !frid.is_list(fd_inner)) return kMalformed;
    const uint32_t K = frid.count(fd_inner);
    const uint32_t fc_inner = fric.child(fc0, 1);
    if (!fric.child(fc0, 0) || !fc_inner || !fric.is_list(fc_inner) || fric.count(fc_inner) != K || !fric.child(fc0, 2))
        return kMalformed;
    if (K > kMaxList) return kMalformed;
    bool mismatch = Q != cfg.n_queries || N != cfg.n_cols || K != cfg.n_layers;
    uint32_t *r = nullptr;
    uint32_t tmp[8 * kMaxList], v, pl;
    if (!mismatch) {
        r = rec;
        memset(r, 0, (size_t)m.words * 4);
    }
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t c = com.child(c0, k);
        if (com.is_list(c) || !get_hash(com, c, tmp)) return kMalformed;  // u256 integers here
        if (r) memcpy(r + 8 * k, tmp, 32);
    }
    {
        uint32_t ct = oods.first_child(ot), cc = oods.first_child(oc);
        for (uint32_t k = 0; k < N; k++) {  // each column is an array of its MAX_COLUMN_OFFSET = 1 samples
            const uint32_t col = oods.next_child(ct);
            if (!col || !oods.is_list(col) || oods.count(col) < 1 || !get_qm31(oods, oods.child(col, 0), tmp)) return kMalformed;
            if (r) memcpy(r + 24 + 4 * k, tmp, 16);
        }
        for (uint32_t k = 0; k < kCp; k++) {
            if (!get_qm31(oods, oods.next_child(cc), tmp)) return kMalformed;
            if (r) memcpy(r + 24 + 4 * m.N + 4 * k, tmp, 16);
        }
    }
    uint32_t lde_log = 0;
    uint32_t dec_c = dec.first_child(d0);
    for (uint32_t q = 0; q < Q; q++) {
        const uint32_t d = dec.next_child(dec_c);
        const uint32_t tpart = d ? dec.child(d, 0) : 0, cpart = d ? dec.child(d, 1) : 0;
        const uint32_t tv = tpart ? dec.child(tpart, 0) : 0, tp = tpart ? dec.child(tpart, 1) : 0;
        const uint32_t cv = cpart ? dec.child(cpart, 0) : 0, cp = cpart ? dec.child(cpart, 1) : 0;
        if (!tv || !tp || !cv || !cp || !dec.is_list(tv) || !dec.is_list(tp) || !dec.is_list(cv) || !dec.is_list(cp))
            return kMalformed;
        if (dec.count(tv) != N || dec.count(cv) != kCp) return kMalformed;
        uint32_t tv_c = dec.first_child(tv), cv_c = dec.first_child(cv);
        for (uint32_t k = 0; k < N; k++) {
            const uint32_t col = dec.next_child(tv_c);
            if (!col || !dec.is_list(col) || dec.count(col) < 1 || !get_u32(dec, dec.child(col, 0), v)) return kMalformed;
            if (r) m.trace_vals(r, q)[k] = v;
        }
        for (uint32_t k = 0; k < kCp; k++) {
            if (!get_u32(dec, dec.next_child(cv_c), v)) return kMalformed;
            if (r) m.cp_vals(r, q)[k] = v;
        }
        if (!get_path(dec, tp, 0, dec.count(tp), m.L, tmp, pl)) return kMalformed;
        if (q == 0) {
            lde_log = pl;
            if (lde_log != cfg.lde_log) { mismatch = true; r = nullptr; }
        }
        if (r) { memcpy(m.trace_path(r, q), tmp, (size_t)(pl < m.L ? pl : m.L) * 32); m.plen(r, 0, q) = pl; }
        if (!get_path(dec, cp, 0, dec.count(cp), m.L, tmp, pl)) return kMalformed;
        if (r) { memcpy(m.cp_path(r, q), tmp, (size_t)(pl < m.L ? pl : m.L) * 32); m.plen(r, 1, q) = pl; }
    }
    uint32_t fc_c = fric.first_child(fc_inner), fd_c = frid.first_child(fd_inner);
    for (uint32_t l = 0; l <= K; l++) {
        const uint32_t root = l == 0 ? fric.child(fc0, 0) : fric.next_child(fc_c);
        if (fric.is_list(root) || !get_hash(fric, root, tmp)) return kMalformed;
        const bool keep = r && l <= m.K;
        if (keep) memcpy(r + 24 + 4 * m.N + 64 + 8 * l, tmp, 32);
        const uint32_t layer = l == 0 ? fd_first : frid.next_child(fd_c);
        if (!layer || !frid.is_list(layer) || frid.count(layer) != Q) return kMalformed;
        uint32_t lq_c = frid.first_child(layer);
        for (uint32_t q = 0; q < Q; q++) {
            const uint32_t x = frid.next_child(lq_c);
            const uint32_t w = x ? frid.child(x, 0) : 0, pth = x ? frid.child(x, 1) : 0;
            if (!w || !pth || !frid.is_list(pth) || !get_qm31(frid, w, tmp)) return kMalformed;
            uint32_t *dst = keep ? m.fri_wit(r, l, q) : nullptr;
            if (dst) memcpy(dst, tmp, 16);
            const uint32_t slot = keep ? m.L - 1 - l : 0;
            if (!get_path(frid, pth, 0, frid.count(pth), slot, tmp, pl)) return kMalformed;
            if (dst) { memcpy(dst + 4, tmp, (size_t)(pl < slot ? pl : slot) * 32); m.plen(r, 2 + l, q) = pl; }
        }
    }
    if (!get_qm31(fric, fric.child(fc0, 2), tmp)) return kMalformed;
    uint64_t nonce;
    if (!get_u64(non, non.resolve(non.root), nonce)) return kMalformed;
    if (mismatch || Q == 0) return kConfigMismatch;
    memcpy(r + 24 + 4 * m.N + 64 + 8 * (m.K + 1), tmp, 16);
    r[m.head - 2] = (uint32_t)(nonce >> 32);
    r[m.head - 1] = (uint32_t)nonce;
    return kParsed;
}

}  // namespace

// The largest witness of a supported config (1024 columns, 64 queries, LDE 2^31) is a few MB of text;
// nothing near this bound is a proof, and a parse tree costs up to ~10 bytes per input byte.
constexpr size_t kMaxTextBytes = (size_t)32 << 20;
constexpr size_t kKeepNodes = (size_t)4 << 20;  // per-thread scratch above this is released after use

static void tree_trim(Tree &t)
{
    if (t.nodes.capacity() > kKeepNodes) { std::vector<Node>().swap(t.nodes); std::vector<U256>().swap(t.bigs); }
}

ParseResult stwo_parse_text(const ss_stwo_cfg &cfg, const char *text, size_t len, int fmt, uint32_t *record)
{
    static thread_local Tree j;
    if (len > kMaxTextBytes) return kMalformed;
    ParseResult r = kMalformed;
    if (parse_json(j, text, len) && !j.nodes.empty() && j.nodes[0].kind == kObj) {
        if (fmt == SS_TEXT_AUTO) fmt = j.member(0, "COMMITMENTS") ? SS_TEXT_WIT : SS_TEXT_JSON;
        r = fmt == SS_TEXT_WIT ? stwo_from_wit(cfg, j, record) : stwo_from_json(cfg, j, record);
    }
    tree_trim(j);
    return r;
}

// ------------------------------------------------------------------ the minimal proof.json, streaming
// A text in the member order of the writers (csrc/ss_text.cpp json_text = formats.stwo_minimal_to_json; any JSON
// whitespace between tokens), with the config the caller expects, read in one pass without a tree: numbers go
// straight into the lists of the record.  It answers "parsed, here is the record" or DECLINES -- every other text
// (another member order, a mismatching or malformed one) is the general reader's to judge, so a decline is never an
// outcome.  What it accepts it reads as stwo_min_from_json does (tests/test_minimal.py: equal on every text both take).
namespace {

struct MinScan {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool ch(char c)
    {
        ws();
        if (p >= end || *p != c) return false;
        p++;
        return true;
    }
    bool peek(char c) { ws(); return p < end && *p == c; }
    bool str(const char *v)  // "v"
    {
        ws();
        const size_t n = strlen(v);
        if ((size_t)(end - p) < n + 2 || p[0] != '"' || memcmp(p + 1, v, n) != 0 || p[n + 1] != '"') return false;
        p += n + 2;
        return true;
    }
    bool key(const char *k) { return str(k) && ch(':'); }  // "k" :
    // a plain JSON integer <= max: digits only, no leading zero, nothing of a fraction or exponent behind it
    bool num(uint64_t max, uint64_t &v)
    {
        ws();
        const char *s = p;
        uint64_t x = 0;
        while (p < end && (unsigned)(*p - '0') < 10u && p - s < 20) { x = x * 10 + (uint64_t)(*p - '0'); p++; }
        const size_t nd = (size_t)(p - s);
        if (nd == 0 || (nd > 1 && *s == '0')) return false;
        if (nd == 20 && (s[0] > '1' || x < 10000000000000000000ull)) return false;  // wrapped past 2^64
        if (p < end && ((unsigned)(*p - '0') < 10u || *p == '.' || *p == 'e' || *p == 'E')) return false;
        v = x;
        return x <= max;
    }
    bool u32(uint32_t &w) { uint64_t v; if (!num(0xffffffffull, v)) return false; w = (uint32_t)v; return true; }
    bool cst(uint64_t want) { uint64_t v; return num(~0ull, v) && v == want; }
    bool hash(uint32_t *w)  // [b0, .., b31] -> 8 words, most significant byte first
    {
        ws();
        if (bytes32_fast(p, end, w)) return true;
        if (!ch('[')) return false;
        for (uint32_t k = 0; k < 32; k++) {
            if (k && !ch(',')) return false;
            ws();
            uint32_t v = 0, nd = 0;
            const char *s = p;
            while (p < end && (unsigned)(*p - '0') < 10u && nd < 4) { v = v * 10 + (uint32_t)(*p - '0'); p++; nd++; }
            if (nd == 0 || nd > 3 || v > 255 || (nd > 1 && *s == '0')) return false;
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) return false;
            w[k >> 2] = (k & 3) ? (w[k >> 2] << 8) | v : v;
        }
        return ch(']');
    }
    bool qm31(uint32_t *w)
    {
        return ch('[') && ch('[') && u32(w[0]) && ch(',') && u32(w[1]) && ch(']') && ch(',') && ch('[') && u32(w[2]) && ch(',') &&
               u32(w[3]) && ch(']') && ch(']');
    }
};

}  // namespace

bool stwo_min_stream(const ss_stwo_cfg &cfg, const char *text, size_t len, std::vector<uint32_t> &out)
{
    uint32_t bits = 0;  // what the text has to declare: the writers' rule (csrc/ss_text.cpp), no text for any other config
    while (bits < 64 && pow_target_of_bits(bits) != cfg.pow_target) bits++;
    if (bits == 64 || cfg.hash > SS_HASH_BLAKE2S ||
        !stwo_cfg_ok(cfg.n_cols, cfg.trace_log, cfg.lde_log, cfg.n_queries, cfg.n_layers, cfg.mode & 1))
        return false;
    const MinMap m = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    const uint32_t N = m.N, Q = m.Q, K = m.K;
    MinScan s{text, text + len};
    // lists in the order of the text; the record wants trace values, composition values, the fri_witness lists, then the
    // hash lists, so they are collected here and copied behind the head at the end
    static thread_local std::vector<uint32_t> vals[2], fw[kMaxList + 1], hw[kMaxList + 3];
    out.assign(m.data, 0);
    uint32_t *head = out.data();
    auto hash_witness = [&](uint32_t tree) -> bool {  // "hash_witness": [hashes], "column_witness": []
        std::vector<uint32_t> &v = hw[tree];
        v.clear();
        const size_t cap = (size_t)Q * min_tree_len(m.L, tree);
        if (!s.key("hash_witness") || !s.ch('[')) return false;
        if (!s.peek(']')) {
            do {
                if (v.size() == 8 * cap) return false;
                v.resize(v.size() + 8);
                if (!s.hash(v.data() + v.size() - 8)) return false;
            } while (s.ch(','));
        }
        return s.ch(']') && s.ch(',') && s.key("column_witness") && s.ch('[') && s.ch(']');
    };
    auto flat = [&](std::vector<uint32_t> &v, uint32_t per) -> bool {  // [u32, ..]: a multiple of `per`, at most Q rows
        v.clear();
        if (!s.ch('[')) return false;
        if (!s.peek(']')) {
            do {
                uint32_t w;
                if (v.size() == (size_t)Q * per || !s.u32(w)) return false;
                v.push_back(w);
            } while (s.ch(','));
        }
        return s.ch(']') && v.size() % per == 0;
    };
    auto layer = [&](uint32_t l) -> bool {
        std::vector<uint32_t> &v = fw[l];
        v.clear();
        if (!s.ch('{') || !s.key("fri_witness") || !s.ch('[')) return false;
        if (!s.peek(']')) {
            do {
                if (v.size() == 4 * (size_t)Q) return false;
                v.resize(v.size() + 4);
                if (!s.qm31(v.data() + v.size() - 4)) return false;
            } while (s.ch(','));
        }
        return s.ch(']') && s.ch(',') && s.key("decommitment") && s.ch('{') && hash_witness(2 + l) && s.ch('}') && s.ch(',') &&
               s.key("commitment") && s.hash(head + 24 + 4 * N + 64 + 8 * l) && s.ch('}');
    };
    bool ok = s.ch('{') && s.key("config") && s.ch('{') && s.key("pow_bits") && s.cst(bits) && s.ch(',') && s.key("fri_config") &&
              s.ch('{') && s.key("log_blowup_factor") && s.cst(cfg.lde_log - cfg.trace_log) && s.ch(',') &&
              s.key("log_last_layer_degree_bound") && s.cst(0) && s.ch(',') && s.key("n_queries") && s.cst(Q) && s.ch('}');
    if (ok && cfg.hash == SS_HASH_BLAKE2S) ok = s.ch(',') && s.key("hash") && s.str("blake2s");  // (the extension member)
    ok = ok && s.ch('}') && s.ch(',') && s.key("commitments") && s.ch('[') && s.hash(head) && s.ch(',') && s.hash(head + 8) &&
         s.ch(',') && s.hash(head + 16) && s.ch(']') && s.ch(',') && s.key("sampled_values") && s.ch('[') && s.ch('[') && s.ch(']') &&
         s.ch(',') && s.ch('[');
    for (uint32_t k = 0; ok && k < N; k++) ok = (k == 0 || s.ch(',')) && s.ch('[') && s.qm31(head + 24 + 4 * k) && s.ch(']');
    ok = ok && s.ch(']') && s.ch(',') && s.ch('[');
    for (uint32_t k = 0; ok && k < kCp; k++) ok = (k == 0 || s.ch(',')) && s.ch('[') && s.qm31(head + 24 + 4 * N + 4 * k) && s.ch(']');
    ok = ok && s.ch(']') && s.ch(']') && s.ch(',') && s.key("decommitments") && s.ch('[') && s.ch('{') && s.key("hash_witness") &&
         s.ch('[') && s.ch(']') && s.ch(',') && s.key("column_witness") && s.ch('[') && s.ch(']') && s.ch('}') && s.ch(',') &&
         s.ch('{') && hash_witness(0) && s.ch('}') && s.ch(',') && s.ch('{') && hash_witness(1) && s.ch('}') && s.ch(']') && s.ch(',') &&
         s.key("queried_values") && s.ch('[') && s.ch('[') && s.ch(']') && s.ch(',') && flat(vals[0], N) && s.ch(',') &&
         flat(vals[1], kCp) && s.ch(']') && s.ch(',') && s.key("proof_of_work");
    uint64_t nonce = 0;
    ok = ok && s.num(~0ull, nonce) && s.ch(',') && s.key("fri_proof") && s.ch('{') && s.key("first_layer") && layer(0) && s.ch(',') &&
         s.key("inner_layers") && s.ch('[');
    for (uint32_t l = 1; ok && l <= K; l++) ok = (l == 1 || s.ch(',')) && layer(l);
    ok = ok && s.ch(']') && s.ch(',') && s.key("last_layer_poly") && s.ch('{') && s.key("coeffs") && s.ch('[') &&
         s.qm31(head + 24 + 4 * N + 64 + 8 * (K + 1)) && s.ch(']') && s.ch(',') && s.key("log_size") && s.cst(0) && s.ch('}') &&
         s.ch('}') && s.ch('}');
    if (ok) { s.ws(); ok = s.p == s.end; }
    if (!ok) { out.clear(); return false; }
    head[m.head - 2] = (uint32_t)(nonce >> 32);
    head[m.head - 1] = (uint32_t)nonce;
    head[m.nv] = (uint32_t)(vals[0].size() / N);
    head[m.nv + 1] = (uint32_t)(vals[1].size() / kCp);
    size_t total = m.data + vals[0].size() + vals[1].size();
    for (uint32_t l = 0; l <= K; l++) { head[m.nfw + l] = (uint32_t)(fw[l].size() / 4); total += fw[l].size(); }
    for (uint32_t t = 0; t < K + 3; t++) { head[m.nhw + t] = (uint32_t)(hw[t].size() / 8); total += hw[t].size(); }
    out.reserve(total);
    out.insert(out.end(), vals[0].begin(), vals[0].end());
    out.insert(out.end(), vals[1].begin(), vals[1].end());
    for (uint32_t l = 0; l <= K; l++) out.insert(out.end(), fw[l].begin(), fw[l].end());
    for (uint32_t t = 0; t < K + 3; t++) out.insert(out.end(), hw[t].begin(), hw[t].end());
    return true;
}

ParseResult stwo_parse_minimal_text(const ss_stwo_cfg &cfg, const char *text, size_t len, std::vector<uint32_t> &out, int route)
{
    static thread_local Tree j;
    out.clear();
    if (len > kMaxTextBytes) return kMalformed;
    if (route != kRouteGeneral && stwo_min_stream(cfg, text, len, out)) return kParsed;
    if (route == kRouteStream) return kDeclined;
    ParseResult r = kMalformed;
    if (parse_json(j, text, len) && !j.nodes.empty() && j.nodes[0].kind == kObj) r = stwo_min_from_json(cfg, j, out);
    if (r != kParsed) out.clear();
    tree_trim(j);
    return r;
}

// ============================================================================= stark101
struct S101Chain {
    uint32_t ev = 0;
    std::vector<uint32_t> path;  // 8 words per sibling
};
struct S101Layer {
    uint32_t root[8], beta;
    S101Chain cpa, cpb;
};
struct S101Parsed {
    uint32_t root[8], last;
    S101Chain evals[3];
    std::vector<S101Layer> layers;
};

namespace {

bool s101_chain(const Tree &t, uint32_t ev, uint32_t path, S101Chain &out)
{
    if (!ev || !path || !get_u32(t, ev, out.ev) || !t.is_list(path) || t.count(path) > kMaxList) return false;
    out.path.resize((size_t)t.count(path) * 8);
    uint32_t pl;
    return get_path(t, path, 0, t.count(path), kMaxList, out.path.data(), pl);
}

// formats._s101_from_parts on (root, evals, layers, last) nodes of possibly different trees
bool s101_from_parts(const Tree &tr, uint32_t root, const Tree &te, uint32_t evals, const Tree &tl, uint32_t layers,
                     const Tree &tz, uint32_t last, S101Parsed &out)
{
    if (tr.is_list(root) || !get_hash(tr, root, out.root)) return false;
    if (!te.is_list(evals) || te.count(evals) != 3 || !tl.is_list(layers) || tl.count(layers) > kMaxList) return false;
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t e = te.child(evals, k);
        if (!e || !s101_chain(te, te.child(e, 0), te.child(e, 1), out.evals[k])) return false;
    }
    out.layers.resize(tl.count(layers));
    for (uint32_t i = 0; i < out.layers.size(); i++) {
        const uint32_t l = tl.child(layers, i);
        if (!l || !tl.is_list(l) || tl.count(l) != 6) return false;
        S101Layer &y = out.layers[i];
        const uint32_t rt = tl.child(l, 0);
        if (tl.is_list(rt) || !get_hash(tl, rt, y.root) || !get_u32(tl, tl.child(l, 1), y.beta)) return false;
        if (!s101_chain(tl, tl.child(l, 2), tl.child(l, 3), y.cpa) || !s101_chain(tl, tl.child(l, 4), tl.child(l, 5), y.cpb))
            return false;
    }
    return get_u32(tz, last, out.last);
}

}  // namespace

S101Parsed *s101_parse_text(const char *text, size_t len, int fmt)
{
    static thread_local Tree j;
    if (len > kMaxTextBytes) return nullptr;
    if (!parse_json(j, text, len) || j.nodes.empty() || j.nodes[0].kind != kObj) { tree_trim(j); return nullptr; }
    if (fmt == SS_TEXT_AUTO) fmt = j.member(0, "P_MT_ROOT") ? SS_TEXT_WIT : SS_TEXT_JSON;
    S101Parsed *p = new S101Parsed();
    bool ok;
    if (fmt == SS_TEXT_WIT) {
        static thread_local Tree a, b, c, d;
        ok = wit_member(j, "P_MT_ROOT", a) && wit_member(j, "P_EVALS", b) && wit_member(j, "FRI_LAYERS", c) &&
             wit_member(j, "FRI_LAST_LAYER", d) &&
             s101_from_parts(a, a.resolve(a.root), b, b.resolve(b.root), c, c.resolve(c.root), d, d.resolve(d.root), *p);
    } else {
        const uint32_t a = j.member(0, "p_mt_root"), b = j.member(0, "evals"), c = j.member(0, "fri_layers"),
                       d = j.member(0, "fri_last_layer");
        ok = a && b && c && d && s101_from_parts(j, a, j, b, j, c, j, d, *p);
    }
    if (!ok) { delete p; return nullptr; }
    return p;
}

void s101_parsed_shape(const S101Parsed *p, uint32_t *n_layers, uint32_t *max_path)
{
    size_t pm = 0;
    for (const auto &e : p->evals) pm = e.path.size() / 8 > pm ? e.path.size() / 8 : pm;
    for (const auto &l : p->layers) {
        pm = l.cpa.path.size() / 8 > pm ? l.cpa.path.size() / 8 : pm;
        pm = l.cpb.path.size() / 8 > pm ? l.cpb.path.size() / 8 : pm;
    }
    *n_layers = (uint32_t)p->layers.size();
    *max_path = (uint32_t)pm;
}

void s101_parsed_record(const S101Parsed *p, const ss_s101_shape &sh, uint32_t *r)
{
    const uint32_t ML = sh.max_layers, PM = sh.max_path;
    memset(r, 0, (size_t)s101_record_words(ML, PM) * 4);
    memcpy(r, p->root, 32);
    r[8] = (uint32_t)p->layers.size();
    r[9] = p->last;
    uint32_t *o = r + 10;
    auto chain = [&](const S101Chain &c) {
        o[0] = c.ev;
        o[1] = (uint32_t)(c.path.size() / 8);
        memcpy(o + 2, c.path.data(), c.path.size() * 4);
        o += 2 + 8 * PM;
    };
    for (const auto &e : p->evals) chain(e);
    for (uint32_t i = 0; i < ML && i < p->layers.size(); i++) {
        memcpy(o, p->layers[i].root, 32);
        o[8] = p->layers[i].beta;
        o += 9;
        chain(p->layers[i].cpa);
        chain(p->layers[i].cpb);
    }
}

void s101_parsed_free(S101Parsed *p) { delete p; }

unsigned effective_cpus()
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota);
            const unsigned cores = (unsigned)((q + period - 1) / period);
            if (cores >= 1 && cores < n) n = cores;
        }
        fclose(f);
    } else {  // cgroup v1
        long quota = -1, period = 0;
        if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(q, "%ld", &quota) != 1) quota = -1;
            fclose(q);
        }
        if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(q, "%ld", &period) != 1) period = 0;
            fclose(q);
        }
        if (quota > 0 && period > 0) {
            const unsigned cores = (unsigned)((quota + period - 1) / period);
            if (cores >= 1 && cores < n) n = cores;
        }
    }
    // Several library processes on one host (one per GPU: bench.py --gpus N, torch.distributed.run) share the granted
    // cores: each is told its share, or every rank would start a pool as large as the whole grant.
    if (const char *e = getenv("SS_HOST_THREADS")) {
        const long v = atol(e);
        if (v >= 1 && (unsigned long)v < n) n = (unsigned)v;
    }
    return n ? n : 1;
}

}  // namespace ss
