// Canonical proof texts and their templates (see ss_text.h).  Host code, no HIP.
#include "ss_text.h"

#include <cstdio>
#include <cstring>

#include "ss_layout.h"
#include "ss_minimal.h"
#include "ss_shared.h"

namespace ss {
namespace {

// word offsets of a record's sections (include/ss_verify.h)
struct Rec {
    uint32_t N, L, Q, K, head, qstride, fbase, tbase, words;
    uint32_t foff[kMaxList + 1];
    explicit Rec(const ss_stwo_cfg &c) : N(c.n_cols), L(c.lde_log), Q(c.n_queries), K(c.n_layers)
    {
        head = 24 + 4 * N + 64 + 8 * (K + 1) + 4 + 2;
        qstride = N + kCp + 16 * L;
        fbase = head + Q * qstride;
        uint32_t o = 0;
        for (uint32_t l = 0; l <= K; l++) { foff[l] = o; o += Q * (4 + 8 * (L - 1 - l)); }
        tbase = fbase + o;
        words = tbase + (K + 3) * Q;
    }
    uint32_t oods_trace() const { return 24; }
    uint32_t oods_cp() const { return 24 + 4 * N; }
    uint32_t fri_root(uint32_t l) const { return 24 + 4 * N + 64 + 8 * l; }
    uint32_t last() const { return fri_root(K + 1); }
    uint32_t nonce() const { return last() + 4; }
    uint32_t trace_vals(uint32_t q) const { return head + q * qstride; }
    uint32_t cp_vals(uint32_t q) const { return trace_vals(q) + N; }
    uint32_t trace_path(uint32_t q) const { return cp_vals(q) + kCp; }
    uint32_t cp_path(uint32_t q) const { return trace_path(q) + 8 * L; }
    uint32_t fri_wit(uint32_t l, uint32_t q) const { return fbase + foff[l] + q * (4 + 8 * (L - 1 - l)); }
    // the hash_witness list of tree t (0 trace, 1 composition, 2 + l FRI layer l): one full path per query
    bool shared() const { return false; }
    uint32_t n_vals(uint32_t) const { return Q; }  // entries of queried_values[1 + t] / of a layer's fri_witness
    uint32_t n_fw(uint32_t) const { return Q; }
    uint32_t tree_len(uint32_t t) const { return t < 2 ? L : L + 1 - t; }
    uint32_t n_entries(uint32_t t) const { return Q * tree_len(t); }
    uint32_t entry(uint32_t t, uint32_t e) const
    {
        const uint32_t len = tree_len(t), q = e / len, l = e - q * len;
        return (t == 0 ? trace_path(q) : t == 1 ? cp_path(q) : fri_wit(t - 2, q) + 4) + 8 * l;
    }
    uint32_t query(uint32_t) const { return 0; }
};

// word offsets of a SHARED record (include/ss_verify.h, csrc/ss_shared.h): the same members, every distinct sibling once
struct SRec {
    uint32_t N, L, Q, K;
    SharedMap m;
    uint32_t count[kMaxTrees], node0[kMaxTrees];
    // capacity = true: the form the GPU reader writes (full-length lists, tree t's nodes at a fixed base)
    SRec(const ss_stwo_cfg &c, const uint32_t *counts) : N(c.n_cols), L(c.lde_log), Q(c.n_queries), K(c.n_layers)
    {
        m = shared_map(N, L, Q, K);
        uint32_t o = m.nodes;
        for (uint32_t t = 0; t < K + 3; t++) {
            count[t] = counts ? counts[t] : Q * shared_tree_len(L, t);
            node0[t] = o;
            o += 8 * count[t];
        }
    }
    uint32_t oods_trace() const { return 24; }
    uint32_t oods_cp() const { return 24 + 4 * N; }
    uint32_t fri_root(uint32_t l) const { return 24 + 4 * N + 64 + 8 * l; }
    uint32_t last() const { return fri_root(K + 1); }
    uint32_t nonce() const { return last() + 4; }
    uint32_t trace_vals(uint32_t q) const { return m.vals + q * (N + kCp); }
    uint32_t cp_vals(uint32_t q) const { return trace_vals(q) + N; }
    uint32_t fri_wit(uint32_t l, uint32_t q) const { return m.wit + (l * Q + q) * 4; }
    bool shared() const { return true; }
    uint32_t n_vals(uint32_t) const { return Q; }
    uint32_t n_fw(uint32_t) const { return Q; }
    uint32_t n_entries(uint32_t t) const { return count[t]; }
    uint32_t entry(uint32_t t, uint32_t e) const { return node0[t] + 8 * e; }
    uint32_t query(uint32_t q) const { return m.qry + q; }
};

// word offsets of a MINIMAL record (include/ss_verify.h, csrc/ss_minimal.h): the lists are as long as the record says
struct MRec {
    uint32_t N, L, Q, K;
    uint32_t nv[2], nfw[kMaxList + 1], nhw[kMaxTrees];
    uint32_t tv, cv, fw[kMaxList + 1], hw[kMaxTrees];
    MRec(const ss_stwo_cfg &c, const uint32_t *rec) : N(c.n_cols), L(c.lde_log), Q(c.n_queries), K(c.n_layers)
    {
        const MinMap m = min_map(N, L, Q, K);
        uint32_t o = m.data;
        nv[0] = rec[m.nv]; nv[1] = rec[m.nv + 1];
        tv = o; o += nv[0] * N;
        cv = o; o += nv[1] * kCp;
        for (uint32_t l = 0; l <= K; l++) { nfw[l] = rec[m.nfw + l]; fw[l] = o; o += 4 * nfw[l]; }
        for (uint32_t t = 0; t < K + 3; t++) { nhw[t] = rec[m.nhw + t]; hw[t] = o; o += 8 * nhw[t]; }
    }
    uint32_t oods_trace() const { return 24; }
    uint32_t oods_cp() const { return 24 + 4 * N; }
    uint32_t fri_root(uint32_t l) const { return 24 + 4 * N + 64 + 8 * l; }
    uint32_t last() const { return fri_root(K + 1); }
    uint32_t nonce() const { return last() + 4; }
    uint32_t trace_vals(uint32_t i) const { return tv + i * N; }
    uint32_t cp_vals(uint32_t i) const { return cv + i * kCp; }
    uint32_t fri_wit(uint32_t l, uint32_t i) const { return fw[l] + 4 * i; }
    bool shared() const { return false; }
    uint32_t n_vals(uint32_t t) const { return nv[t]; }
    uint32_t n_fw(uint32_t l) const { return nfw[l]; }
    uint32_t n_entries(uint32_t t) const { return nhw[t]; }
    uint32_t entry(uint32_t t, uint32_t e) const { return hw[t] + 8 * e; }
    uint32_t query(uint32_t) const { return 0; }
};

bool pow_bits_of(uint64_t target, uint32_t &bits)
{
    if (target == ~(uint64_t)0) { bits = 0; return true; }
    for (uint32_t b = 1; b <= 63; b++)
        if (target == (((uint64_t)1 << (64 - b)) - 1)) { bits = b; return true; }
    return false;  // (64 bits: target 0, which `v < 0` never meets; no config of the verifier's)
}

// The writers are templates over a sink: TextSink prints the record's numbers, SlotSink prints zeros and
// notes where each number would go.  One description of each format serves both.
struct TextSink {
    std::string &out;
    const uint32_t *rec;
    void lit(const char *s) { out += s; }
    void dec(uint64_t v)
    {
        char buf[24];
        int n = 0;
        do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) out.push_back(buf[--n]);
    }
    void u32(uint32_t w) { dec(rec[w]); }
    void byte(uint32_t w, uint32_t k) { dec((rec[w] >> (8 * (3 - k))) & 255); }  // byte k (0..3, big endian) of word w
    void u64(uint32_t w) { dec(((uint64_t)rec[w] << 32) | rec[w + 1]); }
    void hex256(uint32_t w)
    {
        static const char *d = "0123456789abcdef";
        out += "0x";
        for (uint32_t j = 0; j < 8; j++)
            for (int s = 28; s >= 0; s -= 4) out.push_back(d[(rec[w + j] >> s) & 15]);
    }
    void cst(uint32_t v) { dec(v); }
    void list_begin(uint32_t) {}
    void vals_begin(uint32_t) {}
    void fw_begin(uint32_t) {}
    void dec256(uint32_t w)  // 8 words, most significant first, as a decimal integer
    {
        uint32_t limb[8];
        for (int j = 0; j < 8; j++) limb[j] = rec[w + j];
        char buf[80];
        int n = 0;
        for (;;) {
            uint64_t rem = 0;
            bool any = false;
            for (int j = 0; j < 8; j++) {  // limb /= 10^9
                const uint64_t cur = (rem << 32) | limb[j];
                limb[j] = (uint32_t)(cur / 1000000000u);
                rem = cur % 1000000000u;
                any |= limb[j] != 0;
            }
            for (int d = 0; d < 9 && (any || rem); d++) { buf[n++] = (char)('0' + rem % 10); rem /= 10; }
            if (!any) break;
        }
        if (!n) buf[n++] = '0';
        while (n) out.push_back(buf[--n]);
    }
};

struct SlotSink {
    std::string &out;
    std::vector<TextSlot> &slots;
    std::vector<uint32_t> &at;  // byte offset of every number in `out`
    uint32_t list_at[kMaxTrees] = {}, list_tok[kMaxTrees] = {};  // where the first entry of a hash_witness list starts
    void list_begin(uint32_t t) { list_at[t] = (uint32_t)out.size(); list_tok[t] = (uint32_t)slots.size(); }
    // (the minimal form's other lists: the two flat value lists, the fri_witness list of layer l)
    uint32_t vals_at[2] = {}, vals_tok[2] = {}, fw_at[kMaxList + 1] = {}, fw_tok[kMaxList + 1] = {};
    void vals_begin(uint32_t k) { vals_at[k] = (uint32_t)out.size(); vals_tok[k] = (uint32_t)slots.size(); }
    void fw_begin(uint32_t l) { fw_at[l] = (uint32_t)out.size(); fw_tok[l] = (uint32_t)slots.size(); }
    void lit(const char *s) { out += s; }
    void num(uint32_t dst, uint32_t kind, const char *sample)
    {
        at.push_back((uint32_t)out.size());
        slots.push_back({dst, kind});
        out += sample;
    }
    void u32(uint32_t w) { num(w, kSlotU32, "0"); }
    void byte(uint32_t w, uint32_t k) { num(4 * w + 3 - k, kSlotByte, "0"); }  // little-endian byte address of the record
    void u64(uint32_t w) { num(w, kSlotU64, "0"); }
    void hex256(uint32_t w) { num(w, kSlotHex256, "0x0000000000000000000000000000000000000000000000000000000000000000"); }
    void cst(uint32_t v) { num(v, kSlotConst, "0"); }
    void dec256(uint32_t w) { num(w, kSlotDec256, "0"); }
};

// ------------------------------------------------------------------------------ proof.json (format C)
// formats.stwo_to_json: the member order of tests/data/proof.json.
template <class S, class M>
void json_text(const ss_stwo_cfg &cfg, const M &m, uint32_t pow_bits, TextStyle style, S &s)
{
    const char *cm = style == kStylePython ? ", " : ",", *co = style == kStylePython ? ": " : ":";
    auto key = [&](const char *k) { s.lit("\""); s.lit(k); s.lit("\""); s.lit(co); };
    auto hash_bytes = [&](uint32_t w) {  // a hash as a list of its 32 byte values
        s.lit("[");
        for (uint32_t k = 0; k < 32; k++) { if (k) s.lit(cm); s.byte(w + k / 4, k % 4); }
        s.lit("]");
    };
    auto qm31 = [&](uint32_t w) {
        s.lit("[["); s.u32(w); s.lit(cm); s.u32(w + 1); s.lit("]"); s.lit(cm);
        s.lit("["); s.u32(w + 2); s.lit(cm); s.u32(w + 3); s.lit("]]");
    };
    auto hash_witness = [&](uint32_t tree) {  // concatenated over the queries (or, shared: every distinct sibling once)
        key("hash_witness"); s.lit("[");
        s.list_begin(tree);
        for (uint32_t e = 0, n = m.n_entries(tree); e < n; e++) { if (e) s.lit(cm); hash_bytes(m.entry(tree, e)); }
        s.lit("]"); s.lit(cm); key("column_witness"); s.lit("[]");
    };
    s.lit("{"); key("config"); s.lit("{"); key("pow_bits"); s.cst(pow_bits); s.lit(cm); key("fri_config"); s.lit("{");
    key("log_blowup_factor"); s.cst(cfg.lde_log - cfg.trace_log); s.lit(cm);
    key("log_last_layer_degree_bound"); s.cst(0); s.lit(cm); key("n_queries"); s.cst(m.Q); s.lit("}");
    if (cfg.hash == SS_HASH_BLAKE2S) { s.lit(cm); key("hash"); s.lit("\"blake2s\""); }  // extension key; the reference has none
    s.lit("}"); s.lit(cm);
    key("commitments"); s.lit("[");
    for (uint32_t k = 0; k < 3; k++) { if (k) s.lit(cm); hash_bytes(8 * k); }
    s.lit("]"); s.lit(cm);
    key("sampled_values"); s.lit("[[]"); s.lit(cm); s.lit("[");
    for (uint32_t k = 0; k < m.N; k++) { if (k) s.lit(cm); s.lit("["); qm31(m.oods_trace() + 4 * k); s.lit("]"); }
    s.lit("]"); s.lit(cm); s.lit("[");
    for (uint32_t k = 0; k < kCp; k++) { if (k) s.lit(cm); s.lit("["); qm31(m.oods_cp() + 4 * k); s.lit("]"); }
    s.lit("]]"); s.lit(cm);
    key("decommitments"); s.lit("[{"); key("hash_witness"); s.lit("[]"); s.lit(cm); key("column_witness"); s.lit("[]}"); s.lit(cm);
    s.lit("{"); hash_witness(0); s.lit("}"); s.lit(cm);
    s.lit("{"); hash_witness(1); s.lit("}]"); s.lit(cm);
    key("queried_values"); s.lit("[[]"); s.lit(cm); s.lit("[");
    s.vals_begin(0);
    for (uint32_t q = 0, n = m.n_vals(0); q < n; q++)
        for (uint32_t k = 0; k < m.N; k++) { if (q | k) s.lit(cm); s.u32(m.trace_vals(q) + k); }
    s.lit("]"); s.lit(cm); s.lit("[");
    s.vals_begin(1);
    for (uint32_t q = 0, n = m.n_vals(1); q < n; q++)
        for (uint32_t k = 0; k < kCp; k++) { if (q | k) s.lit(cm); s.u32(m.cp_vals(q) + k); }
    s.lit("]]"); s.lit(cm);
    key("proof_of_work"); s.u64(m.nonce()); s.lit(cm);
    auto layer = [&](uint32_t l) {
        s.lit("{"); key("fri_witness"); s.lit("[");
        s.fw_begin(l);
        for (uint32_t q = 0, n = m.n_fw(l); q < n; q++) { if (q) s.lit(cm); qm31(m.fri_wit(l, q)); }
        s.lit("]"); s.lit(cm); key("decommitment"); s.lit("{");
        hash_witness(2 + l);
        s.lit("}"); s.lit(cm); key("commitment"); hash_bytes(m.fri_root(l)); s.lit("}");
    };
    key("fri_proof"); s.lit("{"); key("first_layer"); layer(0); s.lit(cm); key("inner_layers"); s.lit("[");
    for (uint32_t l = 1; l <= m.K; l++) { if (l > 1) s.lit(cm); layer(l); }
    s.lit("]"); s.lit(cm); key("last_layer_poly"); s.lit("{"); key("coeffs"); s.lit("["); qm31(m.last()); s.lit("]"); s.lit(cm);
    key("log_size"); s.cst(0); s.lit("}}");
    if (m.shared()) {  // formats.stwo_to_json(shared=True): the positions, last member
        s.lit(cm); key("queries"); s.lit("[");
        for (uint32_t q = 0; q < m.Q; q++) { if (q) s.lit(cm); s.u32(m.query(q)); }
        s.lit("]");
    }
    s.lit("}");
}

// -------------------------------------------------------------------------------- proof.wit (format D)
// formats.stwo_to_wit = stwo-verifier/scripts/generate_wit.py:106-245, printed by json.dumps(indent=4).
template <class S>
void wit_text(const ss_stwo_cfg &cfg, S &s)
{
    const Rec m(cfg);
    auto qm31 = [&](uint32_t w) {
        s.lit("(("); s.u32(w); s.lit(", "); s.u32(w + 1); s.lit("), ("); s.u32(w + 2); s.lit(", "); s.u32(w + 3); s.lit("))");
    };
    auto lst = [&](uint32_t w, uint32_t len) {
        s.lit("list![");
        for (uint32_t l = 0; l < len; l++) { if (l) s.lit(", "); s.hex256(w + 8 * l); }
        s.lit("]");
    };
    const char *QM = "((u32, u32), (u32, u32))", *MP = "List<u256, ";  // + 32 + ">"
    auto mp = [&]() { s.lit(MP); s.cst(32); s.lit(">"); };
    auto entry = [&](const char *name) { s.lit("    \""); s.lit(name); s.lit("\": {\n        \"value\": \""); };
    auto type = [&]() { s.lit("\",\n        \"type\": \""); };
    auto close = [&](bool last) { s.lit(last ? "\"\n    }\n" : "\"\n    },\n"); };
    s.lit("{\n");
    entry("COMMITMENTS");
    s.lit("("); s.hex256(0); s.lit(", "); s.hex256(8); s.lit(", "); s.hex256(16); s.lit(")");
    type(); s.lit("(u256, u256, u256)"); close(false);
    entry("DECOMMITMENTS");
    s.lit("[");
    for (uint32_t q = 0; q < m.Q; q++) {
        if (q) s.lit(", ");
        s.lit("(([");
        for (uint32_t k = 0; k < m.N; k++) { if (k) s.lit(", "); s.lit("["); s.u32(m.trace_vals(q) + k); s.lit("]"); }
        s.lit("], "); lst(m.trace_path(q), m.L); s.lit("), ([");
        for (uint32_t k = 0; k < kCp; k++) { if (k) s.lit(", "); s.u32(m.cp_vals(q) + k); }
        s.lit("], "); lst(m.cp_path(q), m.L); s.lit("))");
    }
    s.lit("]");
    type();
    s.lit("[(([[u32; "); s.cst(1); s.lit("]; "); s.cst(m.N); s.lit("], "); mp(); s.lit("), ([u32; "); s.cst(16); s.lit("], "); mp();
    s.lit(")); "); s.cst(m.Q); s.lit("]");
    close(false);
    entry("OODS_EVALS");
    s.lit("([");
    for (uint32_t k = 0; k < m.N; k++) { if (k) s.lit(", "); s.lit("["); qm31(m.oods_trace() + 4 * k); s.lit("]"); }
    s.lit("], [");
    for (uint32_t k = 0; k < kCp; k++) { if (k) s.lit(", "); qm31(m.oods_cp() + 4 * k); }
    s.lit("])");
    type();
    s.lit("([["); s.lit(QM); s.lit("; "); s.cst(1); s.lit("]; "); s.cst(m.N); s.lit("], ["); s.lit(QM); s.lit("; "); s.cst(16); s.lit("])");
    close(false);
    entry("FRI_COMMITMENTS");
    s.lit("("); s.hex256(m.fri_root(0)); s.lit(", [");
    for (uint32_t l = 1; l <= m.K; l++) { if (l > 1) s.lit(", "); s.hex256(m.fri_root(l)); }
    s.lit("], "); qm31(m.last()); s.lit(")");
    type();
    s.lit("(u256, [u256; "); s.cst(m.K); s.lit("], "); s.lit(QM); s.lit(")");
    close(false);
    entry("FRI_DECOMMITMENTS");
    auto fl = [&](uint32_t l) {
        s.lit("[");
        for (uint32_t q = 0; q < m.Q; q++) {
            if (q) s.lit(", ");
            s.lit("("); qm31(m.fri_wit(l, q)); s.lit(", "); lst(m.fri_wit(l, q) + 4, m.L - 1 - l); s.lit(")");
        }
        s.lit("]");
    };
    s.lit("("); fl(0); s.lit(", [");
    for (uint32_t l = 1; l <= m.K; l++) { if (l > 1) s.lit(", "); fl(l); }
    s.lit("])");
    type();
    auto fld = [&]() { s.lit("[("); s.lit(QM); s.lit(", "); mp(); s.lit("); "); s.cst(m.Q); s.lit("]"); };
    s.lit("("); fld(); s.lit(", ["); fld(); s.lit("; "); s.cst(m.K); s.lit("])");
    close(false);
    entry("POW_NONCE");
    s.u64(m.nonce());
    type(); s.lit("u64"); close(true);
    s.lit("}");
}

// -------------------------------------------------------------------------------------- stark101
// record of shape {kS101Layers, kS101Path} (include/ss_verify.h): word offsets
struct Rec101 {
    static constexpr uint32_t ML = kS101Layers, PM = kS101Path, chain = 2 + 8 * PM;
    static uint32_t eval(uint32_t k) { return 10 + k * chain; }                       // ev, len, path
    static uint32_t layer(uint32_t i) { return 10 + 3 * chain + i * (9 + 2 * chain); }  // root[8], beta, cpa chain, cpb chain
    static uint32_t words() { return layer(ML); }
    static uint32_t layer_len(uint32_t i) { return PM - i; }
};

// formats.stark101_to_json: {"p_mt_root", "evals", "fri_layers", "fri_last_layer"} (prover.py:108,143-167)
template <class S>
void s101_json_text(TextStyle style, S &s)
{
    typedef Rec101 R;
    const char *cm = style == kStylePython ? ", " : ",", *co = style == kStylePython ? ": " : ":";
    auto key = [&](const char *k) { s.lit("\""); s.lit(k); s.lit("\""); s.lit(co); };
    auto path = [&](uint32_t w, uint32_t len) {
        s.lit("[");
        for (uint32_t l = 0; l < len; l++) { if (l) s.lit(cm); s.dec256(w + 8 * l); }
        s.lit("]");
    };
    s.lit("{"); key("p_mt_root"); s.dec256(0); s.lit(cm);
    key("evals"); s.lit("[");
    for (uint32_t k = 0; k < 3; k++) {
        if (k) s.lit(cm);
        s.lit("["); s.u32(R::eval(k)); s.lit(cm); path(R::eval(k) + 2, R::PM); s.lit("]");
    }
    s.lit("]"); s.lit(cm);
    key("fri_layers"); s.lit("[");
    for (uint32_t i = 0; i < R::ML; i++) {
        const uint32_t b = R::layer(i), a = b + 9, c = a + R::chain, len = R::layer_len(i);
        if (i) s.lit(cm);
        s.lit("["); s.dec256(b); s.lit(cm); s.u32(b + 8); s.lit(cm); s.u32(a); s.lit(cm); path(a + 2, len); s.lit(cm);
        s.u32(c); s.lit(cm); path(c + 2, len); s.lit("]");
    }
    s.lit("]"); s.lit(cm);
    key("fri_last_layer"); s.u32(9); s.lit("}");
}

// formats.stark101_to_wit = stark101/scripts/generate_wit.py:13-30, printed by json.dumps(indent=4)
template <class S>
void s101_wit_text(S &s)
{
    typedef Rec101 R;
    auto lst = [&](uint32_t w, uint32_t len) {
        s.lit("list![");
        for (uint32_t l = 0; l < len; l++) { if (l) s.lit(", "); s.dec256(w + 8 * l); }
        s.lit("]");
    };
    auto mp = [&]() { s.lit("List<u256, "); s.cst(32); s.lit(">"); };
    auto entry = [&](const char *name) { s.lit("    \""); s.lit(name); s.lit("\": {\n        \"value\": \""); };
    auto type = [&]() { s.lit("\",\n        \"type\": \""); };
    auto close = [&](bool last) { s.lit(last ? "\"\n    }\n" : "\"\n    },\n"); };
    s.lit("{\n");
    entry("P_MT_ROOT"); s.dec256(0); type(); s.lit("u256"); close(false);
    entry("P_EVALS");
    s.lit("(");
    for (uint32_t k = 0; k < 3; k++) {
        if (k) s.lit(", ");
        s.lit("("); s.u32(R::eval(k)); s.lit(", "); lst(R::eval(k) + 2, R::PM); s.lit(")");
    }
    s.lit(")");
    type();
    s.lit("(");
    for (uint32_t k = 0; k < 3; k++) { if (k) s.lit(", "); s.lit("(u32, "); mp(); s.lit(")"); }
    s.lit(")");
    close(false);
    entry("FRI_LAYERS");
    s.lit("list![");
    for (uint32_t i = 0; i < R::ML; i++) {
        const uint32_t b = R::layer(i), a = b + 9, c = a + R::chain, len = R::layer_len(i);
        if (i) s.lit(", ");
        s.lit("(("); s.dec256(b); s.lit(", "); s.u32(b + 8); s.lit(", "); s.u32(a); s.lit(", "); lst(a + 2, len); s.lit(", ");
        s.u32(c); s.lit(", "); lst(c + 2, len); s.lit("))");
    }
    s.lit("]");
    type();
    s.lit("List<((u256, u32, u32, "); mp(); s.lit(", u32, "); mp(); s.lit("), "); s.cst(32); s.lit(")");
    close(false);
    entry("FRI_LAST_LAYER"); s.u32(9); type(); s.lit("u32"); close(true);
    s.lit("}");
}

// the words a canonical stark101 text implies: n_layers and every path length
void s101_fixed_words(std::vector<uint32_t> &fixed)
{
    typedef Rec101 R;
    fixed.push_back(8); fixed.push_back(R::ML);
    for (uint32_t k = 0; k < 3; k++) { fixed.push_back(R::eval(k) + 1); fixed.push_back(R::PM); }
    for (uint32_t i = 0; i < R::ML; i++) {
        const uint32_t a = R::layer(i) + 9;
        fixed.push_back(a + 1); fixed.push_back(R::layer_len(i));
        fixed.push_back(a + R::chain + 1); fixed.push_back(R::layer_len(i));
    }
}

bool s101_canonical_record(const uint32_t *rec)
{
    std::vector<uint32_t> fixed;
    s101_fixed_words(fixed);
    for (size_t i = 0; i < fixed.size(); i += 2)
        if (rec[fixed[i]] != fixed[i + 1]) return false;
    return true;
}

bool cfg_writable(const ss_stwo_cfg &c)
{
    return c.hash <= SS_HASH_BLAKE2S && stwo_cfg_ok(c.n_cols, c.trace_log, c.lde_log, c.n_queries, c.n_layers, c.mode & 1);
}

bool uniform_paths(const ss_stwo_cfg &cfg, const uint32_t *rec)
{
    const Rec m(cfg);
    for (uint32_t kind = 0; kind < m.K + 3; kind++) {
        const uint32_t want = kind < 2 ? m.L : m.L - 1 - (kind - 2);
        for (uint32_t q = 0; q < m.Q; q++)
            if (rec[m.tbase + kind * m.Q + q] != want) return false;
    }
    return true;
}

}  // namespace

bool stwo_write_json(const ss_stwo_cfg &cfg, const uint32_t *record, TextStyle style, std::string &out)
{
    uint32_t bits;
    out.clear();
    if (!cfg_writable(cfg) || !pow_bits_of(cfg.pow_target, bits) || !uniform_paths(cfg, record)) return false;
    TextSink s{out, record};
    json_text(cfg, Rec(cfg), bits, style, s);
    return true;
}

bool stwo_write_json_shared(const ss_stwo_cfg &cfg, const uint32_t *shared, size_t words, TextStyle style, std::string &out)
{
    uint32_t bits;
    out.clear();
    if (!cfg_writable(cfg) || !pow_bits_of(cfg.pow_target, bits) || !shared) return false;
    const SharedMap m = shared_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    if (words < m.nodes) return false;
    SharedPlan plan;
    if (!shared_plan(m, shared + m.qry, plan)) return false;
    size_t total = m.nodes;
    for (uint32_t t = 0; t < m.K + 3; t++) {
        if (shared[m.cnt + t] != plan.base[t][m.Q]) return false;
        total += 8 * (size_t)plan.base[t][m.Q];
    }
    if (total != words) return false;
    TextSink s{out, shared};
    json_text(cfg, SRec(cfg, shared + m.cnt), bits, style, s);
    return true;
}

// formats.stwo_minimal_to_json: the same members with the lists as upstream stwo fills them
bool stwo_write_json_minimal(const ss_stwo_cfg &cfg, const uint32_t *rec, size_t words, TextStyle style, std::string &out)
{
    uint32_t bits;
    out.clear();
    if (!cfg_writable(cfg) || !pow_bits_of(cfg.pow_target, bits) || !rec) return false;
    const MinMap m = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    if (words < m.data) return false;
    size_t total = m.data;
    if (rec[m.nv] > m.Q || rec[m.nv + 1] > m.Q) return false;
    total += (size_t)rec[m.nv] * m.N + (size_t)rec[m.nv + 1] * kCp;
    for (uint32_t l = 0; l <= m.K; l++) {
        if (rec[m.nfw + l] > m.Q) return false;
        total += 4 * (size_t)rec[m.nfw + l];
    }
    for (uint32_t t = 0; t < m.K + 3; t++) {
        if (rec[m.nhw + t] > m.Q * min_tree_len(m.L, t)) return false;
        total += 8 * (size_t)rec[m.nhw + t];
    }
    if (total != words) return false;
    TextSink s{out, rec};
    json_text(cfg, MRec(cfg, rec), bits, style, s);
    return true;
}

bool stwo_write_wit(const ss_stwo_cfg &cfg, const uint32_t *record, std::string &out)
{
    out.clear();
    if (!cfg_writable(cfg) || !uniform_paths(cfg, record)) return false;
    TextSink s{out, record};
    wit_text(cfg, s);
    return true;
}

static bool skeleton_of(const std::string &sample, const std::vector<uint32_t> &at, TextTemplateHost &out,
                        const std::vector<uint32_t> *marks = nullptr, std::vector<uint32_t> *mark_skel = nullptr);

bool s101_write_json(const uint32_t *record, TextStyle style, std::string &out)
{
    out.clear();
    if (!s101_canonical_record(record)) return false;
    TextSink s{out, record};
    s101_json_text(style, s);
    return true;
}

bool s101_write_wit(const uint32_t *record, std::string &out)
{
    out.clear();
    if (!s101_canonical_record(record)) return false;
    TextSink s{out, record};
    s101_wit_text(s);
    return true;
}

TextTemplate TextTemplateHost::view() const
{
    TextTemplate t;
    t.skel = skel.data(); t.skel_len = skel_len;
    t.slots = slots.data(); t.n_slots = (uint32_t)slots.size();
    t.record_words = record_words;
    t.tbase = tbase; t.n_trailer = (uint32_t)trailer.size(); t.trailer = trailer.data();
    t.n_fixed = (uint32_t)(fixed.size() / 2); t.fixed = fixed.data();
    return t;
}

void stwo_build_template(const ss_stwo_cfg &cfg, int fmt, TextTemplateHost &out)
{
    out = TextTemplateHost();
    uint32_t bits = 0;
    if (!cfg_writable(cfg) || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT && fmt != SS_TEXT_JSON_SHARED && fmt != SS_TEXT_JSON_MINIMAL)) return;
    if (fmt != SS_TEXT_WIT && !pow_bits_of(cfg.pow_target, bits)) return;  // no proof.json can declare this target
    std::string sample;
    std::vector<uint32_t> at;
    SlotSink s{sample, out.slots, at};
    if (fmt == SS_TEXT_JSON_SHARED) {
        // the full-length text over a capacity-form shared record; where its lists start, in skeleton coordinates
        const SRec sm(cfg, nullptr);
        json_text(cfg, sm, bits, kStyleCompact, s);
        std::vector<uint32_t> marks(s.list_at, s.list_at + sm.K + 3), mark_skel;
        if (!skeleton_of(sample, at, out, &marks, &mark_skel)) { out = TextTemplateHost(); return; }
        SharedTextInfo &I = out.sinfo;
        I.n_trees = sm.K + 3; I.Q = sm.Q; I.L = sm.L; I.K = sm.K;
        I.entry_toks = 32;
        I.entry_skel = 32 + 31 + 2 + 1;
        for (uint32_t t = 0; t < I.n_trees; t++) { I.S[t] = mark_skel[t]; I.T[t] = s.list_tok[t]; I.n[t] = sm.count[t]; }
        out.record_words = sm.m.nodes + 8 * sm.m.max_nodes;
        out.tbase = 0;
        out.ok = true;
        return;
    }
    if (fmt == SS_TEXT_JSON_MINIMAL) {
        // the full-length text over a capacity-form minimal record: every list as long as the config allows
        const MinMap mm = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
        std::vector<uint32_t> head(mm.data, 0);
        head[mm.nv] = head[mm.nv + 1] = mm.Q;
        for (uint32_t l = 0; l <= mm.K; l++) head[mm.nfw + l] = mm.Q;
        for (uint32_t t = 0; t < mm.K + 3; t++) head[mm.nhw + t] = mm.Q * min_tree_len(mm.L, t);
        const MRec mr(cfg, head.data());
        json_text(cfg, mr, bits, kStyleCompact, s);
        MinTextInfo &I = out.minfo;
        I.n_lists = 2 * mm.K + 6; I.N = mm.N; I.Q = mm.Q; I.L = mm.L; I.K = mm.K;
        std::vector<uint32_t> marks, mark_skel;
        auto list = [&](uint32_t j, uint32_t at_byte, uint32_t tok, uint32_t n, uint32_t es, uint32_t et, uint32_t word, uint32_t per) {
            marks.push_back(at_byte);
            I.T[j] = tok; I.n[j] = n; I.es[j] = es; I.et[j] = et; I.word[j] = word; I.per[j] = per;
        };
        const uint32_t hash_skel = 32 + 31 + 2 + 1, qm31_skel = 13 + 1;  // "[m,..,m]," and "[[m,m],[m,m]],"
        list(0, s.list_at[0], s.list_tok[0], mr.nhw[0], hash_skel, 32, mm.nhw + 0, 1);
        list(1, s.list_at[1], s.list_tok[1], mr.nhw[1], hash_skel, 32, mm.nhw + 1, 1);
        list(2, s.vals_at[0], s.vals_tok[0], mm.Q * mm.N, 2, 1, mm.nv, mm.N);
        list(3, s.vals_at[1], s.vals_tok[1], mm.Q * kCp, 2, 1, mm.nv + 1, kCp);
        for (uint32_t l = 0; l <= mm.K; l++) {
            list(4 + 2 * l, s.fw_at[l], s.fw_tok[l], mm.Q, qm31_skel, 4, mm.nfw + l, 1);
            list(5 + 2 * l, s.list_at[2 + l], s.list_tok[2 + l], mr.nhw[2 + l], hash_skel, 32, mm.nhw + 2 + l, 1);
        }
        if (!skeleton_of(sample, at, out, &marks, &mark_skel) || mark_skel.size() != I.n_lists) { out = TextTemplateHost(); return; }
        bool good = true;
        for (uint32_t j = 0; j < I.n_lists && good; j++) {  // the entry sizes above are what the writer prints
            I.S[j] = mark_skel[j];
            const uint32_t end = I.S[j] + I.n[j] * I.es[j] - 1;
            good = I.n[j] >= 1 && end < out.skel_len && out.skel[end] == ']' && out.skel[I.S[j] - 1] == '[' &&
                   (I.n[j] == 1 || out.skel[I.S[j] + I.es[j] - 1] == ',') && (j == 0 || I.S[j] > I.S[j - 1]);
        }
        if (!good) { out = TextTemplateHost(); return; }
        out.record_words = (uint32_t)min_max_words(mm);
        out.tbase = 0;
        out.ok = true;
        return;
    }
    if (fmt == SS_TEXT_JSON) json_text(cfg, Rec(cfg), bits, kStyleCompact, s);
    else wit_text(cfg, s);
    if (!skeleton_of(sample, at, out)) { out = TextTemplateHost(); return; }
    const Rec m(cfg);
    out.record_words = m.words;
    out.tbase = m.tbase;
    for (uint32_t kind = 0; kind < m.K + 3; kind++)
        for (uint32_t q = 0; q < m.Q; q++) out.trailer.push_back(kind < 2 ? m.L : m.L - 1 - (kind - 2));
    out.ok = true;
}

void s101_build_template(int fmt, TextTemplateHost &out)
{
    out = TextTemplateHost();
    if (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT) return;
    std::string sample;
    std::vector<uint32_t> at;
    SlotSink s{sample, out.slots, at};
    if (fmt == SS_TEXT_JSON) s101_json_text(kStyleCompact, s);
    else s101_wit_text(s);
    if (!skeleton_of(sample, at, out)) { out = TextTemplateHost(); return; }
    out.record_words = Rec101::words();
    s101_fixed_words(out.fixed);
    out.ok = true;
}

// the skeleton of the sample text, by the tokenizer itself; its numbers must be exactly the sink's
// (marks: ascending sample offsets whose skeleton positions the caller wants)
bool skeleton_of(const std::string &sample, const std::vector<uint32_t> &at, TextTemplateHost &out,
                 const std::vector<uint32_t> *marks, std::vector<uint32_t> *mark_skel)
{
    uint32_t run = kRunNone, in_str = 0;
    size_t k = 0, mk = 0;
    bool good = true;
    for (size_t i = 0; i < sample.size() && good; i++) {
        while (marks && mk < marks->size() && (*marks)[mk] == i) { mark_skel->push_back((uint32_t)out.skel.size()); mk++; }
        const uint32_t c = (unsigned char)sample[i];
        if (txt_is_bad(c)) { good = false; break; }
        const uint32_t r = scan_byte(c, run, in_str);
        if (r & 2) {
            good = k < at.size() && at[k] == i;
            k++;
            out.skel.push_back(kSkelMark);
        }
        if (r & 1) out.skel.push_back((uint8_t)c);
    }
    good = good && k == at.size() && in_str == 0;
    if (!good) return false;
    out.skel_len = (uint32_t)out.skel.size();
    out.skel.resize(out.skel.size() + kSkelSlack, 0);
    return true;
}

// ------------------------------------------------------------------------- the fast path, scalar
namespace {

// number token body[0..n) (alnum run starting with a digit) -> record, by kind; false = not canonical / out of range
bool place_number(const TextSlot &sl, const unsigned char *body, size_t n, uint32_t *rec)
{
    if (sl.kind == kSlotHex256) {
        if (n != 66 || body[0] != '0' || (body[1] != 'x' && body[1] != 'X')) return false;
        for (uint32_t j = 0; j < 8; j++) {
            uint32_t w = 0;
            for (uint32_t d = 0; d < 8; d++) {
                const uint32_t c = body[2 + 8 * j + d];
                uint32_t v;
                if (c - '0' < 10u) v = c - '0';
                else if ((c | 0x20) - 'a' < 6u) v = (c | 0x20) - 'a' + 10;
                else return false;
                w = (w << 4) | v;
            }
            rec[sl.dst + j] = w;
        }
        return true;
    }
    if (n > 1 && body[0] == '0') return false;  // canonical decimals only
    if (sl.kind == kSlotDec256) {
        if (n > 78) return false;
        uint32_t limb[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // little endian
        for (size_t i = 0; i < n;) {
            const size_t k = n - i < 9 ? n - i : 9;
            uint32_t chunk = 0, mul = 1;
            for (size_t j = 0; j < k; j++) {
                const uint32_t d = body[i + j] - '0';
                if (d > 9) return false;
                chunk = chunk * 10 + d;
                mul *= 10;
            }
            uint64_t carry = chunk;
            for (int l = 0; l < 8; l++) {
                const uint64_t v = (uint64_t)limb[l] * mul + carry;
                limb[l] = (uint32_t)v;
                carry = v >> 32;
            }
            if (carry) return false;  // >= 2^256
            i += k;
        }
        for (int l = 0; l < 8; l++) rec[sl.dst + l] = limb[7 - l];
        return true;
    }
    if (n > 20) return false;  // no u64 has more digits
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t d = body[i] - '0';
        if (d > 9) return false;
        if (v > (~(uint64_t)0 - d) / 10) return false;  // >= 2^64
        v = v * 10 + d;
    }
    switch (sl.kind) {
    case kSlotU32:
        if (v > 0xffffffffull) return false;
        rec[sl.dst] = (uint32_t)v;
        return true;
    case kSlotByte:
        if (v > 255) return false;
        reinterpret_cast<uint8_t *>(rec)[sl.dst] = (uint8_t)v;  // records are little-endian words
        return true;
    case kSlotU64:
        rec[sl.dst] = (uint32_t)(v >> 32);
        rec[sl.dst + 1] = (uint32_t)v;
        return true;
    case kSlotConst:
        return v == sl.dst;
    }
    return false;
}

}  // namespace

bool shared_text_scan_reference(const ss_stwo_cfg &cfg, const TextTemplateHost &th, const char *text, size_t len, uint32_t *record)
{
    if (!th.ok) return false;
    const TextTemplate t = th.view();
    const SharedTextInfo &I = th.sinfo;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text);
    // the positions, from the last KiB; what they imply
    TextHint h;
    const uint32_t tail = len < 1024 ? (uint32_t)len : 1024;
    if (!shared_text_hint(p + (len - tail), tail, I.Q, h.pos)) return false;
    const SharedMap m = shared_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    SharedPlan plan;
    if (!shared_plan(m, h.pos, plan)) return false;
    uint32_t counts[kMaxTrees];
    for (uint32_t k = 0; k < I.n_trees; k++) counts[k] = plan.base[k][I.Q];
    shared_text_gaps(I, counts, t.skel_len, t.n_slots, h.g);
    // the scan of text_scan_reference through the gap maps, into a capacity-form shared record
    std::vector<uint32_t> cap(t.record_words, 0);
    uint32_t run = kRunNone, in_str = 0, sk = 0, tok = 0;
    for (size_t i = 0; i < len; i++) {
        const uint32_t c = p[i];
        if (txt_is_bad(c)) return false;
        const uint32_t r = scan_byte(c, run, in_str);
        if (r & 2) {
            if (sk >= h.g.skel_len || tok >= h.g.n_slots || t.skel[gap_map(h.g.G, h.g.D, I.n_trees, sk)] != kSkelMark) return false;
            sk++;
            size_t e = i;
            while (e < len && e - i <= kMaxTokenBytes && txt_is_alnum(p[e])) e++;
            if (!place_number(t.slots[gap_map(h.g.Gk, h.g.Dk, I.n_trees, tok)], p + i, e - i, cap.data())) return false;
            tok++;
        }
        if (r & 1) {
            if (sk >= h.g.skel_len || t.skel[gap_map(h.g.G, h.g.D, I.n_trees, sk)] != c) return false;
            sk++;
        }
    }
    if (sk != h.g.skel_len || tok != h.g.n_slots || in_str) return false;
    for (uint32_t q = 0; q < I.Q; q++)
        if (cap[m.qry + q] != h.pos[q]) return false;
    shared_expand_host(m, plan, cap.data(), true, record);
    return true;
}

bool minimal_to_capacity(const ss_stwo_cfg &cfg, const uint32_t *rec, size_t words, uint32_t *cap)
{
    const MinMap m = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    if (words < m.data) return false;
    if (rec[m.nv] > m.Q || rec[m.nv + 1] > m.Q) return false;
    size_t o = m.data, c = m.data;
    memcpy(cap, rec, (size_t)m.data * 4);
    auto move = [&](size_t have, size_t room) {
        if (o + have > words) return false;
        memcpy(cap + c, rec + o, have * 4);
        o += have;
        c += room;
        return true;
    };
    if (!move((size_t)rec[m.nv] * m.N, (size_t)m.Q * m.N) || !move((size_t)rec[m.nv + 1] * kCp, (size_t)m.Q * kCp)) return false;
    for (uint32_t l = 0; l <= m.K; l++)
        if (rec[m.nfw + l] > m.Q || !move(4 * (size_t)rec[m.nfw + l], 4 * (size_t)m.Q)) return false;
    for (uint32_t t = 0; t < m.K + 3; t++) {
        const size_t room = (size_t)m.Q * min_tree_len(m.L, t);
        if (rec[m.nhw + t] > room || !move(8 * (size_t)rec[m.nhw + t], 8 * room)) return false;
    }
    return o == words;
}

void minimal_compact(const ss_stwo_cfg &cfg, const uint32_t *cap, std::vector<uint32_t> &out)
{
    const MinMap m = min_map(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers);
    out.assign(cap, cap + m.data);
    size_t c = m.data;
    auto move = [&](size_t have, size_t room) { out.insert(out.end(), cap + c, cap + c + have); c += room; };
    move((size_t)cap[m.nv] * m.N, (size_t)m.Q * m.N);
    move((size_t)cap[m.nv + 1] * kCp, (size_t)m.Q * kCp);
    for (uint32_t l = 0; l <= m.K; l++) move(4 * (size_t)cap[m.nfw + l], 4 * (size_t)m.Q);
    for (uint32_t t = 0; t < m.K + 3; t++) move(8 * (size_t)cap[m.nhw + t], 8 * (size_t)m.Q * min_tree_len(m.L, t));
}

bool minimal_text_scan_reference(const ss_stwo_cfg &cfg, const TextTemplateHost &th, const char *text, size_t len, uint32_t *cap)
{
    if (!th.ok || len > 0xffffffffu) return false;
    const TextTemplate t = th.view();
    const MinTextInfo &I = th.minfo;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text);
    // pass 1: the numbers in front of every landmark (what the device's scan leaves per window, text_landmark_kernel)
    std::vector<uint32_t> lm[3];
    {
        uint32_t run = kRunNone, in_str = 0, tok = 0;
        for (size_t i = 0; i < len; i++) {
            const uint32_t c = p[i];
            if (txt_is_bad(c)) return false;
            const int kind = min_text_landmark(p + i, (uint32_t)(len - i));
            if (kind >= 0) {
                if (lm[kind].size() >= kMaxLandmarks) return false;
                lm[kind].push_back(tok);
            }
            if (scan_byte(c, run, in_str) & 2) tok++;
        }
    }
    uint32_t counts[kMaxTextLists];
    if (!min_text_counts(I, lm[kLmHash].data(), (uint32_t)lm[kLmHash].size(), lm[kLmColumn].data(), (uint32_t)lm[kLmColumn].size(),
                         lm[kLmPow].data(), (uint32_t)lm[kLmPow].size(), counts))
        return false;
    MinTextGaps g;
    min_text_gaps(I, counts, t.skel_len, t.n_slots, g);
    // pass 2: the scan of text_scan_reference through the gap maps
    uint32_t run = kRunNone, in_str = 0, sk = 0, tok = 0;
    for (size_t i = 0; i < len; i++) {
        const uint32_t c = p[i];
        const uint32_t r = scan_byte(c, run, in_str);
        if (r & 2) {
            if (sk >= g.skel_len || tok >= g.n_slots || t.skel[gap_map(g.G, g.D, I.n_lists, sk)] != kSkelMark) return false;
            sk++;
            size_t e = i;
            while (e < len && e - i <= kMaxTokenBytes && txt_is_alnum(p[e])) e++;
            if (!place_number(t.slots[gap_map(g.Gk, g.Dk, I.n_lists, tok)], p + i, e - i, cap)) return false;
            tok++;
        }
        if (r & 1) {
            if (sk >= g.skel_len || t.skel[gap_map(g.G, g.D, I.n_lists, sk)] != c) return false;
            sk++;
        }
    }
    if (sk != g.skel_len || tok != g.n_slots || in_str) return false;
    for (uint32_t j = 0; j < I.n_lists; j++) cap[I.word[j]] = counts[j] / I.per[j];
    return true;
}

bool text_scan_reference(const TextTemplate &t, const char *text, size_t len, uint32_t *rec)
{
    if (!t.skel) return false;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text);
    uint32_t run = kRunNone, in_str = 0, sk = 0, tok = 0;
    for (size_t i = 0; i < len; i++) {
        const uint32_t c = p[i];
        if (txt_is_bad(c)) return false;
        const uint32_t r = scan_byte(c, run, in_str);
        if (r & 2) {
            if (sk >= t.skel_len || t.skel[sk] != kSkelMark || tok >= t.n_slots) return false;
            sk++;
            size_t e = i;
            while (e < len && e - i <= kMaxTokenBytes && txt_is_alnum(p[e])) e++;
            if (!place_number(t.slots[tok], p + i, e - i, rec)) return false;
            tok++;
        }
        if (r & 1) {
            if (sk >= t.skel_len || t.skel[sk] != c) return false;
            sk++;
        }
    }
    if (sk != t.skel_len || tok != t.n_slots) return false;
    for (uint32_t i = 0; i < t.n_trailer; i++) rec[t.tbase + i] = t.trailer[i];
    for (uint32_t i = 0; i < t.n_fixed; i++) rec[t.fixed[2 * i]] = t.fixed[2 * i + 1];
    return true;
}

}  // namespace ss
