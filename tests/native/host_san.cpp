// Sanitizer driver of the library's host code that indexes caller-owned buffers: the record -> batch
// packers (csrc/ss_pack.cpp), the canonical text writers, the template builder and the scalar statement of
// the GPU reader's rule (csrc/ss_text.cpp).  Built by tests/test_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all
// Exact-size heap buffers make any out-of-bounds index an ASan report; UBSan catches shifts / overflows.
//
//   host_san <seed> <mutants> file.json file.wit     (the reference's production proof in both formats)
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <stdexcept>

#include <sys/wait.h>
#include <unistd.h>

#include "ss_ingest.h"
#include "ss_pack.h"
#include "ss_pool.h"
#include "ss_minimal.h"
#include "ss_shared.h"
#include "ss_text.h"

int ss::set_err(int code, const char *, ...) { return code; }  // (the library's lives in ss_api.hip)

static uint64_t rng_state;
static uint32_t rnd()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

static std::string slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static std::string mutate(const std::string &s)
{
    std::string m = s;
    const int n_edits = 1 + rnd() % 3;
    for (int e = 0; e < n_edits && !m.empty(); e++) {
        const size_t at = rnd() % m.size();
        switch (rnd() % 7) {
        case 0: m[at] = (char)(m[at] ^ (1 << (rnd() % 8))); break;
        case 1: m.erase(at, 1 + rnd() % 40); break;
        case 2: m.insert(at, m.substr(at, 1 + rnd() % 60)); break;
        case 3: m.resize(at); break;
        case 4: m[at] = "0123456789"[rnd() % 10]; break;
        case 5: m[at] = "[](){},:\" x_"[rnd() % 12]; break;
        default: m.insert(at, std::string(1 + rnd() % 90, (char)('0' + rnd() % 10))); break;
        }
    }
    return m;
}

static ss_stwo_cfg make_cfg(uint32_t N, uint32_t TL, uint32_t L, uint32_t Q, uint32_t K, uint32_t hash, uint32_t flags)
{
    ss_stwo_cfg c;
    memset(&c, 0, sizeof c);
    c.n_cols = N; c.trace_log = TL; c.lde_log = L; c.n_queries = Q; c.n_layers = K; c.mode = 1;
    c.pow_target = 0x07ffffffffffffffull; c.hash = hash; c.flags = flags;
    return c;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: host_san seed mutants proof.json proof.wit\n"); return 2; }
    rng_state = strtoull(argv[1], nullptr, 10);
    const int mutants = atoi(argv[2]);
    size_t checks = 0;

    // ---- packers: random records, every word of the batch accounted for, exact-size buffers
    const uint32_t shapes[][5] = {{4, 9, 13, 16, 8}, {4, 3, 4, 1, 2}, {1, 1, 2, 1, 0}, {7, 5, 9, 3, 4}, {40, 6, 10, 64, 3}, {4, 16, 20, 32, 15}};
    for (const auto &sh : shapes)
        for (uint32_t hash = 0; hash < 2; hash++)
            for (uint32_t flags = 0; flags < 2; flags++) {
                const ss_stwo_cfg c = make_cfg(sh[0], sh[1], sh[2], sh[3], sh[4], hash, flags);
                const size_t W = ss_stwo_record_words(&c);
                if (!W) { fprintf(stderr, "config refused\n"); return 1; }
                for (size_t n : {(size_t)1, (size_t)5, (size_t)64, (size_t)131}) {
                    std::vector<std::vector<uint32_t>> recs(3, std::vector<uint32_t>(W));
                    for (auto &r : recs)
                        for (auto &w : r) w = rnd() | 1u;  // non-zero: zeros in the batch are padding only
                    std::vector<const uint32_t *> ptrs(n);
                    for (size_t i = 0; i < n; i++) ptrs[i] = recs[i % 3].data();
                    const size_t words = ss_stwo_batch_words(&c, n);
                    uint32_t *batch = new uint32_t[words];  // exact size: ASan sees any overrun
                    if (ss_stwo_pack(&c, n, ptrs.data(), batch) != SS_OK) { fprintf(stderr, "pack failed\n"); return 1; }
                    size_t nonzero = 0;
                    for (size_t i = 0; i < words; i++) nonzero += batch[i] != 0;
                    if (nonzero != n * W) { fprintf(stderr, "pack is not a permutation: %zu of %zu words\n", nonzero, n * W); return 1; }
                    delete[] batch;
                    checks++;
                }
            }
    for (uint32_t ml : {0u, 1u, 10u, 31u})
        for (uint32_t pm : {0u, 4u, 13u, 31u}) {
            ss_s101_shape sh = {ml, pm};
            const size_t W = ss_s101_record_words(&sh);
            std::vector<uint32_t> rec(W);
            for (auto &w : rec) w = rnd() | 1u;
            for (size_t n : {(size_t)1, (size_t)65}) {
                std::vector<const uint32_t *> ptrs(n, rec.data());
                const size_t words = ss_s101_batch_words(&sh, n);
                uint32_t *batch = new uint32_t[words];
                if (ss_s101_pack(&sh, n, ptrs.data(), batch) != SS_OK) { fprintf(stderr, "s101 pack failed\n"); return 1; }
                size_t nonzero = 0;
                for (size_t i = 0; i < words; i++) nonzero += batch[i] != 0;
                if (nonzero != n * W) { fprintf(stderr, "s101 pack is not a permutation\n"); return 1; }
                delete[] batch;
                checks++;
            }
        }

    // ---- writers, templates and the scalar rule, on the reference's proof and its mutants
    const ss_stwo_cfg prod = make_cfg(4, 9, 13, 16, 8, 0, 0);
    const size_t W = ss_stwo_record_words(&prod);
    size_t taken = 0;
    for (int f = 0; f < 2; f++) {
        const int fmt = f == 0 ? SS_TEXT_JSON : SS_TEXT_WIT;
        const std::string base = slurp(argv[3 + f]);
        ss::TextTemplateHost t;
        ss::stwo_build_template(prod, fmt, t);
        if (!t.ok) { fprintf(stderr, "no template\n"); return 1; }
        uint32_t *rec = new uint32_t[W];  // exact size
        if (!ss::text_scan_reference(t.view(), base.data(), base.size(), rec)) { fprintf(stderr, "reference file not canonical\n"); return 1; }
        std::string back;
        const bool ok = fmt == SS_TEXT_JSON ? ss::stwo_write_json(prod, rec, ss::kStyleCompact, back) : ss::stwo_write_wit(prod, rec, back);
        if (!ok || back.size() + 2 < base.size()) { fprintf(stderr, "writer does not reproduce the file\n"); return 1; }
        for (int i = 0; i < mutants; i++) {
            const std::string m = mutate(base);
            char *exact = new char[m.size() ? m.size() : 1];  // no terminator, no slack
            memcpy(exact, m.data(), m.size());
            taken += ss::text_scan_reference(t.view(), exact, m.size(), rec);
            delete[] exact;
            checks++;
        }
        delete[] rec;
    }
    // templates of awkward configs (zero inner layers, one query, many columns, no JSON for an odd pow_target)
    for (const auto &sh : shapes) {
        ss_stwo_cfg c = make_cfg(sh[0], sh[1], sh[2], sh[3], sh[4], 1, 0);
        ss::TextTemplateHost a, b;
        ss::stwo_build_template(c, SS_TEXT_JSON, a);
        ss::stwo_build_template(c, SS_TEXT_WIT, b);
        c.pow_target = 12345;
        ss::TextTemplateHost d;
        ss::stwo_build_template(c, SS_TEXT_JSON, d);
        if (!a.ok || !b.ok || d.ok) { fprintf(stderr, "template of a supported config missing (or made for an odd pow_target)\n"); return 1; }
        checks++;
    }
    // ---- shared records (csrc/ss_sharedrec.cpp): share / unshare through exact-size buffers, random shapes and
    // positions (uniform, clustered, equal), structure mutants into the host expansion; the shared text writer, its
    // template and the scalar rule of the GPU reader for that format
    size_t shared_taken = 0;
    for (const auto &sh : shapes) {
        const ss_stwo_cfg c = make_cfg(sh[0], sh[1], sh[2], sh[3], sh[4], 0, 0);
        const size_t Wc = ss_stwo_record_words(&c), cap = ss_stwo_shared_max_words(&c), fixed = ss_stwo_shared_fixed_words(&c);
        const uint32_t N = c.n_cols, L = c.lde_log, Q = c.n_queries, K = c.n_layers;
        for (int rep = 0; rep < 6; rep++) {
            std::vector<uint32_t> qs(Q);
            const uint32_t base_pos = rnd() & ((1u << L) - 1);
            for (auto &q : qs) q = rep % 3 == 0 ? (rnd() & ((1u << L) - 1)) : rep % 3 == 1 ? (base_pos ^ (rnd() & 7 & ((1u << L) - 1))) : base_pos;
            // a record whose paths agree wherever these positions make them meet: node bytes are a function of (tree, level, position)
            uint32_t *rec = new uint32_t[Wc];
            for (size_t i = 0; i < Wc; i++) rec[i] = rnd() | 1u;
            const uint32_t head = 24 + 4 * N + 64 + 8 * (K + 1) + 6, qstride = N + 16 + 16 * L;
            uint32_t fo = head + Q * qstride;
            for (uint32_t t = 0; t < K + 3; t++) {
                const uint32_t len = t < 2 ? L : L + 1 - t, shift = t < 2 ? 0 : t - 1;
                for (uint32_t q = 0; q < Q; q++)
                    for (uint32_t lvl = 0; lvl < len; lvl++) {
                        uint32_t *dst = t < 2 ? rec + head + q * qstride + N + 16 + t * 8 * L + 8 * lvl : rec + fo + q * (4 + 8 * len) + 4 + 8 * lvl;
                        const uint32_t pos = ((qs[q] >> shift) >> lvl) ^ 1;
                        for (uint32_t w = 0; w < 8; w++) dst[w] = (pos * 2654435761u) ^ (t * 40503u + lvl * 97u + w) ^ 0x9e3779b9u;
                    }
                if (t >= 2) fo += Q * (4 + 8 * len);
            }
            for (uint32_t t = 0; t < K + 3; t++)
                for (uint32_t q = 0; q < Q; q++) rec[fo + t * Q + q] = t < 2 ? L : L + 1 - t;
            size_t words = 0;
            std::vector<uint32_t> big(cap);
            if (ss_stwo_share_record(&c, rec, qs.data(), big.data(), cap, &words) != 0 || words < fixed || words > cap) { fprintf(stderr, "share failed\n"); return 1; }
            uint32_t *shr = new uint32_t[words];  // exact size
            memcpy(shr, big.data(), words * 4);
            uint32_t *back = new uint32_t[Wc];
            if (ss_stwo_unshare_record(&c, shr, words, back) != 0 || memcmp(back, rec, Wc * 4) != 0) { fprintf(stderr, "unshare is not the inverse of share\n"); return 1; }
            if (words > 1 && ss_stwo_share_record(&c, rec, qs.data(), big.data(), words - 1, &words) >= 0) { fprintf(stderr, "short capacity accepted\n"); return 1; }
            for (int m = 0; m < 40; m++) {  // structure mutants: any size, hints, counts
                size_t mw = words;
                std::vector<uint32_t> mut(shr, shr + words);
                switch (rnd() % 5) {
                case 0: mw = rnd() % (words + 1); break;
                case 1: mut[fixed - (K + 3) - Q + rnd() % Q] = rnd(); break;
                case 2: mut[fixed - (K + 3) + rnd() % (K + 3)] += 1 + rnd() % 3; break;
                case 3: mut[rnd() % words] ^= 1u << (rnd() % 32); break;
                default: mut.resize(words + 1 + rnd() % 9, 7); mw = mut.size(); break;
                }
                uint32_t *exact = new uint32_t[mw ? mw : 1];
                memcpy(exact, mut.data(), mw * 4);
                (void)ss_stwo_unshare_record(&c, exact, mw, back);
                delete[] exact;
                checks++;
            }
            // the text of this shared record, its template, the scalar rule
            std::string text;
            ss::TextTemplateHost t3;
            ss::stwo_build_template(c, SS_TEXT_JSON_SHARED, t3);
            if (!t3.ok || !ss::stwo_write_json_shared(c, shr, words, rep & 1 ? ss::kStylePython : ss::kStyleCompact, text)) { fprintf(stderr, "no shared text\n"); return 1; }
            if (!ss::shared_text_scan_reference(c, t3, text.data(), text.size(), back) || memcmp(back, rec, Wc * 4) != 0) { fprintf(stderr, "shared text does not read back\n"); return 1; }
            for (int i = 0; i < mutants / 20; i++) {
                const std::string mt = mutate(text);
                char *exact = new char[mt.size() ? mt.size() : 1];
                memcpy(exact, mt.data(), mt.size());
                shared_taken += ss::shared_text_scan_reference(c, t3, exact, mt.size(), back);
                delete[] exact;
                checks++;
            }
            delete[] back;
            delete[] shr;
            delete[] rec;
        }
    }

    // ---- minimal records (csrc/ss_minimalrec.cpp, ss_text.cpp, ss_ingest.cpp): minimise through exact-size buffers on random
    // shapes and positions (uniform, clustered, equal), the minimal text writer, the host reader on the text and on mutants
    // of it, the writer on structure mutants of the record
    size_t minimal_read = 0, minimal_taken = 0;
    for (const auto &sh : shapes) {
        const ss_stwo_cfg c = make_cfg(sh[0], sh[1], sh[2], sh[3], sh[4], 0, 0);
        const size_t Wc = ss_stwo_record_words(&c), cap = ss_stwo_minimal_max_words(&c), fixed = ss_stwo_minimal_fixed_words(&c);
        const uint32_t N = c.n_cols, L = c.lde_log, Q = c.n_queries, K = c.n_layers;
        for (int rep = 0; rep < 6; rep++) {
            std::vector<uint32_t> qs(Q);
            const uint32_t base_pos = rnd() & ((1u << L) - 1);
            for (auto &q : qs) q = rep % 3 == 0 ? (rnd() & ((1u << L) - 1)) : rep % 3 == 1 ? (base_pos ^ (rnd() & 7 & ((1u << L) - 1))) : base_pos;
            // a record whose queries agree wherever they present the same thing: every word a function of (what, level, position)
            uint32_t *rec = new uint32_t[Wc];
            for (size_t i = 0; i < Wc; i++) rec[i] = rnd() | 1u;
            const uint32_t head = 24 + 4 * N + 64 + 8 * (K + 1) + 6, qstride = N + 16 + 16 * L;
            for (uint32_t q = 0; q < Q; q++)
                for (uint32_t k = 0; k < N + 16; k++) rec[head + q * qstride + k] = (qs[q] * 2246822519u) ^ (k * 3266489917u);
            uint32_t fo = head + Q * qstride;
            for (uint32_t t = 0; t < K + 3; t++) {
                const uint32_t len = t < 2 ? L : L + 1 - t, shift = t < 2 ? 0 : t - 1;
                for (uint32_t q = 0; q < Q; q++) {
                    if (t >= 2)
                        for (uint32_t w = 0; w < 4; w++) rec[fo + q * (4 + 8 * len) + w] = (((qs[q] >> (t - 2)) ^ 1) * 668265263u) ^ (t * 31u + w);
                    for (uint32_t lvl = 0; lvl < len; lvl++) {
                        uint32_t *dst = t < 2 ? rec + head + q * qstride + N + 16 + t * 8 * L + 8 * lvl : rec + fo + q * (4 + 8 * len) + 4 + 8 * lvl;
                        const uint32_t pos = ((qs[q] >> shift) >> lvl) ^ 1;
                        for (uint32_t w = 0; w < 8; w++) dst[w] = (pos * 2654435761u) ^ (t * 40503u + lvl * 97u + w) ^ 0x9e3779b9u;
                    }
                }
                if (t >= 2) fo += Q * (4 + 8 * len);
            }
            for (uint32_t t = 0; t < K + 3; t++)
                for (uint32_t q = 0; q < Q; q++) rec[fo + t * Q + q] = t < 2 ? L : L + 1 - t;
            size_t words = 0;
            std::vector<uint32_t> big(cap);
            if (ss_stwo_minimise_record(&c, rec, qs.data(), big.data(), cap, &words) != 0 || words < fixed || words > cap) { fprintf(stderr, "minimise failed\n"); return 1; }
            uint32_t *mr = new uint32_t[words];  // exact size
            memcpy(mr, big.data(), words * 4);
            if (words > 1 && ss_stwo_minimise_record(&c, rec, qs.data(), big.data(), words - 1, &words) >= 0) { fprintf(stderr, "short capacity accepted\n"); return 1; }
            std::vector<uint32_t> counts(2 + (K + 1) + (K + 3));
            if (ss_stwo_minimal_counts(&c, qs.data(), counts.data()) != 0 || memcmp(counts.data(), mr + fixed - counts.size(), counts.size() * 4) != 0) { fprintf(stderr, "counts differ from the record's\n"); return 1; }
            std::string text;
            if (!ss::stwo_write_json_minimal(c, mr, words, rep & 1 ? ss::kStylePython : ss::kStyleCompact, text)) { fprintf(stderr, "no minimal text\n"); return 1; }
            std::vector<uint32_t> back;
            if (ss::stwo_parse_minimal_text(c, text.data(), text.size(), back) != ss::kParsed || back.size() != words || memcmp(back.data(), mr, words * 4) != 0) { fprintf(stderr, "minimal text does not read back\n"); return 1; }
            // the GPU reader's rule for this form, scalar (template of the full-length text, landmarks, gaps): the text reads
            // back through it, into the capacity form, and compacts to the record; the two forms into each other
            ss::TextTemplateHost t4;
            ss::stwo_build_template(c, SS_TEXT_JSON_MINIMAL, t4);
            uint32_t *capr = new uint32_t[cap];  // exact size
            if (!t4.ok || t4.record_words != cap || !ss::minimal_text_scan_reference(c, t4, text.data(), text.size(), capr)) { fprintf(stderr, "minimal text not taken by the scalar rule\n"); return 1; }
            ss::minimal_compact(c, capr, back);
            if (back.size() != words || memcmp(back.data(), mr, words * 4) != 0) { fprintf(stderr, "scalar rule reads another record\n"); return 1; }
            if (!ss::minimal_to_capacity(c, mr, words, capr)) { fprintf(stderr, "no capacity form\n"); return 1; }
            ss::minimal_compact(c, capr, back);
            if (back.size() != words || memcmp(back.data(), mr, words * 4) != 0) { fprintf(stderr, "capacity form does not compact back\n"); return 1; }
            if (words > fixed && ss::minimal_to_capacity(c, mr, words - 1, capr)) { fprintf(stderr, "short record spread\n"); return 1; }
            for (int i = 0; i < mutants / 20; i++) {
                const std::string mt = mutate(text);
                char *exact = new char[mt.size() ? mt.size() : 1];
                memcpy(exact, mt.data(), mt.size());
                const bool host = ss::stwo_parse_minimal_text(c, exact, mt.size(), back) == ss::kParsed;
                minimal_read += host;
                if (ss::minimal_text_scan_reference(c, t4, exact, mt.size(), capr)) {  // whatever the rule takes, the host reader reads the same
                    std::vector<uint32_t> got;
                    ss::minimal_compact(c, capr, got);
                    if (!host || got != back) { fprintf(stderr, "scalar rule and host reader differ on a mutant\n"); return 1; }
                    minimal_taken++;
                }
                delete[] exact;
                checks++;
            }
            delete[] capr;
            for (int m = 0; m < 40; m++) {  // structure mutants of the record into the writer: any size, any counts
                size_t mw = words;
                std::vector<uint32_t> mut(mr, mr + words);
                switch (rnd() % 4) {
                case 0: mw = rnd() % (words + 1); break;
                case 1: mut[fixed - counts.size() + rnd() % counts.size()] += 1 + rnd() % 3; break;
                case 2: mut[fixed - counts.size() + rnd() % counts.size()] = rnd(); break;
                default: mut.resize(words + 1 + rnd() % 9, 7); mw = mut.size(); break;
                }
                uint32_t *exact = new uint32_t[mw ? mw : 1];
                memcpy(exact, mut.data(), mw * 4);
                (void)ss::stwo_write_json_minimal(c, exact, mw, ss::kStyleCompact, text);
                delete[] exact;
                checks++;
            }
            delete[] mr;
            delete[] rec;
        }
    }

    // ---- the worker pool (csrc/ss_pool.cpp): an exception in an item fails the call on the CALLING thread and leaves the
    // pool usable; a fork()ed child (no worker threads of its own) runs items inline and exits without touching the pool
    {
        bool caught = false;
        try {
            ss::parallel_for(1000, [](size_t i) { if (i == 517) throw std::runtime_error("item 517"); }, 8);
        } catch (const std::runtime_error &e) {
            caught = strcmp(e.what(), "item 517") == 0;
        }
        std::vector<int> seen(2000, 0);
        ss::parallel_for(seen.size(), [&](size_t i) { seen[i] = 1; }, 8);
        size_t done = 0;
        for (int v : seen) done += v;
        if (!caught || done != seen.size()) { fprintf(stderr, "pool: exception not delivered or pool unusable afterwards\n"); return 1; }
        const pid_t child = fork();
        if (child == 0) {
            std::vector<int> c2(100, 0);
            ss::parallel_for(c2.size(), [&](size_t i) { c2[i] = 1; }, 8);
            size_t d2 = 0;
            for (int v : c2) d2 += v;
            // (_exit: LeakSanitizer would report the vanished worker threads' allocations as leaks of the child; the pool
            // itself has no destructor that could run at exit -- it is allocated once and never freed)
            _exit(d2 == c2.size() ? 0 : 3);
        }
        int st = 0;
        if (waitpid(child, &st, 0) != child || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "pool: forked child failed (%d)\n", st); return 1; }
        checks += 3;
    }
    printf("host_san: %zu checks, %zu mutants taken by the scalar rule, %zu shared-text mutants, %zu minimal-text mutants read (%zu taken by the scalar rule)\n", checks, taken, shared_taken, minimal_read, minimal_taken);
    return 0;
}
