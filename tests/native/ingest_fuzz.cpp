// Sanitizer fuzz driver of the native text readers (csrc/ss_ingest.cpp).  Built by
// tests/test_ingest.py with  g++ -fsanitize=address,undefined  and run on the reference's own files:
// every input is parsed as it is and after seeded mutations (byte flips, deletions, duplications,
// truncations, digit / bracket substitutions).  Any out-of-bounds access, overflow or leak aborts the
// process; the outcome itself (parsed / malformed / other config) is only counted.
//
//   ingest_fuzz <seed> <mutants per file> <production|testing|s101> file...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "ss_ingest.h"
#include "ss_layout.h"

static uint64_t rng_state;
static uint32_t rnd()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

static std::string mutate(const std::string &s)
{
    std::string m = s;
    const int n_edits = 1 + rnd() % 3;
    for (int e = 0; e < n_edits && !m.empty(); e++) {
        const size_t at = rnd() % m.size();
        switch (rnd() % 8) {
        case 0: m[at] = (char)(m[at] ^ (1 << (rnd() % 8))); break;
        case 1: m.erase(at, 1 + rnd() % 40); break;
        case 2: m.insert(at, m.substr(at, 1 + rnd() % 60)); break;
        case 3: m.resize(at); break;
        case 4: m[at] = "0123456789"[rnd() % 10]; break;
        case 5: m[at] = "[](){},:\" x_"[rnd() % 12]; break;
        case 6: m.insert(at, std::string(1 + rnd() % 90, (char)('0' + rnd() % 10))); break;
        default: { static const char *ins[3] = {"list![", "0x", "qm31("}; m.insert(at, ins[rnd() % 3]); break; }
        }
    }
    return m;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: ingest_fuzz seed mutants production|testing|s101 file...\n"); return 2; }
    rng_state = strtoull(argv[1], nullptr, 10);
    const int mutants = atoi(argv[2]);
    const bool s101 = strcmp(argv[3], "s101") == 0;
    ss_stwo_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    if (strcmp(argv[3], "testing") == 0) { cfg.n_cols = 4; cfg.trace_log = 3; cfg.lde_log = 4; cfg.n_queries = 1; cfg.n_layers = 2; }
    else { cfg.n_cols = 4; cfg.trace_log = 9; cfg.lde_log = 13; cfg.n_queries = 16; cfg.n_layers = 8; }
    cfg.mode = 1;
    cfg.pow_target = 0x07ffffffffffffffull;
    std::vector<uint32_t> rec(ss::stwo_record_words(cfg.n_cols, cfg.lde_log, cfg.n_queries, cfg.n_layers) + 16, 0xdeadbeefu);
    const size_t W = rec.size() - 16;
    long outcomes[3] = {0, 0, 0}, total = 0;
    for (int f = 4; f < argc; f++) {
        std::ifstream in(argv[f], std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string base = ss.str();
        if (base.empty()) { fprintf(stderr, "%s: empty or unreadable\n", argv[f]); return 2; }
        for (int k = 0; k <= mutants; k++) {
            const std::string text = k ? mutate(base) : base;
            for (int fmt = 0; fmt < 3; fmt++) {  // auto, forced json, forced wit
                if (s101) {
                    ss::S101Parsed *p = ss::s101_parse_text(text.data(), text.size(), fmt);
                    if (p) {
                        uint32_t nl, pm;
                        ss::s101_parsed_shape(p, &nl, &pm);
                        if (nl > 31 || pm > 31) { fprintf(stderr, "shape out of range\n"); return 1; }
                        ss_s101_shape sh = {nl, pm};
                        std::vector<uint32_t> r(ss::s101_record_words(nl, pm));
                        ss::s101_parsed_record(p, sh, r.data());
                        ss::s101_parsed_free(p);
                    }
                    outcomes[p ? 0 : 1]++;
                } else {
                    const int r = (int)ss::stwo_parse_text(cfg, text.data(), text.size(), fmt, rec.data());
                    if (r < 0 || r > 2) { fprintf(stderr, "bad outcome %d\n", r); return 1; }
                    for (size_t g = 0; g < 16; g++)
                        if (rec[W + g] != 0xdeadbeefu) { fprintf(stderr, "wrote past the record\n"); return 1; }
                    if (k == 0 && fmt == 0 && r != 0) { fprintf(stderr, "%s: the unmutated file does not parse (%d)\n", argv[f], r); return 1; }
                    outcomes[r]++;
                }
                total++;
            }
        }
    }
    printf("%ld parses: %ld parsed, %ld malformed, %ld other config\n", total, outcomes[0], outcomes[1], outcomes[2]);
    return 0;
}
