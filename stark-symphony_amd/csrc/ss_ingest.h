// Native readers of the reference's proof / witness text formats (host code, no HIP).
//
// The reference's callers hand the verifier text: `proof.json` (stwo: the schema read by
// stwo-verifier/scripts/generate_wit.py:106-245; stark101: stark101/scripts/fibsquare/prover.py:
// 108,143-167) or `proof.wit` (generate_wit.py:218-243, stark101/scripts/generate_wit.py:13-30;
// consumed by `simfony run --witness`, simfony-cli/src/main.rs:163-209).  These functions turn that
// text straight into the records of include/ss_verify.h, so the drop-in path is bound by PCIe, not
// by a Python parser.  stark-symphony_amd/formats.py is the same grammar in Python (used by the
// conversion tools and as the cross-check of this file in tests/test_ingest.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "ss_abi.h"

namespace ss {

// outcome of parsing one text (the verdict-side meaning is in include/ss_verify.h)
enum ParseResult : int {
    kParsed = 0,
    kMalformed = 1,       // not a witness of the reference's types (`simfony run` exits 1 before running)
    kConfigMismatch = 2,  // well formed, but not the shape / declared parameters the verifier expects
    kDeclined = 3,        // kRouteStream only: the streaming reader does not take this text (no outcome: the general reader judges it)
};

// fmt: SS_TEXT_AUTO sniffs (a .wit is a JSON object with a "COMMITMENTS" / "P_MT_ROOT" member)
ParseResult stwo_parse_text(const ss_stwo_cfg &cfg, const char *text, size_t len, int fmt, uint32_t *record);

// The minimal proof.json (one sorted, deduplicated decommitment per tree; formats.stwo_minimal_from_json): the same
// schema with lists whose lengths are data.  out receives the minimal record (include/ss_verify.h) when parsed.
// Two readers: a streaming one for texts in the writers' member order with the expected config (one pass, no tree) and
// the general one (any member order, judges mismatch / malformed); route picks (tests compare them), kRouteAuto = the
// streaming reader first and the general one for whatever it declines.
enum : int { kRouteAuto = 0, kRouteGeneral = 1, kRouteStream = 2 };
ParseResult stwo_parse_minimal_text(const ss_stwo_cfg &cfg, const char *text, size_t len, std::vector<uint32_t> &out,
                                    int route = kRouteAuto);

// stark101: a proof's shape is data (List<_, 32>), so parsing yields the shape too
struct S101Parsed;
S101Parsed *s101_parse_text(const char *text, size_t len, int fmt);  // nullptr = malformed
void s101_parsed_shape(const S101Parsed *p, uint32_t *n_layers, uint32_t *max_path);
void s101_parsed_record(const S101Parsed *p, const ss_s101_shape &shape, uint32_t *record);
void s101_parsed_free(S101Parsed *p);

unsigned effective_cpus();  // scheduler affinity capped by the cgroup CPU quota

}  // namespace ss
