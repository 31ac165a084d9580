// ss_verify_inputs (include/ss_verify.h, ABI 2.4): ONE host-side entry point for "verify these N witnesses" -- the call a
// cgo / JNI / Rust-FFI binding binds in place of the reference's `simfony run main.simf --witness proof.wit` per proof
// (stark101/Makefile:8-9, stwo-verifier/Makefile:17-18, simfony-cli/src/main.rs:163-209,254-257).  A descriptor names the
// proof family, the form of one input and where the inputs lie; this file checks the combination and hands the call to the
// implementation of that (form, source) pair, which keeps its named export (include/ss_verify_forms.h).  Host-only.
#include <cstring>

#include "../../include/ss_verify.h"
#include "../../include/ss_verify_forms.h"
#include "ss_pack.h"

using namespace ss;

extern "C" int ss_verify_inputs(ss_ctx *ctx, const ss_input_desc *in, uint32_t *status_host, ss_ingest_stats *stats)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
    if (!in || !status_host) return set_err(SS_ERR_ARG, "null argument");
    if (!in->n) return set_err(SS_ERR_ARG, "n == 0 (an empty batch is the caller's no-op)");
    if (stats) memset(stats, 0, sizeof *stats);  // (the record forms have no text to account for)
    const size_t n = in->n;
    const bool host = in->source == SS_SRC_HOST, pinned = in->source == SS_SRC_PINNED, files = in->source == SS_SRC_FILES;
    if (!host && !pinned && !files) return set_err(SS_ERR_ARG, "unknown source %u", in->source);
    if ((host || files) && !in->items) return set_err(SS_ERR_ARG, "items is null");
    if (pinned && !in->blob) return set_err(SS_ERR_ARG, "blob is null");
    if (files && in->form != SS_FORM_TEXT) return set_err(SS_ERR_ARG, "files hold text: form must be SS_FORM_TEXT");
    const uint32_t *const *recs = reinterpret_cast<const uint32_t *const *>(in->items);
    const char *const *texts = reinterpret_cast<const char *const *>(in->items);
    if (in->family == SS_FAMILY_STWO) {
        const ss_stwo_cfg *c = in->cfg;
        if (!c) return set_err(SS_ERR_ARG, "stwo inputs need the config the caller expects");
        switch (in->form) {
        case SS_FORM_RECORDS:
            return host ? ss_stwo_verify_records(ctx, c, n, recs, status_host)
                        : ss_stwo_verify_records_pinned(ctx, c, n, static_cast<const uint32_t *>(in->blob), status_host);
        case SS_FORM_SHARED_RECORDS:
            if (host && !in->lens) return set_err(SS_ERR_ARG, "lens (words per record) is null");
            if (pinned && !in->offs) return set_err(SS_ERR_ARG, "offs (n + 1 word offsets) is null");
            return host ? ss_stwo_verify_shared_records(ctx, c, n, recs, in->lens, status_host)
                        : ss_stwo_verify_shared_records_pinned(ctx, c, n, static_cast<const uint32_t *>(in->blob), in->offs, status_host);
        case SS_FORM_MINIMAL_RECORDS:
            if (host && !in->lens) return set_err(SS_ERR_ARG, "lens (words per record) is null");
            if (pinned && !in->offs) return set_err(SS_ERR_ARG, "offs (n + 1 word offsets) is null");
            return host ? ss_stwo_verify_minimal_records(ctx, c, n, recs, in->lens, status_host)
                        : ss_stwo_verify_minimal_records_pinned(ctx, c, n, static_cast<const uint32_t *>(in->blob), in->offs, status_host);
        case SS_FORM_TEXT:
            if (files) return ss_stwo_verify_files(ctx, c, n, texts, (int)in->text_fmt, status_host, stats);
            if (!in->lens) return set_err(SS_ERR_ARG, "lens (bytes per text) is null");
            if (pinned && !in->offs) return set_err(SS_ERR_ARG, "offs (n + 1 byte offsets) is null");
            if (in->text_fmt == SS_TEXT_JSON_MINIMAL)
                return host ? ss_stwo_verify_minimal_texts(ctx, c, n, texts, in->lens, status_host, stats)
                            : ss_stwo_verify_minimal_texts_pinned(ctx, c, n, in->blob, in->offs, in->lens, status_host, stats);
            return host ? ss_stwo_verify_texts(ctx, c, n, texts, in->lens, (int)in->text_fmt, status_host, stats)
                        : ss_stwo_verify_texts_pinned(ctx, c, n, static_cast<const char *>(in->blob), in->offs, in->lens,
                                                      (int)in->text_fmt, status_host, stats);
        }
        return set_err(SS_ERR_ARG, "unknown form %u", in->form);
    }
    if (in->family == SS_FAMILY_STARK101) {
        switch (in->form) {
        case SS_FORM_RECORDS:
            if (!host) return set_err(SS_ERR_ARG, "stark101 records are taken from host memory (SS_SRC_HOST)");
            if (!in->shape) return set_err(SS_ERR_ARG, "stark101 records need their shape");
            return ss_s101_verify_records(ctx, in->shape, n, recs, status_host);
        case SS_FORM_TEXT:
            if (files) return ss_s101_verify_files(ctx, n, texts, (int)in->text_fmt, status_host, stats);
            if (!in->lens) return set_err(SS_ERR_ARG, "lens (bytes per text) is null");
            if (pinned && !in->offs) return set_err(SS_ERR_ARG, "offs (n + 1 byte offsets) is null");
            return host ? ss_s101_verify_texts(ctx, n, texts, in->lens, (int)in->text_fmt, status_host, stats)
                        : ss_s101_verify_texts_pinned(ctx, n, static_cast<const char *>(in->blob), in->offs, in->lens,
                                                      (int)in->text_fmt, status_host, stats);
        }
        return set_err(SS_ERR_ARG, "stark101 has per-query records and texts only (form %u)", in->form);
    }
    return set_err(SS_ERR_ARG, "unknown family %u", in->family);
}
