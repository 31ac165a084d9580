// Texts / files -> verdicts (ss_stwo_verify_texts / _files of include/ss_verify.h), bound by the host link.
//
// The reference's callers hand the verifier text (stwo-verifier/scripts/generate_wit.py:106-245,218-243;
// `simfony run --witness`, simfony-cli/src/main.rs:163-209).  Per chunk of ~64 MiB of text:
//   host threads   copy (or read) the raw bytes into pinned staging -- no parsing;
//   upload stream  one H2D copy of the chunk (one stream only: two concurrent H2D streams share the link
//                  badly on this platform, profiles/r03_pcie_probe.txt);
//   compute stream GPU reader (ss_textdev.hip) -> records + one outcome word per text, outcomes downloaded;
//   host           texts the GPU reader did not take (outcome 1: not byte-for-byte canonical) go through the
//                  tree parser of ss_ingest.cpp -- the arbiter for parsed / other config / malformed -- and
//                  their records are uploaded over the GPU's;
//   verify stream  re-tile (ss_stwo_pack_dev) + verify the chunk.
// A stager thread keeps up to kTextBufs chunks staged ahead, so the upload stream never waits for the host;
// the GPU reader of chunk c+1 runs beside the verification of chunk c.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "ss_copy.h"
#include "ss_ctx.h"
#include "ss_ingest.h"
#include "ss_layout.h"
#include "ss_textdev.h"

namespace ss {

int grow(GrowBuf &b, size_t bytes, bool pinned)
{
    if (b.p && b.bytes >= bytes) return SS_OK;
    release(b);
    bytes = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;  // 25 % headroom: chunks of one call differ a little
    if (pinned) HIP_TRY(hipHostMalloc(&b.p, bytes, hipHostMallocDefault));
    else HIP_TRY(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    b.pinned = pinned;
    return SS_OK;
}

void release(GrowBuf &b)
{
    if (b.p) { if (b.pinned) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
    b.p = nullptr;
    b.bytes = 0;
}

void text_path_destroy(TextPath &tp)
{
    for (int i = 0; i < kTextBufs; i++) {
        release(tp.text_pin[i]); release(tp.text_dev[i]); release(tp.rec_dev[i]); release(tp.out_dev[i]);
        release(tp.out_pin[i]); release(tp.fix_pin[i]); release(tp.win_dev[i]); release(tp.shrec_dev[i]); release(tp.hint_dev[i]);
        release(tp.fix_dev[i]);
        if (tp.uploaded[i]) (void)hipEventDestroy(tp.uploaded[i]);
        if (tp.parsed[i]) (void)hipEventDestroy(tp.parsed[i]);
        if (tp.fixed[i]) (void)hipEventDestroy(tp.fixed[i]);
        if (tp.packed[i]) (void)hipEventDestroy(tp.packed[i]);
    }
    release(tp.batch_dev); release(tp.ws_dev); release(tp.status_dev);
    for (auto &t : tp.templates) {
        if (t.skel) (void)hipFree(t.skel);
        if (t.slots) (void)hipFree(t.slots);
        if (t.trailer) (void)hipFree(t.trailer);
    }
    tp.templates.clear();
    if (tp.up) (void)hipStreamDestroy(tp.up);
    if (tp.cx) (void)hipStreamDestroy(tp.cx);
    if (tp.vx) (void)hipStreamDestroy(tp.vx);
    tp.up = tp.cx = tp.vx = nullptr;
}

namespace {

bool same_text_cfg(const ss_stwo_cfg &a, const ss_stwo_cfg &b)
{
    return a.n_cols == b.n_cols && a.trace_log == b.trace_log && a.lde_log == b.lde_log &&
           a.n_queries == b.n_queries && a.n_layers == b.n_layers && a.pow_target == b.pow_target && a.hash == b.hash;
}

// the device copy of (cfg, fmt)'s template, built on first use
int template_of(ss_ctx *ctx, const ss_stwo_cfg &cfg, int fmt, hipStream_t s, TextTemplate &view, SharedTextInfo *sinfo = nullptr,
                MinTextInfo *minfo = nullptr)
{
    TextPath &tp = ctx->tp;
    for (size_t i = 0; i < tp.templates.size(); i++) {
        if (tp.templates[i].fmt != fmt || !same_text_cfg(tp.templates[i].cfg, cfg)) continue;
        const DevTemplate t = tp.templates[i];  // most recently used last
        tp.templates.erase(tp.templates.begin() + i);
        tp.templates.push_back(t);
        view = t.ok ? t.view : TextTemplate();
        if (sinfo) *sinfo = t.sinfo;
        if (minfo) *minfo = t.minfo;
        return SS_OK;
    }
    if (tp.templates.size() >= 12) {  // a caller that cycles through many configs: drop the least recently used one
        // -- of ANOTHER config: the views of this config's other formats are in use by the call that asks
        size_t v = 0;
        while (v < tp.templates.size() && same_text_cfg(tp.templates[v].cfg, cfg)) v++;
        if (v < tp.templates.size()) {
            DevTemplate &o = tp.templates[v];
            HIP_TRY(hipStreamSynchronize(s));
            if (o.skel) (void)hipFree(o.skel);
            if (o.slots) (void)hipFree(o.slots);
            if (o.trailer) (void)hipFree(o.trailer);
            tp.templates.erase(tp.templates.begin() + v);
        }
    }
    TextTemplateHost h;
    if (cfg.n_cols == 0) s101_build_template(fmt, h);  // the key s101_ingest_dev uses
    else stwo_build_template(cfg, fmt, h);
    DevTemplate d{};
    d.cfg = cfg; d.fmt = fmt; d.ok = h.ok; d.sinfo = h.sinfo; d.minfo = h.minfo;
    if (sinfo) *sinfo = h.sinfo;
    if (minfo) *minfo = h.minfo;
    if (h.ok) {
        HIP_TRY(hipMalloc(&d.skel, h.skel.size()));
        HIP_TRY(hipMalloc(&d.slots, h.slots.size() * sizeof(TextSlot)));
        const size_t tail_words = h.trailer.size() + h.fixed.size();  // trailer values, then the (word, value) pairs
        HIP_TRY(hipMalloc(&d.trailer, std::max<size_t>(tail_words, 1) * 4));
        HIP_TRY(hipMemcpy(d.skel, h.skel.data(), h.skel.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.slots, h.slots.data(), h.slots.size() * sizeof(TextSlot), hipMemcpyHostToDevice));
        if (!h.trailer.empty())
            HIP_TRY(hipMemcpy(d.trailer, h.trailer.data(), h.trailer.size() * 4, hipMemcpyHostToDevice));
        if (!h.fixed.empty())
            HIP_TRY(hipMemcpy((uint32_t *)d.trailer + h.trailer.size(), h.fixed.data(), h.fixed.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipDeviceSynchronize());  // (once per template: its readers run on streams that do not wait for the null stream)
        d.view = h.view();
        d.view.skel = (const uint8_t *)d.skel;
        d.view.slots = (const TextSlot *)d.slots;
        d.view.trailer = (const uint32_t *)d.trailer;
        d.view.fixed = (const uint32_t *)d.trailer + h.trailer.size();
    }
    tp.templates.push_back(d);
    view = d.ok ? d.view : TextTemplate();
    return SS_OK;
}

int ensure_streams(TextPath &tp)
{
    if (!tp.up) HIP_TRY(hipStreamCreateWithFlags(&tp.up, hipStreamNonBlocking));
    if (!tp.cx) HIP_TRY(hipStreamCreateWithFlags(&tp.cx, hipStreamNonBlocking));
    if (!tp.vx) HIP_TRY(hipStreamCreateWithFlags(&tp.vx, hipStreamNonBlocking));
    for (int i = 0; i < kTextBufs; i++) {
        if (!tp.uploaded[i]) HIP_TRY(hipEventCreateWithFlags(&tp.uploaded[i], hipEventDisableTiming));
        // (the host waits on these two while the stager's threads copy: sleep, do not spin on a core)
        if (!tp.parsed[i]) HIP_TRY(hipEventCreateWithFlags(&tp.parsed[i], hipEventDisableTiming | hipEventBlockingSync));
        if (!tp.fixed[i]) HIP_TRY(hipEventCreateWithFlags(&tp.fixed[i], hipEventDisableTiming | hipEventBlockingSync));
        if (!tp.packed[i]) HIP_TRY(hipEventCreateWithFlags(&tp.packed[i], hipEventDisableTiming));
    }
    return SS_OK;
}

constexpr size_t kChunkTextBytes = (size_t)64 << 20;    // text per chunk (one H2D copy)
constexpr size_t kChunkRecordBytes = (size_t)256 << 20;  // records per chunk
constexpr size_t kMaxTextBytes = (size_t)32 << 20;       // ss_ingest.cpp refuses longer texts

struct Chunk {
    size_t lo = 0, cnt = 0;
    size_t text_bytes = 0;  // aligned text area
};

// A whole file into dst (cap bytes, 16-byte aligned pinned memory); returns its length, or -1 (absent, unreadable,
// longer than cap).  read(2) straight into the pinned buffer would fill it with ordinary stores, which is what the
// streaming copy of ss_copy.h avoids; so the file is read in pieces that stay in the core's cache and each piece
// goes to the staging buffer with streaming stores.  (mmap + streaming copy was tried: the page-table work and the
// TLB shootdowns of munmap on 8 threads cost more than they save for 0.4 MB files.)
long read_into(const char *path, uint8_t *dst, size_t cap)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -1;
    static thread_local std::vector<uint8_t> bounce(128 << 10);
    size_t got = 0;
    for (;;) {
        const ssize_t k = read(fd, bounce.data(), bounce.size());
        if (k < 0) { close(fd); return -1; }
        if (k == 0) break;
        if (got + (size_t)k > cap) { close(fd); return -1; }  // the file grew past its stat size
        copy_streaming(dst + got, bounce.data(), (size_t)k);  // (pieces are multiples of 16 bytes except the last)
        got += (size_t)k;
    }
    close(fd);
    return (long)got;
}

}  // namespace

// What differs between the proof families behind the one pipeline.
struct Family {
    size_t W = 0;                 // words of a record on the device
    TextTemplate tmpl[4];         // device views by kTextJson / kTextWit / kTextShared / kTextMinimal (the last two: stwo only)
    SharedTextInfo sinfo{};       // of tmpl[kTextShared]
    MinTextInfo minfo{};          // of tmpl[kTextMinimal]
    bool minimal = false;         // every text is a minimal proof.json (format 3): records are capacity-form minimal records
    const ss_stwo_cfg *shared_cfg = nullptr;  // not null: shared-path texts are read and expanded on the GPU
    const char *wit_key = "";     // the member name a .wit starts with (format sniffing for the GPU reader's first guess)
    bool zero_records = false;    // records have padding words the GPU reader does not write
    // host reader of text g into dst (W words): 0 = parsed, SS_STATUS_MALFORMED / SS_STATUS_CONFIG_MISMATCH = stage-0
    // verdict (dst zeroed), kDeferred = parsed, but it does not fit a W-word record: the caller deals with it afterwards
    std::function<int(size_t g, const char *text, size_t len, uint32_t *dst)> host_read;
    std::function<size_t(size_t cnt)> batch_words, ws_bytes;
    // re-tile cnt records and verify them, asynchronously on `s`
    std::function<int(size_t cnt, const uint32_t *rec_dev, uint32_t *batch_dev, void *ws, size_t wsb, uint32_t *status_dev,
                      hipStream_t s)> verify;
};
constexpr int kDeferred = 3;

// a caller-pinned text buffer: offsets that leave room for 16-byte reads, memory the DMA engine can read
static int blob_ok(size_t n, const size_t *lens, const uint8_t *blob, const uint64_t *blob_offs)
{
    hipPointerAttribute_t a;
    for (size_t i = 0; i < n; i++)
        if ((blob_offs[i] & 15) || blob_offs[i + 1] < blob_offs[i] + lens[i])
            return set_err(SS_ERR_ARG, "text %zu: offsets are multiples of 16, ascending, at least a text's length apart", i);
    if ((blob_offs[n] & 15)) return set_err(SS_ERR_ARG, "the end offset is a multiple of 16 too (room for 16-byte reads)");
    if (hipPointerGetAttributes(&a, blob) != hipSuccess || a.type != hipMemoryTypeHost ||
        (blob_offs[n] && (hipPointerGetAttributes(&a, blob + blob_offs[n] - 1) != hipSuccess || a.type != hipMemoryTypeHost))) {
        (void)hipGetLastError();
        return set_err(SS_ERR_ARG, "the buffer is not page-locked host memory (hipHostMalloc / ss_host_register)");
    }
    return SS_OK;
}

// blob / blob_offs (optional): the texts lie in ONE page-locked buffer of the caller's, text i at byte blob_offs[i] (a
// multiple of 16, ascending, blob_offs[n] = the end): nothing is staged -- the DMA engine reads the chunk's bytes where they
// are, the stager only writes the chunk's small tables, and the host reader (non-canonical texts) reads the caller's copy.
static int ingest_pipeline(ss_ctx *ctx, const Family &F, size_t n, const char *const *texts, const size_t *lens,
                           const char *const *paths, int fmt, uint32_t *status_host, std::vector<uint8_t> &outcome,
                           ss_ingest_stats *stats, double t0, const uint8_t *blob = nullptr, const uint64_t *blob_offs = nullptr)
{
    TextPath &tp = ctx->tp;
    int rc;
    const size_t W = F.W;
    TextParseArgs args{};
    args.tmpl[kTextJson] = F.tmpl[kTextJson];
    args.tmpl[kTextWit] = F.tmpl[kTextWit];
    args.tmpl[kTextShared] = F.tmpl[kTextShared];
    args.tmpl[kTextMinimal] = F.tmpl[kTextMinimal];
    args.sinfo = F.sinfo;
    args.minfo = F.minfo;
    const bool min_ok = F.minimal && F.tmpl[kTextMinimal].skel;
    args.record_words = (uint32_t)W;
    const bool shared_ok = F.shared_cfg && F.tmpl[kTextShared].skel;
    const size_t SW = shared_ok ? F.tmpl[kTextShared].record_words : 0;  // words of a capacity-form shared record
    const unsigned threads = effective_cpus();
    // staging is a copy: a few threads saturate it, and the thread that drives the GPU needs a core too
    unsigned stage_threads = std::max(1u, std::min(threads > 1 ? threads - 1 : 1u, 8u));
    if (const char *e = getenv("SS_STAGE_THREADS")) stage_threads = (unsigned)std::max(1, atoi(e));  // tuning knob

    // ---- sizes (files: stat) and the chunk plan
    std::vector<uint32_t> tlen(n, 0);
    std::vector<uint8_t> unreadable(n, 0);
    if (paths) {
        parallel_for(n, [&](size_t i) {
            struct stat st;
            if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode) || (uint64_t)st.st_size > kMaxTextBytes) unreadable[i] = 1;
            else tlen[i] = (uint32_t)st.st_size;
        }, threads);
    } else {
        for (size_t i = 0; i < n; i++) {
            if (!texts[i] || lens[i] > kMaxTextBytes) unreadable[i] = 1;  // longer than any witness: malformed by rule
            else tlen[i] = (uint32_t)lens[i];
        }
    }
    auto aligned = [](size_t v) { return (v + 15) & ~(size_t)15; };
    // bytes text i occupies in a chunk's text area: its length rounded up, or (caller-pinned) the distance to the next text
    auto room = [&](size_t i) { return blob ? (size_t)(blob_offs[i + 1] - blob_offs[i]) : aligned(tlen[i]); };
    // Chunks of kChunkTextBytes of text -- smaller at both ends of the call: nothing can overlap the staging and
    // upload of the first chunk, nor the reading and verification of the last one.
    std::vector<Chunk> chunks;
    {
        size_t total = 0;
        for (size_t i = 0; i < n; i++) total += room(i);
        size_t done = 0;
        Chunk cur;
        auto cap_now = [&]() {
            const size_t ramp = std::min(done, total - std::min(total, done)) / 2 + (kChunkTextBytes >> 3);
            return std::min(kChunkTextBytes, std::max(kChunkTextBytes >> 3, ramp));
        };
        size_t cap = cap_now();
        for (size_t i = 0; i < n; i++) {
            const size_t a = room(i);
            if (cur.cnt && (cur.text_bytes + a > cap || (cur.cnt + 1) * W * 4 > kChunkRecordBytes)) {
                chunks.push_back(cur);
                done += cur.text_bytes;
                cur = Chunk{i, 0, 0};
                cap = cap_now();
            }
            cur.cnt++;
            cur.text_bytes += a;
        }
        chunks.push_back(cur);
    }
    const size_t nchunks = chunks.size();
    size_t max_cnt = 0, max_text = 0;
    for (auto &ch : chunks) { max_cnt = std::max(max_cnt, ch.cnt); max_text = std::max(max_text, ch.text_bytes); }
    // staging layout of a chunk:
    //   [texts, 16-byte aligned each][slack][offs u64 x cnt][lens u32 x cnt][win_base u32 x (cnt + 1)][fmt u8 x cnt]
    auto meta_off = [&](const Chunk &ch) { return aligned(ch.text_bytes + kTextSlack); };
    auto stage_bytes = [&](const Chunk &ch) { return meta_off(ch) + ch.cnt * 17 + 4 + 16; };
    const size_t stage_cap = aligned(max_text + kTextSlack) + max_cnt * 17 + 4 + 16;
    const size_t max_windows = (max_text / 1024 + max_cnt + 4) & ~(size_t)3;  // ceil(len / 1024) per text; keeps WinSum 16-byte aligned
    size_t words = 0, wsb = 0;  // (the workspace is not monotone in the batch size: smaller batches get smaller groups)
    for (auto &ch : chunks) {
        words = std::max(words, F.batch_words(ch.cnt));
        wsb = std::max(wsb, F.ws_bytes(ch.cnt));
    }
    for (int b = 0; b < kTextBufs; b++) {
        if ((rc = grow(tp.text_pin[b], stage_cap, true))) return rc;
        if ((rc = grow(tp.text_dev[b], stage_cap, false))) return rc;
        if ((rc = grow(tp.rec_dev[b], max_cnt * W * 4, false))) return rc;
        if ((rc = grow(tp.out_dev[b], max_cnt * 4, false))) return rc;
        if ((rc = grow(tp.out_pin[b], max_cnt * 4, true))) return rc;
        if ((rc = grow(tp.win_dev[b], max_windows * (4 + sizeof(WinSum) + sizeof(WinIn)), false))) return rc;
        if (shared_ok && ((rc = grow(tp.shrec_dev[b], max_cnt * SW * 4, false)) || (rc = grow(tp.hint_dev[b], max_cnt * sizeof(TextHint), false))))
            return rc;
        if (min_ok && (rc = grow(tp.hint_dev[b], max_cnt * sizeof(MinHint), false))) return rc;
    }
    if ((rc = grow(tp.batch_dev, words * 4, false))) return rc;
    if ((rc = grow(tp.ws_dev, wsb, false))) return rc;
    if ((rc = grow(tp.status_dev, n * 4, false))) return rc;

    outcome.assign(n, 0);  // stage-0 verdict of the texts the host reader handled (0 = verified as parsed)
    std::vector<uint32_t> chunk_windows(nchunks, 0);
    double stage_s = 0, parse_s = 0;
    uint64_t text_total = 0, fallbacks = 0;
    uint32_t *status_dev = (uint32_t *)tp.status_dev.p;

    // ---- the stager: raw bytes into pinned memory, a few chunks ahead of the GPU.  No HIP calls on this thread.
    std::mutex m;
    std::condition_variable cv_staged, cv_freed;
    size_t staged = 0, freed = 0;  // chunks staged so far / chunks whose pinned buffer is free again (finished)
    bool abort = false;
    auto stage_chunk = [&](size_t k) {
        const Chunk &ch = chunks[k];
        uint8_t *stage = (uint8_t *)tp.text_pin[k % kTextBufs].p;
        uint64_t *offs = (uint64_t *)(stage + meta_off(ch));
        uint32_t *lens32 = (uint32_t *)(offs + ch.cnt);
        uint32_t *win_base = lens32 + ch.cnt;
        uint8_t *fmts = (uint8_t *)(win_base + ch.cnt + 1);
        {
            size_t o = 0;
            for (size_t i = 0; i < ch.cnt; i++) { offs[i] = o; o += room(ch.lo + i); }
        }
        parallel_for(ch.cnt, [&](size_t i) {
            const size_t g = ch.lo + i;
            const uint8_t *dst = blob ? blob + blob_offs[g] : stage + offs[i];
            uint32_t len = tlen[g];
            if (unreadable[g]) len = 0;
            else if (blob) {
                // (nothing to copy: the chunk's text area IS blob[blob_offs[lo] .. blob_offs[lo + cnt]))
            } else if (paths) {
                const long got = read_into(paths[g], stage + offs[i], len);
                if (got < 0) { unreadable[g] = 1; len = 0; }
                else { len = (uint32_t)got; tlen[g] = len; }  // (a file that shrank since stat)
            } else {
                // (plain memcpy here costs a third of the rate: 84.6k -> 65-71k proofs/s, profiles/r03_text_staging_ab.txt)
                copy_streaming(stage + offs[i], texts[g], len);
            }
            lens32[i] = len;
            // which template to try: a .wit is a JSON object whose first member is COMMITMENTS / P_MT_ROOT.  A wrong
            // guess only costs the fast path -- the host reader sniffs for itself.
            uint32_t f = fmt == SS_TEXT_WIT ? kTextWit : kTextJson;
            if (fmt == SS_TEXT_AUTO) {
                const size_t look = len < 64 ? len : 64;
                const size_t kl = strlen(F.wit_key);
                for (size_t p = 0; p + kl <= look && f == kTextJson; p++)
                    if (memcmp(dst + p, F.wit_key, kl) == 0) f = kTextWit;
            }
            // ... and a shared-path proof.json ends with its "queries" member (at most 64 short numbers: the last KiB)
            if (shared_ok && f == kTextJson && fmt != SS_TEXT_WIT) {
                if (fmt == SS_TEXT_JSON_SHARED) f = kTextShared;
                else {
                    const size_t look = len < 1024 ? len : 1024;
                    if (look >= 9 && memmem(dst + (len - look), look, "\"queries\"", 9)) f = kTextShared;
                }
            }
            fmts[i] = (uint8_t)(F.minimal ? kTextMinimal : f);
        }, stage_threads);
        uint32_t n_windows = 0;  // (after the reads: a file may have shrunk since its stat)
        for (size_t i = 0; i < ch.cnt; i++) {
            win_base[i] = n_windows;
            n_windows += (lens32[i] + 1023) >> 10;
        }
        win_base[ch.cnt] = n_windows;
        chunk_windows[k] = n_windows;
    };
    std::thread stager([&]() {
        for (size_t k = 0; k < nchunks; k++) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_freed.wait(lk, [&] { return abort || k < freed + kTextBufs; });
                if (abort) return;
            }
            const double ts0 = now_s();
            stage_chunk(k);
            const double dt = now_s() - ts0;
            { std::lock_guard<std::mutex> lk(m); staged = k + 1; stage_s += dt; }
            cv_staged.notify_all();
        }
    });
    auto stop_stager = [&]() {
        { std::lock_guard<std::mutex> lk(m); abort = true; }
        cv_freed.notify_all();
        stager.join();
    };

    // What is left once the GPU reader of chunk k is through: texts it did not take go through the host reader and
    // their records over the GPU's; then the chunk is re-tiled and verified on the verify stream.
    auto finish = [&](size_t k) -> int {
        const Chunk &ch = chunks[k];
        const int b = (int)(k % kTextBufs);
        HIP_TRY(hipEventSynchronize(tp.parsed[b]));
        const uint32_t *oc = (const uint32_t *)tp.out_pin[b].p;
        std::vector<uint32_t> todo;
        for (size_t i = 0; i < ch.cnt; i++)
            if (oc[i] != 0) todo.push_back((uint32_t)i);
        HIP_TRY(hipStreamWaitEvent(tp.vx, tp.parsed[b], 0));
        if (!todo.empty()) {
            const double tp0 = now_s();
            fallbacks += todo.size();
            HIP_TRY(hipEventSynchronize(tp.fixed[b]));  // fix_pin[b]'s previous upload is through
            // one block: the re-made records back to back, then the slot each one belongs to
            const size_t fix_bytes = todo.size() * W * 4 + todo.size() * 4;
            if ((rc = grow(tp.fix_pin[b], fix_bytes, true))) return rc;
            if ((rc = grow(tp.fix_dev[b], fix_bytes, false))) return rc;
            uint32_t *fix = (uint32_t *)tp.fix_pin[b].p;
            const uint8_t *stage = (const uint8_t *)tp.text_pin[b].p;
            const uint64_t *offs = (const uint64_t *)(stage + meta_off(ch));
            if (blob) stage = blob + blob_offs[ch.lo];  // (the tables are staged, the texts are where the caller put them)
            try {  // (an exception in a worker -- the host reader allocates its parse tree -- is rethrown here by the pool)
                parallel_for(todo.size(), [&](size_t j) {
                    const size_t i = todo[j], g = ch.lo + i;
                    uint32_t *dst = fix + j * W;
                    int r = (int)SS_STATUS_MALFORMED;
                    if (!unreadable[g]) r = F.host_read(g, (const char *)stage + offs[i], tlen[g], dst);
                    if (r != 0) memset(dst, 0, W * 4);
                    outcome[g] = (uint8_t)r;
                }, threads);
            } catch (const std::exception &e) {
                return set_err(SS_ERR_NOMEM, "host reader: %s", e.what());
            }
            // ONE upload and a scatter kernel, however the host-read texts are spread over the chunk (a copy per text
            // made a batch of mostly non-canonical texts pay a launch per proof on the verify stream: ADVICE r3)
            memcpy(fix + todo.size() * W, todo.data(), todo.size() * 4);
            HIP_TRY(hipMemcpyAsync(tp.fix_dev[b].p, fix, fix_bytes, hipMemcpyHostToDevice, tp.vx));
            HIP_TRY(hipEventRecord(tp.fixed[b], tp.vx));
            const uint32_t *fd = (const uint32_t *)tp.fix_dev[b].p;
            launch_text_scatter(todo.size(), W, fd, fd + todo.size() * W, (uint32_t *)tp.rec_dev[b].p, tp.vx);
            HIP_TRY(hipGetLastError());
            parse_s += now_s() - tp0;
        }
        for (size_t i = 0; i < ch.cnt; i++) text_total += tlen[ch.lo + i];
        { std::lock_guard<std::mutex> lk(m); freed = k + 1; }  // the pinned texts of this chunk are no longer needed
        cv_freed.notify_all();
        rc = F.verify(ch.cnt, (const uint32_t *)tp.rec_dev[b].p, (uint32_t *)tp.batch_dev.p, tp.ws_dev.p, tp.ws_dev.bytes,
                      status_dev + ch.lo, tp.vx);
        // (the verify stream has re-tiled rec_dev[b] by the time anything recorded after this point completes)
        HIP_TRY(hipEventRecord(tp.packed[b], tp.vx));
        return rc;
    };

    auto run = [&]() -> int {
        for (size_t k = 0; k < nchunks; k++) {
            const Chunk &ch = chunks[k];
            const int b = (int)(k % kTextBufs);
            {
                std::unique_lock<std::mutex> lk(m);
                cv_staged.wait(lk, [&] { return staged > k; });
            }
            // ---- upload (the GPU reader of chunk k - kTextBufs has finished with the device buffer: its outcome
            // download completed before finish() freed the pinned buffer this chunk was staged into)
            if (blob) {  // the texts straight from the caller's page-locked buffer, the chunk's tables from the staging buffer
                if (ch.text_bytes)
                    HIP_TRY(hipMemcpyAsync(tp.text_dev[b].p, blob + blob_offs[ch.lo], ch.text_bytes, hipMemcpyHostToDevice, tp.up));
                HIP_TRY(hipMemcpyAsync((uint8_t *)tp.text_dev[b].p + meta_off(ch), (const uint8_t *)tp.text_pin[b].p + meta_off(ch),
                                       stage_bytes(ch) - meta_off(ch), hipMemcpyHostToDevice, tp.up));
            } else {
                HIP_TRY(hipMemcpyAsync(tp.text_dev[b].p, tp.text_pin[b].p, stage_bytes(ch), hipMemcpyHostToDevice, tp.up));
            }
            HIP_TRY(hipEventRecord(tp.uploaded[b], tp.up));
            // ---- GPU reader (after rec_dev[b] has been re-tiled for chunk k - kTextBufs)
            HIP_TRY(hipStreamWaitEvent(tp.cx, tp.uploaded[b], 0));
            HIP_TRY(hipStreamWaitEvent(tp.cx, tp.packed[b], 0));
            if (F.zero_records) HIP_TRY(hipMemsetAsync(tp.rec_dev[b].p, 0, ch.cnt * W * 4, tp.cx));
            const uint8_t *dev = (const uint8_t *)tp.text_dev[b].p;
            args.texts = dev;
            args.offs = (const uint64_t *)(dev + meta_off(ch));
            args.lens = (const uint32_t *)(args.offs + ch.cnt);
            args.win_base = args.lens + ch.cnt;
            args.fmt = (const uint8_t *)(args.win_base + ch.cnt + 1);
            args.win_text = (uint32_t *)tp.win_dev[b].p;
            args.win_sum = (WinSum *)(args.win_text + max_windows);
            args.win_in = (WinIn *)(args.win_sum + max_windows);
            args.records = (uint32_t *)tp.rec_dev[b].p;
            args.outcome = (uint32_t *)tp.out_dev[b].p;
            args.hints = shared_ok ? (TextHint *)tp.hint_dev[b].p : nullptr;
            args.mhints = min_ok ? (MinHint *)tp.hint_dev[b].p : nullptr;
            args.shared_records = shared_ok ? (uint32_t *)tp.shrec_dev[b].p : nullptr;
            args.n = (uint32_t)ch.cnt;
            args.n_windows = chunk_windows[k];
            {
                Timer t(ctx, tp.cx);
                t.begin();
                launch_text_parse(args, tp.cx);
                t.end("stwo_text_parse");
            }
            HIP_TRY(hipGetLastError());
            // shared-path texts were read into capacity-form shared records: expand them into the chunk's records
            if (shared_ok && (rc = shared_expand_launch(ctx, F.shared_cfg, ch.cnt, args.shared_records, nullptr, SW, args.records,
                                                        args.outcome, tp.cx, args.fmt, args.hints->pos,
                                                        (uint32_t)(sizeof(TextHint) / 4))))
                return rc;
            HIP_TRY(hipMemcpyAsync(tp.out_pin[b].p, tp.out_dev[b].p, ch.cnt * 4, hipMemcpyDeviceToHost, tp.cx));
            HIP_TRY(hipEventRecord(tp.parsed[b], tp.cx));
            if (k > 0 && (rc = finish(k - 1))) return rc;
        }
        if ((rc = finish(nchunks - 1))) return rc;
        HIP_TRY(hipMemcpyAsync(status_host, status_dev, n * 4, hipMemcpyDeviceToHost, tp.vx));
        HIP_TRY(hipStreamSynchronize(tp.vx));
        return SS_OK;
    };
    rc = run();
    stop_stager();
    // whatever happened, nothing of this call may still be in flight when the scratch is used again
    (void)hipStreamSynchronize(tp.up);
    (void)hipStreamSynchronize(tp.cx);
    (void)hipStreamSynchronize(tp.vx);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++)
        if (outcome[i] == SS_STATUS_MALFORMED || outcome[i] == SS_STATUS_CONFIG_MISMATCH) status_host[i] = outcome[i];
    if (stats) {
        stats->read_s = stage_s;
        stats->parse_s = parse_s;
        stats->total_s = now_s() - t0;
        stats->text_bytes = text_total;
        stats->record_bytes = (uint64_t)n * W * 4;
        stats->threads = threads;
        stats->host_parsed = (uint32_t)std::min<uint64_t>(fallbacks, 0xffffffffu);
    }
    return SS_OK;
}

int stwo_ingest_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                    const char *const *paths, int fmt, uint32_t *status_host, ss_ingest_stats *stats, const uint8_t *blob,
                    const uint64_t *blob_offs)
{
    if (fmt == SS_TEXT_JSON_MINIMAL)  // (nothing in such a text names its form: the caller does)
        return stwo_minimal_ingest_dev(ctx, c, n, texts, lens, paths, status_host, stats, blob, blob_offs);
    if (!ctx || !status_host || (!texts && !paths && !blob) || ((texts || blob) && !lens) || (blob && !blob_offs))
        return set_err(SS_ERR_ARG, "null argument");
    if (blob) {
        const int bad = blob_ok(n, lens, blob, blob_offs);
        if (bad) return bad;
    }
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (fmt < SS_TEXT_AUTO || fmt > SS_TEXT_JSON_SHARED) return set_err(SS_ERR_ARG, "unknown text format");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    std::lock_guard<std::mutex> lock(ctx->mu);  // the context's scratch: one such call at a time
    const double t0 = now_s();
    SS_DEVICE_GUARD(ctx);
    int rc;
    if ((rc = ensure_streams(ctx->tp))) return rc;
    Family F;
    F.W = ss_stwo_record_words(c);
    if ((rc = template_of(ctx, *c, SS_TEXT_JSON, ctx->tp.cx, F.tmpl[kTextJson]))) return rc;
    if ((rc = template_of(ctx, *c, SS_TEXT_WIT, ctx->tp.cx, F.tmpl[kTextWit]))) return rc;
    if ((rc = template_of(ctx, *c, SS_TEXT_JSON_SHARED, ctx->tp.cx, F.tmpl[kTextShared], &F.sinfo))) return rc;
    F.wit_key = "\"COMMITMENTS\"";
    const ss_stwo_cfg cv = *c;
    F.shared_cfg = &cv;
    const int host_fmt = fmt == SS_TEXT_JSON_SHARED ? SS_TEXT_JSON : fmt;  // (the host reader knows the shared form by its member)
    F.host_read = [&cv, host_fmt](size_t, const char *text, size_t len, uint32_t *dst) {
        const ParseResult r = stwo_parse_text(cv, text, len, host_fmt, dst);
        return r == kParsed ? 0 : r == kConfigMismatch ? (int)SS_STATUS_CONFIG_MISMATCH : (int)SS_STATUS_MALFORMED;
    };
    F.batch_words = [&cv](size_t cnt) { return ss_stwo_batch_words(&cv, cnt); };
    F.ws_bytes = [&cv](size_t cnt) { return ss_stwo_workspace_bytes(&cv, cnt); };
    F.verify = [ctx, &cv](size_t cnt, const uint32_t *rec, uint32_t *batch, void *ws, size_t wsb, uint32_t *status, hipStream_t s) {
        const int r = ss_stwo_pack_dev(ctx, &cv, cnt, rec, batch, s);
        return r ? r : ss_stwo_verify_batch_dev(ctx, &cv, cnt, batch, ws, wsb, status, nullptr, s);
    };
    std::vector<uint8_t> outcome;
    std::vector<const char *> ptrs;
    if (blob) {  // (the pipeline's size / readability rules look at texts[i] and lens[i])
        ptrs.resize(n);
        for (size_t i = 0; i < n; i++) ptrs[i] = (const char *)blob + blob_offs[i];
        texts = ptrs.data();
    }
    return ingest_pipeline(ctx, F, n, texts, lens, paths, fmt, status_host, outcome, stats, t0, blob, blob_offs);
}

// The minimal proof.json.  Same pipeline, another record: the GPU reader fills minimal records in capacity form (the list
// lengths found in the text, csrc/ss_text.h), the host readers' records are spread into that form, and the verification is
// ss_minimal.hip's on records at a fixed stride -- no per-query records anywhere.
int stwo_minimal_ingest_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                            const char *const *paths, uint32_t *status_host, ss_ingest_stats *stats, const uint8_t *blob,
                            const uint64_t *blob_offs)
{
    if (!ctx || !status_host || (!texts && !paths && !blob) || ((texts || blob) && !lens) || (blob && !blob_offs))
        return set_err(SS_ERR_ARG, "null argument");
    if (blob) {
        const int bad = blob_ok(n, lens, blob, blob_offs);
        if (bad) return bad;
    }
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n * (size_t)kMaxQueries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    std::lock_guard<std::mutex> lock(ctx->mu);
    const double t0 = now_s();
    SS_DEVICE_GUARD(ctx);
    int rc;
    if ((rc = ensure_streams(ctx->tp))) return rc;
    Family F;
    F.minimal = true;
    F.W = ss_stwo_minimal_max_words(c);
    {
        SharedTextInfo unused;
        if ((rc = template_of(ctx, *c, SS_TEXT_JSON_MINIMAL, ctx->tp.cx, F.tmpl[kTextMinimal], &unused, &F.minfo))) return rc;
    }
    const ss_stwo_cfg cv = *c;
    const size_t W = F.W;
    F.host_read = [&cv, W](size_t, const char *text, size_t len, uint32_t *dst) {
        static thread_local std::vector<uint32_t> rec;
        const ParseResult r = stwo_parse_minimal_text(cv, text, len, rec);
        if (r != kParsed) return r == kConfigMismatch ? (int)SS_STATUS_CONFIG_MISMATCH : (int)SS_STATUS_MALFORMED;
        memset(dst, 0, W * 4);
        return minimal_to_capacity(cv, rec.data(), rec.size(), dst) ? 0 : (int)SS_STATUS_MALFORMED;
    };
    F.batch_words = [&cv](size_t cnt) { return ss_stwo_minimal_batch_words(&cv, cnt); };
    F.ws_bytes = [&cv](size_t cnt) { return ss_stwo_minimal_workspace_bytes(&cv, cnt); };
    F.verify = [ctx, &cv](size_t cnt, const uint32_t *rec, uint32_t *batch, void *ws, size_t wsb, uint32_t *status, hipStream_t s) {
        return stwo_verify_minimal_any(ctx, &cv, cnt, rec, nullptr, batch, ws, wsb, status, nullptr, SS_PHASE_ALL, s);
    };
    std::vector<uint8_t> outcome;
    std::vector<const char *> ptrs;
    if (blob) {
        ptrs.resize(n);
        for (size_t i = 0; i < n; i++) ptrs[i] = (const char *)blob + blob_offs[i];
        texts = ptrs.data();
    }
    return ingest_pipeline(ctx, F, n, texts, lens, paths, SS_TEXT_JSON, status_host, outcome, stats, t0, blob, blob_offs);
}

// stark101: the protocol's shape has a template (ss_text.h); a proof of another shape is parsed by the host reader
// and, when it does not fit the {10, 13} records of the pipeline, verified afterwards in a batch of its own shape.
int s101_ingest_dev(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, const char *const *paths, int fmt,
                    uint32_t *status_host, ss_ingest_stats *stats, const uint8_t *blob, const uint64_t *blob_offs)
{
    if (!ctx || !status_host || (!texts && !paths && !blob) || ((texts || blob) && !lens) || (blob && !blob_offs))
        return set_err(SS_ERR_ARG, "null argument");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (blob) {
        const int bad = blob_ok(n, lens, blob, blob_offs);
        if (bad) return bad;
    }
    if (fmt < SS_TEXT_AUTO || fmt > SS_TEXT_WIT) return set_err(SS_ERR_ARG, "unknown text format");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    std::lock_guard<std::mutex> lock(ctx->mu);
    const double t0 = now_s();
    SS_DEVICE_GUARD(ctx);
    int rc;
    if ((rc = ensure_streams(ctx->tp))) return rc;
    ss_stwo_cfg key{};  // the template cache is keyed by a config: stark101 uses an impossible one (0 columns)
    key.n_cols = 0; key.lde_log = kS101Path; key.n_layers = kS101Layers;
    Family F;
    const ss_s101_shape sh = {kS101Layers, kS101Path};
    F.W = ss_s101_record_words(&sh);
    if ((rc = template_of(ctx, key, SS_TEXT_JSON, ctx->tp.cx, F.tmpl[kTextJson]))) return rc;
    if ((rc = template_of(ctx, key, SS_TEXT_WIT, ctx->tp.cx, F.tmpl[kTextWit]))) return rc;
    F.wit_key = "\"P_MT_ROOT\"";
    F.zero_records = true;
    std::vector<S101Parsed *> deferred(n, nullptr);
    F.host_read = [&](size_t g, const char *text, size_t len, uint32_t *dst) {
        S101Parsed *p = s101_parse_text(text, len, fmt);
        if (!p) return (int)SS_STATUS_MALFORMED;
        uint32_t nl, pm;
        s101_parsed_shape(p, &nl, &pm);
        if (nl <= sh.max_layers && pm <= sh.max_path) {
            s101_parsed_record(p, sh, dst);
            s101_parsed_free(p);
            return 0;
        }
        deferred[g] = p;  // (one writer per g: the pipeline hands every text to exactly one worker)
        return kDeferred;
    };
    F.batch_words = [&sh](size_t cnt) { return ss_s101_batch_words(&sh, cnt); };
    F.ws_bytes = [&sh](size_t cnt) { return ss_s101_workspace_bytes(&sh, cnt); };
    F.verify = [ctx, &sh](size_t cnt, const uint32_t *rec, uint32_t *batch, void *ws, size_t wsb, uint32_t *status, hipStream_t s) {
        const int r = ss_s101_pack_dev(ctx, &sh, cnt, rec, batch, s);
        return r ? r : ss_s101_verify_batch_dev(ctx, &sh, cnt, batch, ws, wsb, status, nullptr, s);
    };
    std::vector<uint8_t> outcome;
    std::vector<const char *> ptrs0;
    if (blob) {  // (the pipeline's size / readability rules look at texts[i] and lens[i])
        ptrs0.resize(n);
        for (size_t i = 0; i < n; i++) ptrs0[i] = (const char *)blob + blob_offs[i];
        texts = ptrs0.data();
    }
    rc = ingest_pipeline(ctx, F, n, texts, lens, paths, fmt, status_host, outcome, stats, t0, blob, blob_offs);
    // ---- proofs of a larger shape than the protocol's: one more batch, of their own shape
    std::vector<size_t> late;
    ss_s101_shape big = {0, 0};
    for (size_t i = 0; i < n; i++)
        if (deferred[i]) {
            uint32_t nl, pm;
            s101_parsed_shape(deferred[i], &nl, &pm);
            big.max_layers = std::max(big.max_layers, nl);
            big.max_path = std::max(big.max_path, pm);
            late.push_back(i);
        }
    if (!rc && !late.empty()) {
        const size_t Wb = ss_s101_record_words(&big);
        std::vector<uint32_t> recs(late.size() * Wb, 0), st(late.size(), 0xffffffffu);
        std::vector<const uint32_t *> ptrs(late.size());
        for (size_t j = 0; j < late.size(); j++) {
            s101_parsed_record(deferred[late[j]], big, recs.data() + j * Wb);
            ptrs[j] = recs.data() + j * Wb;
        }
        rc = s101_verify_records_locked(ctx, &big, late.size(), ptrs.data(), st.data());
        for (size_t j = 0; j < late.size() && !rc; j++) status_host[late[j]] = st[j];
        if (stats) stats->total_s = now_s() - t0;
    }
    for (size_t i = 0; i < n; i++)
        if (deferred[i]) s101_parsed_free(deferred[i]);
    return rc;
}

// The GPU reader alone (diagnostic / tests): texts -> records + outcome words, synchronous, own buffers.
// (c == nullptr: stark101, records of shape {kS101Layers, kS101Path})
int stwo_read_texts_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                        int fmt, uint32_t *records_host, uint32_t *outcome_host)
{
    if (!ctx || !texts || !lens || !records_host || !outcome_host) return set_err(SS_ERR_ARG, "null argument");
    const bool sh_fmt = c && fmt == SS_TEXT_JSON_SHARED, min_fmt = c && fmt == SS_TEXT_JSON_MINIMAL;
    if ((c && !cfg_ok(c)) || !n || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT && !sh_fmt && !min_fmt)) return set_err(SS_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    SS_DEVICE_GUARD(ctx);
    TextPath &tp = ctx->tp;
    int rc;
    if ((rc = ensure_streams(tp))) return rc;
    ss_stwo_cfg key{};
    key.lde_log = kS101Path; key.n_layers = kS101Layers;  // n_cols 0: the stark101 template (s101_ingest_dev)
    const ss_s101_shape sh = {kS101Layers, kS101Path};
    const size_t W = min_fmt ? ss_stwo_minimal_max_words(c) : c ? ss_stwo_record_words(c) : ss_s101_record_words(&sh);
    TextParseArgs args{};
    if ((rc = template_of(ctx, c ? *c : key, SS_TEXT_JSON, tp.cx, args.tmpl[kTextJson]))) return rc;
    if ((rc = template_of(ctx, c ? *c : key, SS_TEXT_WIT, tp.cx, args.tmpl[kTextWit]))) return rc;
    if (sh_fmt && (rc = template_of(ctx, *c, SS_TEXT_JSON_SHARED, tp.cx, args.tmpl[kTextShared], &args.sinfo))) return rc;
    if (sh_fmt && !args.tmpl[kTextShared].skel) return set_err(SS_ERR_ARG, "no shared-path text exists for this config");
    if (min_fmt) {
        SharedTextInfo unused;
        if ((rc = template_of(ctx, *c, SS_TEXT_JSON_MINIMAL, tp.cx, args.tmpl[kTextMinimal], &unused, &args.minfo))) return rc;
        if (!args.tmpl[kTextMinimal].skel) return set_err(SS_ERR_ARG, "no minimal proof.json exists for this config");
    }
    args.record_words = (uint32_t)W;
    auto aligned = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (!texts[i] || lens[i] > kMaxTextBytes) return set_err(SS_ERR_ARG, "text %zu is null or longer than 32 MiB", i);
        total += aligned(lens[i]);
    }
    const size_t meta = aligned(total + kTextSlack), bytes = meta + n * 17 + 4 + 16;
    std::vector<uint8_t> host(bytes, 0);
    uint64_t *offs = (uint64_t *)(host.data() + meta);
    uint32_t *l32 = (uint32_t *)(offs + n);
    uint32_t *wb = l32 + n;
    uint8_t *f8 = (uint8_t *)(wb + n + 1);
    size_t o = 0;
    uint32_t n_windows = 0;
    for (size_t i = 0; i < n; i++) {
        offs[i] = o;
        memcpy(host.data() + o, texts[i], lens[i]);
        o += aligned(lens[i]);
        l32[i] = (uint32_t)lens[i];
        wb[i] = n_windows;
        n_windows += (uint32_t)((lens[i] + 1023) >> 10);
        f8[i] = (uint8_t)(min_fmt ? kTextMinimal : sh_fmt ? kTextShared : fmt == SS_TEXT_WIT ? kTextWit : kTextJson);
    }
    wb[n] = n_windows;
    GrowBuf text, rec, out, win, shrec, hint;
    auto done = [&](int code) { release(text); release(rec); release(out); release(win); release(shrec); release(hint); return code; };
    if ((rc = grow(text, bytes, false)) || (rc = grow(rec, n * W * 4, false)) || (rc = grow(out, n * 4, false)) ||
        (rc = grow(win, (size_t)(n_windows + 4) * (4 + sizeof(WinSum) + sizeof(WinIn)), false)))
        return done(rc);
    if (sh_fmt && ((rc = grow(shrec, n * (size_t)args.tmpl[kTextShared].record_words * 4, false)) || (rc = grow(hint, n * sizeof(TextHint), false))))
        return done(rc);
    if (min_fmt && (rc = grow(hint, n * sizeof(MinHint), false))) return done(rc);
    args.hints = sh_fmt ? (TextHint *)hint.p : nullptr;
    args.mhints = min_fmt ? (MinHint *)hint.p : nullptr;
    args.shared_records = (uint32_t *)shrec.p;
    const size_t wcap = ((size_t)n_windows + 4) & ~(size_t)3;
    // (the fill on the stream the reader runs on: hipMemset on the null stream is asynchronous to the host and tp.cx does not
    // wait for the null stream -- with more hardware queues than streams the fill overtook the reader's stores)
    if (hipMemcpy(text.p, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemsetAsync(rec.p, c ? 0xee : 0, n * W * 4, tp.cx) != hipSuccess)  // (stark101 records have zero padding the reader leaves alone)
        return done(set_err(SS_ERR_HIP, "upload failed"));
    const uint8_t *dev = (const uint8_t *)text.p;
    args.texts = dev;
    args.offs = (const uint64_t *)(dev + meta);
    args.lens = (const uint32_t *)(args.offs + n);
    args.win_base = args.lens + n;
    args.fmt = (const uint8_t *)(args.win_base + n + 1);
    args.win_text = (uint32_t *)win.p;
    args.win_sum = (WinSum *)(args.win_text + wcap);
    args.win_in = (WinIn *)(args.win_sum + wcap);
    args.records = (uint32_t *)rec.p;
    args.outcome = (uint32_t *)out.p;
    args.n = (uint32_t)n;
    args.n_windows = n_windows;
    launch_text_parse(args, tp.cx);
    if (sh_fmt && (rc = shared_expand_launch(ctx, c, n, args.shared_records, nullptr, args.tmpl[kTextShared].record_words, args.records,
                                             args.outcome, tp.cx, args.fmt, args.hints->pos, (uint32_t)(sizeof(TextHint) / 4))))
        return done(rc);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(tp.cx) != hipSuccess ||
        hipMemcpy(records_host, rec.p, n * W * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(outcome_host, out.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return done(set_err(SS_ERR_HIP, "GPU reader failed: %s", hipGetErrorString(hipGetLastError())));
    return done(SS_OK);
}

}  // namespace ss
