// Private definitions shared by the translation units behind include/ss_verify.h: the context, its
// grow-only scratch, error reporting and kernel timing.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "ss_abi.h"
#include "ss_pack.h"
#include "ss_pool.h"
#include "ss_text.h"

namespace ss {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ss::set_err(SS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// Makes the context's device current for the calling thread while an entry point runs and restores the caller's
// afterwards: a process that holds one context per GPU (the C / Rust caller of INTEGRATION.md) may call any entry
// point of any context from any thread, whatever device is current there.  Not a stream operation: capturable.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err;
    explicit DeviceGuard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define SS_DEVICE_GUARD(ctx)                                                                                   \
    if (!(ctx)) return ss::set_err(SS_ERR_ARG, "ctx is null");                                                 \
    ss::DeviceGuard device_guard_((ctx)->device);                                                              \
    if (device_guard_.err != hipSuccess)                                                                       \
        return ss::set_err(SS_ERR_HIP, "cannot make device %d current: %s", (ctx)->device, hipGetErrorString(device_guard_.err))

struct TimedSpan {
    const char *name;
    hipEvent_t start, stop;
};

// Grow-only scratch of the host-buffer entry points (ss_*_verify_records): pinned staging for
// the chunked upload and the device buffers, so repeated calls do not pay hipMalloc /
// hipHostMalloc again.
struct HostPath {
    void *pinned[2] = {nullptr, nullptr};
    size_t pinned_bytes = 0;
    hipEvent_t pinned_free[2] = {nullptr, nullptr};
    void *dev[7] = {};  // records, batch, ws, status; shared-record chunks (two in flight), their outcomes
    size_t dev_bytes[7] = {};
    hipEvent_t shared_free[2] = {nullptr, nullptr};  // the expansion that read shared chunk buffer b has run
    hipStream_t stream = nullptr;   // uploads
    hipStream_t vstream = nullptr;  // re-tiling and verification of the chunks already uploaded
};

// A config's text template resident on the device (ss_text.h), kept per (config, format).
struct DevTemplate {
    ss_stwo_cfg cfg;
    int fmt;
    bool ok;            // false: no fast path for this config / format
    SharedTextInfo sinfo;  // format SS_TEXT_JSON_SHARED
    MinTextInfo minfo;     // format SS_TEXT_JSON_MINIMAL
    void *skel = nullptr, *slots = nullptr, *trailer = nullptr;
    TextTemplate view;  // device pointers
};

// Scratch of the text entry points (ss_stwo_verify_texts / _files): double-buffered pinned staging and
// device buffers for the raw text, the records the GPU reader writes, its outcomes.
struct GrowBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool pinned = false;
};
constexpr int kTextBufs = 3;  // chunks in flight: one uploading, one being read / verified, one being staged
struct TextPath {
    GrowBuf text_pin[kTextBufs], text_dev[kTextBufs];  // texts + their offsets / lengths / formats behind them
    GrowBuf rec_dev[kTextBufs], out_dev[kTextBufs], out_pin[kTextBufs];
    GrowBuf win_dev[kTextBufs];        // per-window scratch of the GPU reader (ss_textdev.h)
    GrowBuf shrec_dev[kTextBufs], hint_dev[kTextBufs];  // shared-path texts: capacity-form shared records, per-text hints
    GrowBuf fix_pin[kTextBufs];        // records re-made by the host reader, on their way up
    GrowBuf fix_dev[kTextBufs];        // ... as they arrive: one block, scattered into rec_dev by a kernel
    GrowBuf batch_dev, ws_dev, status_dev;
    hipStream_t up = nullptr, cx = nullptr, vx = nullptr;  // upload, GPU reader, re-tile + verify
    hipEvent_t uploaded[kTextBufs] = {};  // H2D of buffer b complete
    hipEvent_t parsed[kTextBufs] = {};    // GPU reader + outcome download of buffer b complete
    hipEvent_t fixed[kTextBufs] = {};     // fix-up uploads from fix_pin[b] complete
    hipEvent_t packed[kTextBufs] = {};    // rec_dev[b] re-tiled: the GPU reader may write it again
    std::vector<DevTemplate> templates;
};

}  // namespace ss

struct ss_ctx {
    int device;
    int timing;
    int cus;                   // compute units of the device
    int top_blocks_per_cu[2][2];  // resident top-kernel blocks per CU: [hash family][0 = with the byte compares, 1 = hash only]
    std::vector<ss::TimedSpan> spans;  // recorded since the last collect
    std::vector<hipEvent_t> pool;      // recycled events
    ss::HostPath hp;
    ss::TextPath tp;
    // Entry points that use the context's own scratch (hp, tp) or its timing list hold this lock for
    // their whole duration: one such call runs at a time per context (include/ss_verify.h, "threads").
    std::mutex mu;
    std::mutex span_mu;  // spans / pool (the device entry points record into them when timing is on)
};

namespace ss {

// Records one (start, stop) event pair around each kernel launch on the launch stream.
struct Timer {
    ss_ctx *c;
    hipStream_t s;
    hipEvent_t cur = nullptr;
    Timer(ss_ctx *c_, hipStream_t s_) : c(c_), s(s_) {}
    void begin();
    void end(const char *name);
};

int hp_reserve(ss_ctx *ctx, int slot, size_t bytes);  // grow-only device buffer `slot` of the host path (ss_api.hip)
int hp_pinned(ss_ctx *ctx, size_t bytes);             // its two pinned staging buffers, streams and events
size_t stage_threads();
int shared_expand_launch(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *shared_dev, const uint64_t *offs_dev,
                         uint64_t capacity_stride, uint32_t *records_dev, uint32_t *outcome_dev, hipStream_t s,
                         const uint8_t *only_fmt = nullptr, const uint32_t *hint_pos = nullptr, uint32_t hint_stride = 0);  // ss_shared.hip

// HEAD half in front of minimal records (ss_minimal.hip): head words, transcript, plan + gather into `batch`, query kernel
int stwo_minimal_head(ss_ctx *ctx, const ss_stwo_cfg *c, const StwoLayout &y, const uint32_t *recs_dev, const uint64_t *offs_dev,
                      uint32_t *batch, uint32_t *ws, uint32_t *status, hipStream_t s);

// ss_stwo_verify_minimal_dev with offs_dev == nullptr allowed: capacity-form records at a stride of ss_stwo_minimal_max_words
int stwo_verify_minimal_any(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *min_dev, const uint64_t *offs_dev,
                            uint32_t *batch_dev, void *workspace, size_t workspace_bytes, uint32_t *status, uint32_t *accept_count,
                            int phases, hipStream_t stream);

int grow(GrowBuf &b, size_t bytes, bool pinned);  // (re)allocates when too small; contents are not kept
void release(GrowBuf &b);

double now_s();
bool read_file(const char *path, std::string &out);

// texts (or files) -> verdicts through the GPU reader (csrc/ss_ingest_dev.hip)
int stwo_ingest_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                    const char *const *paths, int fmt, uint32_t *status_host, ss_ingest_stats *stats,
                    const uint8_t *blob = nullptr, const uint64_t *blob_offs = nullptr);  // blob: the texts in one caller-pinned buffer
// the minimal proof.json (SS_TEXT_JSON_MINIMAL): read into capacity-form minimal records on the GPU, verified from there
int stwo_minimal_ingest_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                            const char *const *paths, uint32_t *status_host, ss_ingest_stats *stats, const uint8_t *blob = nullptr,
                            const uint64_t *blob_offs = nullptr);
int s101_ingest_dev(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, const char *const *paths, int fmt,
                    uint32_t *status_host, ss_ingest_stats *stats, const uint8_t *blob = nullptr, const uint64_t *blob_offs = nullptr);
// ss_s101_verify_records for a caller that already holds ctx->mu
int s101_verify_records_locked(ss_ctx *ctx, const ss_s101_shape *sh, size_t n, const uint32_t *const *records,
                               uint32_t *status_host);
int stwo_read_texts_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts, const size_t *lens,
                        int fmt, uint32_t *records_host, uint32_t *outcome_host);
void text_path_destroy(TextPath &tp);

}  // namespace ss
