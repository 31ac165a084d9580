// C ABI of libss_verify.so (include/ss_verify.h): packers, context, launches, self-test.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <chrono>
#include <string>

#include "ss_abi.h"
#include "ss_copy.h"
#include "ss_ctx.h"
#include "ss_fields.h"
#include "ss_ingest.h"
#include "ss_kernels.h"
#include "ss_layout.h"
#include "ss_minimal.h"
#include "ss_sha256.h"
#include "ss_stwo_checks.h"

using namespace ss;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

int ss::set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int ss_internal_set_err(int code, const char *msg) { return set_err(code, "%s", msg); }

extern "C" int ss_version(void) { return SS_VERSION; }
extern "C" const char *ss_last_error(void) { return g_err; }

extern "C" int ss_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_err(SS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return SS_ERR_NO_DEVICE;
    }
    return n;
}

// ----------------------------------------------------------------------------- context
static constexpr size_t kMaxSpans = 1 << 16;

// Blocks of the persistent top kernel that are resident at once: from the kernel's own register and
// LDS footprint (512 VGPRs per SIMD lane in granules of 8, 160 KB LDS per CU; a block's
// kTopChains / 64 waves go to different SIMDs of the CU).  The grid must not exceed this, or the surplus blocks
// run after the others at a fraction of the occupancy.  Asked once, when the context is created.
static const void *top_kernel(int hf, int hash_only)
{
    return hash_only ? (hf ? (const void *)stwo_top_hash_kernel_b2s : (const void *)stwo_top_hash_kernel_sha)
                     : (hf ? (const void *)stwo_top_kernel_b2s : (const void *)stwo_top_kernel_sha);
}

static int top_blocks_per_cu(int hf, int hash_only)
{
    hipFuncAttributes a;
    const void *fn = top_kernel(hf, hash_only);
    int per_cu = 4;
    if (hipFuncGetAttributes(&a, fn) == hipSuccess && a.numRegs > 0) {
        const int by_regs = 512 / ((a.numRegs + 7) / 8 * 8);
        const int by_lds = a.sharedSizeBytes ? (int)((160u << 10) / a.sharedSizeBytes) : 8;
        per_cu = std::max(1, std::min(4, std::min(by_regs * 4 / (int)(kTopChains / 64), by_lds)));  // kTopMaxBlocks slices
    }
    return per_cu;
}

extern "C" int ss_ctx_create(int device, ss_ctx **out)
{
    if (!out) return set_err(SS_ERR_ARG, "out is null");
    int n = ss_device_count();
    if (n <= 0) return set_err(SS_ERR_NO_DEVICE, "no HIP device visible (%s)", g_err);
    if (device < 0 || device >= n) return set_err(SS_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    DeviceGuard guard(device);  // (the caller's current device is left as it was)
    if (guard.err != hipSuccess) return set_err(SS_ERR_HIP, "cannot make device %d current: %s", device, hipGetErrorString(guard.err));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(SS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    ss_ctx *c = new ss_ctx();
    c->device = device;
    c->timing = 0;
    c->cus = prop.multiProcessorCount;
    for (int hf = 0; hf < 2; hf++)
        for (int ho = 0; ho < 2; ho++) c->top_blocks_per_cu[hf][ho] = top_blocks_per_cu(hf, ho);
    *out = c;
    return SS_OK;
}

extern "C" void ss_ctx_destroy(ss_ctx *ctx)
{
    if (!ctx) return;
    for (auto &sp : ctx->spans) { (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop); }
    for (auto &e : ctx->pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (ctx->hp.pinned[i]) (void)hipHostFree(ctx->hp.pinned[i]);
        if (ctx->hp.pinned_free[i]) (void)hipEventDestroy(ctx->hp.pinned_free[i]);
        if (ctx->hp.shared_free[i]) (void)hipEventDestroy(ctx->hp.shared_free[i]);
    }
    for (auto &d : ctx->hp.dev)
        if (d) (void)hipFree(d);
    if (ctx->hp.stream) (void)hipStreamDestroy(ctx->hp.stream);
    if (ctx->hp.vstream) (void)hipStreamDestroy(ctx->hp.vstream);
    text_path_destroy(ctx->tp);
    delete ctx;
}

extern "C" int ss_ctx_set_timing(ss_ctx *ctx, int enabled)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
    ctx->timing = enabled;
    return SS_OK;
}

// nullptr when the runtime cannot create another event: the span is then skipped, never recorded
// with a null handle.
static hipEvent_t take_event(ss_ctx *c)
{
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

extern "C" int ss_ctx_collect_timing(ss_ctx *ctx, int cap, const char **names, float *total_ms,
                                     uint32_t *launches)
{
    if (!ctx || !names || !total_ms || !launches) return set_err(SS_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->span_mu);
    int k = 0;
    for (auto &sp : ctx->spans) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(sp.stop));
        HIP_TRY(hipEventElapsedTime(&ms, sp.start, sp.stop));
        int j = 0;
        while (j < k && names[j] != sp.name) j++;
        if (j == k) {
            if (k == cap) continue;
            names[k] = sp.name; total_ms[k] = 0.f; launches[k] = 0; k++;
        }
        total_ms[j] += ms;
        launches[j] += 1;
        ctx->pool.push_back(sp.start);
        ctx->pool.push_back(sp.stop);
    }
    ctx->spans.clear();
    return k;
}

void ss::Timer::begin()
{
    if (!c->timing) return;
    std::lock_guard<std::mutex> lk(c->span_mu);
    if (c->spans.size() >= kMaxSpans) return;
    cur = take_event(c);
    if (cur && hipEventRecord(cur, s) != hipSuccess) { c->pool.push_back(cur); cur = nullptr; }
}

void ss::Timer::end(const char *name)
{
    if (!cur) return;
    std::lock_guard<std::mutex> lk(c->span_mu);
    hipEvent_t stop = take_event(c);
    if (!stop || hipEventRecord(stop, s) != hipSuccess) {
        if (stop) c->pool.push_back(stop);
        c->pool.push_back(cur);
    } else {
        c->spans.push_back({name, cur, stop});
    }
    cur = nullptr;
}

// ================================================================================ stwo
static_assert(sizeof(ss_stwo_cfg) == 40, "ss_stwo_cfg is 9 words + tail padding to the u64's alignment");
static_assert(sizeof(ss_s101_shape) == 8, "ss_s101_shape is 2 words");
extern "C" size_t ss_abi_sizeof_cfg(void) { return sizeof(ss_stwo_cfg); }
extern "C" size_t ss_abi_sizeof_shape(void) { return sizeof(ss_s101_shape); }

extern "C" int ss_stwo_ws_layout_of(const ss_stwo_cfg *c, size_t n, ss_stwo_ws_layout *out)
{
    if (!cfg_ok(c) || !n || !out) return set_err(SS_ERR_ARG, "bad argument");
    const StwoLayout y = lay_of(c, n);
    out->np = y.np; out->nip = y.nip;
    out->ctx = y.ws_ctx; out->alpha = y.ws_alpha; out->leaf = y.ws_leaf; out->total_words = y.ws_total_words;
    out->c_queries = y.c_queries; out->c_p = y.c_p; out->c_p2 = y.c_p2; out->c_fold = y.c_fold;
    out->c_m1 = y.c_m1; out->n_pow = y.n_pow;
    out->top_levels = y.T; out->has_plan = y.mchk; out->plan = y.mchk ? y.ws_plan : 0;
    return SS_OK;
}

extern "C" int ss_stwo_read_intermediates(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const void *workspace,
                                          size_t proof, void *stream_, uint32_t *queries, uint32_t *oods_point,
                                          uint32_t *deep_alpha, uint32_t *fold_alphas, uint32_t *fri_answers)
{
    if (!ctx || !workspace) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c) || !n || proof >= n) return set_err(SS_ERR_ARG, "bad config or proof index");
    const StwoLayout y = lay_of(c, n);
    const uint32_t *ws = (const uint32_t *)workspace;
    SS_DEVICE_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    // ctx[w][proof]: `rows` consecutive w of one proof are a column of a (rows x np) matrix
    auto column = [&](uint32_t *dst, const uint32_t *src, size_t pitch_words, size_t rows) {
        return hipMemcpy2D(dst, 4, src, pitch_words * 4, 4, rows, hipMemcpyDeviceToHost);
    };
    const uint32_t *cx = ws + y.ws_ctx + proof;
    std::vector<uint32_t> qs(y.Q);
    HIP_TRY(column(qs.data(), cx + (size_t)y.c_queries * y.np, y.np, y.Q));
    if (queries) memcpy(queries, qs.data(), y.Q * 4);
    if (oods_point) HIP_TRY(column(oods_point, cx + (size_t)y.c_p * y.np, y.np, 8));
    if (fold_alphas) HIP_TRY(column(fold_alphas, cx + (size_t)y.c_fold * y.np, y.np, 4 * (y.K + 1)));
    if (deep_alpha)
        HIP_TRY(hipMemcpy(deep_alpha, ws + y.ws_alpha + proof * y.n_pow * 4, 16, hipMemcpyDeviceToHost));
    if (fri_answers)
        for (uint32_t q = 0; q < y.Q; q++) {  // the query's own member of the layer-0 leaf pair
            const size_t inst = proof * y.Q + q, half = (qs[q] & 1) ? 4 : 0;
            HIP_TRY(column(fri_answers + 4 * q, ws + y.ws_leaf + half * y.nip + inst, y.nip, 4));
        }
    return SS_OK;
}

// TAIL half of a pass: merkle kernel, top (+ cold) kernel, finalize.  `y` is the batch's layout (per-query or minimal).
static int stwo_tail(ss_ctx *ctx, const ss_stwo_cfg *c, const StwoLayout &y, const uint32_t *batch, uint32_t *ws,
                     uint32_t *status, uint32_t *accept_count, hipStream_t s)
{
    Timer t(ctx, s);
    const uint32_t tiles = (y.K + 3) * (y.nip >> 6);
    // the top kernel's group counter, its count of flagged trees and -- when the merkle kernel makes the byte
    // compares (y.mchk) -- the flags it raises, which lie directly behind
    if (y.T) HIP_TRY(hipMemsetAsync(ws + y.ws_counter, 0, y.mchk ? (y.ws_plan - y.ws_counter) * 4 : 8, s));
    t.begin();
    const int hf = c->hash == SS_HASH_BLAKE2S;
    if (y.minimal)
        hipLaunchKernelGGL(hf ? stwo_merkle_min_kernel_b2s : stwo_merkle_min_kernel_sha, dim3((tiles + 3) / 4), dim3(256), 0, s, y,
                           batch, ws, status);
    else
        hipLaunchKernelGGL(hf ? stwo_merkle_kernel_b2s : stwo_merkle_kernel_sha, dim3((tiles + 3) / 4), dim3(256), 0, s, y, batch,
                           ws, status);
    t.end("stwo_merkle");
    if (y.T) {
        t.begin();
        const int ho = y.mchk != 0;
        const uint32_t blocks = std::min<uint32_t>(y.top_blocks, (uint32_t)(ctx->top_blocks_per_cu[hf][ho] * std::min(ctx->cus, 256)));
        void *args[] = {(void *)&y, (void *)&batch, (void *)&ws, (void *)&status};
        // (behind minimal records the hash-only variant that takes computed siblings from their leaders: same footprint)
        const void *fn = y.minimal ? (hf ? (const void *)stwo_top_min_kernel_b2s : (const void *)stwo_top_min_kernel_sha) : top_kernel(hf, ho);
        HIP_TRY(hipLaunchKernel(fn, dim3(blocks), dim3(kTopChains), args, 0, s));
        t.end("stwo_top");
        // trees in which queries disagree about a node (none in an honest batch: the grid reads one word and leaves)
        t.begin();
        hipLaunchKernelGGL(hf ? stwo_top_cold_kernel_b2s : stwo_top_cold_kernel_sha, dim3(4 * ctx->cus), dim3(256), 0, s,
                           y, batch, (const uint32_t *)ws, status);
        t.end("stwo_top_cold");
    }
    t.begin();
    hipLaunchKernelGGL(stwo_finalize_kernel, dim3((y.n + 255) / 256), dim3(256), 0, s, y.n, status,
                       accept_count);
    t.end("stwo_finalize");
    return SS_OK;
}

extern "C" int ss_stwo_verify_phase_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                        const uint32_t *batch, void *workspace, size_t workspace_bytes,
                                        uint32_t *status, uint32_t *accept_count, int phases,
                                        void *stream_)
{
    SS_DEVICE_GUARD(ctx);  // a caller with one context per GPU: launch on THIS context's device whatever is current
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !batch || !workspace || !status) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    if (!(phases & SS_PHASE_ALL)) return set_err(SS_ERR_ARG, "no phase selected");
    const StwoLayout y = lay_of(c, n);
    if (workspace_bytes < y.ws_total_words * 4)
        return set_err(SS_ERR_WORKSPACE, "workspace %zu < %llu bytes", workspace_bytes,
                       (unsigned long long)y.ws_total_words * 4);
    hipStream_t s = (hipStream_t)stream_;
    uint32_t *ws = (uint32_t *)workspace;
    Timer t(ctx, s);
    if (phases & SS_PHASE_HEAD) {
        // (the transcript kernel resets the pass's status words and accept count itself: no memset dispatches)
        t.begin();
        hipLaunchKernelGGL(c->hash == SS_HASH_BLAKE2S ? stwo_transcript_kernel_b2s : stwo_transcript_kernel_sha,
                           dim3((y.n + 63) / 64), dim3(64), 0, s, y, batch, ws, status, accept_count, 1u);
        t.end("stwo_transcript");
        t.begin();
        hipLaunchKernelGGL(stwo_query_kernel, dim3((y.ni + 63) / 64), dim3(64), 2 * (y.K + 3) * 64 * 4, s, y, batch,
                           ws, status);
        t.end("stwo_query");
    }
    if (phases & SS_PHASE_TAIL) {
        const int rc = stwo_tail(ctx, c, y, batch, ws, status, accept_count, s);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

// Minimal records (csrc/ss_minimal.hip): the HEAD half reads the records, fills `batch_dev` and the workspace; the TAIL
// half is the per-query path's, on the minimal layout.
extern "C" int ss_stwo_verify_minimal_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *min_dev,
                                          const uint64_t *offs_dev, uint32_t *batch_dev, void *workspace, size_t workspace_bytes,
                                          uint32_t *status, uint32_t *accept_count, int phases, void *stream_)
{
    if (!offs_dev) return set_err(SS_ERR_ARG, "null/empty argument");
    return stwo_verify_minimal_any(ctx, c, n, min_dev, offs_dev, batch_dev, workspace, workspace_bytes, status, accept_count, phases,
                                   (hipStream_t)stream_);
}

// offs_dev == nullptr: capacity-form records at a fixed stride (the GPU reader's output, csrc/ss_minimal.hip MinArgs)
int ss::stwo_verify_minimal_any(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *min_dev, const uint64_t *offs_dev,
                                uint32_t *batch_dev, void *workspace, size_t workspace_bytes, uint32_t *status,
                                uint32_t *accept_count, int phases, hipStream_t stream_)
{
    SS_DEVICE_GUARD(ctx);
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !min_dev || !batch_dev || !workspace || !status) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n * (size_t)kMaxQueries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    if (!(phases & SS_PHASE_ALL)) return set_err(SS_ERR_ARG, "no phase selected");
    const StwoLayout y = lay_of(c, n, true);
    if (workspace_bytes < y.ws_total_words * 4)
        return set_err(SS_ERR_WORKSPACE, "workspace %zu < %llu bytes", workspace_bytes,
                       (unsigned long long)y.ws_total_words * 4);
    hipStream_t s = (hipStream_t)stream_;
    uint32_t *ws = (uint32_t *)workspace;
    if (phases & SS_PHASE_HEAD) {
        HIP_TRY(hipMemsetAsync(status, 0xff, n * 4, s));
        if (accept_count) HIP_TRY(hipMemsetAsync(accept_count, 0, 4, s));
        const int rc = stwo_minimal_head(ctx, c, y, min_dev, offs_dev, batch_dev, ws, status, s);
        if (rc) return rc;
    }
    if (phases & SS_PHASE_TAIL) {
        const int rc = stwo_tail(ctx, c, y, batch_dev, ws, status, accept_count, s);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_stwo_verify_batch_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                        const uint32_t *batch, void *workspace, size_t workspace_bytes,
                                        uint32_t *status, uint32_t *accept_count, void *stream_)
{
    return ss_stwo_verify_phase_dev(ctx, c, n, batch, workspace, workspace_bytes, status, accept_count,
                                    SS_PHASE_ALL, stream_);
}

// ============================================================================ stark101
extern "C" int ss_s101_verify_phase_dev(ss_ctx *ctx, const ss_s101_shape *sh, size_t n,
                                        const uint32_t *batch, void *workspace, size_t workspace_bytes,
                                        uint32_t *status, uint32_t *accept_count, int phases,
                                        void *stream_)
{
    SS_DEVICE_GUARD(ctx);
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n || !batch || !workspace || !status) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    if (!(phases & SS_PHASE_ALL)) return set_err(SS_ERR_ARG, "no phase selected");
    const S101Layout y = s101_layout(sh->max_layers, sh->max_path, n);
    if (workspace_bytes < y.ws_total_words * 4) return set_err(SS_ERR_WORKSPACE, "workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    uint32_t *ws = (uint32_t *)workspace;
    Timer t(ctx, s);
    if (phases & SS_PHASE_HEAD) {
        // (the transcript kernel resets the pass's status words and accept count itself: no memset dispatches)
        t.begin();
        hipLaunchKernelGGL(s101_transcript_kernel, dim3((y.n + 63) / 64), dim3(64), 0, s, y, batch, ws, status, accept_count);
        t.end("s101_transcript");
    }
    if (phases & SS_PHASE_TAIL) {
        const uint32_t tiles = y.n_types * (y.np >> 6);
        t.begin();
        hipLaunchKernelGGL(s101_merkle_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, y, batch, ws, status);
        t.end("s101_merkle");
        t.begin();
        hipLaunchKernelGGL(stwo_finalize_kernel, dim3((y.n + 255) / 256), dim3(256), 0, s, y.n, status,
                           accept_count);
        t.end("s101_finalize");
    }
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

// Per-stage values of one stark101 proof after a verify call (the twin of ss_stwo_read_intermediates).
extern "C" int ss_s101_read_intermediates(ss_ctx *ctx, const ss_s101_shape *sh, size_t n, const void *workspace, size_t proof,
                                          void *stream_, uint32_t *alphas, uint32_t *idx, uint32_t *x, uint32_t *cp,
                                          uint32_t *folds, uint32_t *state)
{
    if (!ctx || !workspace) return set_err(SS_ERR_ARG, "null argument");
    if (!shape_ok(sh) || !n || proof >= n) return set_err(SS_ERR_ARG, "bad shape or proof index");
    const S101Layout y = s101_layout(sh->max_layers, sh->max_path, n);
    const uint32_t *ws = (const uint32_t *)workspace;
    SS_DEVICE_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    auto rows = [&](uint32_t *dst, uint32_t row0, size_t count) {  // int[row][np]: `count` consecutive rows of one proof
        return hipMemcpy2D(dst, 4, ws + y.ws_int + (size_t)row0 * y.np + proof, (size_t)y.np * 4, 4, count, hipMemcpyDeviceToHost);
    };
    if (idx) HIP_TRY(hipMemcpy(idx, ws + proof, 4, hipMemcpyDeviceToHost));
    if (alphas) HIP_TRY(rows(alphas, 0, 3));
    if (x) HIP_TRY(rows(x, 3, 1));
    if (cp) HIP_TRY(rows(cp, 4, 1));
    if (folds) HIP_TRY(rows(folds, 5, (size_t)y.ML + 1));
    if (state) HIP_TRY(rows(state, 5 + y.ML + 1, 8));
    return SS_OK;
}

extern "C" int ss_s101_verify_batch_dev(ss_ctx *ctx, const ss_s101_shape *sh, size_t n,
                                        const uint32_t *batch, void *workspace, size_t workspace_bytes,
                                        uint32_t *status, uint32_t *accept_count, void *stream_)
{
    return ss_s101_verify_phase_dev(ctx, sh, n, batch, workspace, workspace_bytes, status, accept_count,
                                    SS_PHASE_ALL, stream_);
}

// ============================================================ host-buffer convenience paths
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// ------------------------------------------------------------- device-side packing (stwo)
// records_dev holds n natural-order records back to back; every word of the batch is written
// (padding instances get zeros), destination-major so the stores are contiguous.
namespace ss {

struct StwoRecordMap {
    uint64_t W;        // record words
    uint32_t qstride;  // words per query in the decommitment section
    uint32_t fbase;    // first word of the FRI section
    uint32_t tbase;    // first word of the path-length trailer
    uint32_t foff[kMaxList + 1];  // FRI layer l starts at fbase + foff[l]
};

__device__ __forceinline__ uint32_t rec_word(const uint32_t *rec, const StwoRecordMap &m, uint32_t p,
                                             uint32_t off)
{
    return rec[(uint64_t)p * m.W + off];
}

// head[w][proof], trace_vals[k][inst], cp_vals[k][inst], fri_wit[l][w][inst]: one word per thread
__global__ void stwo_pack_words_kernel(StwoLayout y, StwoRecordMap m, const uint32_t *__restrict__ rec,
                                       uint32_t *__restrict__ out)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= y.off_trace_path) return;
    uint32_t v = 0;
    if (d < y.off_trace_vals) {
        const uint32_t w = (uint32_t)(d / y.np), p = (uint32_t)(d - (uint64_t)w * y.np);
        if (p < y.n) v = rec_word(rec, m, p, w);
    } else {
        uint64_t e;
        uint32_t row, off0;
        if (d < y.off_cp_vals) { e = d - y.off_trace_vals; row = (uint32_t)(e / y.nip); off0 = row; }
        else if (d < y.off_fri_wit) { e = d - y.off_cp_vals; row = (uint32_t)(e / y.nip); off0 = y.N + row; }
        else if (d < y.off_plen) { e = d - y.off_fri_wit; row = (uint32_t)(e / y.nip); off0 = 0; }
        else { e = d - y.off_plen; row = (uint32_t)(e / y.nip); off0 = 0; }
        const uint32_t inst = (uint32_t)(e - (uint64_t)row * y.nip);
        if (inst < y.ni) {
            const uint32_t p = inst / y.Q, q = inst - p * y.Q;
            if (d < y.off_fri_wit) {
                v = rec_word(rec, m, p, y.head_words + q * m.qstride + off0);
            } else if (d >= y.off_plen) {
                v = rec_word(rec, m, p, m.tbase + row * y.Q + q);
            } else {
                const uint32_t l = row >> 2, w = row & 3, len = y.L - 1 - l;
                v = rec_word(rec, m, p, m.fbase + m.foff[l] + q * (4 + 8 * len) + w);
            }
        }
    }
    out[d] = v;
}

// Merkle paths: one 16-byte unit per thread.  Tiles [g][level][half][lane] hold the lowest len - top levels of a
// type, the `top` section [proof][type][level][query][half] the top ones (ss_layout.h).
__global__ void stwo_pack_paths_kernel(StwoLayout y, StwoRecordMap m, const uint32_t *__restrict__ rec,
                                       uint32_t *__restrict__ out)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // uint4 index in the path area
    const uint64_t first = y.off_trace_path >> 2, total = (y.total_words >> 2) - first;
    if (u >= total) return;
    const uint64_t d = (first + u) << 2;  // word offset
    auto src_of = [&](uint32_t type, uint32_t &len, uint32_t &per_q) {  // record offset of level 0 word 0 for query 0
        if (type < 2) { len = y.L; per_q = m.qstride; return y.head_words + y.N + kCp + type * 8 * y.L; }
        const uint32_t l = type - 2;
        len = y.L - 1 - l;
        per_q = 4 + 8 * len;
        return m.fbase + m.foff[l] + 4;
    };
    uint4 v = make_uint4(0, 0, 0, 0);
    if (d >= y.off_top) {
        const uint64_t r = (d - y.off_top) >> 2, per_proof = y.top_words >> 2;
        const uint32_t p = per_proof ? (uint32_t)(r / per_proof) : y.n;
        if (p < y.n) {  // (behind the last proof: alignment padding)
            const uint32_t in_p = (uint32_t)(r - (uint64_t)p * per_proof) << 2;  // word inside the proof's part
            uint32_t type = 0;
            while (type + 1 < y.K + 3 && in_p >= y.top_off[type + 1]) type++;
            uint32_t len, per_q;
            const uint32_t src0 = src_of(type, len, per_q);
            const uint32_t top = y.T < len ? y.T : len;
            const uint32_t e = (in_p - y.top_off[type]) >> 2;  // (level * Q + query) * 2 + half
            const uint32_t half = e & 1, lq = e >> 1, lvl = lq / y.Q, q = lq - lvl * y.Q;
            const uint32_t *s = rec + (uint64_t)p * m.W + src0 + q * per_q + (len - top + lvl) * 8 + half * 4;
            v = make_uint4(s[0], s[1], s[2], s[3]);
        }
    } else {
        uint32_t type;
        uint64_t base;
        if (d < y.off_cp_path) { type = 0; base = y.off_trace_path; }
        else if (d < y.off_fri_path[0]) { type = 1; base = y.off_cp_path; }
        else {
            uint32_t l = 0;
            while (l < y.K && d >= y.off_fri_path[l + 1]) l++;
            type = 2 + l;
            base = y.off_fri_path[l];
        }
        uint32_t len, per_q;
        const uint32_t src0 = src_of(type, len, per_q);
        const uint32_t low = len - (y.T < len ? y.T : len);  // levels in the tile (> 0: the unit lies in this section)
        const uint64_t r = (d - base) >> 2;           // uint4 units inside the section
        const uint32_t lane = (uint32_t)(r & 63);
        const uint64_t r2 = r >> 6;
        const uint32_t half = (uint32_t)(r2 & 1);
        const uint64_t r3 = r2 >> 1;                   // g * low + level
        const uint32_t g = (uint32_t)(r3 / low), level = (uint32_t)(r3 - (uint64_t)g * low);
        const uint32_t inst = g * 64 + lane;
        if (inst < y.ni) {
            const uint32_t p = inst / y.Q, q = inst - p * y.Q;
            const uint32_t *s = rec + (uint64_t)p * m.W + src0 + q * per_q + level * 8 + half * 4;
            v = make_uint4(s[0], s[1], s[2], s[3]);
        }
    }
    reinterpret_cast<uint4 *>(out)[first + u] = v;
}

}  // namespace ss

static StwoRecordMap record_map(const StwoLayout &y)
{
    StwoRecordMap m{};
    m.W = stwo_record_words(y.N, y.L, y.Q, y.K);
    m.qstride = y.N + kCp + 16 * y.L;
    m.fbase = y.head_words + y.Q * m.qstride;
    uint32_t o = 0;
    for (uint32_t l = 0; l <= y.K; l++) {
        m.foff[l] = o;
        o += y.Q * (4 + 8 * (y.L - 1 - l));
    }
    m.tbase = m.fbase + o;
    return m;
}

extern "C" int ss_stwo_pack_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *records_dev,
                                uint32_t *batch_dev, void *stream_)
{
    SS_DEVICE_GUARD(ctx);
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !records_dev || !batch_dev) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const StwoLayout y = lay_of(c, n);
    const StwoRecordMap m = record_map(y);
    hipStream_t s = (hipStream_t)stream_;
    Timer t(ctx, s);
    t.begin();
    hipLaunchKernelGGL(stwo_pack_words_kernel, dim3((unsigned)((y.off_trace_path + 255) / 256)), dim3(256), 0,
                       s, y, m, records_dev, batch_dev);
    const uint64_t units = (y.total_words - y.off_trace_path) >> 2;
    hipLaunchKernelGGL(stwo_pack_paths_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, y, m,
                       records_dev, batch_dev);
    t.end("stwo_pack");
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

// ------------------------------------------------------------ device-side packing (stark101)
namespace ss {
// one word of the batch per thread (ss_layout.h: head[w][proof], leaf / len[type][proof], path tiles per type)
__global__ void s101_pack_kernel(S101Layout y, const uint32_t *__restrict__ rec, uint32_t *__restrict__ out)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= y.total_words) return;
    const uint32_t chain = 2 + 8 * y.PM;
    const uint64_t W = s101_record_words(y.ML, y.PM);
    auto chain_base = [&](uint32_t type) {  // record offset of chain `type`: ev, len, path
        if (type < 3) return 10 + type * chain;
        const uint32_t i = (type - 3) >> 1, which = (type - 3) & 1;
        return 10 + 3 * chain + i * (9 + 2 * chain) + 9 + which * chain;
    };
    uint32_t v = 0;
    if (d < y.off_leaf) {
        const uint32_t w = (uint32_t)(d / y.np), p = (uint32_t)(d - (uint64_t)w * y.np);
        if (p < y.n) {
            uint32_t src;
            if (w < y.h_layer) src = w;  // root[8], n_layers, last
            else { const uint32_t i = (w - y.h_layer) / 9, j = (w - y.h_layer) - 9 * i; src = 10 + 3 * chain + i * (9 + 2 * chain) + j; }
            v = rec[(uint64_t)p * W + src];
        }
    } else if (d < y.off_path) {
        const bool is_len = d >= y.off_len;
        const uint64_t e = d - (is_len ? y.off_len : y.off_leaf);
        const uint32_t type = (uint32_t)(e / y.np), p = (uint32_t)(e - (uint64_t)type * y.np);
        if (p < y.n) v = rec[(uint64_t)p * W + chain_base(type) + (is_len ? 1 : 0)];
    } else {
        const uint64_t e = d - y.off_path;
        const uint32_t type = (uint32_t)(e / y.path_stride);
        const uint64_t r = e - (uint64_t)type * y.path_stride;
        const uint32_t lane4 = (uint32_t)(r & 255), lane = lane4 >> 2, wlo = lane4 & 3;
        const uint64_t r2 = r >> 8;
        const uint32_t half = (uint32_t)(r2 & 1);
        const uint64_t r3 = r2 >> 1;  // g * PM + level
        const uint32_t g = (uint32_t)(r3 / y.PM), level = (uint32_t)(r3 - (uint64_t)g * y.PM);
        const uint32_t p = g * 64 + lane;
        if (p < y.n) v = rec[(uint64_t)p * W + chain_base(type) + 2 + level * 8 + half * 4 + wlo];
    }
    out[d] = v;
}
}  // namespace ss

extern "C" int ss_s101_pack_dev(ss_ctx *ctx, const ss_s101_shape *sh, size_t n, const uint32_t *records_dev,
                                uint32_t *batch_dev, void *stream_)
{
    SS_DEVICE_GUARD(ctx);
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n || !records_dev || !batch_dev) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const S101Layout y = s101_layout(sh->max_layers, sh->max_path, n);
    if (!y.total_words) return SS_OK;
    hipStream_t s = (hipStream_t)stream_;
    Timer t(ctx, s);
    t.begin();
    hipLaunchKernelGGL(s101_pack_kernel, dim3((unsigned)((y.total_words + 255) / 256)), dim3(256), 0, s, y, records_dev,
                       batch_dev);
    t.end("s101_pack");
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

// threads of a staging copy into pinned memory (a few saturate it; SS_STAGE_THREADS: tuning knob of the evidence scripts)
size_t ss::stage_threads()
{
    static const size_t n = [] {
        const char *e = getenv("SS_STAGE_THREADS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? (size_t)v : (size_t)8;
    }();
    return n;
}

int ss::hp_reserve(ss_ctx *ctx, int slot, size_t bytes)
{
    HostPath &hp = ctx->hp;
    if (hp.dev_bytes[slot] >= bytes) return SS_OK;
    if (hp.dev[slot]) { HIP_TRY(hipFree(hp.dev[slot])); hp.dev[slot] = nullptr; hp.dev_bytes[slot] = 0; }
    HIP_TRY(hipMalloc(&hp.dev[slot], bytes));
    hp.dev_bytes[slot] = bytes;
    return SS_OK;
}

int ss::hp_pinned(ss_ctx *ctx, size_t bytes)
{
    HostPath &hp = ctx->hp;
    if (!hp.stream) HIP_TRY(hipStreamCreateWithFlags(&hp.stream, hipStreamNonBlocking));
    if (!hp.vstream) HIP_TRY(hipStreamCreateWithFlags(&hp.vstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++)
        if (!hp.pinned_free[i]) HIP_TRY(hipEventCreateWithFlags(&hp.pinned_free[i], hipEventDisableTiming));
    if (hp.pinned_bytes >= bytes) return SS_OK;
    for (int i = 0; i < 2; i++) {
        if (hp.pinned[i]) { HIP_TRY(hipHostFree(hp.pinned[i])); hp.pinned[i] = nullptr; }
        HIP_TRY(hipHostMalloc(&hp.pinned[i], bytes, hipHostMallocDefault));
    }
    hp.pinned_bytes = bytes;
    return SS_OK;
}

// Host records -> verdicts: records are gathered into pinned staging in chunks (host threads),
// uploaded asynchronously (two buffers in flight); every chunk is re-tiled ON THE GPU (ss_stwo_pack_dev) and
// verified on a second stream as soon as it has arrived, so only the last chunk's verification is not hidden
// behind an upload.  PCIe-bound.
extern "C" int ss_stwo_verify_records(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                      const uint32_t *const *records, uint32_t *status_host)
{
    if (!ctx || !status_host || !records) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    std::lock_guard<std::mutex> lock(ctx->mu);  // the context's scratch: one such call at a time
    SS_DEVICE_GUARD(ctx);
    const size_t W = ss_stwo_record_words(c);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, (64u << 20) / (W * 4)));
    // (the first chunks are small and double: nothing overlaps the staging of the first one)
    std::vector<size_t> counts;
    for (size_t lo = 0, step = std::max<size_t>(1, chunk / 16); lo < n; step = std::min(chunk, step * 2)) {
        counts.push_back(std::min(step, n - lo));
        lo += counts.back();
    }
    size_t words = 0, wsb = 0;  // (neither is monotone in the batch size: smaller batches get smaller top-kernel groups)
    for (size_t cnt : counts) {
        words = std::max(words, ss_stwo_batch_words(c, cnt));
        wsb = std::max(wsb, ss_stwo_workspace_bytes(c, cnt));
    }
    int rc;
    if ((rc = hp_reserve(ctx, 0, n * W * 4))) return rc;
    if ((rc = hp_reserve(ctx, 1, words * 4))) return rc;
    if ((rc = hp_reserve(ctx, 2, wsb))) return rc;
    if ((rc = hp_reserve(ctx, 3, n * 4))) return rc;
    if ((rc = hp_pinned(ctx, chunk * W * 4))) return rc;
    HostPath &hp = ctx->hp;
    hipStream_t s = hp.stream, vs = hp.vstream;
    uint32_t *rec_dev = (uint32_t *)hp.dev[0], *status_dev = (uint32_t *)hp.dev[3];
    auto run = [&]() -> int {
        int buf = 0;
        size_t lo = 0;
        for (size_t cnt : counts) {
            HIP_TRY(hipEventSynchronize(hp.pinned_free[buf]));  // previous upload from this buffer done
            uint32_t *stage = (uint32_t *)hp.pinned[buf];
            parallel_for(cnt, [&](size_t i) { copy_streaming(stage + i * W, records[lo + i], W * 4); },
                         std::max<size_t>(1, std::min<size_t>(stage_threads(), cnt * W * 4 / (1u << 20))));
            HIP_TRY(hipMemcpyAsync(rec_dev + lo * W, stage, cnt * W * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(hp.pinned_free[buf], s));
            // the chunk that has just been queued: re-tile and verify it behind its upload (batch and workspace are
            // shared by the chunks, which the verify stream takes one after the other)
            HIP_TRY(hipStreamWaitEvent(vs, hp.pinned_free[buf], 0));
            int r = ss_stwo_pack_dev(ctx, c, cnt, rec_dev + lo * W, (uint32_t *)hp.dev[1], vs);
            if (r) return r;
            r = ss_stwo_verify_batch_dev(ctx, c, cnt, (const uint32_t *)hp.dev[1], hp.dev[2], wsb, status_dev + lo, nullptr, vs);
            if (r) return r;
            lo += cnt;
            buf ^= 1;
        }
        HIP_TRY(hipMemcpyAsync(status_host, status_dev, n * 4, hipMemcpyDeviceToHost, vs));
        return SS_OK;
    };
    rc = run();
    // whatever happened, nothing of this call may still be in flight when the scratch is used again
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(vs);
    if (rc) return rc;
    HIP_TRY(e1);
    HIP_TRY(e2);
    return SS_OK;
}

// Host records -> verdicts (stark101): host pack into pinned staging, one upload, verify, download.
// Scratch (pinned + device) is the context's grow-only HostPath, as for the stwo twin.
extern "C" int ss_s101_verify_records(ss_ctx *ctx, const ss_s101_shape *sh, size_t n,
                                      const uint32_t *const *records, uint32_t *status_host)
{
    if (!ctx) return set_err(SS_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return s101_verify_records_locked(ctx, sh, n, records, status_host);
}

int ss::s101_verify_records_locked(ss_ctx *ctx, const ss_s101_shape *sh, size_t n, const uint32_t *const *records,
                                   uint32_t *status_host)
{
    if (!ctx || !status_host || !records) return set_err(SS_ERR_ARG, "null argument");
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    SS_DEVICE_GUARD(ctx);
    const size_t words = ss_s101_batch_words(sh, n), wsb = ss_s101_workspace_bytes(sh, n);
    int rc;
    if ((rc = hp_reserve(ctx, 1, words * 4))) return rc;
    if ((rc = hp_reserve(ctx, 2, wsb))) return rc;
    if ((rc = hp_reserve(ctx, 3, n * 4))) return rc;
    if ((rc = hp_pinned(ctx, words * 4))) return rc;
    HostPath &hp = ctx->hp;
    hipStream_t s = hp.stream;
    HIP_TRY(hipEventSynchronize(hp.pinned_free[0]));
    if ((rc = ss_s101_pack(sh, n, records, (uint32_t *)hp.pinned[0]))) return rc;
    HIP_TRY(hipMemcpyAsync(hp.dev[1], hp.pinned[0], words * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(hp.pinned_free[0], s));
    rc = ss_s101_verify_batch_dev(ctx, sh, n, (const uint32_t *)hp.dev[1], hp.dev[2], wsb, (uint32_t *)hp.dev[3],
                                  nullptr, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(status_host, hp.dev[3], n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SS_OK;
}

// ======================================================================= text ingestion
extern "C" int ss_stwo_parse(const ss_stwo_cfg *c, const char *text, size_t len, int fmt, uint32_t *record_out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!text || !record_out || fmt < SS_TEXT_AUTO || fmt > SS_TEXT_WIT) return set_err(SS_ERR_ARG, "bad argument");
    ParseResult r;
    try {
        r = stwo_parse_text(*c, text, len, fmt, record_out);
    } catch (const std::exception &e) {
        return set_err(SS_ERR_NOMEM, "host reader: %s", e.what());
    }
    if (r != kParsed) memset(record_out, 0, ss_stwo_record_words(c) * 4);
    return r == kParsed ? 0 : r == kConfigMismatch ? (int)SS_STATUS_CONFIG_MISMATCH : (int)SS_STATUS_MALFORMED;
}

extern "C" int ss_s101_parse(const char *text, size_t len, int fmt, ss_s101_shape *shape, uint32_t *record_out)
{
    if (!text || !shape || fmt < SS_TEXT_AUTO || fmt > SS_TEXT_WIT) return set_err(SS_ERR_ARG, "bad argument");
    S101Parsed *p;
    try {
        p = s101_parse_text(text, len, fmt);
    } catch (const std::exception &e) {
        return set_err(SS_ERR_NOMEM, "host reader: %s", e.what());
    }
    if (!p) return (int)SS_STATUS_MALFORMED;
    uint32_t nl, pm;
    s101_parsed_shape(p, &nl, &pm);
    if (record_out && shape_ok(shape) && nl <= shape->max_layers && pm <= shape->max_path)
        s101_parsed_record(p, *shape, record_out);
    else { shape->max_layers = nl; shape->max_path = pm; }
    s101_parsed_free(p);
    return 0;
}

extern "C" size_t ss_stwo_write_text(const ss_stwo_cfg *c, const uint32_t *record, int fmt, int python_separators,
                                     char *buf, size_t cap)
{
    if (!cfg_ok(c) || !record || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT)) { set_err(SS_ERR_ARG, "bad argument"); return 0; }
    std::string out;
    const bool ok = fmt == SS_TEXT_JSON ? stwo_write_json(*c, record, python_separators ? kStylePython : kStyleCompact, out)
                                        : stwo_write_wit(*c, record, out);
    if (!ok) { set_err(SS_ERR_ARG, "record cannot be written in this format (path lengths / pow_target)"); return 0; }
    if (buf && out.size() <= cap) memcpy(buf, out.data(), out.size());
    return out.size();
}

// The minimal proof.json (formats.stwo_minimal_to_json) <-> minimal record; host only.
extern "C" int ss_stwo_parse_minimal(const ss_stwo_cfg *c, const char *text, size_t len, uint32_t *minimal_out, size_t cap_words,
                                     size_t *words_out)
{
    return ss_stwo_parse_minimal_route(c, text, len, SS_READER_AUTO, minimal_out, cap_words, words_out);
}

extern "C" int ss_stwo_parse_minimal_route(const ss_stwo_cfg *c, const char *text, size_t len, int reader, uint32_t *minimal_out,
                                           size_t cap_words, size_t *words_out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!text || !words_out || reader < SS_READER_AUTO || reader > SS_READER_STREAM) return set_err(SS_ERR_ARG, "bad argument");
    *words_out = 0;
    try {
        std::vector<uint32_t> rec;
        const ParseResult r = stwo_parse_minimal_text(*c, text, len, rec, reader);
        if (r == kDeclined) return SS_READER_DECLINED;
        if (r != kParsed) return r == kConfigMismatch ? (int)SS_STATUS_CONFIG_MISMATCH : (int)SS_STATUS_MALFORMED;
        *words_out = rec.size();
        if (!minimal_out || cap_words < rec.size()) return set_err(SS_ERR_ARG, "minimal record needs %zu words", rec.size());
        memcpy(minimal_out, rec.data(), rec.size() * 4);
        return 0;
    } catch (const std::exception &e) {
        return set_err(SS_ERR_NOMEM, "host reader: %s", e.what());
    }
}

extern "C" size_t ss_stwo_write_minimal_text(const ss_stwo_cfg *c, const uint32_t *minimal, size_t words, int python_separators,
                                             char *buf, size_t cap)
{
    if (!cfg_ok(c) || !minimal) { set_err(SS_ERR_ARG, "bad argument"); return 0; }
    try {
        std::string out;
        if (!stwo_write_json_minimal(*c, minimal, words, python_separators ? kStylePython : kStyleCompact, out)) {
            set_err(SS_ERR_ARG, "not a minimal record of this config (or a pow_target no proof.json can declare)");
            return 0;
        }
        if (buf && out.size() <= cap) memcpy(buf, out.data(), out.size());
        return out.size();
    } catch (const std::exception &) {
        set_err(SS_ERR_NOMEM, "out of host memory");
        return 0;
    }
}

// Minimal proof.json texts -> verdicts: read by the host reader on the library's worker threads (the GPU reader has no
// template for a text whose list lengths depend on the queries), verified as minimal records.
extern "C" int ss_stwo_verify_minimal_texts(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts,
                                            const size_t *lens, uint32_t *status_host, ss_ingest_stats *stats)
{
    if (!texts) return set_err(SS_ERR_ARG, "null argument");
    try {
        return stwo_minimal_ingest_dev(ctx, c, n, texts, lens, nullptr, status_host, stats);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_verify_minimal_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const void *blob,
                                                   const uint64_t *offs, const size_t *lens, uint32_t *status_host,
                                                   ss_ingest_stats *stats)
{
    if (!blob || !offs) return set_err(SS_ERR_ARG, "null argument");
    try {
        return stwo_minimal_ingest_dev(ctx, c, n, nullptr, lens, nullptr, status_host, stats, (const uint8_t *)blob, offs);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" size_t ss_stwo_write_shared_text(const ss_stwo_cfg *c, const uint32_t *shared, size_t words, int python_separators,
                                            char *buf, size_t cap)
{
    if (!cfg_ok(c) || !shared) { set_err(SS_ERR_ARG, "bad argument"); return 0; }
    try {
        std::string out;
        if (!stwo_write_json_shared(*c, shared, words, python_separators ? kStylePython : kStyleCompact, out)) {
            set_err(SS_ERR_ARG, "not a shared record of this config (or a pow_target no proof.json can declare)");
            return 0;
        }
        if (buf && out.size() <= cap) memcpy(buf, out.data(), out.size());
        return out.size();
    } catch (const std::exception &) {
        set_err(SS_ERR_NOMEM, "out of host memory");
        return 0;
    }
}

extern "C" int ss_stwo_text_is_canonical(const ss_stwo_cfg *c, const char *text, size_t len, int fmt, uint32_t *record_out)
{
    if (!cfg_ok(c) || !text || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT && fmt != SS_TEXT_JSON_SHARED && fmt != SS_TEXT_JSON_MINIMAL))
        return set_err(SS_ERR_ARG, "bad argument");
    try {
        // the template of the last (config, format) asked about is kept per thread: building one walks the whole text
        static thread_local TextTemplateHost h;
        static thread_local ss_stwo_cfg h_cfg;
        static thread_local int h_fmt = -1;
        if (h_fmt != fmt || memcmp(&h_cfg, c, sizeof h_cfg) != 0) {
            h_fmt = -1;  // (nothing is cached if building throws)
            stwo_build_template(*c, fmt, h);
            memcpy(&h_cfg, c, sizeof h_cfg);  // (ctypes / C callers zero the struct's padding; a mismatch only rebuilds)
            h_fmt = fmt;
        }
        if (!h.ok) return 0;
        std::vector<uint32_t> scratch;
        uint32_t *rec = record_out;
        if (!rec) { scratch.resize(std::max<size_t>(h.record_words, ss_stwo_record_words(c))); rec = scratch.data(); }
        if (fmt == SS_TEXT_JSON_SHARED) return shared_text_scan_reference(*c, h, text, len, rec) ? 1 : 0;
        if (fmt == SS_TEXT_JSON_MINIMAL) return minimal_text_scan_reference(*c, h, text, len, rec) ? 1 : 0;  // (capacity form)
        return text_scan_reference(h.view(), text, len, rec) ? 1 : 0;
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_minimal_from_capacity(const ss_stwo_cfg *c, const uint32_t *capacity, uint32_t *minimal_out, size_t cap_words,
                                             size_t *words_out)
{
    if (!cfg_ok(c) || !capacity || !words_out) return set_err(SS_ERR_ARG, "bad argument");
    const MinMap m = min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    bool ok = capacity[m.nv] <= m.Q && capacity[m.nv + 1] <= m.Q;
    for (uint32_t l = 0; l <= m.K; l++) ok &= capacity[m.nfw + l] <= m.Q;
    for (uint32_t t = 0; t < m.K + 3; t++) ok &= capacity[m.nhw + t] <= m.Q * min_tree_len(m.L, t);
    if (!ok) return set_err(SS_ERR_ARG, "list lengths beyond the capacity of the config");
    try {
        std::vector<uint32_t> out;
        minimal_compact(*c, capacity, out);
        *words_out = out.size();
        if (!minimal_out || cap_words < out.size()) return set_err(SS_ERR_ARG, "minimal record needs %zu words", out.size());
        memcpy(minimal_out, out.data(), out.size() * 4);
        return SS_OK;
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_minimal_to_capacity(const ss_stwo_cfg *c, const uint32_t *minimal, size_t words, uint32_t *capacity_out)
{
    if (!cfg_ok(c) || !minimal || !capacity_out) return set_err(SS_ERR_ARG, "bad argument");
    return minimal_to_capacity(*c, minimal, words, capacity_out) ? SS_OK : set_err(SS_ERR_ARG, "not a minimal record of this config");
}

extern "C" size_t ss_s101_write_text(const uint32_t *record, int fmt, int python_separators, char *buf, size_t cap)
{
    if (!record || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT)) { set_err(SS_ERR_ARG, "bad argument"); return 0; }
    std::string out;
    const bool ok = fmt == SS_TEXT_JSON ? s101_write_json(record, python_separators ? kStylePython : kStyleCompact, out)
                                        : s101_write_wit(record, out);
    if (!ok) { set_err(SS_ERR_ARG, "the record's path lengths are not the protocol's"); return 0; }
    if (buf && out.size() <= cap) memcpy(buf, out.data(), out.size());
    return out.size();
}

extern "C" int ss_s101_text_is_canonical(const char *text, size_t len, int fmt, uint32_t *record_out)
{
    if (!text || (fmt != SS_TEXT_JSON && fmt != SS_TEXT_WIT)) return set_err(SS_ERR_ARG, "bad argument");
    TextTemplateHost h;
    s101_build_template(fmt, h);
    if (!h.ok) return 0;
    std::vector<uint32_t> scratch(h.record_words, 0);
    uint32_t *rec = record_out ? record_out : scratch.data();
    if (record_out) memset(record_out, 0, (size_t)h.record_words * 4);  // shorter paths leave zero padding
    return text_scan_reference(h.view(), text, len, rec) ? 1 : 0;
}

double ss::now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool ss::read_file(const char *path, std::string &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    out.clear();
    char buf[1 << 16];
    size_t k;
    bool too_big = false;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) {
        out.append(buf, k);
        if (out.size() > ((size_t)32 << 20)) { too_big = true; break; }  // ss_ingest.cpp refuses texts above 32 MiB
    }
    const bool ok = !ferror(f) && !too_big;
    fclose(f);
    return ok;
}

extern "C" int ss_stwo_verify_texts(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts,
                                    const size_t *lens, int fmt, uint32_t *status_host, ss_ingest_stats *stats)
{
    try {
        return stwo_ingest_dev(ctx, c, n, texts, lens, nullptr, fmt, status_host, stats);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

// the same with the texts in ONE page-locked buffer of the caller's (text i at byte offs[i], a multiple of 16): no staging copy
extern "C" int ss_stwo_verify_texts_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *blob, const uint64_t *offs,
                                           const size_t *lens, int fmt, uint32_t *status_host, ss_ingest_stats *stats)
{
    if (!blob || !offs) return set_err(SS_ERR_ARG, "null argument");
    try {
        return stwo_ingest_dev(ctx, c, n, nullptr, lens, nullptr, fmt, status_host, stats, (const uint8_t *)blob, offs);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_read_texts(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *texts,
                                  const size_t *lens, int fmt, uint32_t *records_host, uint32_t *outcome_host)
{
    return stwo_read_texts_dev(ctx, c, n, texts, lens, fmt, records_host, outcome_host);
}

extern "C" int ss_s101_read_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                                  uint32_t *records_host, uint32_t *outcome_host)
{
    return stwo_read_texts_dev(ctx, nullptr, n, texts, lens, fmt, records_host, outcome_host);
}

extern "C" int ss_stwo_verify_files(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const char *const *paths, int fmt,
                                    uint32_t *status_host, ss_ingest_stats *stats)
{
    try {
        return stwo_ingest_dev(ctx, c, n, nullptr, nullptr, paths, fmt, status_host, stats);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_s101_verify_texts(ss_ctx *ctx, size_t n, const char *const *texts, const size_t *lens, int fmt,
                                    uint32_t *status_host, ss_ingest_stats *stats)
{
    try {
        return s101_ingest_dev(ctx, n, texts, lens, nullptr, fmt, status_host, stats);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_s101_verify_texts_pinned(ss_ctx *ctx, size_t n, const char *blob, const uint64_t *offs, const size_t *lens, int fmt,
                                           uint32_t *status_host, ss_ingest_stats *stats)
{
    if (!blob || !offs) return set_err(SS_ERR_ARG, "null argument");
    try {
        return s101_ingest_dev(ctx, n, nullptr, lens, nullptr, fmt, status_host, stats, (const uint8_t *)blob, offs);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_s101_verify_files(ss_ctx *ctx, size_t n, const char *const *paths, int fmt, uint32_t *status_host,
                                    ss_ingest_stats *stats)
{
    try {
        return s101_ingest_dev(ctx, n, nullptr, nullptr, paths, fmt, status_host, stats);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

// =========================================================================== self-test
namespace ss {
__global__ void selftest_kernel(int op, uint32_t n, const uint32_t *in, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
    case 0: {
        uint32_t l[8], r[8], o[8];
        for (int j = 0; j < 8; j++) { l[j] = in[16 * i + j]; r[j] = in[16 * i + 8 + j]; }
        sha256_pair(l, r, o);
        for (int j = 0; j < 8; j++) out[8 * i + j] = o[j];
        break;
    }
    case 1: {
        uint32_t a = in[2 * i], b = in[2 * i + 1], inv;
        out[4 * i] = m31_add(a, b);
        out[4 * i + 1] = m31_sub(a, b);
        out[4 * i + 2] = m31_mul(a, b);
        out[4 * i + 3] = m31_inv(a, inv) ? inv : 0xffffffffu;
        break;
    }
    case 2: {
        QM31 a = {in[8 * i], in[8 * i + 1], in[8 * i + 2], in[8 * i + 3]};
        QM31 b = {in[8 * i + 4], in[8 * i + 5], in[8 * i + 6], in[8 * i + 7]};
        QM31 m = qm31_mul(a, b), v;
        bool ok = qm31_inv(a, v);
        if (!ok) v = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        out[8 * i] = m.a; out[8 * i + 1] = m.b; out[8 * i + 2] = m.c; out[8 * i + 3] = m.d;
        out[8 * i + 4] = v.a; out[8 * i + 5] = v.b; out[8 * i + 6] = v.c; out[8 * i + 7] = v.d;
        break;
    }
    case 3: {
        M31Point pt = circle_point(in[i]);
        out[2 * i] = pt.x;
        out[2 * i + 1] = pt.y;
        break;
    }
    case 4: {
        uint32_t a = in[2 * i], b = in[2 * i + 1], d;
        out[4 * i] = f101_add(a, b);
        out[4 * i + 1] = f101_sub(a, b);
        out[4 * i + 2] = f101_mul(a, b);
        out[4 * i + 3] = f101_div(a, b, d) ? d : 0xffffffffu;
        break;
    }
    case 5: {  // lazily reduced forms on operands in [0, P]: mul, sqr, mul by (0 + y u), canonical add/sub/mul
        QM31 a = {in[8 * i], in[8 * i + 1], in[8 * i + 2], in[8 * i + 3]};
        QM31 b = {in[8 * i + 4], in[8 * i + 5], in[8 * i + 6], in[8 * i + 7]};
        const QM31 m = qm31_mul_c(a, b), s = qm31_sqr_c(a), h = qm31_mul_im_c(a, q_im(b));
        uint32_t *o = out + 16 * i;
        o[0] = m.a; o[1] = m.b; o[2] = m.c; o[3] = m.d;
        o[4] = s.a; o[5] = s.b; o[6] = s.c; o[7] = s.d;
        o[8] = h.a; o[9] = h.b; o[10] = h.c; o[11] = h.d;
        const uint32_t x = m31_red(a.a), y = m31_red(b.a);  // strictly canonical for the min-based forms
        o[12] = m31_add_c(x, y); o[13] = m31_sub_c(x, y); o[14] = m31_mul_c(x, y);
        o[15] = m31_red64(((uint64_t)a.b << 32) | b.b);
        break;
    }
    case 6: {  // the asserts behind the FRI layer loop, as the query kernel evaluates them (ss_stwo_checks.h); 0 = none fails
        const uint32_t *x = in + 13 * i;
        const QM31 eval = {x[5], x[6], x[7], x[8]}, last = {x[9], x[10], x[11], x[12]};
        const uint32_t c = stwo_last_layer_code(x[0], x[1], x[2], x[3], x[4], eval, last);
        out[i] = c == 0xffffffffu ? 0 : c;
        break;
    }
    }
}
}  // namespace ss

extern "C" int ss_selftest(ss_ctx *ctx, int op, size_t n, const uint32_t *in_host, uint32_t *out_host)
{
    static const int in_w[7] = {16, 2, 8, 1, 2, 8, 13}, out_w[7] = {8, 4, 8, 2, 4, 16, 1};
    if (!ctx || !in_host || !out_host || op < 0 || op > 6 || !n) return set_err(SS_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    SS_DEVICE_GUARD(ctx);
    DevBuf a, b;
    HIP_TRY(hipMalloc(&a.p, n * in_w[op] * 4));
    HIP_TRY(hipMalloc(&b.p, n * out_w[op] * 4));
    HIP_TRY(hipMemcpy(a.p, in_host, n * in_w[op] * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ss::selftest_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (uint32_t)n,
                       (const uint32_t *)a.p, (uint32_t *)b.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, b.p, n * out_w[op] * 4, hipMemcpyDeviceToHost));
    return SS_OK;
}
