// C ABI of libss_verify.so (include/ss_verify.h): packers, context, launches, self-test.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ss_verify.h"
#include "ss_fields.h"
#include "ss_kernels.h"
#include "ss_layout.h"
#include "ss_sha256.h"

using namespace ss;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return set_err(SS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

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
struct TimedSpan {
    const char *name;
    hipEvent_t start, stop;
};

struct ss_ctx {
    int device;
    int timing;
    std::vector<TimedSpan> spans;   // recorded since the last collect
    std::vector<hipEvent_t> pool;   // recycled events
};

static constexpr size_t kMaxSpans = 1 << 16;

extern "C" int ss_ctx_create(int device, ss_ctx **out)
{
    if (!out) return set_err(SS_ERR_ARG, "out is null");
    int n = ss_device_count();
    if (n <= 0) return set_err(SS_ERR_NO_DEVICE, "no HIP device visible (%s)", g_err);
    if (device < 0 || device >= n) return set_err(SS_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(SS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    ss_ctx *c = new ss_ctx();
    c->device = device;
    c->timing = 0;
    *out = c;
    return SS_OK;
}

extern "C" void ss_ctx_destroy(ss_ctx *ctx)
{
    if (!ctx) return;
    for (auto &sp : ctx->spans) { (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop); }
    for (auto &e : ctx->pool) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" int ss_ctx_set_timing(ss_ctx *ctx, int enabled)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
    ctx->timing = enabled;
    return SS_OK;
}

static hipEvent_t take_event(ss_ctx *c)
{
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

extern "C" int ss_ctx_collect_timing(ss_ctx *ctx, int cap, const char **names, float *total_ms,
                                     uint32_t *launches)
{
    if (!ctx || !names || !total_ms || !launches) return set_err(SS_ERR_ARG, "null argument");
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

// Records one (start, stop) event pair around each kernel launch on the launch stream.
struct Timer {
    ss_ctx *c;
    hipStream_t s;
    hipEvent_t cur = nullptr;
    Timer(ss_ctx *c_, hipStream_t s_) : c(c_), s(s_) {}
    void begin()
    {
        if (!c->timing || c->spans.size() >= kMaxSpans) return;
        cur = take_event(c);
        (void)hipEventRecord(cur, s);
    }
    void end(const char *name)
    {
        if (!cur) return;
        hipEvent_t stop = take_event(c);
        (void)hipEventRecord(stop, s);
        c->spans.push_back({name, cur, stop});
        cur = nullptr;
    }
};

// ------------------------------------------------------------------------ host threads
template <class F>
static void parallel_for(size_t n, F f)
{
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, std::min<size_t>(n, 32)));
    if (nt == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++)
        th.emplace_back([=]() { for (size_t i = t; i < n; i += nt) f(i); });
    for (auto &x : th) x.join();
}

// ================================================================================ stwo
static bool cfg_ok(const ss_stwo_cfg *c)
{
    return c && stwo_cfg_ok(c->n_cols, c->trace_log, c->lde_log, c->n_queries, c->n_layers, c->mode);
}
static StwoLayout lay_of(const ss_stwo_cfg *c, size_t n)
{
    return stwo_layout(c->n_cols, c->trace_log, c->lde_log, c->n_queries, c->n_layers, c->mode,
                       c->pow_target, n);
}

extern "C" size_t ss_stwo_record_words(const ss_stwo_cfg *c)
{
    return cfg_ok(c) ? (size_t)stwo_record_words(c->n_cols, c->lde_log, c->n_queries, c->n_layers) : 0;
}
extern "C" size_t ss_stwo_batch_words(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n).total_words : 0;
}
extern "C" size_t ss_stwo_workspace_bytes(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n).ws_total_words * 4 : 0;
}

extern "C" int ss_stwo_pack(const ss_stwo_cfg *c, size_t n, const uint32_t *const *records,
                            uint32_t *out)
{
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !records || !out) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const StwoLayout y = lay_of(c, n);
    memset(out, 0, (size_t)y.total_words * 4);
    const uint32_t N = y.N, L = y.L, Q = y.Q, K = y.K;
    parallel_for(n, [&](size_t p) {
        const uint32_t *r = records[p];
        for (uint32_t w = 0; w < y.head_words; w++) out[y.off_head + (uint64_t)w * y.np + p] = r[w];
        r += y.head_words;
        for (uint32_t q = 0; q < Q; q++) {
            const uint64_t inst = (uint64_t)p * Q + q;
            for (uint32_t k = 0; k < N; k++) out[y.off_trace_vals + (uint64_t)k * y.nip + inst] = *r++;
            for (uint32_t k = 0; k < kCp; k++) out[y.off_cp_vals + (uint64_t)k * y.nip + inst] = *r++;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t w = 0; w < 8; w++) out[tile_word(y.off_trace_path, L, inst, l, w)] = *r++;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t w = 0; w < 8; w++) out[tile_word(y.off_cp_path, L, inst, l, w)] = *r++;
        }
        for (uint32_t l = 0; l <= K; l++) {
            const uint32_t len = L - 1 - l;
            for (uint32_t q = 0; q < Q; q++) {
                const uint64_t inst = (uint64_t)p * Q + q;
                for (uint32_t w = 0; w < 4; w++)
                    out[y.off_fri_wit + ((uint64_t)l * 4 + w) * y.nip + inst] = *r++;
                for (uint32_t lv = 0; lv < len; lv++)
                    for (uint32_t w = 0; w < 8; w++) out[tile_word(y.off_fri_path[l], len, inst, lv, w)] = *r++;
            }
        }
    });
    return SS_OK;
}

extern "C" int ss_stwo_verify_phase_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                        const uint32_t *batch, const uint32_t *shape_status,
                                        void *workspace, size_t workspace_bytes, uint32_t *status,
                                        uint32_t *accept_count, int phases, void *stream_)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
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
        HIP_TRY(hipMemsetAsync(status, 0xff, n * 4, s));
        if (accept_count) HIP_TRY(hipMemsetAsync(accept_count, 0, 4, s));
        t.begin();
        hipLaunchKernelGGL(stwo_transcript_kernel, dim3((y.n + 63) / 64), dim3(64), 0, s, y, batch, ws, status);
        t.end("stwo_transcript");
        t.begin();
        hipLaunchKernelGGL(stwo_query_kernel, dim3((y.ni + 63) / 64), dim3(64), 0, s, y, batch, ws, status);
        t.end("stwo_query");
    }
    if (phases & SS_PHASE_TAIL) {
        const uint32_t tiles = (y.K + 3) * (y.nip >> 6);
        t.begin();
        hipLaunchKernelGGL(stwo_merkle_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, y, batch, ws, status);
        t.end("stwo_merkle");
        t.begin();
        hipLaunchKernelGGL(stwo_finalize_kernel, dim3((y.n + 255) / 256), dim3(256), 0, s, y.n, status,
                           shape_status, accept_count);
        t.end("stwo_finalize");
    }
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

extern "C" int ss_stwo_verify_batch_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                        const uint32_t *batch, const uint32_t *shape_status,
                                        void *workspace, size_t workspace_bytes, uint32_t *status,
                                        uint32_t *accept_count, void *stream_)
{
    return ss_stwo_verify_phase_dev(ctx, c, n, batch, shape_status, workspace, workspace_bytes, status,
                                    accept_count, SS_PHASE_ALL, stream_);
}

// ============================================================================ stark101
static bool shape_ok(const ss_s101_shape *sh) { return sh && s101_shape_ok(sh->max_layers, sh->max_path); }

extern "C" size_t ss_s101_record_words(const ss_s101_shape *sh)
{
    return shape_ok(sh) ? (size_t)s101_record_words(sh->max_layers, sh->max_path) : 0;
}
extern "C" size_t ss_s101_batch_words(const ss_s101_shape *sh, size_t n)
{
    return shape_ok(sh) && n ? (size_t)s101_layout(sh->max_layers, sh->max_path, n).total_words : 0;
}
extern "C" size_t ss_s101_workspace_bytes(const ss_s101_shape *sh, size_t n)
{
    return shape_ok(sh) && n ? (size_t)s101_layout(sh->max_layers, sh->max_path, n).ws_total_words * 4 : 0;
}

extern "C" int ss_s101_pack(const ss_s101_shape *sh, size_t n, const uint32_t *const *records,
                            uint32_t *out)
{
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n || !records || !out) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const S101Layout y = s101_layout(sh->max_layers, sh->max_path, n);
    memset(out, 0, (size_t)y.total_words * 4);
    const uint32_t ML = y.ML, PM = y.PM;
    parallel_for(n, [&](size_t p) {
        const uint32_t *r = records[p];
        auto head = [&](uint32_t w) -> uint32_t & { return out[y.off_head + (uint64_t)w * y.np + p]; };
        auto chain = [&](uint32_t type, const uint32_t *&rr) {
            out[y.off_leaf + (uint64_t)type * y.np + p] = *rr++;
            uint32_t len = *rr++;
            out[y.off_len + (uint64_t)type * y.np + p] = len;
            const uint64_t base = y.off_path + type * y.path_stride;
            for (uint32_t l = 0; l < PM; l++)
                for (uint32_t w = 0; w < 8; w++) out[tile_word(base, PM, p, l, w)] = *rr++;
        };
        for (uint32_t w = 0; w < 8; w++) head(y.h_root + w) = *r++;
        head(y.h_nlayers) = *r++;
        head(y.h_last) = *r++;
        for (uint32_t k = 0; k < 3; k++) chain(k, r);
        for (uint32_t i = 0; i < ML; i++) {
            for (uint32_t w = 0; w < 9; w++) head(y.h_layer + 9 * i + w) = *r++;  // root[8], beta
            chain(3 + 2 * i, r);
            chain(4 + 2 * i, r);
        }
    });
    return SS_OK;
}

extern "C" int ss_s101_verify_phase_dev(ss_ctx *ctx, const ss_s101_shape *sh, size_t n,
                                        const uint32_t *batch, void *workspace, size_t workspace_bytes,
                                        uint32_t *status, uint32_t *accept_count, int phases,
                                        void *stream_)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
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
        HIP_TRY(hipMemsetAsync(status, 0xff, n * 4, s));
        if (accept_count) HIP_TRY(hipMemsetAsync(accept_count, 0, 4, s));
        t.begin();
        hipLaunchKernelGGL(s101_transcript_kernel, dim3((y.n + 63) / 64), dim3(64), 0, s, y, batch, ws, status);
        t.end("s101_transcript");
    }
    if (phases & SS_PHASE_TAIL) {
        const uint32_t tiles = y.n_types * (y.np >> 6);
        t.begin();
        hipLaunchKernelGGL(s101_merkle_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, y, batch, ws, status);
        t.end("s101_merkle");
        t.begin();
        hipLaunchKernelGGL(stwo_finalize_kernel, dim3((y.n + 255) / 256), dim3(256), 0, s, y.n, status,
                           (const uint32_t *)nullptr, accept_count);
        t.end("s101_finalize");
    }
    HIP_TRY(hipGetLastError());
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

extern "C" int ss_stwo_verify_records(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n,
                                      const uint32_t *const *records, const uint32_t *shape_status_host,
                                      uint32_t *status_host)
{
    if (!ctx || !status_host) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t words = ss_stwo_batch_words(c, n), wsb = ss_stwo_workspace_bytes(c, n);
    std::vector<uint32_t> host(words);
    int rc = ss_stwo_pack(c, n, records, host.data());
    if (rc) return rc;
    DevBuf b, w, st, sh;
    HIP_TRY(hipMalloc(&b.p, words * 4));
    HIP_TRY(hipMalloc(&w.p, wsb));
    HIP_TRY(hipMalloc(&st.p, n * 4));
    HIP_TRY(hipMemcpy(b.p, host.data(), words * 4, hipMemcpyHostToDevice));
    if (shape_status_host) {
        HIP_TRY(hipMalloc(&sh.p, n * 4));
        HIP_TRY(hipMemcpy(sh.p, shape_status_host, n * 4, hipMemcpyHostToDevice));
    }
    rc = ss_stwo_verify_batch_dev(ctx, c, n, (const uint32_t *)b.p, (const uint32_t *)sh.p, w.p, wsb,
                                  (uint32_t *)st.p, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(status_host, st.p, n * 4, hipMemcpyDeviceToHost));
    return SS_OK;
}

extern "C" int ss_s101_verify_records(ss_ctx *ctx, const ss_s101_shape *sh, size_t n,
                                      const uint32_t *const *records, uint32_t *status_host)
{
    if (!ctx || !status_host) return set_err(SS_ERR_ARG, "null argument");
    if (!shape_ok(sh)) return set_err(SS_ERR_ARG, "unsupported stark101 shape");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t words = ss_s101_batch_words(sh, n), wsb = ss_s101_workspace_bytes(sh, n);
    std::vector<uint32_t> host(words);
    int rc = ss_s101_pack(sh, n, records, host.data());
    if (rc) return rc;
    DevBuf b, w, st;
    HIP_TRY(hipMalloc(&b.p, words * 4));
    HIP_TRY(hipMalloc(&w.p, wsb));
    HIP_TRY(hipMalloc(&st.p, n * 4));
    HIP_TRY(hipMemcpy(b.p, host.data(), words * 4, hipMemcpyHostToDevice));
    rc = ss_s101_verify_batch_dev(ctx, sh, n, (const uint32_t *)b.p, w.p, wsb, (uint32_t *)st.p, nullptr,
                                  nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(status_host, st.p, n * 4, hipMemcpyDeviceToHost));
    return SS_OK;
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
    }
}
}  // namespace ss

extern "C" int ss_selftest(ss_ctx *ctx, int op, size_t n, const uint32_t *in_host, uint32_t *out_host)
{
    static const int in_w[5] = {16, 2, 8, 1, 2}, out_w[5] = {8, 4, 8, 2, 4};
    if (!ctx || !in_host || !out_host || op < 0 || op > 4 || !n) return set_err(SS_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
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
