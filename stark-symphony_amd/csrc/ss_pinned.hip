// Host entry points WITHOUT the staging copy: the caller's buffer is page-locked (hipHostMalloc, or any memory passed to
// ss_host_register) and the DMA engine reads it directly, chunk by chunk, while the previous chunk is re-tiled /
// expanded and verified on a second stream.  The staged twins (ss_stwo_verify_records, .._shared_records,
// .._minimal_records) copy every input into the library's own pinned buffers first, which costs host cores: at one
// process per GPU and eight GPUs per host a rank has two of the 16 granted cores (bench.py), and a streaming copy into
// pinned memory needs four to feed a 55 GB/s link (profiles/r03_pcie_probe.txt).  Here the host does nothing per byte.
//
// Reference anchor: the reference's caller hands ONE file per process (stwo-verifier/Makefile:17-18 `simfony run
// --witness`); a batch caller that already holds its records in one buffer is what SURVEY.md 8(e) "each rank reads only
// its slice" amounts to on the host side.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "ss_abi.h"
#include "ss_ctx.h"
#include "ss_layout.h"
#include "ss_minimal.h"
#include "ss_shared.h"

namespace ss {

__global__ void stwo_shared_outcome_kernel(uint32_t n, const uint32_t *__restrict__ outcome, uint32_t *__restrict__ status);  // ss_shared.hip

static bool is_locked_host(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// kind 0 per-query records (offs == nullptr: n records of W words back to back), 1 shared, 2 minimal
static int verify_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *flat, const uint64_t *offs, int kind,
                         uint32_t *status_host)
{
    if (!ctx || !flat || !status_host || (kind && !offs)) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n * (size_t)kMaxQueries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const size_t W = ss_stwo_record_words(c);
    if (kind)
        for (size_t i = 0; i < n; i++)
            if (offs[i + 1] < offs[i]) return set_err(SS_ERR_ARG, "offsets must ascend");
    const size_t total_words = kind ? (size_t)offs[n] : n * W;
    if (!is_locked_host(flat) || (total_words && !is_locked_host(flat + total_words - 1)))
        return set_err(SS_ERR_ARG, "the buffer is not page-locked host memory (hipHostMalloc / ss_host_register)");
    std::lock_guard<std::mutex> lock(ctx->mu);
    SS_DEVICE_GUARD(ctx);
    auto start = [&](size_t i) { return kind ? (size_t)offs[i] : i * W; };
    // chunks of <= 256 MiB of input (the first ones small and doubling: nothing overlaps the first upload), <= 16384 records
    const size_t budget = (256u << 20) / 4;  // (no staging to balance against: few, large chunks)
    std::vector<size_t> first;
    {
        size_t lo = 0, step = budget / 16;
        while (lo < n) {
            first.push_back(lo);
            size_t hi = lo + 1;
            while (hi < n && start(hi + 1) - start(lo) <= step && hi - lo < 16384) hi++;
            lo = hi;
            step = std::min(budget, step * 2);
        }
        first.push_back(n);
    }
    size_t chunk_words = 0, chunk_n = 0, bwords = 0, wsb = 0;
    for (size_t k = 0; k + 1 < first.size(); k++) {
        const size_t cnt = first[k + 1] - first[k];
        chunk_words = std::max(chunk_words, start(first[k + 1]) - start(first[k]) + 2 * (cnt + 1) + 2);
        chunk_n = std::max(chunk_n, cnt);
        bwords = std::max(bwords, kind == 2 ? ss_stwo_minimal_batch_words(c, cnt) : ss_stwo_batch_words(c, cnt));
        wsb = std::max(wsb, kind == 2 ? ss_stwo_minimal_workspace_bytes(c, cnt) : ss_stwo_workspace_bytes(c, cnt));
    }
    HostPath &hp = ctx->hp;
    int rc;
    if (kind == 1 && (rc = hp_reserve(ctx, 0, chunk_n * W * 4))) return rc;   // the expansion's records
    if ((rc = hp_reserve(ctx, 1, bwords * 4))) return rc;
    if ((rc = hp_reserve(ctx, 2, wsb))) return rc;
    if ((rc = hp_reserve(ctx, 3, n * 4))) return rc;
    if ((rc = hp_reserve(ctx, 4, chunk_words * 4))) return rc;
    if ((rc = hp_reserve(ctx, 5, chunk_words * 4))) return rc;
    if (kind == 1 && (rc = hp_reserve(ctx, 6, n * 4))) return rc;
    if ((rc = hp_pinned(ctx, std::max<size_t>(4096, 8 * (chunk_n + 2))))) return rc;  // (only the chunks' offset tables are staged)
    for (int i = 0; i < 2; i++)
        if (!hp.shared_free[i]) HIP_TRY(hipEventCreateWithFlags(&hp.shared_free[i], hipEventDisableTiming));
    hipStream_t s = hp.stream, vs = hp.vstream;
    uint32_t *status_dev = (uint32_t *)hp.dev[3], *outcome_dev = (uint32_t *)hp.dev[6];
    auto run = [&]() -> int {
        int buf = 0;
        for (size_t k = 0; k + 1 < first.size(); k++) {
            const size_t lo = first[k], cnt = first[k + 1] - lo;
            const size_t words = start(lo + cnt) - start(lo);
            uint32_t *dev = (uint32_t *)hp.dev[4 + buf];
            uint64_t head = 0;  // words in front of the records: the chunk's offset table (shared / minimal)
            HIP_TRY(hipStreamWaitEvent(s, hp.shared_free[buf], 0));  // the work that last read this device buffer
            if (kind) {
                HIP_TRY(hipEventSynchronize(hp.pinned_free[buf]));
                uint64_t *o = (uint64_t *)hp.pinned[buf];
                head = 2 * (cnt + 1);
                head += head & 1;
                for (size_t i = 0; i <= cnt; i++) o[i] = head + (offs[lo + i] - offs[lo]);
                HIP_TRY(hipMemcpyAsync(dev, o, 8 * (cnt + 1), hipMemcpyHostToDevice, s));
            }
            if (words) HIP_TRY(hipMemcpyAsync(dev + head, flat + start(lo), words * 4, hipMemcpyHostToDevice, s));  // straight from the caller's memory
            HIP_TRY(hipEventRecord(hp.pinned_free[buf], s));  // the chunk is up (and its offset table's staging is free again)
            HIP_TRY(hipStreamWaitEvent(vs, hp.pinned_free[buf], 0));
            int r;
            if (kind == 0) {
                if ((r = ss_stwo_pack_dev(ctx, c, cnt, dev, (uint32_t *)hp.dev[1], vs))) return r;
                r = ss_stwo_verify_batch_dev(ctx, c, cnt, (const uint32_t *)hp.dev[1], hp.dev[2], wsb, status_dev + lo, nullptr, vs);
            } else if (kind == 1) {
                if ((r = shared_expand_launch(ctx, c, cnt, dev, (const uint64_t *)dev, 0, (uint32_t *)hp.dev[0], outcome_dev + lo, vs))) return r;
                if ((r = ss_stwo_pack_dev(ctx, c, cnt, (const uint32_t *)hp.dev[0], (uint32_t *)hp.dev[1], vs))) return r;
                r = ss_stwo_verify_batch_dev(ctx, c, cnt, (const uint32_t *)hp.dev[1], hp.dev[2], wsb, status_dev + lo, nullptr, vs);
            } else {
                r = ss_stwo_verify_minimal_dev(ctx, c, cnt, dev, (const uint64_t *)dev, (uint32_t *)hp.dev[1], hp.dev[2], wsb,
                                               status_dev + lo, nullptr, SS_PHASE_ALL, vs);
            }
            if (r) return r;
            HIP_TRY(hipEventRecord(hp.shared_free[buf], vs));
            buf ^= 1;
        }
        if (kind == 1)
            hipLaunchKernelGGL(stwo_shared_outcome_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vs, (uint32_t)n, outcome_dev,
                               status_dev);
        HIP_TRY(hipMemcpyAsync(status_host, status_dev, n * 4, hipMemcpyDeviceToHost, vs));
        return SS_OK;
    };
    rc = run();
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(vs);
    if (rc) return rc;
    HIP_TRY(e1);
    HIP_TRY(e2);
    return SS_OK;
}

}  // namespace ss

using namespace ss;

extern "C" int ss_host_register(ss_ctx *ctx, void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return set_err(SS_ERR_ARG, "null argument");
    SS_DEVICE_GUARD(ctx);
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return SS_OK;
}

extern "C" int ss_host_unregister(ss_ctx *ctx, void *ptr)
{
    if (!ptr) return set_err(SS_ERR_ARG, "null argument");
    SS_DEVICE_GUARD(ctx);
    HIP_TRY(hipHostUnregister(ptr));
    return SS_OK;
}

extern "C" int ss_stwo_verify_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *records, uint32_t *status_host)
{
    try {
        return verify_pinned(ctx, c, n, records, nullptr, 0, status_host);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_verify_shared_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *flat,
                                                    const uint64_t *offs, uint32_t *status_host)
{
    try {
        return verify_pinned(ctx, c, n, flat, offs, 1, status_host);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}

extern "C" int ss_stwo_verify_minimal_records_pinned(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *flat,
                                                     const uint64_t *offs, uint32_t *status_host)
{
    try {
        return verify_pinned(ctx, c, n, flat, offs, 2, status_host);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}
