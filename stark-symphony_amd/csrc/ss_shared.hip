// Shared-path records on the GPU: the expansion kernel (shared -> per-query records, a pure gather) and the
// host-buffer entry point that uploads the SMALLER shared records and expands them behind the link.
//
// Anchors: stwo-verifier/src/fri/queries.simf:41 (the reference does not deduplicate) and
// stwo-verifier/scripts/generate_wit.py:36-42 (its adapter splits the witness lists per query): the per-query
// record IS the reference's witness; a shared record is the same bytes with every repeated sibling stored once.
// The rule is stated in ss_shared.h; the host twin is ss_stwo_unshare_record (ss_sharedrec.cpp).
//
// HBM-bound word shuffling, no MFMA.  One 256-thread block per proof: wave 0 turns the Q hinted positions
// into the plan (pairwise bit lengths by shuffles, per-tree prefix sums), all four waves then copy -- a
// sibling is four consecutive lanes (eight where a side is not 8-byte aligned) reading 32 contiguous bytes of
// the node list and writing 32 contiguous bytes of the record, so both sides are full 32-byte sectors; a lane
// keeps kGather siblings in flight (3.2 TB/s with the chip full, profiles/r04_shared_expand_probe.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "ss_abi.h"
#include "ss_copy.h"
#include "ss_ctx.h"
#include "ss_layout.h"
#include "ss_shared.h"

namespace ss {

struct SharedArgs {
    SharedMap m;
    uint32_t record_words, qstride, fbase, tbase;
    uint32_t foff[kMaxList + 1];
    uint32_t n;
    uint32_t capacity;        // 1: fixed-capacity form (tree t's nodes at nodes + 8 Q sum_{t' < t} len_t', no counts, no size check)
    uint64_t stride;          // capacity form: words between records
    const uint64_t *offs;     // compact form: n + 1 word offsets of the records inside `shared`
    const uint32_t *shared;
    uint32_t *records;
    uint32_t *outcome;        // 0 = expanded, `refuse` = no shared record of this config
    uint32_t refuse;          // SS_STATUS_MALFORMED; the GPU text reader: 1 = "the host reader decides"
    // the GPU text reader (capacity form): only texts of format 2 whose outcome is still 0 are expanded, and the positions
    // the place pass stored must be the ones the template was cut with (hint_pos + p * hint_stride words)
    const uint8_t *only_fmt;
    const uint32_t *hint_pos;
    uint32_t hint_stride;
};

#ifndef SS_GATHER
#define SS_GATHER 4
#endif
constexpr uint32_t kGather = SS_GATHER;  // siblings a lane keeps in flight in the copy phase

__global__ void __launch_bounds__(256) stwo_shared_expand_kernel(SharedArgs a)
{
    __shared__ uint8_t s_d[kMaxQueries][kMaxQueries];    // d(q, e) for e < q
    __shared__ uint8_t s_lead[kMaxQueries][32];          // lead(q, absolute level)
    __shared__ uint16_t s_base[kMaxList + 3][kMaxQueries];
    __shared__ uint32_t s_node0[kMaxList + 3];           // first word of tree t's nodes inside the record
    __shared__ uint32_t s_short, s_bad;
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    if (p >= a.n) return;
    if (a.only_fmt && (a.only_fmt[p] != 2 || a.outcome[p] != 0)) return;
    const SharedMap &m = a.m;
    const uint32_t N = m.N, L = m.L, Q = m.Q, K = m.K;
    const uint64_t off = a.capacity ? (uint64_t)p * a.stride : a.offs[p];
    const uint64_t words = a.capacity ? a.stride : a.offs[p + 1] - off;
    const uint32_t *sh = a.shared + off;
    uint32_t *rec = a.records + (uint64_t)p * a.record_words;
    if (tid == 0) s_short = words < m.nodes;
    __syncthreads();
    if (s_short) {  // (too short to hold even the fixed words: nothing of it is read)
        for (uint32_t i = tid; i < a.record_words; i += 256) rec[i] = 0;
        if (tid == 0) a.outcome[p] = a.refuse;
        return;
    }
    // ---- the plan (wave 0; lane = query)
    if (tid < 64) {
        const uint32_t q = tid;
        const uint32_t pos = q < Q ? sh[m.qry + q] : 0;
        bool bad = (pos >> L) != 0;
        if (a.hint_pos && q < Q) bad |= pos != a.hint_pos[(uint64_t)p * a.hint_stride + q];
        uint32_t s = 32;
        for (uint32_t e = 0; e < Q; e++) {
            const uint32_t other = __shfl(pos, e);
            if (e < q) {
                const uint32_t x = pos ^ other;
                const uint32_t d = x ? 32 - __clz(x) : 0;
                s_d[q][e] = (uint8_t)d;
                s = d < s ? d : s;
            }
        }
        uint64_t total = m.nodes;
        uint32_t cap0 = m.nodes;
        for (uint32_t t = 0; t < K + 3; t++) {
            const uint32_t fresh = q < Q ? shared_fresh(s, L, t) : 0;
            uint32_t incl = fresh;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(incl, d);
                if (q >= d) incl += y;
            }
            if (q < Q) s_base[t][q] = (uint16_t)(incl - fresh);
            const uint32_t count = __shfl(incl, 63);
            if (q == 0) s_node0[t] = a.capacity ? cap0 : (uint32_t)total;
            if (!a.capacity) bad |= sh[m.cnt + t] != count;
            total += 8ull * count;
            cap0 += 8 * Q * shared_tree_len(L, t);
        }
        if (!a.capacity) bad |= total != words;
        const bool any = __ballot(bad) != 0;
        if (q == 0) s_bad = any;
    }
    __syncthreads();
    if (s_bad) {
        for (uint32_t i = tid; i < a.record_words; i += 256) rec[i] = 0;
        if (tid == 0) a.outcome[p] = a.refuse;
        return;
    }
    // lead(q, a) = the first e < q with d(q, e) <= a, else q
    for (uint32_t i = tid; i < Q * 32; i += 256) {
        const uint32_t q = i >> 5, lv = i & 31;
        uint32_t e = 0;
        while (e < q && s_d[q][e] > lv) e++;
        s_lead[q][lv] = (uint8_t)e;
    }
    // ---- words that only move: head, queried values, FRI witnesses, the path-length trailer
    for (uint32_t i = tid; i < m.head; i += 256) rec[i] = sh[i];
    {
        const uint32_t per_q = N + kCp;
        for (uint32_t i = tid; i < Q * per_q; i += 256) {
            const uint32_t q = i / per_q, r = i - q * per_q;
            rec[m.head + q * a.qstride + r] = sh[m.vals + i];
        }
    }
    for (uint32_t i = tid; i < 4 * Q * (K + 1); i += 256) {
        const uint32_t w = i & 3, lq = i >> 2, l = lq / Q, q = lq - l * Q;
        rec[a.fbase + a.foff[l] + q * (4 + 8 * (L - 1 - l)) + w] = sh[m.wit + i];
    }
    for (uint32_t i = tid; i < (K + 3) * Q; i += 256) rec[a.tbase + i] = shared_tree_len(L, i / Q);
    __syncthreads();
    // ---- the siblings
    const uint32_t sub = tid & 7, grp = tid >> 3;
    for (uint32_t t = 0; t < K + 3; t++) {
        const uint32_t len = shared_tree_len(L, t), shift = shared_tree_shift(t);
        const uint32_t *nodes = sh + s_node0[t];
        uint32_t dst0, per_q;
        if (t < 2) { dst0 = m.head + N + kCp + t * 8 * L; per_q = a.qstride; }
        else { const uint32_t l = t - 2; dst0 = a.fbase + a.foff[l] + 4; per_q = 4 + 8 * len; }
        // (8-byte accesses, four lanes per node, wherever both sides are 8-byte aligned -- every layout with an even
        // number of columns and queries: all BASELINE configs; word accesses, eight lanes per node, otherwise)
        const bool wide = ((((uintptr_t)nodes) | ((uintptr_t)(rec + dst0))) & 7) == 0 && (per_q & 1) == 0;
        if (wide) {
            // kGather nodes per lane in flight: a lone load -> store chain per iteration is bound by memory latency
            const uint32_t sub2 = tid & 3, total = Q * len;
            for (uint32_t i0 = tid >> 2; i0 < total; i0 += 64 * kGather) {
                uint2 v[kGather];
                uint32_t dst[kGather];
#pragma unroll
                for (uint32_t u = 0; u < kGather; u++) {
                    const uint32_t i = i0 + 64 * u;
                    const uint32_t ii = i < total ? i : i0;  // (the tail repeats its first node: same bytes to the same place)
                    const uint32_t q = ii / len, lvl = ii - q * len;
                    const uint32_t src = (uint32_t)s_base[t][s_lead[q][shift + lvl]] + lvl;
                    dst[u] = dst0 + q * per_q + 8 * lvl;
                    v[u] = reinterpret_cast<const uint2 *>(nodes + 8 * (uint64_t)src)[sub2];
                }
#pragma unroll
                for (uint32_t u = 0; u < kGather; u++) reinterpret_cast<uint2 *>(rec + dst[u])[sub2] = v[u];
            }
        } else {
            for (uint32_t i = grp; i < Q * len; i += 32) {
                const uint32_t q = i / len, lvl = i - q * len;
                const uint32_t src = (uint32_t)s_base[t][s_lead[q][shift + lvl]] + lvl;
                rec[dst0 + q * per_q + 8 * lvl + sub] = nodes[8 * (uint64_t)src + sub];
            }
        }
    }
    if (tid == 0) a.outcome[p] = 0;
}

// status[i] = outcome[i] wherever the expansion refused a record (stage-0 code, smaller than every assert's)
__global__ void stwo_shared_outcome_kernel(uint32_t n, const uint32_t *__restrict__ outcome, uint32_t *__restrict__ status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && outcome[i]) status[i] = outcome[i];
}

static SharedArgs shared_args(const ss_stwo_cfg *c, size_t n)
{
    SharedArgs a{};
    a.m = shared_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    a.record_words = (uint32_t)stwo_record_words(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    a.qstride = c->n_cols + kCp + 16 * c->lde_log;
    a.fbase = a.m.head + c->n_queries * a.qstride;
    uint32_t o = 0;
    for (uint32_t l = 0; l <= c->n_layers; l++) { a.foff[l] = o; o += c->n_queries * (4 + 8 * (c->lde_log - 1 - l)); }
    a.tbase = a.fbase + o;
    a.n = (uint32_t)n;
    return a;
}

int shared_expand_launch(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *shared_dev, const uint64_t *offs_dev,
                         uint64_t capacity_stride, uint32_t *records_dev, uint32_t *outcome_dev, hipStream_t s,
                         const uint8_t *only_fmt, const uint32_t *hint_pos, uint32_t hint_stride)
{
    SharedArgs a = shared_args(c, n);
    a.capacity = offs_dev ? 0 : 1;
    a.stride = capacity_stride;
    a.offs = offs_dev;
    a.shared = shared_dev;
    a.records = records_dev;
    a.outcome = outcome_dev;
    a.refuse = only_fmt ? 1u : SS_STATUS_MALFORMED;
    a.only_fmt = only_fmt;
    a.hint_pos = hint_pos;
    a.hint_stride = hint_stride;
    Timer t(ctx, s);
    t.begin();
    hipLaunchKernelGGL(stwo_shared_expand_kernel, dim3((unsigned)n), dim3(256), 0, s, a);
    t.end("stwo_shared_expand");
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

}  // namespace ss

using namespace ss;

extern "C" int ss_stwo_expand_shared_dev(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *shared_dev,
                                         const uint64_t *offs_dev, uint32_t *records_dev, uint32_t *outcome_dev, void *stream)
{
    if (!ctx) return set_err(SS_ERR_ARG, "ctx is null");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n || !shared_dev || !offs_dev || !records_dev || !outcome_dev) return set_err(SS_ERR_ARG, "null/empty argument");
    if (n > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    SS_DEVICE_GUARD(ctx);
    return shared_expand_launch(ctx, c, n, shared_dev, offs_dev, 0, records_dev, outcome_dev, (hipStream_t)stream);
}

// Host shared records -> verdicts.  The twin of ss_stwo_verify_records with 9-21 % fewer bytes on the link: the
// variable-length records go back to back into pinned staging (streaming stores), each chunk is uploaded with its
// offset table behind it, expanded, re-tiled and verified on the second stream while the next chunk uploads.
static int verify_shared_records_impl(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *const *shared,
                                             const size_t *words, uint32_t *status_host)
{
    if (!ctx || !status_host || !shared || !words) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n * (size_t)c->n_queries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    const size_t W = ss_stwo_record_words(c), max_words = ss_stwo_shared_max_words(c);
    for (size_t i = 0; i < n; i++)
        if (!shared[i]) return set_err(SS_ERR_ARG, "record %zu is null", i);
    std::lock_guard<std::mutex> lock(ctx->mu);  // the context's scratch: one such call at a time
    SS_DEVICE_GUARD(ctx);
    // A record longer than any shared record of this config can be is malformed whatever it holds: only its
    // fixed words travel (the kernel then sees a size that cannot match and refuses it).
    const size_t fixed = ss_stwo_shared_fixed_words(c);
    auto sent = [&](size_t i) { return words[i] > max_words ? std::min(words[i], fixed) : words[i]; };
    // chunks of <= 64 MiB; the first ones small and doubling (nothing overlaps the staging of the first)
    const size_t budget = (64u << 20) / 4;
    std::vector<size_t> first;  // first record of every chunk, then n
    {
        size_t lo = 0, step_words = std::max<size_t>(budget / 16, max_words);
        while (lo < n) {
            first.push_back(lo);
            size_t w = 0, hi = lo;
            // (and at most 256 MiB of expanded records per chunk, however short -- e.g. malformed -- the inputs are)
            while (hi < n && (hi == lo || (w + sent(hi) <= step_words && (hi - lo + 1) * W * 4 <= ((size_t)256 << 20)))) w += sent(hi++);
            lo = hi;
            step_words = std::min(budget, step_words * 2);
        }
        first.push_back(n);
    }
    size_t chunk_words = 0, chunk_n = 0, bwords = 0, wsb = 0;
    for (size_t k = 0; k + 1 < first.size(); k++) {
        size_t w = 0;
        for (size_t i = first[k]; i < first[k + 1]; i++) w += sent(i);
        const size_t cnt = first[k + 1] - first[k];
        chunk_words = std::max(chunk_words, w + 2 * (cnt + 1) + 2);  // + the u64 offset table
        chunk_n = std::max(chunk_n, cnt);
        bwords = std::max(bwords, ss_stwo_batch_words(c, cnt));
        wsb = std::max(wsb, ss_stwo_workspace_bytes(c, cnt));
    }
    HostPath &hp = ctx->hp;
    int rc;
    if ((rc = hp_reserve(ctx, 0, chunk_n * W * 4))) return rc;
    if ((rc = hp_reserve(ctx, 1, bwords * 4))) return rc;
    if ((rc = hp_reserve(ctx, 2, wsb))) return rc;
    if ((rc = hp_reserve(ctx, 3, n * 4))) return rc;
    if ((rc = hp_reserve(ctx, 4, chunk_words * 4))) return rc;
    if ((rc = hp_reserve(ctx, 5, chunk_words * 4))) return rc;
    if ((rc = hp_reserve(ctx, 6, n * 4))) return rc;
    if ((rc = hp_pinned(ctx, chunk_words * 4))) return rc;
    for (int i = 0; i < 2; i++)
        if (!hp.shared_free[i]) HIP_TRY(hipEventCreateWithFlags(&hp.shared_free[i], hipEventDisableTiming));
    hipStream_t s = hp.stream, vs = hp.vstream;
    uint32_t *rec_dev = (uint32_t *)hp.dev[0], *status_dev = (uint32_t *)hp.dev[3], *outcome_dev = (uint32_t *)hp.dev[6];
    auto run = [&]() -> int {
        int buf = 0;
        for (size_t k = 0; k + 1 < first.size(); k++) {
            const size_t lo = first[k], cnt = first[k + 1] - lo;
            HIP_TRY(hipEventSynchronize(hp.pinned_free[buf]));  // previous upload from this staging buffer done
            uint32_t *stage = (uint32_t *)hp.pinned[buf];
            // layout of a chunk: offs[cnt + 1] (u64, in words from the chunk start) | records back to back
            uint64_t *offs = (uint64_t *)stage;
            uint64_t o = 2 * (cnt + 1);
            o += o & 1;
            for (size_t i = 0; i < cnt; i++) { offs[i] = o; o += sent(lo + i); }
            offs[cnt] = o;
            parallel_for(cnt, [&](size_t i) { copy_streaming(stage + offs[i], shared[lo + i], sent(lo + i) * 4); },
                         std::max<size_t>(1, std::min<size_t>(stage_threads(), o * 4 / (1u << 20))));
            for (size_t i = 0; i < cnt; i++)
                if (words[lo + i] > max_words)  // only the fixed words were sent: an impossible count makes the kernel refuse them
                    stage[offs[i] + shared_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers).cnt] = 0xffffffffu;
            uint32_t *sh_dev = (uint32_t *)hp.dev[4 + buf];
            HIP_TRY(hipStreamWaitEvent(s, hp.shared_free[buf], 0));  // the expansion that last read this device buffer
            HIP_TRY(hipMemcpyAsync(sh_dev, stage, o * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(hp.pinned_free[buf], s));
            HIP_TRY(hipStreamWaitEvent(vs, hp.pinned_free[buf], 0));
            int r = shared_expand_launch(ctx, c, cnt, sh_dev, (const uint64_t *)sh_dev, 0, rec_dev, outcome_dev + lo, vs);
            if (r) return r;
            HIP_TRY(hipEventRecord(hp.shared_free[buf], vs));
            r = ss_stwo_pack_dev(ctx, c, cnt, rec_dev, (uint32_t *)hp.dev[1], vs);
            if (r) return r;
            r = ss_stwo_verify_batch_dev(ctx, c, cnt, (const uint32_t *)hp.dev[1], hp.dev[2], wsb, status_dev + lo, nullptr, vs);
            if (r) return r;
            buf ^= 1;
        }
        hipLaunchKernelGGL(stwo_shared_outcome_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vs, (uint32_t)n,
                           outcome_dev, status_dev);
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

extern "C" int ss_stwo_verify_shared_records(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *const *shared,
                                             const size_t *words, uint32_t *status_host)
{
    try {  // (std::vector / std::function allocations: a std::bad_alloc must not cross the C ABI -- ADVICE r4)
        return verify_shared_records_impl(ctx, c, n, shared, words, status_host);
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}
