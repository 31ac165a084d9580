// Minimal records on the GPU ("sorted multi-proof Merkle (real stwo format)"; ss_minimal.h).
//
// Anchors: stwo-verifier/src/fri/queries.simf:41 (the reference does not sort or deduplicate its queries),
// stwo-verifier/scripts/generate_wit.py:36-42 (its adapter cuts the prover's lists into one path per query),
// stwo-verifier/src/merkle.simf:22-44 (the per-path fold this replaces).  A minimal record holds what upstream
// stwo's prover sends: values once per distinct queried position, and only the siblings / fold-pair evaluations
// that no other query's chain produces.  PARITY UNPINNED (no bytes of the form in the reference); pinned instead to
// the per-query path through R(M) (the test checker's so_stwo_minimal_expand).
//
// No expansion pass that hashes, no hint from the prover: the transcript kernel draws the verifier's own queries,
// stwo_min_expand_kernel (one block per proof) turns them into the plan of ss_minimal.h and GATHERS the record's
// lists straight into the batch layout of the per-query kernels -- values by position rank, witnesses by their rank
// in the level, holes (zeros) where the record omits a sibling -- and leaves, per chain, the lane that will produce
// each omitted sibling.  The query kernel takes a hole's fold partner from that lane, the merkle kernel a hole's
// sibling (the chains of a proof run the same tree in lockstep in one wavefront), the top kernel from the node its
// sibling's leader has just stored (ss_stwo.hip, `lay.minimal`).  A tree whose lists do not have the lengths the
// queries imply gets path length 0 for all its chains: the code of `path == 1` (merkle.simf:42), as in R(M).
//
// HBM-bound word shuffling (the gather) + the integer-ALU-bound kernels of ss_stwo.hip; no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "ss_abi.h"
#include "ss_copy.h"
#include "ss_ctx.h"
#include "ss_kernels.h"
#include "ss_layout.h"
#include "ss_minimal.h"

namespace ss {

struct MinArgs {
    MinMap m;
    const uint32_t *recs;  // minimal records, record p at word offset offs[p], offs[p + 1] - offs[p] words
    const uint64_t *offs;  // nullptr: CAPACITY form -- record p at p * stride, every list at the base it has when all lists
                           // have their largest length (what the GPU reader of the minimal proof.json writes: ss_text.h)
    uint64_t stride;
    uint32_t n;
};

// head[w][proof] of the batch from the records' first words, so that the transcript kernel runs as for any batch
__global__ void stwo_min_head_kernel(StwoLayout y, MinArgs a, uint32_t *__restrict__ batch)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= (uint64_t)y.head_words * y.np) return;
    const uint32_t w = (uint32_t)(d / y.np), p = (uint32_t)(d - (uint64_t)w * y.np);
    uint32_t v = 0;
    if (p < a.n) {
        const uint64_t off = a.offs ? a.offs[p] : p * a.stride;
        if (!a.offs || a.offs[p + 1] - off >= a.m.data) v = a.recs[off + w];  // (shorter than the fixed words: nothing of it is read)
    }
    batch[y.off_head + d] = v;
}

__global__ void __launch_bounds__(256)
stwo_min_expand_kernel(StwoLayout y, MinArgs a, uint32_t *__restrict__ batch, uint32_t *__restrict__ ws,
                       uint32_t *__restrict__ status)
{
    constexpr uint32_t NT = kMaxList + 3;
    __shared__ uint8_t s_sib[kMaxQueries][32], s_widx[kMaxQueries][32], s_vidx[kMaxQueries];
    __shared__ uint16_t s_cum[33];
    __shared__ uint32_t s_nodes0;
    __shared__ uint32_t s_tv, s_cv, s_fw[kMaxList + 1], s_hw[NT];  // word offsets of the lists inside the record
    __shared__ uint32_t s_ok[NT];                                  // the tree's lists have the lengths the queries imply
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    if (p >= a.n) return;
    const MinMap &m = a.m;
    const uint32_t N = y.N, L = y.L, Q = y.Q, Qd = y.Qd, K = y.K, np = y.np, nip = y.nip;
    const bool capacity = a.offs == nullptr;
    const uint64_t off = capacity ? p * a.stride : a.offs[p];
    const uint64_t words = capacity ? a.stride : a.offs[p + 1] - off;
    const uint32_t *rec = a.recs + off;
    const uint32_t inst0 = p * Q;

    // ---- the plan (wave 0; lane = query): ss_minimal.h's closed form on the verifier's own queries
    if (tid < 64) {
        const uint32_t q = tid;
        uint32_t *qw = ws + y.ws_ctx + (size_t)y.c_queries * np + p;
        const uint32_t pos = q < Qd ? qw[(size_t)q * np] : qw[0];
        if (q >= Qd && q < Q) qw[(size_t)q * np] = pos;  // chains Qd .. Q-1 repeat query 0 (ss_layout.h)
        const bool in = q < Q;
        if (q == 0) s_cum[0] = 0;
        uint32_t cum = 0;
        for (uint32_t lv = 0; lv < 32; lv++) {
            if (lv >= L) { if (in) { s_sib[q][lv] = kMinNone; s_widx[q][lv] = 0; } continue; }
            const uint32_t x = pos >> lv;
            bool rep = in;
            uint32_t sb = kMinNone;
            for (uint32_t e = 0; e < Q; e++) {
                const uint32_t xe = (uint32_t)__shfl((int)x, (int)e);
                if (e < q && xe == x) rep = false;
                if (sb == kMinNone && xe == (x ^ 1)) sb = e;
            }
            const uint64_t lone = __ballot(rep && sb == kMinNone), reps = __ballot(rep);
            uint32_t w = 0, v = 0;
            for (uint32_t e = 0; e < Q; e++) {
                const uint32_t xe = (uint32_t)__shfl((int)x, (int)e);
                w += ((lone >> e) & 1) && xe < x;
                v += ((reps >> e) & 1) && xe < x;
            }
            if (in) { s_sib[q][lv] = (uint8_t)sb; s_widx[q][lv] = (uint8_t)w; }
            if (lv == 0) {
                if (in) s_vidx[q] = (uint8_t)v;
                if (q == 0) s_nodes0 = (uint32_t)__popcll(reps);
            }
            cum += (uint32_t)__popcll(lone);
            if (q == 0) s_cum[lv + 1] = (uint16_t)cum;
        }
        if (q == 0)
            for (uint32_t lv = L; lv < 32; lv++) s_cum[lv + 1] = (uint16_t)cum;
    }
    __syncthreads();
    // ---- the record's own list lengths: where its lists start, whether it is a minimal record of this config at all,
    // and which trees have the lengths the queries imply
    if (tid == 0) {
        bool malformed = words < m.data;
        uint64_t o = m.data;
        if (!malformed) {
            const uint32_t Qr = m.Q;  // the config's query count bounds every list
            malformed |= rec[m.nv] > Qr || rec[m.nv + 1] > Qr;
            s_tv = (uint32_t)o; o += (uint64_t)(capacity ? Qr : malformed ? 0 : rec[m.nv]) * N;
            s_cv = (uint32_t)o; o += (uint64_t)(capacity ? Qr : malformed ? 0 : rec[m.nv + 1]) * kCp;
            for (uint32_t l = 0; l <= K && !malformed; l++) {
                malformed |= rec[m.nfw + l] > Qr;
                s_fw[l] = (uint32_t)o;
                o += 4ull * (capacity ? Qr : malformed ? 0 : rec[m.nfw + l]);
            }
            for (uint32_t t = 0; t < K + 3 && !malformed; t++) {
                malformed |= rec[m.nhw + t] > Qr * min_tree_len(L, t);
                s_hw[t] = (uint32_t)o;
                o += 8ull * (capacity ? Qr * min_tree_len(L, t) : malformed ? 0 : rec[m.nhw + t]);
            }
            malformed |= !capacity && o != words;
        }
        for (uint32_t t = 0; t < K + 3; t++) {
            bool ok = !malformed;
            if (ok) {
                const uint32_t sh = min_tree_shift(t);
                ok = rec[m.nhw + t] == (uint32_t)(s_cum[L] - s_cum[sh]);
                if (t < 2) ok &= rec[m.nv + t] == s_nodes0;
                else ok &= rec[m.nfw + t - 2] == (uint32_t)(s_cum[t - 1] - s_cum[t - 2]);
            }
            s_ok[t] = ok;
        }
        if (malformed) atomicMin(&status[p], (uint32_t)SS_STATUS_MALFORMED);
    }
    __syncthreads();

    // ---- the gather; destination-major (query fastest), so that the stores of a wave are contiguous runs
    for (uint32_t i = tid; i < Q * N; i += 256) {
        const uint32_t k = i / Q, q = i - k * Q;
        batch[y.off_trace_vals + (size_t)k * nip + inst0 + q] = s_ok[0] ? rec[s_tv + (uint32_t)s_vidx[q] * N + k] : 0;
    }
    for (uint32_t i = tid; i < Q * kCp; i += 256) {
        const uint32_t k = i / Q, q = i - k * Q;
        batch[y.off_cp_vals + (size_t)k * nip + inst0 + q] = s_ok[1] ? rec[s_cv + (uint32_t)s_vidx[q] * kCp + k] : 0;
    }
    for (uint32_t i = tid; i < (K + 1) * 4 * Q; i += 256) {
        const uint32_t row = i / Q, q = i - row * Q, l = row >> 2, w = row & 3;
        const bool have = s_ok[2 + l] && s_sib[q][l] == kMinNone;  // else: the partner is another chain's value (query kernel)
        batch[y.off_fri_wit + (size_t)row * nip + inst0 + q] = have ? rec[s_fw[l] + 4 * (uint32_t)s_widx[q][l] + w] : 0;
    }
    for (uint32_t i = tid; i < (K + 3) * Q; i += 256) {
        const uint32_t t = i / Q, q = i - t * Q;
        batch[y.off_plen + (size_t)t * nip + inst0 + q] = s_ok[t] ? min_tree_len(L, t) : 0;
    }
    for (uint32_t i = tid; i < Q * 8; i += 256) {
        const uint32_t q = i >> 3, w = i & 7;
        ws[y.ws_sib + (size_t)(inst0 + q) * 8 + w] = (uint32_t)s_sib[q][4 * w] | (uint32_t)s_sib[q][4 * w + 1] << 8 |
                                                    (uint32_t)s_sib[q][4 * w + 2] << 16 | (uint32_t)s_sib[q][4 * w + 3] << 24;
    }
    for (uint32_t t = 0; t < K + 3; t++) {
        const uint32_t len = min_tree_len(L, t), sh = min_tree_shift(t);
        const uint32_t top = y.T < len ? y.T : len, low = len - top;
        const uint64_t base = t == 0 ? y.off_trace_path : t == 1 ? y.off_cp_path : y.off_fri_path[t - 2];
        const bool ok = s_ok[t];
        for (uint32_t i = tid; i < len * 2 * Q; i += 256) {  // one 16-byte half of a sibling per thread
            const uint32_t q = i % Q, r = i / Q, half = r & 1, lvl = r >> 1, lv = sh + lvl;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok && s_sib[q][lv] == kMinNone) {
                const uint32_t *src = rec + s_hw[t] + 8 * ((uint32_t)(s_cum[lv] - s_cum[sh]) + s_widx[q][lv]) + 4 * half;
                v = make_uint4(src[0], src[1], src[2], src[3]);
            }
            const uint64_t dst = lvl < low ? tile_word(base, low, inst0 + q, lvl, half * 4)
                                           : y.off_top + (uint64_t)p * y.top_words + y.top_off[t] + ((uint64_t)(lvl - low) * Q + q) * 8 + half * 4;
            *reinterpret_cast<uint4 *>(batch + dst) = v;
        }
    }
}

static MinArgs min_args(const ss_stwo_cfg *c, size_t n, const uint32_t *recs, const uint64_t *offs)
{
    MinArgs a{};
    a.m = min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers);
    a.recs = recs;
    a.offs = offs;
    a.stride = min_max_words(a.m);
    a.n = (uint32_t)n;
    return a;
}

// the HEAD half in front of minimal records: head words -> transcript -> plan + gather -> query kernel
int stwo_minimal_head(ss_ctx *ctx, const ss_stwo_cfg *c, const StwoLayout &y, const uint32_t *recs_dev, const uint64_t *offs_dev,
                      uint32_t *batch, uint32_t *ws, uint32_t *status, hipStream_t s)
{
    const MinArgs a = min_args(c, y.n, recs_dev, offs_dev);
    Timer t(ctx, s);
    t.begin();
    const uint64_t hw = (uint64_t)y.head_words * y.np;
    hipLaunchKernelGGL(stwo_min_head_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, y, a, batch);
    t.end("stwo_min_head");
    t.begin();
    hipLaunchKernelGGL(c->hash == SS_HASH_BLAKE2S ? stwo_transcript_kernel_b2s : stwo_transcript_kernel_sha,
                       dim3((y.n + 63) / 64), dim3(64), 0, s, y, (const uint32_t *)batch, ws, status, (uint32_t *)nullptr, 0u);  // (reset by the caller: stwo_min_head may have written verdicts)
    t.end("stwo_transcript");
    t.begin();
    hipLaunchKernelGGL(stwo_min_expand_kernel, dim3(y.n), dim3(256), 0, s, y, a, batch, ws, status);
    t.end("stwo_min_expand");
    t.begin();
    hipLaunchKernelGGL(stwo_query_kernel, dim3((y.ni + 63) / 64), dim3(64), 2 * (y.K + 3) * 64 * 4, s, y, (const uint32_t *)batch,
                       ws, status);
    t.end("stwo_query");
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

}  // namespace ss

using namespace ss;

extern "C" size_t ss_stwo_minimal_batch_words(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n, true).total_words : 0;
}

extern "C" size_t ss_stwo_minimal_workspace_bytes(const ss_stwo_cfg *c, size_t n)
{
    return cfg_ok(c) && n ? (size_t)lay_of(c, n, true).ws_total_words * 4 : 0;
}

// Host minimal records -> verdicts: the twin of ss_stwo_verify_shared_records.  The variable-length records go back to
// back into pinned staging with their offset table in front; each chunk is uploaded and verified on the second stream
// while the next one is staged and uploaded.
extern "C" int ss_stwo_verify_minimal_records(ss_ctx *ctx, const ss_stwo_cfg *c, size_t n, const uint32_t *const *recs,
                                              const size_t *words, uint32_t *status_host)
{
    if (!ctx || !status_host || !recs || !words) return set_err(SS_ERR_ARG, "null argument");
    if (!cfg_ok(c)) return set_err(SS_ERR_ARG, "unsupported stwo config");
    if (!n) return set_err(SS_ERR_ARG, "empty batch");
    if (n * (size_t)kMaxQueries > 0x7fffffffu) return set_err(SS_ERR_ARG, "batch too large");
    for (size_t i = 0; i < n; i++)
        if (!recs[i]) return set_err(SS_ERR_ARG, "record %zu is null", i);
    try {
        std::lock_guard<std::mutex> lock(ctx->mu);  // the context's scratch: one such call at a time
        SS_DEVICE_GUARD(ctx);
        const size_t max_words = ss_stwo_minimal_max_words(c), fixed = ss_stwo_minimal_fixed_words(c);
        // a record longer than any minimal record of the config is malformed whatever it holds: only its fixed words
        // travel, and the kernel refuses a record whose size is not the one its counts give
        auto sent = [&](size_t i) { return words[i] > max_words ? std::min(words[i], fixed) : words[i]; };
        const size_t budget = (64u << 20) / 4;  // (staged: equal chunks, so that staging chunk k + 1 takes as long as uploading chunk k)
        std::vector<size_t> first;
        {
            size_t lo = 0, step_words = std::max<size_t>(budget / 16, max_words);
            while (lo < n) {
                first.push_back(lo);
                size_t w = 0, hi = lo;
                while (hi < n && (hi == lo || (w + sent(hi) <= step_words && hi - lo < 16384))) w += sent(hi++);
                lo = hi;
                step_words = std::min(budget, step_words * 2);
            }
            first.push_back(n);
        }
        size_t chunk_words = 0, bwords = 0, wsb = 0;
        for (size_t k = 0; k + 1 < first.size(); k++) {
            size_t w = 0;
            for (size_t i = first[k]; i < first[k + 1]; i++) w += sent(i);
            const size_t cnt = first[k + 1] - first[k];
            chunk_words = std::max(chunk_words, w + 2 * (cnt + 1) + 2);
            bwords = std::max(bwords, ss_stwo_minimal_batch_words(c, cnt));
            wsb = std::max(wsb, ss_stwo_minimal_workspace_bytes(c, cnt));
        }
        HostPath &hp = ctx->hp;
        int rc;
        if ((rc = hp_reserve(ctx, 1, bwords * 4))) return rc;
        if ((rc = hp_reserve(ctx, 2, wsb))) return rc;
        if ((rc = hp_reserve(ctx, 3, n * 4))) return rc;
        if ((rc = hp_reserve(ctx, 4, chunk_words * 4))) return rc;
        if ((rc = hp_reserve(ctx, 5, chunk_words * 4))) return rc;
        if ((rc = hp_pinned(ctx, chunk_words * 4))) return rc;
        for (int i = 0; i < 2; i++)
            if (!hp.shared_free[i]) HIP_TRY(hipEventCreateWithFlags(&hp.shared_free[i], hipEventDisableTiming));
        hipStream_t s = hp.stream, vs = hp.vstream;
        uint32_t *status_dev = (uint32_t *)hp.dev[3];
        auto run = [&]() -> int {
            int buf = 0;
            for (size_t k = 0; k + 1 < first.size(); k++) {
                const size_t lo = first[k], cnt = first[k + 1] - lo;
                HIP_TRY(hipEventSynchronize(hp.pinned_free[buf]));
                uint32_t *stage = (uint32_t *)hp.pinned[buf];
                uint64_t *offs = (uint64_t *)stage;  // offs[cnt + 1] (u64, words from the chunk start) | records
                uint64_t o = 2 * (cnt + 1);
                o += o & 1;
                for (size_t i = 0; i < cnt; i++) { offs[i] = o; o += sent(lo + i); }
                offs[cnt] = o;
                parallel_for(cnt, [&](size_t i) { copy_streaming(stage + offs[i], recs[lo + i], sent(lo + i) * 4); },
                             std::max<size_t>(1, std::min<size_t>(stage_threads(), o * 4 / (1u << 20))));
                for (size_t i = 0; i < cnt; i++)
                    if (words[lo + i] > max_words && sent(lo + i) >= fixed)  // an impossible count: the kernel refuses the record
                        stage[offs[i] + min_map(c->n_cols, c->lde_log, c->n_queries, c->n_layers).nv] = 0xffffffffu;
                uint32_t *dev = (uint32_t *)hp.dev[4 + buf];
                HIP_TRY(hipStreamWaitEvent(s, hp.shared_free[buf], 0));  // the verification that last read this device buffer
                HIP_TRY(hipMemcpyAsync(dev, stage, o * 4, hipMemcpyHostToDevice, s));
                HIP_TRY(hipEventRecord(hp.pinned_free[buf], s));
                HIP_TRY(hipStreamWaitEvent(vs, hp.pinned_free[buf], 0));
                const int r = ss_stwo_verify_minimal_dev(ctx, c, cnt, dev, (const uint64_t *)dev, (uint32_t *)hp.dev[1], hp.dev[2], wsb,
                                                         status_dev + lo, nullptr, SS_PHASE_ALL, vs);
                if (r) return r;
                HIP_TRY(hipEventRecord(hp.shared_free[buf], vs));
                buf ^= 1;
            }
            HIP_TRY(hipMemcpyAsync(status_host, status_dev, n * 4, hipMemcpyDeviceToHost, vs));
            return SS_OK;
        };
        rc = run();
        const hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(vs);
        if (rc) return rc;
        HIP_TRY(e1);
        HIP_TRY(e2);
        return SS_OK;
    } catch (const std::exception &) {
        return set_err(SS_ERR_NOMEM, "out of host memory");
    }
}
