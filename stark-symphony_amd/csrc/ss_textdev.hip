// Text -> record on the GPU: the fast path of ss_text.h for gfx950.
//
// Replaces, for canonical texts, the host tree parser in front of the verifier (the reference's callers hand
// over text: stwo-verifier/scripts/generate_wit.py:106-245,218-243; simfony-cli/src/main.rs:163-209).
//
// The unit of work is a 1 KiB WINDOW of a text, one wavefront per window, 16 bytes per lane -- a chunk of
// 64 MiB of text is 65 536 independent waves whatever the number of texts.  The tokenizer of ss_text.h is
// sequential only through three small states (inside a JSON string? inside which kind of alnum run? how many
// skeleton bytes / numbers so far?), and each composes associatively, over lanes and over windows alike:
//   string state   parity of the quotes before (a backslash anywhere takes the text off the fast path, so
//                  quotes are never escaped): ballot + popcount;
//   run state      a lane / window that holds any non-alnum byte decides its outgoing state alone; an
//                  all-alnum one passes on what enters it.  The state entering position i is the outgoing state
//                  of the nearest decided position below (ballot + count-leading-zeros + one shuffle) or, if
//                  that is "no run", the class of the first byte behind it;
//   positions      prefix sums of the skeleton bytes / numbers each position contributes -- which depend on
//                  the entering states only through the window's leading alnum run and through whether its
//                  whitespace sits inside a string, so a window can count both cases without knowing them.
// Three kernels per chunk (plus, for the minimal proof.json, text_landmark_kernel and text_minhint_kernel between the scan
// and the place pass: its list lengths are found in the text, ss_text.h):
//   text_summary_kernel  wave = window: WinSum (counts for either string state, leading run, flags);
//   text_scan_kernel     wave = text:   scans its windows' summaries -> WinIn (entering states and positions),
//                        checks the totals against the template (skeleton length, number count, string closed);
//   text_place_kernel    wave = window: replays its bytes with the true states (scan_byte of ss_text.h),
//                        compares what they contribute to the skeleton with the template's (staged in LDS) and
//                        converts the numbers that start in the window into the record (digits read from the
//                        LDS copy of this window and the next).
// Any mismatch, non-canonical number or out-of-range value sets the text's outcome to 1: it goes to the host
// reader (ss_ingest.cpp), which alone produces parsed / other-config / malformed.  HBM-bound byte work: no
// MFMA; coalesced 16-byte loads; the text is read twice (1 KiB windows stay in L2 / MALL between the passes);
// scattered 1..32-byte stores into each text's own record.
#include <hip/hip_runtime.h>

#include "ss_text.h"
#include "ss_textdev.h"
#include "ss_shared.h"

namespace ss {

namespace {

__device__ __forceinline__ uint32_t byte_of(const uint4 &v, uint32_t j)
{
    const uint32_t w = j < 8 ? (j < 4 ? v.x : v.y) : (j < 12 ? v.z : v.w);
    return (w >> (8 * (j & 3))) & 255;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (uint32_t d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// what a lane's bytes say on their own
struct LaneLocal {
    bool bad, det;              // a byte that ends the fast path; a non-alnum byte (the lane decides its outgoing run state)
    uint32_t qpar, rout, ftype; // quote parity; run state behind the lane (if det); class of the first byte (kRunNone: not alnum)
};

__device__ __forceinline__ LaneLocal lane_local(const uint4 &cur, uint32_t nvalid)
{
    LaneLocal L{false, false, 0, kRunNone, kRunNone};
    bool in_alnum = false;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t c = byte_of(cur, j);
        if (j < nvalid) {
            L.bad |= txt_is_bad(c);
            const bool an = txt_is_alnum(c);
            const uint32_t cls = txt_is_digit(c) ? kRunToken : kRunIdent;
            if (j == 0 && an) L.ftype = cls;
            if (an) { if (!in_alnum) L.rout = cls; }
            else { L.det = true; L.rout = kRunNone; }
            in_alnum = an;
            L.qpar ^= c == '"';
        }
    }
    if (!L.det) L.rout = kRunNone;  // an all-alnum lane has no state of its own
    return L;
}

// Run state entering position `lane` of 64 positions (lanes of a window, or windows of a text) given every
// position's (decided?, outgoing state, class of first byte) and the state entering position 0.
// All lanes must call it (shuffles).
__device__ __forceinline__ uint32_t entering_run(uint64_t det_mask, uint32_t rout, uint32_t ftype, uint32_t carry,
                                                 uint32_t lane)
{
    const uint64_t lower = det_mask & ((1ull << lane) - 1);
    const int j = lower ? 63 - __builtin_clzll(lower) : -1;  // nearest decided position below
    const uint32_t base = __shfl(rout, j < 0 ? 0 : j);
    const uint32_t first_after = __shfl(ftype, j + 1 > 63 ? 63 : j + 1);
    const uint32_t from = j < 0 ? carry : base;
    // "no run" behind position j: the run, if any, starts with the first byte of position j + 1
    return from != kRunNone ? from : ((uint32_t)(j + 1) < lane ? first_after : kRunNone);
}
// ... and the state behind position 63
__device__ __forceinline__ uint32_t leaving_run(uint64_t det_mask, uint32_t rout, uint32_t ftype, uint32_t carry)
{
    const int j = det_mask ? 63 - __builtin_clzll(det_mask) : -1;
    const uint32_t base = __shfl(rout, j < 0 ? 0 : j);
    const uint32_t first_after = __shfl(ftype, j + 1 > 63 ? 63 : j + 1);
    const uint32_t from = j < 0 ? carry : base;
    return from != kRunNone ? from : (j + 1 < 64 ? first_after : kRunNone);
}

// skeleton bytes / numbers the leading alnum run of a window (or lane) contributes, by the run state entering it
__device__ __forceinline__ void lead_counts(uint32_t run_in, uint32_t ftype, uint32_t lead, uint32_t &emit, uint32_t &tok)
{
    emit = tok = 0;
    if (!lead) return;
    if (run_in == kRunIdent) emit = lead;
    else if (run_in == kRunNone) {
        if (ftype == kRunToken) { emit = 1; tok = 1; }  // one marker
        else emit = lead;
    }
}

}  // namespace

// window -> text
__global__ void __launch_bounds__(64) text_index_kernel(TextParseArgs a)
{
    const uint32_t t = blockIdx.x;
    if (t >= a.n) return;
    for (uint32_t w = a.win_base[t] + threadIdx.x; w < a.win_base[t + 1]; w += 64) a.win_text[w] = t;
    if (a.mhints && a.fmt[t] == kTextMinimal && threadIdx.x < 3) a.mhints[t].n_lm[threadIdx.x] = 0;
}

// Format 2 only: the positions a shared-path text names (read backwards from its last KiB by one lane: <= 64 short
// numbers), the distinct siblings per tree they imply (the closed form of ss_shared.h: lane = query) and the gaps those
// counts cut out of the full-length template.  A text whose tail is not `[p0, .., pQ-1] }` gets totals that nothing matches.
__global__ void __launch_bounds__(64) text_hint_kernel(TextParseArgs a)
{
    __shared__ uint4 s_tail4[68];
    __shared__ uint32_t s_pos[64], s_cnt[kMaxTrees], s_ok;
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.n || a.fmt[t] != kTextShared) return;
    const SharedTextInfo &I = a.sinfo;
    const uint32_t len = a.lens[t];
    const uint32_t base = len > 1024 ? (len - 1024) & ~15u : 0;
    const uint4 *text4 = reinterpret_cast<const uint4 *>(a.texts + a.offs[t] + base);  // (kTextSlack readable behind the chunk)
    s_tail4[lane] = text4[lane];
    if (lane < 4) s_tail4[64 + lane] = text4[64 + lane];
    __syncthreads();
    if (lane == 0) {
        const uint32_t tail = len < 1024 ? len : 1024;
        bool ok = a.tmpl[kTextShared].skel != nullptr && shared_text_hint(reinterpret_cast<const uint8_t *>(s_tail4) + (len - base - tail), tail, I.Q, s_pos);
        for (uint32_t q = 0; q < I.Q && ok; q++) ok = (s_pos[q] >> I.L) == 0;
        s_ok = ok;
    }
    __syncthreads();
    TextHint *h = a.hints + t;
    if (!s_ok) {
        if (lane == 0) { h->g.skel_len = 0xffffffffu; h->g.n_slots = 0xffffffffu; }
        return;
    }
    const uint32_t q = lane, pos = q < I.Q ? s_pos[q] : 0;
    uint32_t s = 32;
    for (uint32_t e = 0; e < q && e < I.Q; e++) {
        const uint32_t x = pos ^ s_pos[e], d = x ? 32 - __clz(x) : 0;
        s = d < s ? d : s;
    }
    for (uint32_t k = 0; k < I.n_trees; k++) {
        const uint32_t c = wave_sum(q < I.Q ? shared_fresh(s, I.L, k) : 0);
        if (lane == 0) s_cnt[k] = c;
    }
    __syncthreads();
    if (lane == 0) shared_text_gaps(I, s_cnt, a.tmpl[kTextShared].skel_len, a.tmpl[kTextShared].n_slots, h->g);
    if (q < I.Q) h->pos[q] = pos;
}

__global__ void __launch_bounds__(64) text_summary_kernel(TextParseArgs a)
{
    const uint32_t gw = blockIdx.x, lane = threadIdx.x;
    if (gw >= a.n_windows) return;
    const uint32_t t = a.win_text[gw], w = gw - a.win_base[t], len = a.lens[t];
    const uint4 *text4 = reinterpret_cast<const uint4 *>(a.texts + a.offs[t]);
    const uint4 cur = text4[(size_t)w * 64 + lane];
    const uint32_t pos0 = (w << 10) + lane * 16;
    const uint32_t nvalid = pos0 >= len ? 0 : (len - pos0 < 16 ? len - pos0 : 16);

    // one pass: the leading alnum run of the lane, and behind it (where the run state is known) what the
    // bytes contribute; whitespace counted by the quote parity relative to the lane start
    bool bad = false, det = false, in_lead = true;
    uint32_t qpar = 0, run = kRunNone, ftype = kRunNone, lead = 0, fixed = 0, toks = 0, ws0 = 0, ws1 = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t c = byte_of(cur, j);
        if (j < nvalid) {
            bad |= txt_is_bad(c);
            const bool an = txt_is_alnum(c);
            const uint32_t cls = txt_is_digit(c) ? kRunToken : kRunIdent;
            if (j == 0 && an) ftype = cls;
            if (in_lead && an) { lead++; }
            else {
                in_lead = false;
                if (an) {
                    if (run == kRunNone) { run = cls; if (cls == kRunToken) { fixed++; toks++; } }
                    if (run == kRunIdent) fixed++;
                } else {
                    run = kRunNone;
                    det = true;
                    if (c == '"') { qpar ^= 1; fixed++; }
                    else if (txt_is_ws(c)) { if (qpar) ws1++; else ws0++; }
                    else fixed++;
                }
            }
        }
    }
    const uint32_t rout = det ? run : kRunNone;
    const uint64_t det_mask = __ballot(det), q_mask = __ballot(qpar);
    const uint64_t below = (1ull << lane) - 1;
    // lanes behind the window's first decided lane know the run state that enters them; the leading runs of the
    // others form the window's leading run
    const uint32_t run_in = entering_run(det_mask, rout, ftype, kRunNone, lane);
    uint32_t lead_outer = 0;
    if ((det_mask & below) == 0) lead_outer = lead;
    else {
        uint32_t e, k;
        lead_counts(run_in, ftype, lead, e, k);
        fixed += e;
        toks += k;
    }
    const uint32_t lpar = (uint32_t)__popcll(q_mask & below) & 1;  // quote parity at the lane start, from the window start
    const uint32_t ws_even = lpar ? ws1 : ws0, ws_odd = lpar ? ws0 : ws1;
    const uint32_t s_fixed = wave_sum(fixed | (toks << 16));
    const uint32_t s_ws = wave_sum(ws_even | (ws_odd << 16));
    const uint32_t s_lead = wave_sum(lead_outer);
    const uint32_t w_rout = leaving_run(det_mask, rout, ftype, kRunNone);
    const uint32_t w_ftype = __shfl(ftype, 0);
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0) {
        WinSum s;
        s.flags = (any_bad ? 1u : 0u) | (((uint32_t)__popcll(q_mask) & 1) << 1) | ((det_mask ? 1u : 0u) << 2) | (w_rout << 3) |
                  (w_ftype << 5);
        s.fixed = s_fixed;
        s.ws = s_ws;
        s.lead = s_lead;
        a.win_sum[gw] = s;
    }
}

__global__ void __launch_bounds__(64) text_scan_kernel(TextParseArgs a)
{
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.n) return;
    const uint32_t w0 = a.win_base[t], nwin = a.win_base[t + 1] - w0;
    const uint32_t f = a.fmt[t];
    const bool wit = f == kTextWit;
    const uint8_t *skel = a.tmpl[f].skel;
    // (a shared-path text has the totals its own positions imply: text_hint_kernel)
    const uint32_t skel_len = f == kTextShared ? a.hints[t].g.skel_len : a.tmpl[f].skel_len;
    const uint32_t n_slots = f == kTextShared ? a.hints[t].g.n_slots : a.tmpl[f].n_slots;
    uint32_t carry_run = kRunNone, carry_str = 0, skel_pos = 0, tok_pos = 0;
    bool bad = skel == nullptr || nwin == 0 || skel_len == 0xffffffffu || (f == kTextMinimal && !a.mhints);
    const uint64_t below = (1ull << lane) - 1;
    for (uint32_t g = 0; g < nwin && !bad; g += 64) {
        const uint32_t w = g + lane;
        WinSum s{0, 0, 0, 0};
        if (w < nwin) s = a.win_sum[w0 + w];
        const bool det = (s.flags >> 2) & 1;
        const uint32_t rout = (s.flags >> 3) & 3, ftype = (s.flags >> 5) & 3;
        const uint64_t det_mask = __ballot(det), q_mask = __ballot((s.flags >> 1) & 1);
        const uint32_t in_str = carry_str ^ ((uint32_t)__popcll(q_mask & below) & 1);
        const uint32_t run_in = entering_run(det_mask, rout, ftype, carry_run, lane);
        carry_run = leaving_run(det_mask, rout, ftype, carry_run);
        carry_str ^= (uint32_t)__popcll(q_mask) & 1;
        uint32_t e, k;
        lead_counts(run_in, ftype, s.lead, e, k);
        // whitespace is kept where the string state (entering state ^ local parity) is 1
        e += (s.fixed & 0xffff) + (in_str ? (s.ws & 0xffff) : (s.ws >> 16));
        k += s.fixed >> 16;
        uint32_t ie = e, ik = k;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t ye = __shfl_up(ie, d), yk = __shfl_up(ik, d);
            if (lane >= d) { ie += ye; ik += yk; }
        }
        if (w < nwin) {
            WinIn in;
            in.skel_pos = skel_pos + ie - e;
            in.tok_pos = tok_pos + ik - k;
            in.state = in_str | (run_in << 1);
            a.win_in[w0 + w] = in;
        }
        skel_pos += __shfl(ie, 63);
        tok_pos += __shfl(ik, 63);
        bad = __ballot(s.flags & 1) != 0 || skel_pos > skel_len || tok_pos > n_slots;
    }
    // (a minimal proof.json has the totals its list lengths imply, and those are found next: text_minhint_kernel compares;
    // here the full-length template's totals are upper bounds)
    const bool good = !bad && (f == kTextMinimal || (skel_pos == skel_len && tok_pos == n_slots)) && carry_str == 0;
    if (f == kTextMinimal && lane == 0 && a.mhints) { a.mhints[t].seen_skel = skel_pos; a.mhints[t].seen_tok = tok_pos; }
    if (good && f < kTextShared) {  // the path-length trailer of a canonical text is the config's (format 2: written by the expansion)
        const TextTemplate &T = wit ? a.tmpl[kTextWit] : a.tmpl[kTextJson];
        uint32_t *rec = a.records + (size_t)t * a.record_words;
        for (uint32_t i = lane; i < T.n_trailer; i += 64) rec[T.tbase + i] = T.trailer[i];
        for (uint32_t i = lane; i < T.n_fixed; i += 64) rec[T.fixed[2 * i]] = T.fixed[2 * i + 1];
    }
    if (lane == 0) a.outcome[t] = good ? 0u : 1u;
}

// Format 3 only, wave = window: the member names that stand next to the lists of a minimal proof.json (ss_text.h,
// min_text_landmark) and, for each, the numbers in front of it -- the replay of the place kernel up to the marks.
__global__ void __launch_bounds__(64) text_landmark_kernel(TextParseArgs a)
{
    __shared__ uint4 s_text4[64 + 1];  // this window and the first 16 bytes of the next
    const uint8_t *s_text = reinterpret_cast<const uint8_t *>(s_text4);
    const uint32_t gw = blockIdx.x, lane = threadIdx.x;
    if (gw >= a.n_windows) return;
    const uint32_t t = a.win_text[gw];
    if (a.fmt[t] != kTextMinimal || a.outcome[t] != 0) return;
    const uint32_t w = gw - a.win_base[t], nwin = a.win_base[t + 1] - a.win_base[t], len = a.lens[t];
    const uint4 *text4 = reinterpret_cast<const uint4 *>(a.texts + a.offs[t]);
    const WinIn in = a.win_in[gw];
    const uint4 cur = text4[(size_t)w * 64 + lane];
    s_text4[lane] = cur;
    if (lane == 0) s_text4[64] = w + 1 < nwin ? text4[(size_t)(w + 1) * 64] : make_uint4(0, 0, 0, 0);
    const uint32_t win0 = w << 10, pos0 = win0 + lane * 16;
    const uint32_t nvalid = pos0 >= len ? 0 : (len - pos0 < 16 ? len - pos0 : 16);
    const LaneLocal L = lane_local(cur, nvalid);
    const uint64_t det_mask = __ballot(L.det), q_mask = __ballot(L.qpar);
    const uint64_t below = (1ull << lane) - 1;
    uint32_t r_str = (in.state & 1) ^ ((uint32_t)__popcll(q_mask & below) & 1);
    uint32_t r_run = entering_run(det_mask, L.rout, L.ftype, in.state >> 1, lane);
    uint32_t mark_mask = 0, quote_mask = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        if (j < nvalid) {
            const uint32_t c = byte_of(cur, j);
            mark_mask |= (scan_byte(c, r_run, r_str) >> 1) << j;
            quote_mask |= (uint32_t)(c == '"') << j;
        }
    }
    const uint32_t mine = __popc(mark_mask);
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    const uint32_t tok0 = in.tok_pos + incl - mine;
    __syncthreads();
    MinHint *h = a.mhints + t;
    while (quote_mask) {
        const uint32_t j = __ffs(quote_mask) - 1;
        quote_mask &= quote_mask - 1;
        const int kind = min_text_landmark(s_text + lane * 16 + j, len - (pos0 + j));
        if (kind >= 0) {
            const uint32_t slot = atomicAdd(&h->n_lm[kind], 1u);
            if (slot < kMaxLandmarks) h->lm[kind][slot] = tok0 + __popc(mark_mask & ((1u << j) - 1));
        }
    }
}

// Format 3 only, wave = text: the landmarks in order, the list lengths they give, the gaps those cut out of the full-length
// template, and the lengths into the record.  A text whose landmarks are not a minimal proof.json's, or whose totals are not
// the ones its lengths imply, goes to the host reader.
__global__ void __launch_bounds__(64) text_minhint_kernel(TextParseArgs a)
{
    __shared__ uint32_t s_lm[3][kMaxLandmarks], s_sorted[3][kMaxLandmarks], s_counts[kMaxTextLists], s_ok;
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.n || a.fmt[t] != kTextMinimal || a.outcome[t] != 0) return;
    MinHint *h = a.mhints + t;
    const MinTextInfo &I = a.minfo;
    const uint32_t n_lm[3] = {h->n_lm[0], h->n_lm[1], h->n_lm[2]};
    bool ok = n_lm[kLmHash] == I.K + 4 && n_lm[kLmColumn] == I.K + 4 && n_lm[kLmPow] == 1;  // (<= kMaxLandmarks: all stored)
    if (ok) {
        for (uint32_t k = 0; k < 3; k++)
            if (lane < n_lm[k]) s_lm[k][lane] = h->lm[k][lane];
        __syncthreads();
        for (uint32_t k = 0; k < 3; k++)
            if (lane < n_lm[k]) {
                const uint32_t v = s_lm[k][lane];
                uint32_t rank = 0;
                for (uint32_t e = 0; e < n_lm[k]; e++) rank += s_lm[k][e] < v || (s_lm[k][e] == v && e < lane);
                s_sorted[k][rank] = v;
            }
        __syncthreads();
        if (lane == 0) {
            bool good = min_text_counts(I, s_sorted[kLmHash], n_lm[kLmHash], s_sorted[kLmColumn], n_lm[kLmColumn], s_sorted[kLmPow],
                                        n_lm[kLmPow], s_counts);
            if (good) {
                min_text_gaps(I, s_counts, a.tmpl[kTextMinimal].skel_len, a.tmpl[kTextMinimal].n_slots, h->g);
                good = h->g.skel_len == h->seen_skel && h->g.n_slots == h->seen_tok;
            }
            s_ok = good;
        }
        __syncthreads();
        ok = s_ok != 0;
        if (ok) {
            uint32_t *rec = a.records + (size_t)t * a.record_words;
            for (uint32_t j = lane; j < I.n_lists; j += 64) rec[I.word[j]] = s_counts[j] / I.per[j];
        }
    }
    if (!ok && lane == 0) a.outcome[t] = 1;
}

__global__ void __launch_bounds__(64) text_place_kernel(TextParseArgs a)
{
    __shared__ uint4 s_text4[2 * 64];   // this window and the next
    __shared__ uint4 s_skel4[2 * 64];   // 2 KiB of the template's skeleton from skel_pos & ~15
    __shared__ uint2 s_slot[8 * 64];    // the slots of the numbers that start in this window
    const uint8_t *s_text = reinterpret_cast<const uint8_t *>(s_text4);
    const uint8_t *s_skel = reinterpret_cast<const uint8_t *>(s_skel4);

    const uint32_t gw = blockIdx.x, lane = threadIdx.x;
    if (gw >= a.n_windows) return;
    const uint32_t t = a.win_text[gw];
    if (a.outcome[t] != 0) return;  // the scan (or another window) has already sent this text to the host reader
    const uint32_t w = gw - a.win_base[t], nwin = a.win_base[t + 1] - a.win_base[t], len = a.lens[t];
    const uint32_t f = a.fmt[t];
    const bool shared = f >= kTextShared;  // a text whose lists are shorter than the template's: shared-path (2) or minimal (3) proof.json
    const uint8_t *skel = a.tmpl[f].skel;
    const TextSlot *slots = a.tmpl[f].slots;
    const uint32_t n_slots = f == kTextShared ? a.hints[t].g.n_slots : f == kTextMinimal ? a.mhints[t].g.n_slots : a.tmpl[f].n_slots;
    uint32_t *rec = f == kTextShared ? a.shared_records + (size_t)t * a.tmpl[kTextShared].record_words : a.records + (size_t)t * a.record_words;
    // formats 2 and 3: positions in this text -> positions in the full-length template (ss_text.h, TextGaps / MinTextGaps)
    __shared__ uint32_t s_G[kMaxTextLists], s_D[kMaxTextLists], s_Gk[kMaxTextLists], s_Dk[kMaxTextLists];
    const uint32_t n_trees = f == kTextMinimal ? a.minfo.n_lists : a.sinfo.n_trees;  // (gaps)
    if (shared) {
        const uint32_t *gG = f == kTextShared ? a.hints[t].g.G : a.mhints[t].g.G, *gD = f == kTextShared ? a.hints[t].g.D : a.mhints[t].g.D;
        const uint32_t *gGk = f == kTextShared ? a.hints[t].g.Gk : a.mhints[t].g.Gk, *gDk = f == kTextShared ? a.hints[t].g.Dk : a.mhints[t].g.Dk;
        for (uint32_t i = lane; i < n_trees; i += 64) { s_G[i] = gG[i]; s_D[i] = gD[i]; s_Gk[i] = gGk[i]; s_Dk[i] = gDk[i]; }
        __syncthreads();
    }
    const uint4 *text4 = reinterpret_cast<const uint4 *>(a.texts + a.offs[t]);  // 16-byte aligned by the host
    const WinIn in = a.win_in[gw];

    const uint4 cur = text4[(size_t)w * 64 + lane];
    uint4 nxt = make_uint4(0, 0, 0, 0);
    // a number that starts in this window may run into the next: its first kMaxTokenBytes bytes are enough
    if (w + 1 < nwin && lane * 16 < kMaxTokenBytes) nxt = text4[(size_t)(w + 1) * 64 + lane];
    const uint32_t win0 = w << 10, pos0 = win0 + lane * 16;
    const uint32_t nvalid = pos0 >= len ? 0 : (len - pos0 < 16 ? len - pos0 : 16);
    s_text4[lane] = cur;
    s_text4[64 + lane] = nxt;
    // the template's skeleton from where this window starts in it (zero padded behind its end)
    const uint32_t sk_base = in.skel_pos & ~15u;
    if (!shared) {
        const uint4 *src = reinterpret_cast<const uint4 *>(skel + sk_base);
        s_skel4[lane] = src[lane];
        s_skel4[64 + lane] = src[64 + lane];
    } else {
        // the same 2 KiB through the gap map: a 16-byte group that no gap cuts is one (unaligned) load
        uint8_t *dst = reinterpret_cast<uint8_t *>(s_skel4);
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t j = lane + 64 * h, p0 = sk_base + 16 * j;
            const uint32_t m0 = gap_map(s_G, s_D, n_trees, p0), m1 = gap_map(s_G, s_D, n_trees, p0 + 15);
            if (m1 - m0 == 15) {
                uint4 v;
                __builtin_memcpy(&v, skel + m0, 16);
                s_skel4[j] = v;
            } else {
                for (uint32_t b = 0; b < 16; b++) dst[16 * j + b] = skel[gap_map(s_G, s_D, n_trees, p0 + b)];
            }
        }
    }

    // ---- the states entering this lane
    const LaneLocal L = lane_local(cur, nvalid);
    const uint64_t det_mask = __ballot(L.det), q_mask = __ballot(L.qpar);
    const uint64_t below = (1ull << lane) - 1;
    const uint32_t in_str = (in.state & 1) ^ ((uint32_t)__popcll(q_mask & below) & 1);
    const uint32_t run = entering_run(det_mask, L.rout, L.ftype, in.state >> 1, lane);

    // ---- replay with the true states
    uint32_t emit_mask = 0, mark_mask = 0;
    {
        uint32_t r_run = run, r_str = in_str;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            if (j < nvalid) {
                const uint32_t r = scan_byte(byte_of(cur, j), r_run, r_str);
                emit_mask |= (r & 1) << j;
                mark_mask |= (r >> 1) << j;
            }
        }
    }
    // ---- positions inside the window
    const uint32_t n_mark = __popc(mark_mask);
    const uint32_t mine = (__popc(emit_mask) + n_mark) | (n_mark << 16);
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    const uint32_t excl = incl - mine, n_tok = __shfl(incl, 63) >> 16;
    const uint32_t sk0 = in.skel_pos + (excl & 0xffff), tk0 = excl >> 16;  // tk0: index inside the window
    bool mism = in.tok_pos + n_tok > n_slots;
    if (!mism)
        for (uint32_t i = lane; i < n_tok; i += 64) {
            const uint32_t k = in.tok_pos + i;
            const TextSlot sl = slots[shared ? gap_map(s_Gk, s_Dk, n_trees, k) : k];
            s_slot[i] = make_uint2(sl.dst, sl.kind);
        }
    __syncthreads();

    // ---- the skeleton this lane contributes against the template's
    {
        uint32_t o = sk0 - sk_base;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            if ((mark_mask >> j) & 1) mism |= s_skel[o++] != kSkelMark;
            if ((emit_mask >> j) & 1) mism |= s_skel[o++] != byte_of(cur, j);
        }
    }
    // ---- the numbers that start in this lane's bytes
    {
        const uint32_t have = 1024 + kMaxTokenBytes;                  // text bytes present in s_text
        const uint32_t lim = len - win0 < have ? len - win0 : have;
        uint32_t k = tk0, mm = mark_mask;
        while (mm && !mism) {
            const uint32_t j = __ffs(mm) - 1;
            mm &= mm - 1;
            const uint2 sl = s_slot[k++];
            const uint32_t dst = sl.x, kind = sl.y;
            const uint32_t p = lane * 16 + j;
            uint32_t n = 0;
            while (n <= kMaxTokenBytes && p + n < lim && txt_is_alnum(s_text[p + n])) n++;
            if (kind == kSlotDec256) {  // decimal below 2^256: nine digits at a time into eight limbs
                bool ok = n >= 1 && n <= 78 && !(n > 1 && s_text[p] == '0');
                uint32_t limb[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // little endian
                for (uint32_t i = 0; i < n && ok;) {
                    const uint32_t k = n - i < 9 ? n - i : 9;
                    uint32_t chunk = 0, mul = 1;
                    for (uint32_t j = 0; j < k; j++) {
                        const uint32_t d = s_text[p + i + j] - '0';
                        ok &= d <= 9;
                        chunk = chunk * 10 + d;
                        mul *= 10;
                    }
                    uint64_t carry = chunk;
#pragma unroll
                    for (int l = 0; l < 8; l++) {
                        const uint64_t v = (uint64_t)limb[l] * mul + carry;
                        limb[l] = (uint32_t)v;
                        carry = v >> 32;
                    }
                    ok &= carry == 0;  // >= 2^256
                    i += k;
                }
                if (ok) {
#pragma unroll
                    for (int l = 0; l < 8; l++) rec[dst + l] = limb[7 - l];
                }
                mism |= !ok;
                continue;
            }
            if (kind == kSlotHex256) {
                bool ok = n == 66 && s_text[p] == '0' && (s_text[p + 1] | 0x20) == 'x';
                if (ok) {
                    for (uint32_t wd = 0; wd < 8; wd++) {
                        uint32_t v = 0;
#pragma unroll
                        for (uint32_t d = 0; d < 8; d++) {
                            const uint32_t c = s_text[p + 2 + 8 * wd + d];
                            uint32_t h;
                            if (c - '0' < 10u) h = c - '0';
                            else if ((c | 0x20) - 'a' < 6u) h = (c | 0x20) - 'a' + 10;
                            else { h = 0; ok = false; }
                            v = (v << 4) | h;
                        }
                        rec[dst + wd] = v;
                    }
                }
                mism |= !ok;
                continue;
            }
            // canonical decimal: digits only, no leading zero, at most 20 digits, below 2^64
            bool ok = n >= 1 && n <= 20 && !(n > 1 && s_text[p] == '0');
            uint64_t v = 0;
            for (uint32_t i = 0; i < n && ok; i++) {
                const uint32_t d = s_text[p + i] - '0';
                ok = d <= 9 && v <= (~(uint64_t)0 - d) / 10;
                v = v * 10 + d;
            }
            if (ok) {
                switch (kind) {
                case kSlotU32:
                    ok = v <= 0xffffffffull;
                    if (ok) rec[dst] = (uint32_t)v;
                    break;
                case kSlotByte:
                    ok = v <= 255;
                    if (ok) reinterpret_cast<uint8_t *>(rec)[dst] = (uint8_t)v;
                    break;
                case kSlotU64:
                    rec[dst] = (uint32_t)(v >> 32);
                    rec[dst + 1] = (uint32_t)v;
                    break;
                default:  // kSlotConst
                    ok = v == dst;
                    break;
                }
            }
            mism |= !ok;
        }
    }
    if (mism) a.outcome[t] = 1;
}

__global__ void __launch_bounds__(256) text_scatter_kernel(uint32_t record_words, const uint32_t *__restrict__ src,
                                                           const uint32_t *__restrict__ idx, uint32_t *__restrict__ dst)
{
    const uint32_t *from = src + (size_t)blockIdx.x * record_words;
    uint32_t *to = dst + (size_t)idx[blockIdx.x] * record_words;
    for (uint32_t i = threadIdx.x; i < record_words; i += 256) to[i] = from[i];
}

void launch_text_scatter(size_t n, size_t record_words, const uint32_t *src, const uint32_t *idx, uint32_t *dst, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(text_scatter_kernel, dim3((unsigned)n), dim3(256), 0, s, (uint32_t)record_words, src, idx, dst);
}

void launch_text_parse(const TextParseArgs &a, hipStream_t s)
{
    if (!a.n) return;
    hipLaunchKernelGGL(text_index_kernel, dim3(a.n), dim3(64), 0, s, a);
    if (a.hints) hipLaunchKernelGGL(text_hint_kernel, dim3(a.n), dim3(64), 0, s, a);
    if (a.n_windows) hipLaunchKernelGGL(text_summary_kernel, dim3(a.n_windows), dim3(64), 0, s, a);
    hipLaunchKernelGGL(text_scan_kernel, dim3(a.n), dim3(64), 0, s, a);
    if (a.mhints) {
        if (a.n_windows) hipLaunchKernelGGL(text_landmark_kernel, dim3(a.n_windows), dim3(64), 0, s, a);
        hipLaunchKernelGGL(text_minhint_kernel, dim3(a.n), dim3(64), 0, s, a);
    }
    if (a.n_windows) hipLaunchKernelGGL(text_place_kernel, dim3(a.n_windows), dim3(64), 0, s, a);
}

}  // namespace ss
