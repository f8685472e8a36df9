// iknn_score.hip -- item-kNN scoring for a BATCH of queries, gfx950.
//
// Stands in for `score_explicit` / `score_implicit` (src/accel/knn/item_score.rs:23-111)
// and `ScoreAccumulator` (src/accel/knn/accum.rs:16-239), which score ONE query per call
// on one CPU thread (the reference's "batch" is a Python loop,
// src/lenskit/batch/_runner.py:283-308).
//
// Per query: for every reference (history) item r, in history order, its similarity
// row S_r is scattered into per-target accumulators that keep the `max_nbrs` largest
// similarities (an entry only displaces the current minimum when STRICTLY larger,
// accum.rs:108); score_t = sum(s*v)/sum(s) (explicit) or sum(s) (implicit) over the kept
// entries, NaN when fewer than `min_nbrs`; count_t = entries kept.  Null (negative)
// reference items are skipped (the reference would read out of bounds, SURVEY.md section 8a
// row a14); null targets give NaN / -1.  A NaN similarity is an error
// (accum.rs:146-151 -> ValueError("similarity is null")).
//
// One workgroup per query.  History items are taken one at a time (a similarity row has
// distinct columns, so the threads of the workgroup never collide inside one row and the
// per-target state is updated with plain loads/stores; a barrier separates rows).  The
// per-target slots live in a per-workgroup slab of the workspace (L2-resident):
// cnt[t] (-1 = not a target), minval[t], and max_nbrs (s, v) slots.
//
// That kernel (`iknn_score_kernel`) costs a query one barrier per history item and an
// n_items-wide reset, whatever the number of targets: 13 ms for the cfg3 batch (10 000 queries
// x 100 targets).  Calls in which no query has more than KF_NT_MAX targets -- the `score a
// candidate list` case -- take `iknn_score_fast_kernel` instead (and max_nbrs <= KF_NBR_MAX):
// the targets go into an LDS hash table, the history rows are streamed concurrently (a wave takes
// KF_U rows at a time, coalesced; an LDS probe per entry), every hit is appended to its target's
// list (LDS counters, the (history position, weight, value) triples in an L2-resident slab).
// Histories are cut into rounds of KF_CAP rows, so a list never holds more than KF_CAP hits;
// after a round one wave per target sorts the hits back into history order and feeds them to the
// accumulator -- which is the reference's, step for step: a vector while there is room, then
// std's BinaryHeap push / pop replayed on an LDS copy, so the SAME one of several equal weights
// is evicted, the array ends in the same order, and the sequential sums over it (product and sum
// rounded separately) give the reference's bits.  Queries with long histories are started first
// and split into parts by target (see the kernel).  1.3 ms for the cfg3 batch.
#include "common.h"

// The reference's sums are plain f32 multiplies and adds (Rust never contracts a * b + c): no
// FMA contraction anywhere in this file, so that `weight * value` is rounded before it is added.
#pragma clang fp contract(off)

namespace lk {

constexpr int KS_THREADS = 512;
constexpr int KS_MAX_WGS = 256;

// per workgroup: cnt[t] (-1 = not a target), the heap flag of t, and max_nbrs + 1 (weight, value)
// slots per item (a push precedes the pop)
__host__ __device__ inline size_t ks_slab_bytes(int64_t n_items, int max_nbrs)
{
    size_t per = (size_t)n_items * (sizeof(int32_t) + sizeof(int32_t)) +
                 (size_t)n_items * ((size_t)max_nbrs + 1) * 2 * sizeof(float);
    return (per + 255) / 256 * 256;
}

// ---- the reference's accumulator, restated for ONE lane (arrays in LDS or in the slab) ---------
// `Full(BinaryHeap<AccEntry>)` with the REVERSED ordering of accum.rs:170-184: a min-heap on
// the weight.  std's BinaryHeap::push = sift_up(0, old_len); pop = swap the last element into
// the root, sift_down_to_bottom(0), sift_up.  Followed step for step so that WHICH of several
// equal weights is evicted, and the array order the sums run over, are the reference's.
struct KfHeap {
    float *w, *v;
    int len;
    __device__ __forceinline__ void sift_up(int pos, float ew, float ev)
    {
        while (pos > 0) {
            const int parent = (pos - 1) >> 1;
            if (ew >= w[parent]) break;  // hole.element() <= hole.get(parent) in reversed order
            w[pos] = w[parent];
            v[pos] = v[parent];
            pos = parent;
        }
        w[pos] = ew;
        v[pos] = ev;
    }
    __device__ __forceinline__ void push(float ew, float ev) { sift_up(len++, ew, ev); }
    __device__ __forceinline__ void pop()
    {
        const float ew = w[len - 1], ev = v[len - 1];
        --len;
        if (len == 0) return;
        const int end = len;
        int pos = 0, child = 1;
        const int limit = end >= 2 ? end - 2 : 0;
        while (child <= limit && end >= 2) {
            if (w[child] >= w[child + 1]) child += 1;  // hole.get(child) <= hole.get(child + 1)
            w[pos] = w[child];
            v[pos] = v[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            w[pos] = w[child];
            v[pos] = v[child];
            pos = child;
        }
        sift_up(pos, ew, ev);
    }
};


// SWAP = false: item-kNN (matrix = similarities, entry value = weight, the reference row's
// rating = value).  SWAP = true: user-kNN (`user_score_items_*`,
// src/accel/knn/user_score.rs:21-98): matrix = the users' ratings, the reference rows are the
// neighbours, their similarity (ref_rates) = weight, the entry value (rating; none for
// implicit feedback: s_val null) = value.  `n_rows` = rows of the matrix, `n_items` = columns.
template <bool SWAP>
__global__ __launch_bounds__(KS_THREADS) void iknn_score_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_rows, int64_t n_items, int64_t n_queries,
    const int64_t *__restrict__ ref_ptr, const int32_t *__restrict__ ref_items,
    const float *__restrict__ ref_rates, const int64_t *__restrict__ tgt_ptr,
    const int32_t *__restrict__ tgt_items, int max_nbrs, int min_nbrs, char *__restrict__ ws,
    size_t slab_bytes, float *__restrict__ out_scores, int32_t *__restrict__ out_counts,
    int *__restrict__ status)
{
    char *slab = ws + (size_t)blockIdx.x * slab_bytes;
    int32_t *cnt = reinterpret_cast<int32_t *>(slab);
    int32_t *is_heap = reinterpret_cast<int32_t *>(slab + (size_t)n_items * 4);
    const size_t stride = (size_t)max_nbrs + 1;
    float *slot_s = reinterpret_cast<float *>(slab + (size_t)n_items * 8);
    float *slot_v = slot_s + (size_t)n_items * stride;
    const int tid = threadIdx.x;
    const bool explicit_ = SWAP ? (s_val != nullptr) : (ref_rates != nullptr);

    for (int64_t q = blockIdx.x; q < n_queries; q += gridDim.x) {
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        const int64_t tb = tgt_ptr[q], te = tgt_ptr[q + 1];
        // enable the targets (ScoreAccumulator::new_array, accum.rs:30-37)
        for (int64_t t = tid; t < n_items; t += KS_THREADS) cnt[t] = -1;
        __syncthreads();
        for (int64_t j = tb + tid; j < te; j += KS_THREADS) {
            const int t = tgt_items[j];
            if (t >= 0 && t < n_items) {
                cnt[t] = 0;
                is_heap[t] = 0;
            }
        }
        __syncthreads();
        // scatter the history rows, one reference item at a time, in history order
        for (int64_t r = rb; r < re; ++r) {
            const int ri = ref_items[r];
            if (ri < 0 || ri >= n_rows) continue;  // wave-uniform: null reference row
            const float rscalar = (SWAP || explicit_) ? ref_rates[r] : 0.f;
            const int64_t sb = s_ptr[ri], se = s_ptr[ri + 1];
            for (int64_t e = sb + tid; e < se; e += KS_THREADS) {
                const int t = s_idx[e];
                const int c = cnt[t];
                if (c < 0) continue;
                // (weight, value) of this contribution
                const float s = SWAP ? rscalar : s_val[e];
                const float rv = SWAP ? (explicit_ ? s_val[e] : 0.f) : rscalar;
                if (s != s) {
                    atomicCAS(status, 0, 1);
                    continue;
                }
                float *ss = slot_s + (size_t)t * stride;
                float *sv = slot_v + (size_t)t * stride;
                const bool heap = is_heap[t] != 0;
                if (!heap && c < max_nbrs) {  // Partial(vec): push (accum.rs:103-104)
                    ss[c] = s;
                    sv[c] = rv;
                    cnt[t] = c + 1;
                } else {  // Full(heap): the reference's BinaryHeap, step for step
                    KfHeap h{ss, sv, c};
                    if (!heap) {
                        // Partial -> Full (heap_mut, accum.rs:76-83): the vector is popped from the
                        // back and every element pushed onto an empty heap = reverse it in place,
                        // then sift element k up into the heap formed by the k before it
                        for (int i = 0, j = c - 1; i < j; ++i, --j) {
                            const float a = ss[i], b = sv[i];
                            ss[i] = ss[j];
                            sv[i] = sv[j];
                            ss[j] = a;
                            sv[j] = b;
                        }
                        for (int kk = 1; kk < c; ++kk) h.sift_up(kk, ss[kk], sv[kk]);
                        is_heap[t] = 1;
                    }
                    if (s > ss[0]) {  // accum.rs:108: strictly greater than the minimum
                        h.push(s, rv);
                        while (h.len > max_nbrs) h.pop();
                    }
                }
            }
            __syncthreads();
        }
        // collect (collect_items_averaged / _summed / _counts, accum.rs:186-239)
        for (int64_t j = tb + tid; j < te; j += KS_THREADS) {
            const int t = tgt_items[j];
            float score = __builtin_nanf("");
            int count = -1;
            if (t >= 0 && t < n_items) {
                const int c = cnt[t];
                count = c;
                if (c >= min_nbrs && c > 0) {
                    // (sequential sums in array order, product rounded before it is added)
                    const float *ss = slot_s + (size_t)t * stride;
                    const float *sv = slot_v + (size_t)t * stride;
                    float tw = 0.f, wsum = 0.f;
                    for (int i = 0; i < c; ++i) {
                        tw += ss[i];
                        wsum += ss[i] * sv[i];
                    }
                    score = explicit_ ? wsum / tw : tw;
                }
            }
            out_scores[j] = score;
            out_counts[j] = count;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Candidate-list path (see the file header).
// ---------------------------------------------------------------------------
constexpr int KF_THREADS = 256;
constexpr int KF_WAVES = KF_THREADS / 64;
constexpr int KF_HT = 2048;       // hash slots (load factor <= 0.5)
constexpr int KF_NT_MAX = 1024;   // targets per query
constexpr int KF_CAP = 256;       // history rows per round = the most hits one target collects in it
constexpr int KF_NBR_MAX = 255;   // max_nbrs the per-wave accumulator buffer holds
constexpr int KF_MAX_WGS = 1024;
constexpr int KF_U = 16;          // history rows a wave has in flight
constexpr int KF_HEAVY = 2048;    // queries with more history rows than this are started first ...
constexpr int KF_SPLIT = 8;       // ... and scored in this many parts (by target)

// per workgroup: the round's hits (history position, weight, value) per target, and -- for
// queries with more than KF_CAP history rows -- the accumulators between rounds
__host__ __device__ inline size_t kf_slab_bytes(int64_t nt_max, int max_nbrs)
{
    return ((size_t)nt_max * (KF_CAP * 12 + (size_t)(max_nbrs + 1) * 8) + 255) / 256 * 256;
}

__device__ __forceinline__ int kf_hash(int t) { return (int)(((unsigned)t * 2654435761u) >> 21); }
__device__ __forceinline__ int kf_part(int t, int split)  // split: a power of two
{
    return (int)((((unsigned)t * 0x85ebca6bu) >> 13) & (unsigned)(split - 1));
}

__device__ __forceinline__ int64_t kf_readlane64(int64_t v, int l)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// pre-pass of the list kernel: the longest target list (status[2]) and the queries with more than
// KF_HEAVY history rows (heavy[0 .. status[6]))
__global__ __launch_bounds__(1024) void seg_max_kernel(const int64_t *__restrict__ ptr, int64_t n,
                                                       int *__restrict__ status,
                                                       const int64_t *__restrict__ ref_ptr,
                                                       int32_t *__restrict__ heavy, int heavy_len)
{
    long long m = 0;
    for (int64_t q = threadIdx.x; q < n; q += 1024) {
        const long long d = ptr[q + 1] - ptr[q];
        m = d > m ? d : m;
        if (ref_ptr[q + 1] - ref_ptr[q] > heavy_len) heavy[atomicAdd(&status[6], 1)] = (int32_t)q;
    }
    if (m > 0x7fffffffLL) m = 0x7fffffffLL;
    atomicMax(&status[2], (int)m);
}

template <bool SWAP>
__global__ __launch_bounds__(KF_THREADS) void iknn_score_fast_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_rows, int64_t n_items, int64_t n_queries,
    const int64_t *__restrict__ ref_ptr, const int32_t *__restrict__ ref_items,
    const float *__restrict__ ref_rates, const int64_t *__restrict__ tgt_ptr,
    const int32_t *__restrict__ tgt_items, int max_nbrs, int min_nbrs, char *__restrict__ ws,
    int nt_max, float *__restrict__ out_scores, int32_t *__restrict__ out_counts,
    int *__restrict__ status, const int32_t *__restrict__ heavy, int heavy_len, int heavy_split)
{
    __shared__ int hkey[KF_HT];              // item number, -1 = empty
    __shared__ unsigned short hpos[KF_HT];   // the target position that owns the item's list
    __shared__ unsigned short tc[KF_NT_MAX + 2];  // hits of the current round per target (32-bit atomics
                                                  // on pairs: see hit())
    __shared__ short canon[KF_NT_MAX];       // owner position of target position j, -1 = null target
    __shared__ int acc_len[KF_NT_MAX];       // accumulator: entries | (heap state << 16)
    __shared__ float res_s[KF_NT_MAX];
    __shared__ unsigned char rbuf[KF_WAVES][KF_CAP];  // history position of a round hit (< KF_CAP)
    __shared__ float sw[KF_WAVES][KF_CAP], sv[KF_WAVES][KF_CAP];  // the round's hits in history order
    __shared__ float hw[KF_WAVES][KF_CAP], hv[KF_WAVES][KF_CAP];  // the accumulator being updated
    __shared__ float tw_[KF_WAVES][KF_CAP], tv_[KF_WAVES][KF_CAP];  // vector -> heap staging
    char *slab = ws + (size_t)blockIdx.x * kf_slab_bytes(nt_max, max_nbrs);
    unsigned char *slab_r = reinterpret_cast<unsigned char *>(slab);                  // [nt_max][CAP]
    float *slab_w = reinterpret_cast<float *>(slab + (size_t)nt_max * KF_CAP);        // 256-aligned
    float *slab_v = slab_w + (size_t)nt_max * KF_CAP;
    float *acc_w = slab_v + (size_t)nt_max * KF_CAP;  // [nt_max][max_nbrs + 1]
    float *acc_v = acc_w + (size_t)nt_max * (max_nbrs + 1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool explicit_ = SWAP ? (s_val != nullptr) : (ref_rates != nullptr);
    const float nanf_ = __builtin_nanf("");
    int *tc32 = reinterpret_cast<int *>(tc);
    __shared__ int s_q;

    // Work is handed out from two counters (status[4], status[5]): first the queries with more
    // than KF_HEAVY history rows (listed by the pre-pass: heavy[0 .. status[6])), then the rest.  A long history is many rounds, one after the
    // other (a target's hits must be applied in history order): the heaviest user of a batch
    // would otherwise be the whole launch (measured: 2.4 ms for 10 000 ML-25M users, of which a
    // light query takes ~30 us).  So heavy queries (i) start first and (ii) are cut into KF_SPLIT
    // PARTS BY TARGET: part g owns the targets whose item number hashes to g, streams the whole
    // history, and feeds only its own accumulators -- targets are independent of each other.
    for (int phase = 0; phase < 2; ++phase)
    for (;;) {
        __syncthreads();
        if (tid == 0) s_q = atomicAdd(&status[4 + phase], 1);
        __syncthreads();
        const int split = phase == 0 ? heavy_split : 1;
        const int part = s_q % split;
        if (s_q / split >= (phase == 0 ? (int64_t)status[6] : n_queries)) break;
        const int64_t q = phase == 0 ? (int64_t)heavy[s_q / split] : (int64_t)s_q;
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        if (phase == 1 && (re - rb) > heavy_len) continue;
        const int64_t tb = tgt_ptr[q];
        const int nt = (int)(tgt_ptr[q + 1] - tb);  // <= nt_max <= KF_NT_MAX (checked by the host)
        for (int i = tid; i < KF_HT; i += KF_THREADS) hkey[i] = -1;
        for (int j = tid; j < (KF_NT_MAX + 2) / 2; j += KF_THREADS) tc32[j] = 0;
        for (int j = tid; j < nt; j += KF_THREADS) acc_len[j] = 0;
        __syncthreads();
        // the targets (ScoreAccumulator::new_array, accum.rs:30-37); a repeated item shares one list
        for (int j = tid; j < nt; j += KF_THREADS) {
            const int t = tgt_items[tb + j];
            if (t < 0 || t >= n_items) continue;
            if (split > 1 && kf_part(t, split) != part) continue;  // another part's target
            int slot = kf_hash(t);
            for (;;) {
                const int old = atomicCAS(&hkey[slot], -1, t);
                if (old == -1) {
                    hpos[slot] = (unsigned short)j;
                    break;
                }
                if (old == t) break;
                slot = (slot + 1) & (KF_HT - 1);
            }
        }
        __syncthreads();
        for (int j = tid; j < nt; j += KF_THREADS) {
            const int t = tgt_items[tb + j];
            int c = -1;  // null target
            if (t >= 0 && t < n_items) {
                if (split > 1 && kf_part(t, split) != part) {
                    c = -2;  // scored (and written) by another part
                } else {
                    int slot = kf_hash(t);
                    while (hkey[slot] != t) slot = (slot + 1) & (KF_HT - 1);
                    c = hpos[slot];
                }
            }
            canon[j] = (short)c;
        }
        __syncthreads();

        // one hit: the entry (weight s, value rv) of history row number `rr` of this round lands in
        // the list of the target that holds item t
        auto hit = [&](int t, int64_t e, float rscalar, int rr) {
            if (t < 0) return;
            int slot = kf_hash(t);
            int kk;
            while ((kk = hkey[slot]) != -1 && kk != t) slot = (slot + 1) & (KF_HT - 1);
            if (kk != t) return;
            const float s = SWAP ? rscalar : s_val[e];
            const float rv = SWAP ? (explicit_ ? s_val[e] : 0.f) : rscalar;
            if (s != s) {
                atomicCAS(status, 0, 1);
                return;
            }
            const int j = hpos[slot];
            // 16-bit counter inside a 32-bit LDS atomic (counts stay below 2^16: no carry)
            const int old = atomicAdd(&tc32[j >> 1], (j & 1) ? 0x10000 : 1);
            const int pos = (j & 1) ? (int)((unsigned)old >> 16) : (old & 0xffff);
            if (pos >= KF_CAP) {  // a matrix row that names a column twice
                atomicCAS(&status[1], 0, 1);
                return;
            }
            const size_t at = (size_t)j * KF_CAP + pos;
            slab_r[at] = (unsigned char)rr;
            slab_w[at] = s;
            slab_v[at] = rv;
        };

        for (int64_t r0 = rb;; r0 += KF_CAP) {
            const int64_t r1 = re - r0 > KF_CAP ? r0 + KF_CAP : re;
            const bool last = r1 >= re;
            // ---- stream the history rows [r0, r1): a wave takes KF_U rows at a time -----------
            // (row descriptors -- history item -> its row's extent, two dependent loads -- are
            // fetched one sweep AHEAD, under the column loads and the probing of the current one)
            auto load_desc = [&](int64_t rbase, int64_t &b, int64_t &e, float &rate) {
                b = e = 0;
                rate = 0.f;
                if (lane < KF_U && rbase + lane < r1) {
                    const int ri = ref_items[rbase + lane];
                    if (ri >= 0 && ri < n_rows) {  // (null reference rows stay empty)
                        b = s_ptr[ri];
                        e = s_ptr[ri + 1];
                    }
                    if (SWAP || explicit_) rate = ref_rates[rbase + lane];
                }
            };
            int64_t myb, mye;
            float myrate;
            load_desc(r0 + (int64_t)wave * KF_U, myb, mye, myrate);
            for (int64_t rbase = r0 + (int64_t)wave * KF_U; rbase < r1; rbase += KF_WAVES * KF_U) {
                int64_t sb[KF_U];
                int ln[KF_U], t0[KF_U], t1[KF_U];
                float rate[KF_U];
#pragma unroll
                for (int u = 0; u < KF_U; ++u) {
                    sb[u] = kf_readlane64(myb, u);
                    ln[u] = (int)(kf_readlane64(mye, u) - sb[u]);
                    rate[u] = bcast(myrate, u);
                    t0[u] = lane < ln[u] ? s_idx[sb[u] + lane] : -1;
                    t1[u] = lane + 64 < ln[u] ? s_idx[sb[u] + 64 + lane] : -1;
                }
                int64_t nb, ne;
                float nrate;
                load_desc(rbase + KF_WAVES * KF_U, nb, ne, nrate);  // (past r1: empty)
#pragma unroll
                for (int u = 0; u < KF_U; ++u) {
                    const int rr = (int)(rbase - r0) + u;
                    hit(t0[u], sb[u] + lane, rate[u], rr);
                    hit(t1[u], sb[u] + 64 + lane, rate[u], rr);
                    for (int e = 128 + lane; e < ln[u]; e += 64)
                        hit(s_idx[sb[u] + e], sb[u] + e, rate[u], rr);
                }
                myb = nb;
                mye = ne;
                myrate = nrate;
            }
            __threadfence_block();
            __syncthreads();
            // ---- feed the round's hits to the accumulators, in history order: wave per target ----
            // (owner positions only: a null target or a repeat is copied from its owner at the
            // end; the hits of the NEXT target are fetched while this one is worked on)
            auto next_owner = [&](int j) {
                while (j < nt && (canon[j] != j || (tc[j] == 0 && !last))) j += KF_WAVES;
                return j;
            };
            auto load_hits = [&](int j, int c, int (&rr)[KF_CAP / 64], float (&w)[KF_CAP / 64],
                                 float (&v)[KF_CAP / 64]) {
#pragma unroll
                for (int e = 0; e < KF_CAP / 64; ++e) {
                    const int i = lane + 64 * e;
                    rr[e] = 0;
                    w[e] = v[e] = 0.f;
                    if (i < c) {
                        const size_t at = (size_t)j * KF_CAP + i;
                        rr[e] = slab_r[at];
                        w[e] = slab_w[at];
                        v[e] = slab_v[at];
                    }
                }
            };
            int j = next_owner(wave);
            int c = j < nt ? tc[j] : 0;
            int rr_[KF_CAP / 64];
            float w_[KF_CAP / 64], v_[KF_CAP / 64];
            load_hits(j < nt ? j : 0, j < nt ? c : 0, rr_, w_, v_);
            while (j < nt) {
                const int jn = next_owner(j + KF_WAVES);
                const int cn = jn < nt ? tc[jn] : 0;
                int nrr[KF_CAP / 64];
                float nw[KF_CAP / 64], nv[KF_CAP / 64];
                load_hits(jn < nt ? jn : 0, jn < nt ? cn : 0, nrr, nw, nv);

                int len = acc_len[j] & 0xffff;
                bool full = (acc_len[j] >> 16) != 0;
                // the accumulator as the previous round left it
                if (r0 > rb && len > 0) {
                    for (int i = lane; i < len; i += 64) {
                        hw[wave][i] = acc_w[(size_t)j * (max_nbrs + 1) + i];
                        hv[wave][i] = acc_v[(size_t)j * (max_nbrs + 1) + i];
                    }
                }
                // sort the hits by history position (distinct): rank = smaller positions
                int rank[KF_CAP / 64];
#pragma unroll
                for (int e = 0; e < KF_CAP / 64; ++e) {
                    rank[e] = 0;
                    if (lane + 64 * e < c) rbuf[wave][lane + 64 * e] = (unsigned char)rr_[e];
                }
                __builtin_amdgcn_wave_barrier();
                for (int jj = 0; jj < c; ++jj) {
                    const int x = rbuf[wave][jj];
#pragma unroll
                    for (int e = 0; e < KF_CAP / 64; ++e) rank[e] += (x < rr_[e]) ? 1 : 0;
                }
#pragma unroll
                for (int e = 0; e < KF_CAP / 64; ++e) {
                    if (lane + 64 * e < c) {
                        sw[wave][rank[e]] = w_[e];
                        sv[wave][rank[e]] = v_[e];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const float *fw = hw[wave], *fv = hv[wave];  // where the final array is
                if (last && len == 0 && !full && c <= max_nbrs) {
                    // the common case -- one round, everything fits: the sorted hits ARE the vector
                    fw = sw[wave];
                    fv = sv[wave];
                    len = c;
                } else {
                    // Partial(vec): push while there is room (accum.rs:103-104) -- all lanes at once
                    int i0 = 0;
                    if (!full) {
                        const int room = max_nbrs - len;
                        i0 = c < room ? c : room;
                        for (int i = lane; i < i0; i += 64) {
                            hw[wave][len + i] = sw[wave][i];
                            hv[wave][len + i] = sv[wave][i];
                        }
                        len += i0;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i0 < c) {  // the rest meets a full accumulator: one lane, entry by entry
                        if (!full) {
                            // Partial -> Full (heap_mut, accum.rs:76-83): the vector is popped from
                            // the back and every element pushed onto an empty heap
                            for (int i = lane; i < len; i += 64) {
                                tw_[wave][i] = hw[wave][i];
                                tv_[wave][i] = hv[wave][i];
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                        if (lane == 0) {
                            KfHeap h{hw[wave], hv[wave], len};
                            if (!full) {
                                h.len = 0;
                                for (int i = len - 1; i >= 0; --i) h.push(tw_[wave][i], tv_[wave][i]);
                            }
                            for (int i = i0; i < c; ++i) {
                                const float ew = sw[wave][i];
                                if (ew > h.w[0]) {  // accum.rs:108: strictly greater than the minimum
                                    h.push(ew, sv[wave][i]);
                                    while (h.len > max_nbrs) h.pop();
                                }
                            }
                        }
                        full = true;  // (len stays max_nbrs)
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (!last) {
                    for (int i = lane; i < len; i += 64) {
                        acc_w[(size_t)j * (max_nbrs + 1) + i] = hw[wave][i];
                        acc_v[(size_t)j * (max_nbrs + 1) + i] = hv[wave][i];
                    }
                    if (lane == 0) {
                        acc_len[j] = len | (full ? 0x10000 : 0);
                        tc[j] = 0;
                    }
                } else {
                    // total_weight / weighted_sum (accum.rs:121-140): sequential f32 sums over the
                    // array, the product rounded before it is added (no contraction: see the
                    // pragma).  The array goes into registers (lane = position), the running sums
                    // take one element after the other by readlane.
                    float xw[KF_CAP / 64], xp[KF_CAP / 64];
#pragma unroll
                    for (int e = 0; e < KF_CAP / 64; ++e) {
                        const int i = lane + 64 * e;
                        xw[e] = i < len ? fw[i] : 0.f;
                        xp[e] = i < len ? xw[e] * fv[i] : 0.f;
                    }
                    float tw = 0.f, wsum = 0.f;
                    if (len >= min_nbrs) {
#pragma unroll
                        for (int e = 0; e < KF_CAP / 64; ++e) {
                            const int m = len - 64 * e < 64 ? len - 64 * e : 64;
                            for (int l = 0; l < m; ++l) {
                                tw = tw + bcast(xw[e], l);
                                wsum = wsum + bcast(xp[e], l);
                            }
                        }
                    }
                    if (lane == 0) {
                        acc_len[j] = len;
                        res_s[j] = (len >= min_nbrs && len > 0) ? (explicit_ ? wsum / tw : tw) : nanf_;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                j = jn;
                c = cn;
#pragma unroll
                for (int e = 0; e < KF_CAP / 64; ++e) {
                    rr_[e] = nrr[e];
                    w_[e] = nw[e];
                    v_[e] = nv[e];
                }
            }
            __threadfence_block();
            __syncthreads();
            if (last) break;
        }
        for (int j = tid; j < nt; j += KF_THREADS) {
            const int jo = canon[j];
            if (jo == -2) continue;
            out_scores[tb + j] = jo >= 0 ? res_s[jo] : nanf_;
            out_counts[tb + j] = jo >= 0 ? (acc_len[jo] & 0xffff) : -1;
        }
        __syncthreads();
    }
}

}  // namespace lk

namespace lk {

// last call of the calling thread: queries on the list kernel, on the slot kernel, max targets
static thread_local int64_t g_score_stats[3] = {0, 0, 0};

static inline size_t ks_list_bytes(int64_t n_queries)
{
    return ((size_t)(n_queries > 0 ? n_queries : 1) * sizeof(int32_t) + 255) / 256 * 256;
}
static inline int64_t ks_slot_wgs(int64_t n_queries)
{
    int64_t wgs = n_queries < KS_MAX_WGS ? n_queries : KS_MAX_WGS;
    return wgs < 1 ? 1 : wgs;
}

// the slab region serves either kernel: slot slabs for up to KS_MAX_WGS workgroups, and room for
// the list kernel to run at least min(queries, 256) workgroups at the longest admissible target
// list (more workgroups when the lists are shorter)
static inline size_t ks_region_bytes(int64_t n_items, int64_t n_queries, int max_nbrs)
{
    const size_t a = (size_t)ks_slot_wgs(n_queries) * ks_slab_bytes(n_items, max_nbrs);
    const size_t b = max_nbrs <= KF_NBR_MAX
                         ? (size_t)ks_slot_wgs(n_queries) * kf_slab_bytes(KF_NT_MAX, max_nbrs)
                         : 0;
    return a > b ? a : b;
}

static bool score_fast_enabled()
{
    const char *e = getenv("LK_KNN_SCORE_LISTS");  // A/B knob and test hook (read per call)
    return !(e && e[0] == '0');
}

// both scorers: header (status words) | per-workgroup slabs
template <bool SWAP>
static int score_batch(const char *what, const int64_t *d_ptr, const int32_t *d_idx,
                       const float *d_val, int64_t n_rows, int64_t n_items, int64_t n_queries,
                       const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                       const float *d_ref_rates, const int64_t *d_tgt_ptr,
                       const int32_t *d_tgt_items, int32_t max_nbrs, int32_t min_nbrs, void *d_ws,
                       float *d_out_scores, int32_t *d_out_counts, hipStream_t st)
{
    char *ws = static_cast<char *>(d_ws);
    // [0] NaN seen, [1] repeated column, [2] max targets, [4] [5] work counters, [6] heavy queries
    int *status = reinterpret_cast<int *>(ws);
    int32_t *heavy = reinterpret_cast<int32_t *>(ws + 256);
    char *slabs = ws + 256 + ks_list_bytes(n_queries);
    const size_t slot_slab = ks_slab_bytes(n_items, max_nbrs);
    const int64_t slot_wgs = ks_slot_wgs(n_queries);
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 256, st));
    int h[3] = {0, 0, 0};
    g_score_stats[0] = g_score_stats[1] = g_score_stats[2] = 0;
    bool lists = score_fast_enabled() && max_nbrs <= KF_NBR_MAX;
    // tuning knobs (read per call): history length from which a query is split, and into how
    // many parts (a power of two)
    int heavy_len = KF_HEAVY, heavy_split = KF_SPLIT;
    if (const char *e = getenv("LK_KNN_SCORE_HEAVY")) heavy_len = atoi(e) > 0 ? atoi(e) : heavy_len;
    if (const char *e = getenv("LK_KNN_SCORE_SPLIT")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) heavy_split = v;
    }
    int64_t fast_wgs = 0;
    if (lists) {
        hipLaunchKernelGGL(seg_max_kernel, dim3(1), dim3(1024), 0, st, d_tgt_ptr, n_queries,
                           status, d_ref_ptr, heavy, heavy_len);
        LK_HIP_CHECK(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, st));
        LK_HIP_CHECK(hipStreamSynchronize(st));
        g_score_stats[2] = h[2];
        // (a heavy query is up to KF_SPLIT work items)
        fast_wgs = n_queries * heavy_split < KF_MAX_WGS ? n_queries * heavy_split : KF_MAX_WGS;
        if (h[2] > 0) {
            const int64_t fit = (int64_t)(ks_region_bytes(n_items, n_queries, max_nbrs) /
                                          kf_slab_bytes(h[2], max_nbrs));
            if (fit < fast_wgs) fast_wgs = fit;
        }
        lists = h[2] <= KF_NT_MAX && fast_wgs >= 1;
    }
    if (lists) {
        g_score_stats[0] = n_queries;
        hipLaunchKernelGGL(iknn_score_fast_kernel<SWAP>, dim3((unsigned)fast_wgs),
                           dim3(KF_THREADS), 0, st, d_ptr, d_idx, d_val, n_rows, n_items,
                           n_queries, d_ref_ptr, d_ref_items, d_ref_rates, d_tgt_ptr, d_tgt_items,
                           max_nbrs, min_nbrs, slabs, h[2] > 0 ? h[2] : 1, d_out_scores,
                           d_out_counts, status, heavy, heavy_len, heavy_split);
    } else {
        g_score_stats[1] = n_queries;
        hipLaunchKernelGGL(iknn_score_kernel<SWAP>, dim3((unsigned)slot_wgs), dim3(KS_THREADS), 0,
                           st, d_ptr, d_idx, d_val, n_rows, n_items, n_queries, d_ref_ptr,
                           d_ref_items, d_ref_rates, d_tgt_ptr, d_tgt_items, max_nbrs, min_nbrs,
                           slabs, slot_slab, d_out_scores, d_out_counts, status);
    }
    LK_HIP_CHECK(hipGetLastError());
    LK_HIP_CHECK(hipMemcpyAsync(h, status, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    if (h[0] != 0) {
        set_error("similarity is null");
        return LK_E_NAN_SIM;
    }
    if (h[1] != 0) {
        set_error("%s: a matrix row names the same column more than once", what);
        return LK_E_INVALID;
    }
    return LK_OK;
}

}  // namespace lk

extern "C" size_t lk_iknn_score_workspace_bytes(int64_t n_items, int64_t n_queries,
                                                int32_t max_nbrs)
{
    if (n_items < 0 || max_nbrs < 1) return 0;
    return 256 + lk::ks_list_bytes(n_queries) + lk::ks_region_bytes(n_items, n_queries, max_nbrs);
}

extern "C" void lk_knn_score_last_stats(int64_t *out3)
{
    if (out3)
        for (int i = 0; i < 3; ++i) out3[i] = lk::g_score_stats[i];
}

extern "C" int lk_iknn_score_batch(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                                   const float *d_sim_values, int64_t n_items, int64_t n_queries,
                                   const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                                   const float *d_ref_rates, const int64_t *d_tgt_ptr,
                                   const int32_t *d_tgt_items, int32_t max_nbrs, int32_t min_nbrs,
                                   void *d_ws, float *d_out_scores, int32_t *d_out_counts,
                                   void *stream)
{
    LK_REQUIRE(max_nbrs >= 1 && min_nbrs >= 1, "lk_iknn_score_batch: max_nbrs/min_nbrs must be >= 1");
    LK_REQUIRE(n_items >= 0 && n_queries >= 0, "lk_iknn_score_batch: negative size");
    if (n_queries == 0) return LK_OK;
    LK_REQUIRE(d_sim_indptr && d_ref_ptr && d_tgt_ptr && d_ws && d_out_scores && d_out_counts,
               "lk_iknn_score_batch: null pointer");
    return lk::score_batch<false>("lk_iknn_score_batch", d_sim_indptr, d_sim_indices,
                                  d_sim_values, n_items, n_items, n_queries, d_ref_ptr,
                                  d_ref_items, d_ref_rates, d_tgt_ptr, d_tgt_items, max_nbrs,
                                  min_nbrs, d_ws, d_out_scores, d_out_counts,
                                  lk::as_stream(stream));
}

extern "C" int lk_uknn_score_batch(const int64_t *d_rat_indptr, const int32_t *d_rat_indices,
                                   const float *d_rat_values, int64_t n_users, int64_t n_items,
                                   int64_t n_queries, const int64_t *d_nbr_ptr,
                                   const int32_t *d_nbr_rows, const float *d_nbr_sims,
                                   const int64_t *d_tgt_ptr, const int32_t *d_tgt_items,
                                   int32_t max_nbrs, int32_t min_nbrs, void *d_ws,
                                   float *d_out_scores, int32_t *d_out_counts, void *stream)
{
    LK_REQUIRE(max_nbrs >= 1 && min_nbrs >= 1, "lk_uknn_score_batch: max_nbrs/min_nbrs must be >= 1");
    LK_REQUIRE(n_users >= 0 && n_items >= 0 && n_queries >= 0, "lk_uknn_score_batch: negative size");
    if (n_queries == 0) return LK_OK;
    LK_REQUIRE(d_rat_indptr && d_nbr_ptr && d_tgt_ptr && d_ws && d_out_scores && d_out_counts,
               "lk_uknn_score_batch: null pointer");
    return lk::score_batch<true>("lk_uknn_score_batch", d_rat_indptr, d_rat_indices, d_rat_values,
                                 n_users, n_items, n_queries, d_nbr_ptr, d_nbr_rows, d_nbr_sims,
                                 d_tgt_ptr, d_tgt_items, max_nbrs, min_nbrs, d_ws, d_out_scores,
                                 d_out_counts, lk::as_stream(stream));
}

// ---------------------------------------------------------------------------
// Neighbour similarities of user-kNN: sims[q][u] = <user_vectors[u], x_q> for a batch of dense
// query vectors (`nbr_sims = self.user_vectors @ ratings`, src/lenskit/knn/user.py:196): one
// wave per matrix row, lane = query (blocks of 64 queries), the row's entries broadcast.
// x is [n_items x ld_x] (item-major: the queries of one item are contiguous).
// ---------------------------------------------------------------------------
namespace lk {
template <bool IS64>
__global__ __launch_bounds__(256) void csr_rows_dot_kernel(
    const typename IndPtr<IS64>::type *__restrict__ ptr, const int32_t *__restrict__ idx,
    const float *__restrict__ val, int64_t n_rows, const float *__restrict__ x, int64_t ld_x,
    int64_t n_queries, float *__restrict__ out, int64_t ld_out)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = ptr[row], e = ptr[row + 1];
    for (int64_t q0 = 0; q0 < n_queries; q0 += 64) {
        const int64_t q = q0 + lane;
        float acc = 0.f;
        for (int64_t base = b; base < e; base += 64) {
            // 64 entries of the row per coalesced load, then broadcast one by one
            const int64_t me = base + lane;
            const int it = me < e ? idx[me] : 0;
            const float v = me < e ? val[me] : 0.f;
            const int n = (e - base) < 64 ? (int)(e - base) : 64;
            for (int j = 0; j < n; ++j) {
                const int itj = __shfl(it, j, 64);
                const float vj = __shfl(v, j, 64);
                if (q < n_queries) acc = fmaf(vj, x[(int64_t)itj * ld_x + q], acc);
            }
        }
        if (q < n_queries) out[q * ld_out + row] = acc;
    }
}
}  // namespace lk

extern "C" int lk_csr_rows_dot(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                               const float *d_values, int64_t n_rows, const float *d_x,
                               int64_t ld_x, int64_t n_queries, float *d_out, int64_t ld_out,
                               void *stream)
{
    LK_REQUIRE(n_rows >= 0 && n_queries >= 0 && ld_x >= n_queries && ld_out >= n_rows,
               "lk_csr_rows_dot: bad shape");
    if (n_rows == 0 || n_queries == 0) return LK_OK;
    LK_REQUIRE(d_indptr && d_x && d_out, "lk_csr_rows_dot: null pointer");
    hipStream_t st = lk::as_stream(stream);
    const dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::csr_rows_dot_kernel<true>, grid, block, 0, st,
                           static_cast<const int64_t *>(d_indptr), d_indices, d_values, n_rows,
                           d_x, ld_x, n_queries, d_out, ld_out);
    else
        hipLaunchKernelGGL(lk::csr_rows_dot_kernel<false>, grid, block, 0, st,
                           static_cast<const int32_t *>(d_indptr), d_indices, d_values, n_rows,
                           d_x, ld_x, n_queries, d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ---------------------------------------------------------------------------
// EASE (src/lenskit/knn/ease.py).
//
// lk_ease_gram: the dense, regularised co-occurrence Gramian the model inverts,
//     G = X^T X + reg I   (X = the binary users x items matrix; ease.py:111-119),
// from the device similarity build run on unit values (off-diagonal cells: every shared user
// adds 1.0f, exact integers) plus the item counts on the diagonal.  One lane per stored cell.
//
// lk_ease_score_batch: scores = q_vec @ weights (ease.py:161-168) for a batch of queries:
// q_vec is the 0/1 indicator of the query's history, so a query's scores are the SUM of the
// history items' weight rows.  Workgroup = (query, 256-column strip), rows added in history
// order (plain f32 adds: deterministic).
// ---------------------------------------------------------------------------
namespace lk {
__global__ void ease_gram_fill_kernel(const int64_t *__restrict__ ptr,
                                      const int32_t *__restrict__ idx,
                                      const float *__restrict__ val, int64_t n,
                                      float *__restrict__ g, int64_t ld)
{
    const int64_t row = blockIdx.x;
    const int64_t b = ptr[row], e = ptr[row + 1];
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) g[row * ld + idx[i]] = val[i];
}

__global__ void ease_gram_diag_kernel(const int32_t *__restrict__ counts, int64_t n, float reg,
                                      float *__restrict__ g, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i * ld + i] = (float)counts[i] + reg;
}

__global__ __launch_bounds__(256) void ease_score_kernel(
    const int64_t *__restrict__ hist_ptr, const int32_t *__restrict__ hist_items,
    const float *__restrict__ w, int64_t n_items, int64_t ld_w, float *__restrict__ out,
    int64_t ld_out)
{
    const int64_t q = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_items) return;
    const int64_t b = hist_ptr[q], e = hist_ptr[q + 1];
    float acc = 0.f;
    for (int64_t i = b; i < e; ++i) {
        const int32_t it = hist_items[i];
        if (it >= 0 && it < n_items) acc += w[(int64_t)it * ld_w + c];
    }
    out[q * ld_out + c] = acc;
}
}  // namespace lk

extern "C" int lk_ease_gram(const int64_t *d_cooc_indptr, const int32_t *d_cooc_indices,
                            const float *d_cooc_values, const int32_t *d_item_counts,
                            int64_t n_items, float reg, float *d_out, int64_t ld_out, void *stream)
{
    LK_REQUIRE(n_items >= 0 && ld_out >= n_items, "lk_ease_gram: bad shape");
    if (n_items == 0) return LK_OK;
    LK_REQUIRE(d_cooc_indptr && d_item_counts && d_out, "lk_ease_gram: null pointer");
    hipStream_t st = lk::as_stream(stream);
    LK_HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t)n_items * ld_out * sizeof(float), st));
    hipLaunchKernelGGL(lk::ease_gram_fill_kernel, dim3((unsigned)n_items), dim3(256), 0, st,
                       d_cooc_indptr, d_cooc_indices, d_cooc_values, n_items, d_out, ld_out);
    hipLaunchKernelGGL(lk::ease_gram_diag_kernel, dim3((unsigned)((n_items + 255) / 256)),
                       dim3(256), 0, st, d_item_counts, n_items, reg, d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_ease_score_batch(const int64_t *d_hist_ptr, const int32_t *d_hist_items,
                                   int64_t n_queries, const float *d_weights, int64_t n_items,
                                   int64_t ld_w, float *d_out, int64_t ld_out, void *stream)
{
    LK_REQUIRE(n_queries >= 0 && n_items >= 0 && ld_w >= n_items && ld_out >= n_items,
               "lk_ease_score_batch: bad shape");
    if (n_queries == 0 || n_items == 0) return LK_OK;
    LK_REQUIRE(d_hist_ptr && d_weights && d_out, "lk_ease_score_batch: null pointer");
    LK_REQUIRE(n_queries <= 65535, "lk_ease_score_batch: at most 65535 queries per call");
    hipLaunchKernelGGL(lk::ease_score_kernel,
                       dim3((unsigned)((n_items + 255) / 256), (unsigned)n_queries), dim3(256), 0,
                       lk::as_stream(stream), d_hist_ptr, d_hist_items, d_weights, n_items, ld_w,
                       d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
