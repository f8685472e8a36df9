// iknn_recommend.hip -- item-kNN "score every item + top-N" for a BATCH of queries, gfx950.
//
// What `pipelines/iknn-explicit.toml`'s recommender does per query, one query per pipeline run
// (src/lenskit/batch/_runner.py:283-308): candidates = all training items minus the query's
// own (src/lenskit/basic/candidates.py:77-94), `ItemKNNScorer.__call__` over them
// (src/lenskit/knn/item.py:231-295 -> `score_explicit` / `score_implicit`,
// src/accel/knn/item_score.rs:23-111, one `ScoreAccumulator` per item, accum.rs), item means
// added back (item.py:282), then `TopNRanker` (src/lenskit/basic/topn.py:45-69): the n largest
// non-NaN scores, descending.
//
// The scores are the reference accumulator's BIT FOR BIT (iknn_score.hip explains why that
// needs its evaluation order: the hits of a target in history order, a vector while there is
// room, std's BinaryHeap replayed beyond `max_nbrs`, sequential sums with product and sum
// rounded separately).  `iknn_score_kernel` gets that order from one workgroup barrier per
// history row; here NOTHING waits on a barrier:
//
//  * the catalogue is cut into WINDOWS of RW = 4096 targets; a task is (query, window) and
//    belongs to ONE WAVE.  A wave executes its LDS operations in program order, so when it walks
//    the history rows in order, per-target cursors in its private 16 KB of LDS come out in
//    history order by construction -- no atomics between waves, no sort, no barrier;
//  * the part of a similarity row that falls into a window is found through a per-call table
//    (`row_windows_kernel`: lower bounds of the window borders in every row -- rows are sorted
//    by column), so a wave touches only its own entries;
//  * pass 1 counts the window's hits per target (`ds_add_u32`, fire and forget), an in-wave
//    scan turns counts into list offsets, pass 2 re-walks the rows and drops every hit
//    (weight, value) at its target's cursor (`ds_add_rtn_u32`) -- the lists of a window are
//    contiguous, in target order, each in history order;
//  * pass 3: lane l owns targets 64 l .. 64 l + 63 and streams their lists (one contiguous run
//    per lane): <= max_nbrs hits = the vector's sequential sums; more = the BinaryHeap replay of
//    iknn_score.hip on a per-lane scratch; the score (+ item mean) replaces the cursor in LDS and
//    the window leaves as one coalesced row segment of the score panel (NaN where nothing was
//    scored: no panel memset).
//  Segments of <= 64 entries (row x window) are loaded 16 at a time, ahead of their use.
//
// The panel rows then lose the query's own items (NaN) and go through the dense path's selection
// (`row_topn_kernel` / full sort of topk.hip): n largest non-NaN, descending, ties by lower item.
//
// Bound: the similarity rows of every history item are streamed twice (count + fill: 16 B per hit
// of algorithmic traffic, L2 / MALL resident for a truncated model) and every hit costs two LDS
// atomics; the per-window fixed cost (zero, scan, score sweep, copy-out) is what a short history
// pays.  bench.py reports the call against the HBM roofline on those bytes.
//
// Round 6: the kernel described above (`iknn_score_all_kernel`, "the list kernel") is the FALLBACK.
// The product path is `iknn_score_acc_kernel` further down ("the accumulating kernel"): the same
// tasks and windows, but no lists -- every weight is ADDED to its target's LDS cell by
// `ds_add_f32`, whose same-address adds the LDS applies in lane order (probed on the device like
// the cursor order), so the cell is the vector's sequential sum bit for bit; targets beyond
// max_nbrs hits are queued, their lists built by `iknn_heavy_gather_kernel` and replayed by
// `iknn_heap_replay_kernel`.  7.0 -> 5.0 ms on the cfg3 batch.
#include <type_traits>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)  // weight * value is rounded before it is added (accum.rs:128-130)

namespace lk {
// selection over a score panel (topk.hip)
size_t panel_topn_workspace_bytes(int64_t rows, int64_t n_items, int32_t n);
int panel_topn(const float *panel, int64_t ld_s, int64_t rows, int64_t n_items, int32_t n,
               void *sort_ws, int32_t *out_idx, float *out_score, hipStream_t st,
               const unsigned *class_max = nullptr, int classes_per_row = 0);

namespace rec {

#ifndef LK_REC_RW
#define LK_REC_RW 4096
#endif
constexpr int RW = LK_REC_RW;   // targets per window = per wave (a multiple of 512)
static_assert(RW % 512 == 0 && RW <= 4096,
              "phase B's queue words pack (count << 12) | target: a window holds at most 4096 targets");
constexpr int TPL = RW / 64;    // targets per lane in the scan / sweep (lane-owned runs)
static_assert(RW % 512 == 0, "window = 64 lanes x groups of 8 targets");
constexpr int RWAVES = 4;       // waves per workgroup (each on its own window task)
constexpr int RTHREADS = RWAVES * 64;
constexpr int RD = 16;          // segments in flight per wave
constexpr int RPAD = RW + 64;  // cursor array: index t + t / TPL (conflict-free lane-owned runs)

__device__ __forceinline__ int cidx(int t) { return t + t / TPL; }

#ifdef LK_REC_PHASES
// Diagnostic build only (tools/knnrec_phases.py): shader-clock cycles per phase summed over all
// wave tasks -- [0] task set-up, [1] zero, [2] count walk, [3] scan + region, [4] fill walk,
// [5] score sweep, [6] copy-out, [7] tasks
__device__ unsigned long long *lk_rec_phase_buf;
#define LK_RP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#ifdef LK_REC_WALKPH  // (the slots hold walk_acc's inner phases instead: a one-off diagnostic)
#define LK_RP_ADD(i, a, b) (void)(a)
#else
#define LK_RP_ADD(i, a, b) ph[i] += (b) - (a)
#endif
#else
#define LK_RP_T(var)
#define LK_RP_ADD(i, a, b)
#endif

// a target with more than max_nbrs hits: its accumulator is the reference's BinaryHeap replay --
// a long chain of dependent reads and writes of the heap array.  On a per-lane scratch in HBM
// that chain is ~10 k memory latencies for a 780-hit target: 12 of the 13 ms of the heaviest
// cfg3 batch.  The sweep therefore only QUEUES such targets; iknn_heap_replay_kernel replays
// them, one lane per target, on heaps in LDS.
struct OvfEntry {
    int ql, item, cnt, pad;
    unsigned long long list;  // offset of the target's hit list in the hits buffer
};

__device__ __forceinline__ int64_t readlane64(int64_t v, int l)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// woff[r * (nwin + 1) + b] = entries of similarity row r with column < b * RW
__global__ void row_windows_kernel(const int64_t *__restrict__ s_ptr,
                                   const int32_t *__restrict__ s_idx, int64_t n_rows, int nwin,
                                   unsigned *__restrict__ woff)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * (nwin + 1)) return;
    const int64_t r = i / (nwin + 1);
    const int64_t key = (i % (nwin + 1)) * RW;
    const int64_t base = s_ptr[r];
    int64_t lo = base, hi = s_ptr[r + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)s_idx[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    woff[i] = (unsigned)(lo - base);
}

// the reference's accumulator beyond max_nbrs entries, for ONE lane on its own scratch (the code
// of iknn_score.hip's KfHeap; see there for the correspondence with std's BinaryHeap)
struct LaneHeap {
    float *w, *v;
    int len;
    __device__ __forceinline__ void sift_up(int pos, float ew, float ev)
    {
        while (pos > 0) {
            const int parent = (pos - 1) >> 1;
            if (ew >= w[parent]) break;
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
            if (w[child] >= w[child + 1]) child += 1;
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

// One walk over the history rows of a query for one window.  FILL = false: count the hits per
// target; FILL = true: drop (weight, value) at the target's cursor.  The (row x window) pieces are
// cut into segments of <= 64 entries; the column (and weight) loads of RD segments are in flight
// ahead of the LDS updates, which happen strictly in history order.
template <bool FILL, bool EXPL>
__device__ __forceinline__ void walk(unsigned *__restrict__ c, const int64_t *__restrict__ s_ptr,
                                     const int32_t *__restrict__ s_idx,
                                     const float *__restrict__ s_val,
                                     const unsigned *__restrict__ woff, int nwin, int win,
                                     int64_t n_items, const int32_t *__restrict__ ref_items,
                                     const float *__restrict__ ref_rates, int64_t rb, int64_t re,
                                     float2 *__restrict__ hits, int *__restrict__ status, int lane)
{
    const int w0 = win * RW;
    for (int64_t r0 = rb; r0 < re; r0 += 64) {
        // descriptors of up to 64 history rows: lane = row
        int64_t a = 0;
        int n = 0;
        float rate = 0.f;
        if (r0 + lane < re) {
            const int ri = ref_items[r0 + lane];
            if (ri >= 0 && ri < n_items) {  // null reference rows are skipped (item_score.rs:38-49)
                const unsigned *wo = woff + (int64_t)ri * (nwin + 1) + win;
                const unsigned o0 = wo[0], o1 = wo[1];
                a = s_ptr[ri] + o0;
                n = (int)(o1 - o0);
            }
            if (EXPL) rate = ref_rates[r0 + lane];
        }
        unsigned long long mask = __ballot(n > 0);
        if (!mask) continue;
        // issue cursor over the segments of the rows in `mask`, ascending = history order
        int j = __builtin_ctzll(mask);
        int nj = __builtin_amdgcn_readlane(n, j);
        int64_t aj = readlane64(a, j);
        float rj = bcast(rate, j);
        int base = 0;
        bool more = true;
        int t_[RD], cnt_[RD];
        float s_[RD], r_[RD];
#pragma unroll
        for (int d = 0; d < RD; ++d) cnt_[d] = 0;
        // (every load UNCONDITIONAL: a load inside a branch makes hipcc drain the whole memory
        // queue before it; a slot issued past the end re-reads the last row's first entry and is
        // marked empty)
        auto issue = [&](int d) {
            const int left = more ? nj - base : 0;
            const int cn = left < 64 ? left : 64;
            const int off = more ? base + (lane < cn ? lane : cn - 1) : 0;
            const int64_t e = aj + off;
            t_[d] = s_idx[e] - w0;
            if (FILL) s_[d] = s_val[e];
            r_[d] = rj;
            cnt_[d] = cn;
            if (more) {  // wave-uniform cursor update (no memory operation inside)
                base += 64;
                if (base >= nj) {
                    mask &= mask - 1;
                    if (mask) {
                        j = __builtin_ctzll(mask);
                        nj = __builtin_amdgcn_readlane(n, j);
                        aj = readlane64(a, j);
                        rj = bcast(rate, j);
                        base = 0;
                    } else {
                        more = false;
                        base = 0;
                    }
                }
            }
        };
        auto consume = [&](int d) {
            if (lane < cnt_[d]) {
                if (!FILL) {
                    atomicAdd(&c[cidx(t_[d])], 1u);  // ds_add_u32, no return
                } else {
                    const unsigned pos = atomicAdd(&c[cidx(t_[d])], 1u);
                    if (s_[d] != s_[d]) atomicCAS(status, 0, 1);  // accum.rs:146-151
                    hits[pos] = float2{s_[d], r_[d]};
                }
            }
            cnt_[d] = 0;
        };
        // batches of RD segments: all loads of a batch are issued back to back, then consumed in
        // order -- one memory latency per RD segments.  (A rolling ring, re-issuing a slot right
        // after it is consumed, makes hipcc wait for ALL outstanding loads -- vmcnt(0) -- at every
        // consume, i.e. one full latency per segment: 26 ms instead of 3 for the cfg3 batch.)
        while (more) {
#pragma unroll
            for (int d = 0; d < RD; ++d) issue(d);
#pragma unroll
            for (int d = 0; d < RD; ++d)
                if (cnt_[d] > 0) consume(d);  // wave-uniform
        }
    }
}

// ---- the same walk with the (row x window) pieces PACKED (round 5) -----------------------------
// With a truncated model (save_nbrs = 100) a history row leaves ~6 of its 100 entries in a
// 4096-target window: the walk above spends a whole 64-lane load and a whole 64-lane LDS atomic on
// each such piece -- 10 % of the lanes busy, and the call is bound by the NUMBER of pieces (one
// memory latency per RD of them).  Here the pieces of a 64-row chunk are one stream of
// T = sum n_j entries, taken 64 at a time: lane l of a batch holds stream position p, finds its
// row j (the last row whose exclusive prefix is <= p: a 6-step binary search over the chunk's 64
// prefixes in LDS) and its entry a_j + (p - prefix_j).  Counting needs no order.  Filling needs
// the hits of a TARGET in history order: stream order is history order, batches follow each other
// in program order, and inside a batch two lanes that hit the same target take their cursor
// values in LANE order -- the LDS resolves same-address atomics of one instruction in ascending
// lane order on this hardware; `lds_atomic_order_probe_kernel` checks exactly that on the device
// before the packed walk is ever used, and without it the walk above serves (LK_REC_PACKED=0
// forces that).  Scores stay the reference accumulator's, bit for bit (tests/
// test_gpu_iknn_recommend.py, the bench's recommend parity).
#ifndef LK_REC_RDP
#define LK_REC_RDP 8
#endif
constexpr int RDP = LK_REC_RDP;  // batches (of 64 entries) in flight per wave
constexpr int LQ_CAP = 192;  // per-wave queue of "longer" targets of a window (sweep, phase B)

// What a packed walk does with every entry: count it / drop it at its target's cursor / add its
// weight / add weight x rating to the target's LDS cell (the last two: the accumulating kernel below)
constexpr int WALK_COUNT = 0, WALK_FILL = 1, WALK_ADD_W = 2, WALK_ADD_WV = 3;

// ds_add_f32 without return: same-address adds of one instruction are applied in ascending lane
// order, IEEE round-to-nearest, denormals kept (probed on the device before the kernel is used)
__device__ __forceinline__ void lds_fadd(unsigned *cell, float v)
{
    (void)__hip_atomic_fetch_add(reinterpret_cast<float *>(cell), v, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE, bool EXPL>
__device__ __forceinline__ void walk_packed(
    unsigned *__restrict__ c, unsigned *__restrict__ dsc, const int64_t *__restrict__ s_ptr,
    const int32_t *__restrict__ s_idx, const float *__restrict__ s_val,
    const unsigned *__restrict__ woff, int nwin, int win, int64_t n_items,
    const int32_t *__restrict__ ref_items, const float *__restrict__ ref_rates, int64_t rb,
    int64_t re, float2 *__restrict__ hits, int *__restrict__ status, int lane)
{
    const int w0 = win * RW;
    // descriptor table of the chunk (wave-private LDS): [0..63] exclusive prefix, [64..127] /
    // [128..191] low / high word of (piece address - prefix), [192..255] the row's rating
    unsigned *d_ex = dsc, *d_lo = dsc + 64, *d_hi = dsc + 128;
    float *d_rt = reinterpret_cast<float *>(dsc + 192);
    for (int64_t r0 = rb; r0 < re; r0 += 64) {
        int64_t a = 0;
        int n = 0;
        float rate = 0.f;
        if (r0 + lane < re) {
            const int ri = ref_items[r0 + lane];
            if (ri >= 0 && ri < n_items) {  // null reference rows are skipped (item_score.rs:38-49)
                const unsigned *wo = woff + (int64_t)ri * (nwin + 1) + win;
                const unsigned o0 = wo[0], o1 = wo[1];
                a = s_ptr[ri] + o0;
                n = (int)(o1 - o0);
            }
            if (EXPL) rate = ref_rates[r0 + lane];
        }
        unsigned incl = (unsigned)n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0) continue;  // (wave-uniform)
        const unsigned excl = incl - (unsigned)n;
        const int64_t rel = a - (int64_t)excl;
        wave_lds_sync();  // the previous chunk's readers are done
        d_ex[lane] = excl;
        d_lo[lane] = (unsigned)rel;
        d_hi[lane] = (unsigned)((unsigned long long)rel >> 32);
        d_rt[lane] = rate;
        wave_lds_sync();
        for (unsigned p0 = 0; p0 < total; p0 += 64u * RDP) {
            int t_[RDP];
            float s_[RDP], r_[RDP];
            bool live_[RDP];
            // (every load unconditional: positions past the end re-read the stream's last entry)
#pragma unroll
            for (int d = 0; d < RDP; ++d) {
                const unsigned p = p0 + 64u * d + lane;
                live_[d] = p < total;
                const unsigned pc = live_[d] ? p : total - 1;
                // the last row j with prefix_j <= pc (rows without entries share the prefix of
                // the next row with entries, which comes after them)
                int lo = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (d_ex[lo + step] <= pc) lo += step;
                const int64_t e = (int64_t)(((unsigned long long)d_hi[lo] << 32) | d_lo[lo]) +
                                  (int64_t)pc;
                t_[d] = s_idx[e] - w0;
                if (MODE != WALK_COUNT) s_[d] = s_val[e];
                if (MODE == WALK_FILL || MODE == WALK_ADD_WV) r_[d] = d_rt[lo];
            }
#pragma unroll
            for (int d = 0; d < RDP; ++d) {
                if (p0 + 64u * d >= total) break;  // (wave-uniform)
                if (live_[d]) {
                    if (MODE == WALK_COUNT) {
                        atomicAdd(&c[cidx(t_[d])], 1u);  // ds_add_u32, no return
                    } else if (MODE == WALK_FILL) {
                        const unsigned pos = atomicAdd(&c[cidx(t_[d])], 1u);  // lane order
                        if (s_[d] != s_[d]) atomicCAS(status, 0, 1);  // accum.rs:146-151
                        hits[pos] = float2{s_[d], r_[d]};
                    } else if (MODE == WALK_ADD_W) {
                        if (s_[d] != s_[d]) atomicCAS(status, 0, 1);  // accum.rs:146-151
                        lds_fadd(&c[cidx(t_[d])], s_[d]);  // lane order
                    } else {
                        lds_fadd(&c[cidx(t_[d])], s_[d] * r_[d]);  // product rounded, then added
                    }
                }
            }
        }
    }
}

// The packed walk of the accumulating kernel: the same stream of entries, with the chunk
// descriptors PIPELINED -- a walk is a chain of dependent loads per 64-row chunk (history rows ->
// their window offsets and row starts -> the entries), one memory latency each, and that chain is
// what a walk costs.  Here the history rows of chunk k + 2 and the descriptors of chunk k + 1 are
// requested before chunk k's entries, so a chunk costs ONE latency after a two-latency prologue;
// and the first walk of a task RECORDS the descriptors of its first KC chunks in registers
// (CACHE = 1), which the second and third walk replay (CACHE = 2) without loading anything but
// entries.
#ifndef LK_REC_ACC_WAVES
#define LK_REC_ACC_WAVES 2  // waves per SIMD the LDS lets the accumulating kernel have (RW = 4096)
#endif
constexpr int ACC_CAPB = 8192;                  // stream positions the first-position bitmap covers
static_assert(ACC_CAPB >= RW, "a single row piece (<= RW entries) must fit the bitmap");
#ifndef LK_REC_ACC_TAGS
#define LK_REC_ACC_TAGS 256
#endif
constexpr int ACC_TAGS = LK_REC_ACC_TAGS;                     // slots of the duplicate detector (see walk_acc)
constexpr int ACC_DSC = 192 + ACC_CAPB / 32 + 4 + ACC_TAGS;  // words of wave-private table + bitmap + tags
constexpr int KC = 4;
struct DescCache {  // (scalars, not arrays: an array indexed by the chunk number goes to scratch)
    int64_t a0, a1, a2, a3;
    int n0, n1, n2, n3;
    float r0, r1, r2, r3;
};
struct ChunkDesc {
    int64_t a;
    int n;
    float rate;
};
__device__ __forceinline__ ChunkDesc cache_get(const DescCache &dc, int k)
{
    // (the candidates pass through an empty asm: left alone, the optimiser folds the selects into
    // ONE load at a selected address -- and the twelve scalars into a scratch array)
    int64_t a0 = dc.a0, a1 = dc.a1, a2 = dc.a2, a3 = dc.a3;
    int n0 = dc.n0, n1 = dc.n1, n2 = dc.n2, n3 = dc.n3;
    float r0 = dc.r0, r1 = dc.r1, r2 = dc.r2, r3 = dc.r3;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3));
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    ChunkDesc d;
    d.a = k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
    d.n = k == 0 ? n0 : k == 1 ? n1 : k == 2 ? n2 : n3;
    d.rate = k == 0 ? r0 : k == 1 ? r1 : k == 2 ? r2 : r3;
    return d;
}
__device__ __forceinline__ DescCache cache_set(DescCache dc, int k, const ChunkDesc &d)
{
    dc.a0 = k == 0 ? d.a : dc.a0;
    dc.a1 = k == 1 ? d.a : dc.a1;
    dc.a2 = k == 2 ? d.a : dc.a2;
    dc.a3 = k == 3 ? d.a : dc.a3;
    dc.n0 = k == 0 ? d.n : dc.n0;
    dc.n1 = k == 1 ? d.n : dc.n1;
    dc.n2 = k == 2 ? d.n : dc.n2;
    dc.n3 = k == 3 ? d.n : dc.n3;
    dc.r0 = k == 0 ? d.rate : dc.r0;
    dc.r1 = k == 1 ? d.rate : dc.r1;
    dc.r2 = k == 2 ? d.rate : dc.r2;
    dc.r3 = k == 3 ? d.rate : dc.r3;
    return dc;
}

// inclusive scan over the 64 lanes in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16,
// row_bcast 15 / 31 across them) -- `__shfl_up` is a ds_bpermute per step: six trips through the
// LDS queue, the busiest unit of this kernel
__device__ __forceinline__ unsigned wave_incl_scan_dpp(unsigned x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return (unsigned)v;
}

template <int MODE, bool EXPL, int CACHE>
__device__ __forceinline__ DescCache walk_acc(
    unsigned *__restrict__ c, unsigned *__restrict__ dsc, const int64_t *__restrict__ s_ptr,
    const int32_t *__restrict__ s_idx, const float *__restrict__ s_val,
    const unsigned *__restrict__ woff, int nwin, int win, int64_t n_items,
    const int32_t *__restrict__ ref_items, const float *__restrict__ ref_rates, int64_t rb,
    int64_t re, int *__restrict__ status, int lane, DescCache dc, bool mark_own
#ifdef LK_REC_WALKPH
    , unsigned long long *wph
#endif
    )
{
#ifdef LK_REC_WALKPH
#define WPH_T(v) unsigned long long v = 0; if (MODE == WALK_ADD_W) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); }
#define WPH_ADD(i, a, b) if (MODE == WALK_ADD_W) wph[i] += (b) - (a)
#else
#define WPH_T(v)
#define WPH_ADD(i, a, b)
#endif
    static_assert(MODE != WALK_FILL, "the accumulating kernel keeps no lists");
    const int w0 = win * RW;
    // wave-private LDS: [0..127] (piece address - first position) of the rows with entries, by
    // rank; [128..191] their ratings; then the bitmap of first positions
    uint2 *d_rel = reinterpret_cast<uint2 *>(dsc);
    float *d_rt = reinterpret_cast<float *>(dsc + 128);
    unsigned *d_bm = dsc + 192;
    // `ds_add_f32` costs the LDS ~2.3 cycles per ACTIVE LANE (144 per full instruction; an integer
    // atomic, a read or a write of 64 random cells: 6 ... 7 -- tools/probes/lds_atomic_rate.hip), so a
    // float walk adds with it only where it must: where two lanes of the instruction may hit the same
    // cell.  Every lane posts (batch number, lane) to a slot of a small tag table chosen by its target
    // (`ds_max_u32`) and reads the slot back: the lane that finds its own stamp is the HIGHEST lane
    // of its slot -- it adds by read / v_add / write; the others (a lower lane with the same target,
    // or with a target that shares the slot: ~7 of 64) add atomically FIRST.  For one cell that is
    // the lower lanes in lane order, then the highest: the instruction's lane order, as before.
    unsigned *d_tag = dsc + 192 + ACC_CAPB / 32 + 4;
    unsigned tag_seq = 1u;
    if (MODE != WALK_COUNT) {
        for (int i = lane; i < ACC_TAGS; i += 64) d_tag[i] = 0u;
    }
    // the history row (and rating) of this lane in the chunk at r0; -1: none
    auto load_row = [&](int64_t r0, int &ri, float &rate) {
        ri = -1;
        rate = 0.f;
        if (r0 + lane < re) {
            ri = ref_items[r0 + lane];
            if (EXPL) rate = ref_rates[r0 + lane];
        }
    };
    auto load_desc = [&](int ri, float rate) {
        ChunkDesc d{0, 0, rate};
        // (the counting walk also MARKS the query's own items of this window -- bit 31 of the
        // count cell: never scored, never queued; candidates.py:77-94)
        if (MODE == WALK_COUNT && mark_own && ri >= w0 && ri < w0 + RW)
            atomicOr(&c[ri - w0], 0x80000000u);
        if (ri >= 0 && ri < n_items) {  // null reference rows are skipped (item_score.rs:38-49)
            const unsigned *wo = woff + (int64_t)ri * (nwin + 1) + win;
            const unsigned o0 = wo[0], o1 = wo[1];
            d.a = s_ptr[ri] + o0;
            d.n = (int)(o1 - o0);
        }
        return d;
    };
    bool nan_seen = false;
    ChunkDesc cur{0, 0, 0.f};
    int ri1 = -1;
    float rt1 = 0.f;
    if (CACHE == 2) {
        cur = cache_get(dc, 0);
        if (KC < 2 && rb + 64 < re) load_row(rb + 64, ri1, rt1);
    } else {
        int ri0;
        float rt0;
        load_row(rb, ri0, rt0);
        if (rb + 64 < re) load_row(rb + 64, ri1, rt1);
        cur = load_desc(ri0, rt0);
    }
    int k = 0;
    for (int64_t r0 = rb; r0 < re; r0 += 64, ++k) {
        WPH_T(w0_);
        ChunkDesc nxt{0, 0, 0.f};
        int ri2 = -1;
        float rt2 = 0.f;
        if (r0 + 64 < re) {
            if (CACHE == 2 && k + 1 < KC)
                nxt = cache_get(dc, k + 1);
            else
                nxt = load_desc(ri1, rt1);
        }
        if (r0 + 128 < re && !(CACHE == 2 && k + 2 < KC)) load_row(r0 + 128, ri2, rt2);
        if (CACHE == 1 && k < KC) dc = cache_set(dc, k, cur);
        const int n = cur.n;
        const unsigned incl = wave_incl_scan_dpp((unsigned)n);
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        // The row of stream position p.  The list kernel's walk finds it by a 6-step binary search
        // over the prefixes (six dependent LDS reads per 64 entries); here the rows WITH entries are
        // numbered (rank), a bitmap holds a 1 at every row's first position, and the row of p is
        // (ones at positions <= p) - 1: one broadcast read of the bitmap's 64 bits at the batch's
        // start, an and + popcount, and the table read.  The bitmap covers ACC_CAPB positions; a
        // chunk with more (a full similarity matrix) is walked in runs of whole rows that fit.
        const unsigned excl = incl - (unsigned)n;
        unsigned done_pos = 0u;
        int j0 = 0;
        while (done_pos < total) {  // (wave-uniform; one turn unless the chunk exceeds the bitmap)
            const unsigned long long over =
                __ballot(lane >= j0 && incl - done_pos > (unsigned)ACC_CAPB);
            const int j1 = over != 0ull ? (int)__builtin_ctzll(over) : 64;
            if (j1 == j0) {  // a row piece beyond the bitmap: impossible (a piece has <= RW entries,
                             // columns are unique) unless the matrix is malformed -- refused
                atomicCAS(status, 0, 2);
                break;
            }
            const unsigned sub_end =
                j1 == 64 ? total : (unsigned)__builtin_amdgcn_readlane((int)excl, j1);
            const unsigned sub_total = sub_end - done_pos;
            const bool in_sub = lane >= j0 && lane < j1 && n > 0;
            const unsigned long long rows = __ballot(in_sub);
            const int nrows = (int)__popcll(rows);
            const unsigned sx = excl - done_pos;  // first position of the lane's row in this run
            const int64_t rel = cur.a - (int64_t)sx;
            wave_lds_sync();  // the previous run's readers are done
            for (unsigned wd = lane; wd < (sub_total >> 5) + 3u; wd += 64u) d_bm[wd] = 0u;
            wave_lds_sync();
            if (in_sub) {
                const int rk = (int)__popcll(rows & ((1ull << lane) - 1ull));
                d_rel[rk] = uint2{(unsigned)rel, (unsigned)((unsigned long long)rel >> 32)};
                if (MODE == WALK_ADD_WV) d_rt[rk] = cur.rate;
                atomicOr(&d_bm[sx >> 5], 1u << (sx & 31u));  // ds_or_b32, no return
            }
            wave_lds_sync();
            WPH_T(w1_);
            WPH_ADD(0, w0_, w1_);
            unsigned before = 0u;  // ones at positions < the batch's start
            for (unsigned p0 = 0; p0 < sub_total; p0 += 64u * RDP) {
                WPH_T(w2_);
                int t_[RDP];
                float s_[RDP], r_[RDP];
                bool live_[RDP];
                // the bitmap words of all RDP batches in ONE read (lane l: word l of the group), handed
                // out by `v_readlane`: a read per batch was a trip through the LDS queue per batch,
                // each waited for before the table read could even be addressed
                unsigned bmw;
                {
                    const unsigned wq = (p0 >> 5) + (unsigned)(lane < 2 * RDP ? lane : 0);
                    bmw = d_bm[wq < (unsigned)(ACC_CAPB / 32 + 3) ? wq : (unsigned)(ACC_CAPB / 32 + 3)];
                }
                // (three passes over the group's batches -- ranks, table reads, entry loads -- so that
                // the eight table reads leave together and are waited for once)
                int rk_[RDP];
                unsigned pc_[RDP];
#pragma unroll
                for (int d = 0; d < RDP; ++d) {
                    const unsigned pb = p0 + 64u * d;
                    const unsigned p = pb + lane;
                    live_[d] = p < sub_total;
                    pc_[d] = live_[d] ? p : sub_total - 1;
                    // (positions past the end re-read the stream's last entry: every load is
                    // unconditional; a batch past the end sees whatever the bitmap holds there --
                    // its rank is overridden below)
                    const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)bmw, 2 * d);
                    const unsigned m1 = (unsigned)__builtin_amdgcn_readlane((int)bmw, 2 * d + 1);
                    const unsigned le0 = lane < 32 ? (0xffffffffu >> (31 - lane)) : 0xffffffffu;
                    const unsigned le1 = lane < 32 ? 0u : (0xffffffffu >> (63 - lane));
                    int rk = (int)before + __builtin_popcount(m0 & le0) +
                             __builtin_popcount(m1 & le1) - 1;
                    rk = rk < nrows - 1 ? rk : nrows - 1;
                    rk_[d] = pb < sub_total ? rk : nrows - 1;
                    before += (unsigned)(__builtin_popcount(m0) + __builtin_popcount(m1));
                }
                uint2 rl_[RDP];
#pragma unroll
                for (int d = 0; d < RDP; ++d) {
                    rl_[d] = d_rel[rk_[d]];
                    if (MODE == WALK_ADD_WV) r_[d] = d_rt[rk_[d]];
                }
#pragma unroll
                for (int d = 0; d < RDP; ++d) {
                    const int64_t e =
                        (int64_t)(((unsigned long long)rl_[d].y << 32) | rl_[d].x) + (int64_t)pc_[d];
                    t_[d] = s_idx[e] - w0;
                    if (MODE != WALK_COUNT) s_[d] = s_val[e];
                }
                WPH_T(w3_);
#pragma unroll
                for (int d = 0; d < RDP; ++d) {
                    if (p0 + 64u * d >= sub_total) break;  // (wave-uniform)
                    // (lanes past the end are masked, not given a dummy cell: the LDS serves atomics
                    // at about a lane per clock, so an idle lane is time saved)
                    if (MODE == WALK_COUNT) {
                        if (live_[d]) atomicAdd(&c[t_[d]], 1u);  // ds_add_u32, no return
                    } else {
                        float val;
                        if (MODE == WALK_ADD_W) {
                            val = s_[d];
                            nan_seen |= live_[d] && val != val;  // accum.rs:146-151
                        } else {
                            val = s_[d] * r_[d];  // product rounded, then added
                        }
                        const unsigned stamp = (tag_seq << 6) | (unsigned)lane;
                        ++tag_seq;
                        unsigned *slot = &d_tag[(unsigned)t_[d] & (unsigned)(ACC_TAGS - 1)];
                        if (live_[d]) atomicMax(slot, stamp);  // ds_max_u32, no return
                        wave_lds_sync();
                        const bool win = live_[d] && *slot == stamp;
                        if (live_[d] && !win) lds_fadd(&c[t_[d]], val);  // (the few, lane order)
                        wave_lds_sync();  // (free; keeps the compiler from moving the read above)
                        if (win) {
                            const float old_ = __builtin_bit_cast(float, c[t_[d]]);
                            c[t_[d]] = __builtin_bit_cast(unsigned, old_ + val);
                        }
                    }
                }
                WPH_T(w4_);
                WPH_ADD(1, w2_, w3_);
                WPH_ADD(2, w3_, w4_);
#ifdef LK_REC_WALKPH
                if (MODE == WALK_ADD_W) wph[4] += 1;
#endif
            }
            done_pos = sub_end;
            j0 = j1;
        }
        cur = nxt;
        ri1 = ri2;
        rt1 = rt2;
#ifdef LK_REC_WALKPH
        if (MODE == WALK_ADD_W) wph[3] += 1;
#endif
    }
    if (MODE == WALK_ADD_W && __ballot(nan_seen) != 0ull) {
        if (lane == 0) atomicCAS(status, 0, 1);  // "similarity is null" (accum.rs:146-151)
    }
    return dc;
#undef WPH_T
#undef WPH_ADD
}

// Does the LDS hand out the old values of same-address `ds_add_rtn_u32`s of ONE instruction in
// ascending lane order?  (What walk_packed's fill pass relies on.)  Several conflict patterns;
// *ok = 0 as soon as a lane's old value differs from the number of lower lanes with its address.
__global__ void lds_atomic_order_probe_kernel(int *__restrict__ ok)
{
    __shared__ unsigned cell[64];
    const int lane = threadIdx.x;
    for (int pat = 0; pat < 12; ++pat) {
        cell[lane] = 0u;
        __syncthreads();
        int addr;
        switch (pat) {
            case 0: addr = 0; break;
            case 1: addr = lane & 1; break;
            case 2: addr = lane % 5; break;
            case 3: addr = (lane * 7) % 13; break;
            case 4: addr = lane >> 4; break;
            case 5: addr = (lane >> 1) & 7; break;
            case 6: addr = (lane * lane) % 11; break;
            case 7: addr = lane & 31; break;       // the two halves of the wave collide
            case 8: addr = 63 - (lane & 7); break;
            case 9: addr = (lane ^ 21) % 3; break;
            case 10: addr = lane < 40 ? 3 : lane;  break;
            default: addr = (lane * 37 + 11) % 17; break;
        }
        const unsigned old = atomicAdd(&cell[addr], 1u);
        __syncthreads();
        // number of lower lanes with the same address
        unsigned want = 0;
        for (int l = 0; l < 64; ++l) {
            const int al = __shfl(addr, l, 64);
            if (l < lane && al == addr) ++want;
        }
        if (old != want) atomicExch(ok, 0);
        __syncthreads();
    }
}

// The same question for `ds_add_f32` (what iknn_score_acc_kernel relies on): is a cell, after the
// same-address float adds of several instructions, the SEQUENTIAL f32 sum of the addends in
// (instruction, lane) order -- round-to-nearest, denormals kept?  Addends of very different
// magnitudes (and a denormal range), so that any other order or rounding shows in the bits.
__global__ void lds_fadd_order_probe_kernel(int *__restrict__ ok)
{
    __shared__ unsigned cell[64];
    const int lane = threadIdx.x;
    for (int pat = 0; pat < 12; ++pat) {
        cell[lane] = 0u;
        __syncthreads();
        float want = 0.f;  // of cell[lane]
        for (int round = 0; round < 3; ++round) {
            int addr;
            switch ((pat + 5 * round) % 12) {
                case 0: addr = 0; break;
                case 1: addr = lane & 1; break;
                case 2: addr = lane % 5; break;
                case 3: addr = (lane * 7) % 13; break;
                case 4: addr = lane >> 4; break;
                case 5: addr = (lane >> 1) & 7; break;
                case 6: addr = (lane * lane) % 11; break;
                case 7: addr = lane & 31; break;
                case 8: addr = 63 - (lane & 7); break;
                case 9: addr = (lane ^ 21) % 3; break;
                case 10: addr = lane < 40 ? 3 : lane; break;
                default: addr = (lane * 37 + 11) % 17; break;
            }
            const int ex = pat < 8 ? ((lane * 5 + round * 3 + pat) % 23) - 11   // 2^-11 .. 2^11
                                   : -140 + ((lane * 3 + round) % 16);           // denormal sums
            float v = __builtin_ldexpf(1.f + 0.37f * (float)((lane * 11 + pat) % 17), ex);
            if ((lane + round) % 3 == 0) v = -v;
            lds_fadd(&cell[addr], v);
            for (int l = 0; l < 64; ++l) {
                const int al = __shfl(addr, l, 64);
                const float vl = __shfl(v, l, 64);
                if (al == lane) want = want + vl;
            }
        }
        __syncthreads();
        if (cell[lane] != __builtin_bit_cast(unsigned, want)) atomicExch(ok, 0);
        __syncthreads();
    }
}

// Task = (query of the batch, window); one wave per task, taken from an atomic counter.
template <bool EXPL, bool PACKED = false>
__global__ __launch_bounds__(RTHREADS) void iknn_score_all_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_items, int nwin, const unsigned *__restrict__ woff,
    int64_t q0, int64_t nq, const int64_t *__restrict__ ref_ptr,
    const int32_t *__restrict__ ref_items, const float *__restrict__ ref_rates,
    const float *__restrict__ item_bias, int max_nbrs, int min_nbrs, float2 *__restrict__ hits,
    const int64_t *__restrict__ q_hit_base, unsigned long long *__restrict__ q_cursor,
    float *__restrict__ panel, int64_t ld, float *__restrict__ heap_scratch,
    int *__restrict__ task_counter, int *__restrict__ status, OvfEntry *__restrict__ ovf,
    int ovf_cap, int *__restrict__ ovf_count)
{
    __shared__ unsigned cur[RWAVES][RPAD];
    __shared__ unsigned dsc_all[PACKED ? RWAVES : 1][PACKED ? 256 : 1];
    // targets with 3 .. max_nbrs hits, left to a second phase of the sweep (see there): [0] the
    // count, then (beg, cnt << 12 | target) pairs
    __shared__ unsigned lq_all[RWAVES][2 + 2 * LQ_CAP];
#ifdef LK_REC_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned *c = cur[wave];
    unsigned *lq = lq_all[wave];
    const int64_t n_tasks = nq * nwin;
    float *hw = heap_scratch +
                ((size_t)(blockIdx.x * RWAVES + wave) * 64 + lane) * (size_t)(max_nbrs + 1) * 2;
    float *hv = hw + (max_nbrs + 1);
    const float nanf_ = __builtin_nanf("");

    for (;;) {
        LK_RP_T(p0);
        int task = 0;
        if (lane == 0) task = atomicAdd(task_counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= n_tasks) break;
        const int64_t ql = task / nwin;  // query-major: the windows of a query run side by side
        const int win = task % nwin;
        const int64_t q = q0 + ql;
        const int w0 = win * RW;
        const int wn = (int)((n_items - w0) < RW ? (n_items - w0) : RW);
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        float *prow = panel + ql * ld + w0;
        if (re == rb) {  // no history: nothing is scored (item.py:238-245)
            for (int i = lane; i < wn; i += 64) prow[i] = nanf_;
            continue;
        }
        LK_RP_T(p1);
        for (int i = lane; i < RPAD; i += 64) c[i] = 0u;
        if (lane == 0) lq[0] = 0u;
        LK_RP_T(p2);

        // ---- pass 1: hits per target ----------------------------------------------------
        if constexpr (PACKED)
            walk_packed<WALK_COUNT, EXPL>(c, dsc_all[wave], s_ptr, s_idx, s_val, woff, nwin, win,
                                     n_items, ref_items, ref_rates, rb, re, nullptr, status, lane);
        else
            walk<false, EXPL>(c, s_ptr, s_idx, s_val, woff, nwin, win, n_items, ref_items,
                              ref_rates, rb, re, nullptr, status, lane);
        LK_RP_T(p3);

        // ---- counts -> list offsets: lane l owns targets 64 l .. 64 l + 63 --------------------
        unsigned run = 0;
        {
            unsigned v[TPL];
#pragma unroll
            for (int i = 0; i < TPL; ++i) v[i] = c[lane * (TPL + 1) + i];
#pragma unroll
            for (int i = 0; i < TPL; ++i) {
                const unsigned x = v[i];
                v[i] = run;
                run += x;
            }
            // exclusive scan of the lanes' totals
            unsigned incl = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const unsigned excl = incl - run;
#pragma unroll
            for (int i = 0; i < TPL; ++i) c[lane * (TPL + 1) + i] = v[i] + excl;
            run = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);  // hits of the window
        }
        const unsigned total = run;
        if (total == 0) {
            for (int i = lane; i < wn; i += 64) prow[i] = nanf_;
            continue;
        }
        // the window's share of the query's hit region
        unsigned long long hb = 0;
        // (128-byte granules: no cache line is shared between the lists of two waves)
        if (lane == 0)
            hb = q_hit_base[ql] +
                 atomicAdd(&q_cursor[ql], (unsigned long long)((total + 15u) & ~15u));
        hb = (unsigned long long)readlane64((int64_t)hb, 0);
        float2 *lists = hits + hb;
        LK_RP_T(p4);

        // ---- pass 2: the hits, each at its target's cursor (history order by construction) ----
        if constexpr (PACKED)
            walk_packed<WALK_FILL, EXPL>(c, dsc_all[wave], s_ptr, s_idx, s_val, woff, nwin, win,
                                    n_items, ref_items, ref_rates, rb, re, lists, status, lane);
        else
            walk<true, EXPL>(c, s_ptr, s_idx, s_val, woff, nwin, win, n_items, ref_items,
                             ref_rates, rb, re, lists, status, lane);
        LK_RP_T(p5);
        // the lists were written by OTHER lanes of this wave: complete the stores before they are
        // read.  WORKGROUP scope -- writer and reader share the CU's L1; an agent-scope release
        // is a `buffer_wbl2` (write back the XCD's whole L2) per task: 15 ms of a 22 ms call
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // ---- pass 3: scores.  After the fill c[t] = END of t's list = start of t + 1's ----------
        // Eight targets at a time: the first two hits of each are requested together (unconditional
        // loads; most lists are that short), so a group costs one memory latency, not one per
        // non-empty target (the sweep was 47 % of the kernel when every target waited by itself).
        // (round 5) Targets are INTERLEAVED over the lanes here -- lane l takes targets l, l + 64,
        // ... -- not the lane-owned runs of the scan: with runs every lane of a load touched a cache
        // line of its own (item means 256 bytes apart, lists far apart), 24 such loads per group,
        // and the CU's address path (one line per cycle, shared by its eight waves) was what the
        // sweep ran at: 88 k cycles per task.  Interleaved, a load's 64 addresses are neighbours
        // (consecutive targets' lists follow each other in the hit region).  A target's list
        // starts where its predecessor's ends: the neighbour lane's `end` (lane 0: the carry).
        unsigned carry = 0u;
        for (int g = 0; g < TPL / 8; ++g) {
            unsigned en[8], bg[8];
            float2 f0[8], f1[8];
            float bias[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) en[u] = c[cidx(64 * (8 * g + u) + lane)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned prev = __shfl_up(en[u], 1, 64);
                const unsigned first =
                    u == 0 ? carry : (unsigned)__builtin_amdgcn_readlane((int)en[u > 0 ? u - 1 : 0], 63);
                bg[u] = lane == 0 ? first : prev;
            }
            carry = (unsigned)__builtin_amdgcn_readlane((int)en[7], 63);
            // (the item means of the group's targets as well: a load inside the per-target branch
            // is one more serialised latency per scored target)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int it = w0 + 64 * (8 * g + u) + lane;
                bias[u] = item_bias ? item_bias[it < n_items ? it : 0] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned cn = en[u] - bg[u];
                const unsigned a0 = cn > 0 ? bg[u] : 0u, a1 = cn > 1 ? bg[u] + 1 : a0;
                f0[u] = lists[a0];
                f1[u] = lists[a1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 64 * (8 * g + u) + lane;
                const unsigned beg = bg[u], end = en[u];
                const int cnt = (int)(end - beg);
                float score = nanf_;
                const int kept = cnt < max_nbrs ? cnt : max_nbrs;
                if (cnt > 0 && kept >= min_nbrs && t < wn) {
                    float tw = 0.f, ws = 0.f;
                    const float2 *l = lists + beg;
                    bool queued = false;
                    if (cnt > max_nbrs && ovf != nullptr) {
                        // Full(heap): left to iknn_heap_replay_kernel when the queue has room
                        const int slot = atomicAdd(ovf_count, 1);
                        if (slot < ovf_cap) {
                            ovf[slot] = OvfEntry{(int)ql, w0 + t, cnt, 0, hb + beg};
                            queued = true;
                        }
                    }
                    bool later = false;
                    if (!queued && cnt > 2 && cnt <= max_nbrs && cnt < (1 << 20)) {
                        // 3 .. max_nbrs hits: one more memory latency per four hits if summed
                        // here, in a loop the whole wave waits in (lanes diverge on the list
                        // length: the sweep was 56 % of the kernel, nearly all of it these
                        // waits).  Left to phase B below, where every lane streams a list of its
                        // own with eight loads in flight.
                        const unsigned slot = atomicAdd(&lq[0], 1u);
                        if (slot < (unsigned)LQ_CAP) {
                            lq[2 + 2 * slot] = beg;
                            lq[3 + 2 * slot] = ((unsigned)cnt << 12) | (unsigned)t;
                            later = true;
                        }
                    }
                    if (queued || later) {
                        // (the panel cell is written by the replay kernel / the cursor cell by
                        // phase B; NaN until then)
                    } else if (cnt <= max_nbrs) {  // Partial(vec): sums in insertion order
                        tw += f0[u].x;
                        if (EXPL) ws += f0[u].x * f0[u].y;
                        if (cnt > 1) {
                            tw += f1[u].x;
                            if (EXPL) ws += f1[u].x * f1[u].y;
                        }
                        int x = 2;
                        for (; x + 4 <= cnt; x += 4) {  // four independent loads, summed in order
                            const float2 h0 = l[x], h1 = l[x + 1], h2 = l[x + 2], h3 = l[x + 3];
                            tw += h0.x;
                            tw += h1.x;
                            tw += h2.x;
                            tw += h3.x;
                            if (EXPL) {
                                ws += h0.x * h0.y;
                                ws += h1.x * h1.y;
                                ws += h2.x * h2.y;
                                ws += h3.x * h3.y;
                            }
                        }
                        for (; x < cnt; ++x) {
                            const float2 h = l[x];
                            tw += h.x;
                            if (EXPL) ws += h.x * h.y;
                        }
                    } else {  // Full(heap) on the per-lane HBM scratch (queue full / no queue)
                        for (int x = 0; x < max_nbrs; ++x) {  // vec.pop() from the back, push each
                            const float2 h = l[max_nbrs - 1 - x];
                            hw[x] = h.x;
                            hv[x] = h.y;
                        }
                        LaneHeap hp{hw, hv, max_nbrs};
                        for (int kk = 1; kk < max_nbrs; ++kk) hp.sift_up(kk, hw[kk], hv[kk]);
                        for (int x = max_nbrs; x < cnt; ++x) {
                            const float2 h = l[x];
                            if (h.x > hw[0]) {  // strictly greater than the minimum (accum.rs:108)
                                hp.push(h.x, h.y);
                                while (hp.len > max_nbrs) hp.pop();
                            }
                        }
                        for (int x = 0; x < max_nbrs; ++x) {
                            tw += hw[x];
                            if (EXPL) ws += hw[x] * hv[x];
                        }
                    }
                    if (!queued && !later) {
                        score = EXPL ? ws / tw : tw;
                        if (item_bias) score = score + bias[u];  // item.py:282 (f32 add)
                    }
                }
                c[cidx(t)] = __builtin_bit_cast(unsigned, score);
            }
        }
        // ---- phase B: the queued targets, a lane per list, eight hits in flight ----------------
        wave_lds_sync();
        {
            const unsigned nq_ = lq[0] < (unsigned)LQ_CAP ? lq[0] : (unsigned)LQ_CAP;
            for (unsigned qi = lane; qi < ((nq_ + 63u) & ~63u); qi += 64) {
                const bool on = qi < nq_;
                const unsigned b0 = on ? lq[2 + 2 * qi] : 0u;
                const unsigned w = on ? lq[3 + 2 * qi] : 0u;
                const int t = (int)(w & 4095u), cnt = (int)(w >> 12);
                const float2 *l = lists + b0;
                float tw = 0.f, ws = 0.f;
                for (int x = 0; x < cnt; x += 8) {  // (lanes diverge on cnt: ONE loop for all lists)
                    float2 h[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) h[u] = l[x + u < cnt ? x + u : cnt - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (x + u < cnt) {  // insertion order (accum.rs: the vector's sums)
                            tw += h[u].x;
                            if (EXPL) ws += h[u].x * h[u].y;
                        }
                    }
                }
                if (on) {
                    float score = EXPL ? ws / tw : tw;
                    if (item_bias) score = score + item_bias[w0 + t];  // item.py:282 (f32 add)
                    c[cidx(t)] = __builtin_bit_cast(unsigned, score);
                }
            }
        }
        wave_lds_sync();
        LK_RP_T(p6);
        // ---- the window's segment of the panel row, coalesced -----------------------------------
        for (int i = lane; i < wn; i += 64) prow[i] = __builtin_bit_cast(float, c[cidx(i)]);
        LK_RP_T(p7);
        LK_RP_ADD(0, p0, p1);
        LK_RP_ADD(1, p1, p2);
        LK_RP_ADD(2, p2, p3);
        LK_RP_ADD(3, p3, p4);
        LK_RP_ADD(4, p4, p5);
        LK_RP_ADD(5, p5, p6);
        LK_RP_ADD(6, p6, p7);
#ifdef LK_REC_PHASES
        ph[7] += 1;
#endif
    }
#ifdef LK_REC_PHASES
    if (lane == 0 && lk_rec_phase_buf)
        for (int i = 0; i < 8; ++i) atomicAdd(&lk_rec_phase_buf[i], ph[i]);
#endif
}

// ---- round 6: the accumulating kernel ---------------------------------------------------------
// The lists of the kernel above exist for ONE reason: a target's hits must be summed in history
// order.  But the LDS applies the same-address `ds_add_f32`s of an instruction in ascending lane
// order and a wave's instructions in program order (probed, like the cursor order the packed walk
// relies on), so a walk that ADDS every weight into its target's LDS cell yields the vector's
// sequential sum bit for bit -- no list, no hit region in HBM, no fence, no sweep that waits on
// list loads (58 % of the list kernel).  Per (query, window) task:
//   walk 1  counts the hits per target (`ds_add_u32`) and ORs bit 31 into the cells of the query's
//           own items (struck here: they are never scored and never queued);
//   start   lane l owns targets 256 g + 4 l + j (four cells per LDS instruction).  The counts become
//   values  the cells' START VALUES: 0 where the vector's sums are the score (min_nbrs <= hits <=
//           max_nbrs), NaN everywhere else -- no hit, too few, own item, or the BinaryHeap case
//           (hits > max_nbrs: queued for iknn_heavy_replay_kernel, which finds the target's hits
//           again and replays them).  NaN + w stays NaN, which is what such a panel cell holds;
//   walk 2  cell[t] += weight;  the 64 sums of a lane move to registers, the cells are zeroed;
//   walk 3  (explicit feedback) cell[t] += weight * rating (product rounded first, accum.rs:128);
//   sweep   score = ws / tw (+ item mean); the window's segment of the panel row leaves straight
//           from the registers, 16 bytes per lane and store, and the lane's largest score goes to
//           the selection kernel as one of the row's class maxima.
// (How a float walk adds: `ds_add_f32` occupies the LDS for 144 cycles per 64-lane instruction, so
// only the lanes that may share a cell with another lane of the instruction use it -- the others
// read, add and write; walk_acc has the tag table that tells them apart, in the same lane order.)
// Three walks instead of two, each of the cheap kind (walk_acc: rows found by bitmap rank, the
// descriptors of the first chunks recorded by walk 1 and replayed by the others).  With the LDS
// holding 16 KiB of cells per wave the kernel runs two waves per SIMD: what a task costs is its
// instruction count and the LDS's atomic rate (DESIGN.md 4.9 has the counters).
template <bool EXPL>
__global__ __launch_bounds__(RTHREADS) __attribute__((amdgpu_waves_per_eu(LK_REC_ACC_WAVES, LK_REC_ACC_WAVES)))
void iknn_score_acc_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_items, int nwin, const unsigned *__restrict__ woff,
    int64_t q0, int64_t nq, const int64_t *__restrict__ ref_ptr,
    const int32_t *__restrict__ ref_items, const float *__restrict__ ref_rates,
    const float *__restrict__ item_bias, int max_nbrs, int min_nbrs, float *__restrict__ panel,
    int64_t ld, int *__restrict__ task_counter, int *__restrict__ status,
    OvfEntry *__restrict__ ovf, int ovf_cap, int *__restrict__ ovf_count, int exclude_refs,
    unsigned *__restrict__ premax)
{
    // one cell per target, no padding: the sweeps move FOUR cells per LDS instruction (lane l owns
    // targets 256 g + 4 l + j) -- with two waves per SIMD a task's time is its instruction count
    __shared__ uint4 cur4[RWAVES][RW / 4];
    __shared__ uint2 dsc_all2[RWAVES][ACC_DSC / 2];
#ifdef LK_REC_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 *c4 = cur4[wave];
    unsigned *c = reinterpret_cast<unsigned *>(c4);
    unsigned *dsc = reinterpret_cast<unsigned *>(dsc_all2[wave]);
    const int64_t n_tasks = nq * nwin;
    const float nanf_ = __builtin_nanf("");
    const unsigned nan_bits = __builtin_bit_cast(unsigned, nanf_);
    constexpr int NG = RW / 256;  // quads per lane
    // a target is scored from the vector's sums iff lo_thr <= hits <= max_nbrs (kept = min(hits,
    // max_nbrs) >= min_nbrs, accum.rs); beyond max_nbrs: the heap's replay
    const unsigned lo_thr = min_nbrs > 1 ? (unsigned)min_nbrs : 1u;
    const unsigned span = (unsigned)max_nbrs >= lo_thr ? (unsigned)max_nbrs - lo_thr : 0u;
    const bool light_possible = (unsigned)max_nbrs >= lo_thr;
    const bool bias16 = (reinterpret_cast<uintptr_t>(item_bias) & 15u) == 0;

    // (the NEXT task's number is requested while this one runs: a returning global atomic is a
    // memory round trip at the head of every task otherwise)
    int next_task = 0;
    if (lane == 0) next_task = atomicAdd(task_counter, 1);
    for (;;) {
        LK_RP_T(p0);
        const int task = __builtin_amdgcn_readfirstlane(next_task);
        if (task >= n_tasks) break;
        if (lane == 0) next_task = atomicAdd(task_counter, 1);
        const int64_t ql = task / nwin;  // query-major: the windows of a query run side by side
        const int win = task % nwin;
        const int64_t q = q0 + ql;
        const int w0 = win * RW;
        const int wn = (int)((n_items - w0) < RW ? (n_items - w0) : RW);
        const int wpad = (int)((ld - w0) < RW ? (ld - w0) : RW);  // (the row's padding: whole quads)
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        float *prow = panel + ql * ld + w0;
        // the window's 64 class maxima (lane l: targets 256 g + 4 l + j) for the selection kernel:
        // order-preserving keys, 0 = nothing valid
        unsigned *pmax = premax ? premax + ((int64_t)ql * nwin + win) * 64 : nullptr;
        if (re == rb) {  // no history: nothing is scored (item.py:238-245)
            for (int i = lane; i < wn; i += 64) prow[i] = nanf_;
            if (pmax) pmax[lane] = 0u;
            continue;
        }
        LK_RP_T(p1);
#pragma unroll
        for (int g = 0; g < NG; ++g) c4[64 * g + lane] = uint4{0u, 0u, 0u, 0u};
        LK_RP_T(p2);

        // ---- walk 1: hits per target --------------------------------------------------------
        const DescCache dc = walk_acc<WALK_COUNT, EXPL, 1>(
            c, dsc, s_ptr, s_idx, s_val, woff, nwin, win, n_items, ref_items, ref_rates, rb, re,
            status, lane, DescCache{0, 0, 0, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f}, exclude_refs != 0
#ifdef LK_REC_WALKPH
            , ph
#endif
            );
        wave_lds_sync();
        LK_RP_T(p3);

        // ---- counts -> the cells' start values: 0 where the vector's sums are the score, NaN
        // everywhere else (no hit, fewer than min_nbrs, or the heap's case) -- NaN + w stays NaN,
        // so the sums of such a target come out as the NaN its panel cell must hold anyway ------
        {
            uint4 cn[NG];
            unsigned mx = 0u;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                cn[g] = c4[64 * g + lane];
                // (bit 31: an own item of the query -- as good as no hit)
                cn[g].x = (int)cn[g].x < 0 ? 0u : cn[g].x;
                cn[g].y = (int)cn[g].y < 0 ? 0u : cn[g].y;
                cn[g].z = (int)cn[g].z < 0 ? 0u : cn[g].z;
                cn[g].w = (int)cn[g].w < 0 ? 0u : cn[g].w;
                mx = max(max(mx, max(cn[g].x, cn[g].y)), max(cn[g].z, cn[g].w));
            }
            if (__ballot(mx != 0u) == 0ull) {  // no hit in this window
                for (int i = lane; i < wn; i += 64) prow[i] = nanf_;
                if (pmax) pmax[lane] = 0u;
                continue;
            }
            // Full(heap) targets (rare): found again and replayed by iknn_heavy_replay_kernel (the
            // queue holds every such target of the batch: the host cuts batches by hits /
            // (max_nbrs + 1))
            if (__ballot(mx > (unsigned)max_nbrs && max_nbrs >= min_nbrs) != 0ull) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const unsigned v[4] = {cn[g].x, cn[g].y, cn[g].z, cn[g].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (v[j] > (unsigned)max_nbrs && max_nbrs >= min_nbrs) {
                            const int t = 256 * g + 4 * lane + j;
                            const int slot = atomicAdd(ovf_count, 1);
                            if (slot < ovf_cap)
                                ovf[slot] = OvfEntry{(int)ql, w0 + t, (int)v[j], 0, 0ull};
                            else
                                atomicCAS(status, 0, 2);  // cannot happen (host bound)
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint4 z;
                z.x = (light_possible && cn[g].x - lo_thr <= span) ? 0u : nan_bits;
                z.y = (light_possible && cn[g].y - lo_thr <= span) ? 0u : nan_bits;
                z.z = (light_possible && cn[g].z - lo_thr <= span) ? 0u : nan_bits;
                z.w = (light_possible && cn[g].w - lo_thr <= span) ? 0u : nan_bits;
                c4[64 * g + lane] = z;
            }
        }
        wave_lds_sync();
        LK_RP_T(p4);

        // ---- walk 2: total weights, in history order by construction ------------------------------
        (void)walk_acc<WALK_ADD_W, EXPL, 2>(c, dsc, s_ptr, s_idx, s_val, woff, nwin, win, n_items,
                                            ref_items, ref_rates, rb, re, status, lane, dc, false
#ifdef LK_REC_WALKPH
                                            , ph
#endif
                                            );
        wave_lds_sync();
        LK_RP_T(p5);
        float4 tw[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint4 x = c4[64 * g + lane];
            tw[g] = float4{__builtin_bit_cast(float, x.x), __builtin_bit_cast(float, x.y),
                           __builtin_bit_cast(float, x.z), __builtin_bit_cast(float, x.w)};
        }
        if constexpr (EXPL) {
#pragma unroll
            for (int g = 0; g < NG; ++g) c4[64 * g + lane] = uint4{0u, 0u, 0u, 0u};
            wave_lds_sync();
            // ---- walk 3: weighted sums ------------------------------------------------------------
            (void)walk_acc<WALK_ADD_WV, EXPL, 2>(c, dsc, s_ptr, s_idx, s_val, woff, nwin, win,
                                                 n_items, ref_items, ref_rates, rb, re, status, lane,
                                                 dc, false
#ifdef LK_REC_WALKPH
                                                 , ph
#endif
                                                 );
            wave_lds_sync();
        }
        LK_RP_T(p6);
        // ---- scores: ws / tw (+ item mean), straight from the registers to the panel row; the item
        // means of the NEXT four quads are requested before this four's stores (a load behind a
        // store waits for the store) -----------------------------------------------------------------
        // (two copies of the sweep, each free of branches: FULL = a whole window and 16-byte
        // aligned means -- every window but the catalogue's last)
        auto sweep = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            auto load_bias = [&](int g) {
                float4 b{0.f, 0.f, 0.f, 0.f};
                if (item_bias) {  // (kernel-uniform)
                    const int t0 = 256 * g + 4 * lane;
                    if constexpr (FULL) {
                        b = *reinterpret_cast<const float4 *>(item_bias + w0 + t0);
                    } else {  // clamped addresses; what they deliver beyond wn is never stored
                        const int last = wn - 1;
                        b.x = item_bias[w0 + (t0 < last ? t0 : last)];
                        b.y = item_bias[w0 + (t0 + 1 < last ? t0 + 1 : last)];
                        b.z = item_bias[w0 + (t0 + 2 < last ? t0 + 2 : last)];
                        b.w = item_bias[w0 + (t0 + 3 < last ? t0 + 3 : last)];
                    }
                }
                return b;
            };
            float best = nanf_;  // fmaxf(NaN, x) = x: stays NaN while nothing valid was seen
            float4 bnext[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bnext[u] = load_bias(u);
#pragma unroll
            for (int sg = 0; sg < NG / 4; ++sg) {
                float4 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = bnext[u];
                if (sg + 1 < NG / 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) bnext[u] = load_bias(4 * (sg + 1) + u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = 4 * sg + u;
                    float4 sc = tw[g];
                    if constexpr (EXPL) {
                        const uint4 x = c4[64 * g + lane];
                        sc.x = __builtin_bit_cast(float, x.x) / tw[g].x;
                        sc.y = __builtin_bit_cast(float, x.y) / tw[g].y;
                        sc.z = __builtin_bit_cast(float, x.z) / tw[g].z;
                        sc.w = __builtin_bit_cast(float, x.w) / tw[g].w;
                    }
                    if (item_bias) {  // item.py:282 (f32 add)
                        sc.x = sc.x + b[u].x;
                        sc.y = sc.y + b[u].y;
                        sc.z = sc.z + b[u].z;
                        sc.w = sc.w + b[u].w;
                    }
                    // (NaN cells: one bit pattern, the one the other kernels write)
                    sc.x = sc.x != sc.x ? nanf_ : sc.x;
                    sc.y = sc.y != sc.y ? nanf_ : sc.y;
                    sc.z = sc.z != sc.z ? nanf_ : sc.z;
                    sc.w = sc.w != sc.w ? nanf_ : sc.w;
                    const int t0 = 256 * g + 4 * lane;
                    if (FULL || t0 < wpad) *reinterpret_cast<float4 *>(prow + t0) = sc;
                    best = __builtin_fmaxf(__builtin_fmaxf(best, sc.x), __builtin_fmaxf(sc.y, sc.z));
                    best = __builtin_fmaxf(best, sc.w);
                }
            }
            if (pmax) pmax[lane] = best == best ? f2key(best) : 0u;
        };
        if (bias16 && wn == RW)
            sweep(std::true_type{});
        else
            sweep(std::false_type{});
        LK_RP_T(p7);
        LK_RP_ADD(0, p0, p1);
        LK_RP_ADD(1, p1, p2);
        LK_RP_ADD(2, p2, p3);
        LK_RP_ADD(3, p3, p4);
        LK_RP_ADD(4, p4, p5);
        LK_RP_ADD(5, p5, p6);
        LK_RP_ADD(6, p6, p7);
#ifdef LK_REC_PHASES
        ph[7] += 1;
#endif
    }
#ifdef LK_REC_PHASES
    if (lane == 0 && lk_rec_phase_buf)
        for (int i = 0; i < 8; ++i) atomicAdd(&lk_rec_phase_buf[i], ph[i]);
#endif
}

// A queued target (more than max_nbrs hits) of the accumulating kernel: its hits are FOUND and
// REPLAYED by one wave, no list in between.  The wave walks the query's history 256 rows at a time
// (four 64-row steps whose chains of dependent loads -- history row -> window offsets and row
// start -> the search's probes -> the weight -- run side by side); a lane looks its row's window
// piece up (`woff`) and searches the target's column in it (rows are sorted by column); the hits
// of a step are taken in lane order = history order (`v_readlane` with the ballot's next set bit)
// and lane 0 feeds them to the reference's accumulator on a heap in the wave's LDS: the first
// max_nbrs fill the vector, the next one turns it into the heap (accum.rs:76-83: vec.pop() from
// the back, push each), every later one is offered (accum.rs:100-118).  Writes the panel cell.
constexpr int GWAVES = 4;
template <bool EXPL>
__global__ __launch_bounds__(GWAVES * 64) void iknn_heavy_replay_kernel(
    const OvfEntry *__restrict__ ovf, const int *__restrict__ ovf_count, int ovf_cap,
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_items, int nwin, const unsigned *__restrict__ woff,
    int64_t q0, const int64_t *__restrict__ ref_ptr, const int32_t *__restrict__ ref_items,
    const float *__restrict__ ref_rates, const float *__restrict__ item_bias, int max_nbrs,
    float *__restrict__ panel, int64_t ld, int *__restrict__ status)
{
    extern __shared__ float heap_lds[];  // [GWAVES][2][max_nbrs + 1]
    int n = *ovf_count;
    if (n > ovf_cap) n = ovf_cap;
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    float *hw = heap_lds + (size_t)wave * 2 * (max_nbrs + 1), *hv = hw + (max_nbrs + 1);
    for (int i = blockIdx.x * GWAVES + wave; i < n; i += (int)gridDim.x * GWAVES) {
        const OvfEntry e = ovf[i];
        const int64_t q = q0 + e.ql;
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        const int win = e.item / RW;
        // ---- the accumulator (lane 0 only) ---------------------------------------------------
        auto sift_up = [&](int pos, float ew, float ev) {
            while (pos > 0) {
                const int parent = (pos - 1) >> 1;
                if (ew >= hw[parent]) break;
                hw[pos] = hw[parent];
                hv[pos] = hv[parent];
                pos = parent;
            }
            hw[pos] = ew;
            hv[pos] = ev;
        };
        int fed = 0;
        float wmin = 0.f;
        auto feed = [&](float hx, float hy) {
            if (fed < max_nbrs) {  // Partial(vec); stored back to front: the order the heap is built in
                hw[max_nbrs - 1 - fed] = hx;
                hv[max_nbrs - 1 - fed] = hy;
                if (fed == max_nbrs - 1) {
                    for (int kk = 1; kk < max_nbrs; ++kk) sift_up(kk, hw[kk], hv[kk]);
                    wmin = hw[0];
                }
            } else if (hx > wmin) {  // strictly greater than the minimum (accum.rs:108)
                // push (sift_up(0, len)), then pop: swap the last element into the root,
                // sift_down_to_bottom(0), sift_up -- std's BinaryHeap, as in iknn_score.hip
                sift_up(max_nbrs, hx, hy);
                const float ew = hw[max_nbrs], ev = hv[max_nbrs];
                const int end = max_nbrs;
                int pos = 0, child = 1;
                const int limit = end >= 2 ? end - 2 : 0;
                while (child <= limit && end >= 2) {
                    if (hw[child] >= hw[child + 1]) child += 1;
                    hw[pos] = hw[child];
                    hv[pos] = hv[child];
                    pos = child;
                    child = 2 * pos + 1;
                }
                if (child == end - 1) {
                    hw[pos] = hw[child];
                    hv[pos] = hv[child];
                    pos = child;
                }
                sift_up(pos, ew, ev);
                wmin = hw[0];
            }
            ++fed;
        };
        // ---- the walk ------------------------------------------------------------------------
        unsigned seen = 0u;
        constexpr int GU = 4;
        for (int64_t r0 = rb; r0 < re; r0 += 64 * GU) {
            int ri[GU];
            int64_t lo[GU], hi[GU], end[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int64_t r = r0 + 64 * u + lane;
                ri[u] = ref_items[r < re ? r : re - 1];
                if (!(r < re)) ri[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const bool ok = ri[u] >= 0 && ri[u] < n_items;
                const int rr = ok ? ri[u] : 0;
                const unsigned *wo = woff + (int64_t)rr * (nwin + 1) + win;
                const int64_t b = s_ptr[rr];
                lo[u] = b + wo[0];
                hi[u] = end[u] = b + wo[1];
                if (!ok) hi[u] = end[u] = lo[u];
            }
            // lower bound of the target's column in each piece; the four searches advance together
            for (;;) {
                bool any = false;
                int64_t mid[GU];
                int col[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    mid[u] = (lo[u] + hi[u]) >> 1;
                    col[u] = s_idx[lo[u] < hi[u] ? mid[u] : 0];  // (unconditional loads: one wait)
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    if (lo[u] < hi[u]) {
                        if (col[u] < e.item)
                            lo[u] = mid[u] + 1;
                        else
                            hi[u] = mid[u];
                        any = any || lo[u] < hi[u];
                    }
                }
                if (__ballot(any) == 0ull) break;
            }
            int colf[GU];
            float w[GU], rate[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int64_t at = lo[u] < end[u] ? lo[u] : 0;
                colf[u] = s_idx[at];
                w[u] = s_val[at];
                if (!(lo[u] < end[u])) colf[u] = -1;
                const int64_t r = r0 + 64 * u + lane;
                rate[u] = ref_rates ? ref_rates[r < re ? r : re - 1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                unsigned long long m = __ballot(colf[u] == e.item);
                // once the heap is full only a weight above its minimum can change it, and the
                // minimum only rises: the hits that cannot pass are not even handed to lane 0
                // (~max_nbrs ln(hits / max_nbrs) of a long list's hits do pass)
                if (seen >= (unsigned)max_nbrs) {
                    const float wm = __builtin_bit_cast(
                        float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wmin)));
                    const unsigned long long pass = __ballot(colf[u] == e.item && w[u] > wm);
                    if (lane == 0) fed += (int)__popcll(m & ~pass);
                    seen += (unsigned)__popcll(m);
                    m = pass;
                } else {
                    seen += (unsigned)__popcll(m);
                }
                while (m != 0ull) {  // (wave-uniform)
                    const int j = (int)__builtin_ctzll(m);
                    m &= m - 1ull;
                    const float hx = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w[u]), j));
                    const float hy = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rate[u]), j));
                    if (lane == 0) feed(hx, hy);
                }
            }
        }
        if (lane == 0) {
            if (seen != (unsigned)e.cnt || fed <= max_nbrs) {
                atomicCAS(status, 0, 2);  // (an internal error: the count walk saw other hits)
            } else {
                float tw = 0.f, ws = 0.f;
                for (int x = 0; x < max_nbrs; ++x) {
                    tw += hw[x];
                    if (EXPL) ws += hw[x] * hv[x];
                }
                float score = EXPL ? ws / tw : tw;
                if (item_bias) score = score + item_bias[e.item];
                panel[(int64_t)e.ql * ld + e.item] = score;
            }
        }
    }
}

// One LANE per queued target: the reference's accumulator replayed on a heap in LDS (slot-major:
// hw[slot * 64 + lane]), the hit list streamed from the hits buffer.  Writes the panel cell.
template <bool EXPL>
__global__ __launch_bounds__(64) void iknn_heap_replay_kernel(
    const OvfEntry *__restrict__ ovf, const int *__restrict__ ovf_count, int ovf_cap,
    const float2 *__restrict__ hits, const float *__restrict__ item_bias, int max_nbrs,
    float *__restrict__ panel, int64_t ld)
{
    extern __shared__ float heap_lds[];  // [2][(max_nbrs + 1)][64]
    int n = *ovf_count;
    if (n > ovf_cap) n = ovf_cap;
    const int lane = threadIdx.x;
    // Targets per wave: the lanes of a wave run in lockstep through loops whose trip counts differ
    // from target to target, and every step waits for its hits to arrive from HBM.  While there
    // are fewer targets than waves (2 276 in the heaviest cfg3 batch), a wave takes ONE target:
    // its 64 lanes fetch the next 64 hits together (one coalesced request, the one after it
    // already in flight) and lane 0 replays them from the registers (`v_readlane`) -- the chain
    // of a 7 817-hit target waits for 122 requests instead of 1 950.
    const int waves = (int)gridDim.x;
    int tpw = (n + waves - 1) / waves;
    tpw = tpw < 1 ? 1 : (tpw > 64 ? 64 : tpw);
    const bool solo = tpw == 1;  // (kernel-uniform)
    for (int i0 = blockIdx.x * tpw; i0 < n; i0 += waves * tpw) {
        const int i = solo ? i0 : i0 + lane;
        const bool mine = solo ? lane == 0 : (lane < tpw && i < n);  // this lane replays a target
        const OvfEntry e = ovf[i < n ? i : n - 1];
        const float2 *l = hits + e.list;
        float *hw = heap_lds + lane, *hv = heap_lds + (size_t)(max_nbrs + 1) * 64 + lane;
        auto W = [&](int s) -> float & { return hw[s * 64]; };
        auto V = [&](int s) -> float & { return hv[s * 64]; };
        auto sift_up = [&](int pos, float ew, float ev) {
            while (pos > 0) {
                const int parent = (pos - 1) >> 1;
                if (ew >= W(parent)) break;
                W(pos) = W(parent);
                V(pos) = V(parent);
                pos = parent;
            }
            W(pos) = ew;
            V(pos) = ev;
        };
        // one hit offered to the full heap (accum.rs:100-118): strictly greater than the minimum
        // -> push (sift_up(0, len)), then pop: swap the last element into the root,
        // sift_down_to_bottom(0), sift_up -- std's BinaryHeap, as in iknn_score.hip
        float wmin = 0.f;
        auto offer = [&](float hx, float hy) {
            if (hx > wmin) {
                sift_up(max_nbrs, hx, hy);
                const float ew = W(max_nbrs), ev = V(max_nbrs);
                const int end = max_nbrs;
                int pos = 0, child = 1;
                const int limit = end >= 2 ? end - 2 : 0;
                while (child <= limit && end >= 2) {
                    if (W(child) >= W(child + 1)) child += 1;
                    W(pos) = W(child);
                    V(pos) = V(child);
                    pos = child;
                    child = 2 * pos + 1;
                }
                if (child == end - 1) {
                    W(pos) = W(child);
                    V(pos) = V(child);
                    pos = child;
                }
                sift_up(pos, ew, ev);
                wmin = W(0);
            }
        };
        if (mine) {
            // Partial -> Full (accum.rs:76-83): vec.pop() from the back, push each
            for (int x = 0; x < max_nbrs; ++x) {
                const float2 h = l[max_nbrs - 1 - x];
                W(x) = h.x;
                V(x) = h.y;
            }
            for (int kk = 1; kk < max_nbrs; ++kk) sift_up(kk, W(kk), V(kk));
            wmin = W(0);
        }
        if (solo) {
            const int cnt = e.cnt;  // (wave-uniform: every lane read the same entry)
            auto fetch = [&](int x0) {
                const int x = x0 + lane;
                return l[x < cnt ? x : cnt - 1];
            };
            float2 nxt = fetch(max_nbrs);
            for (int x0 = max_nbrs; x0 < cnt; x0 += 64) {
                const float2 cur = nxt;
                if (x0 + 64 < cnt) nxt = fetch(x0 + 64);
                const int m = cnt - x0 < 64 ? cnt - x0 : 64;
                for (int j = 0; j < m; ++j) {
                    const float hx = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.x), j));
                    const float hy = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.y), j));
                    if (mine) offer(hx, hy);
                }
            }
        } else if (mine) {
            for (int x0 = max_nbrs; x0 < e.cnt; x0 += 4) {
                float2 h[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) h[u] = l[x0 + u < e.cnt ? x0 + u : e.cnt - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (x0 + u < e.cnt) offer(h[u].x, h[u].y);
            }
        }
        if (mine) {
            float tw = 0.f, ws = 0.f;
            for (int x = 0; x < max_nbrs; ++x) {
                tw += W(x);
                if (EXPL) ws += W(x) * V(x);
            }
            float score = EXPL ? ws / tw : tw;
            if (item_bias) score = score + item_bias[e.item];
            panel[(int64_t)e.ql * ld + e.item] = score;
        }
    }
}

// panel[q][own item] = NaN (candidates = all items minus the query's, candidates.py:77-94)
__global__ void mask_refs_kernel(const int64_t *__restrict__ ref_ptr,
                                 const int32_t *__restrict__ ref_items, int64_t q0, int64_t nq,
                                 int64_t n_items, float *__restrict__ panel, int64_t ld)
{
    const int64_t ql = blockIdx.x;
    if (ql >= nq) return;
    const int64_t s = ref_ptr[q0 + ql], e = ref_ptr[q0 + ql + 1];
    for (int64_t r = s + threadIdx.x; r < e; r += blockDim.x) {
        const int it = ref_items[r];
        if (it >= 0 && it < n_items) panel[ql * ld + it] = __builtin_nanf("");
    }
}

constexpr int64_t REC_PANEL_ROWS = 6144;            // queries per batch at most (cfg3, 10 000 queries: 4.43 ms at
                                                    // 4096, 4.28 ... 4.34 at 5120 ... 8192, 4.45 in one batch)
constexpr int64_t REC_HITS_MIN = (int64_t)1 << 28;  // hit capacity of a batch (2 GiB) unless a
                                                    // single query needs more
constexpr int REC_MAX_WGS = 512 * (4096 / RW > 1 ? 4096 / RW : 1);  // persistent grid: what the LDS lets a CU hold

static inline int64_t ld_items(int64_t n_items) { return (n_items + 63) / 64 * 64; }
static inline int nwindows(int64_t n_items) { return (int)((n_items + RW - 1) / RW); }

struct Layout {
    size_t off_status, off_woff, off_base, off_cursor, off_heap, off_ovf, off_panel, off_hits,
        off_sort, off_pmax, bytes, panel_bytes, ovf_bytes, pmax_bytes;
    int64_t rows, hit_cap, ovf_cap;
    int sets;  // 2: a second score panel and target queue -- batch b + 1 is scored while batch b's
               // panel is still being replayed into / masked / selected from (accumulating kernel)
};
constexpr int REC_OVF_CAP = 1 << 20;  // queued heap targets per batch (more: HBM-scratch path)
// the accumulating kernel queues EVERY target with more than max_nbrs hits (it keeps no lists to
// fall back on): a query has at most min(n_items, hits / (max_nbrs + 1)) of them, the batches
// are cut so that the sum stays inside the queue, and the queue is never longer than this
constexpr int64_t REC_OVF_CAP_ACC = (int64_t)1 << 23;

static inline int64_t heavy_bound(int64_t query_hits, int64_t n_items, int32_t max_nbrs)
{
    const int64_t b = query_hits / ((int64_t)max_nbrs + 1);
    return b < n_items ? b : n_items;
}

static Layout layout(int64_t n_items, int64_t n_queries, int64_t max_query_hits, int32_t max_nbrs,
                     int32_t n)
{
    Layout L;
    int64_t panel_rows = REC_PANEL_ROWS;
    if (const char *e = getenv("LK_REC_PANEL_ROWS")) {  // tuning knob: queries per batch at most
        const long v = atol(e);
        if (v >= 64) panel_rows = v;
    }
    L.rows = n_queries < panel_rows ? (n_queries > 0 ? n_queries : 1) : panel_rows;
    // the score panel is rows x n_items floats: beyond BASELINE's 62 k items the batch shrinks so
    // that the panel stays inside a byte budget (LK_REC_PANEL_GB, default 8: 2 000 queries per
    // batch at 10^6 items) instead of growing with the catalogue
    {
        const char *e = getenv("LK_REC_PANEL_GB");
        const double gb = e && atof(e) > 0 ? atof(e) : 8.0;
        int64_t fit = (int64_t)(gb * (double)(1ull << 30) / ((double)ld_items(n_items) * 4.0));
        // (more queries than one budget's rows: two panels of half the budget)
        if (n_queries > fit || n_queries > panel_rows) fit /= 2;
        if (fit < 64) fit = 64;
        if (L.rows > fit) L.rows = fit;
    }
    L.sets = n_queries > L.rows ? 2 : 1;
    // a batch holds up to REC_HITS_MIN hits (never less than the heaviest query's; a small call
    // does not pay for more than all of its queries could need)
    // (a query's region = its hits + 16 per window: every window's share is rounded up to a
    // 128-byte granule)
    const int64_t per_q = max_query_hits + 16 * (int64_t)nwindows(n_items);
    int64_t cap = REC_HITS_MIN;
    if (n_queries > 0 && cap / per_q >= n_queries) cap = per_q * n_queries;
    L.hit_cap = per_q > cap ? per_q : cap;
    size_t off = 0;
    L.off_status = off;
    off += 256;
    L.off_woff = off;
    off += align_up((size_t)n_items * (nwindows(n_items) + 1) * sizeof(unsigned), 256);
    L.off_base = off;
    off += align_up((size_t)(n_queries > 0 ? n_queries : 1) * sizeof(int64_t), 256);
    L.off_cursor = off;
    off += align_up((size_t)(n_queries > 0 ? n_queries : 1) * sizeof(unsigned long long), 256);
    L.off_heap = off;
    off += align_up((size_t)REC_MAX_WGS * RWAVES * 64 * (size_t)(max_nbrs + 1) * 2 * sizeof(float),
                    256);
    {
        const int64_t per = heavy_bound(max_query_hits, n_items, max_nbrs);
        int64_t need = per * (n_queries < L.rows ? n_queries : L.rows);
        if (need > REC_OVF_CAP_ACC) need = REC_OVF_CAP_ACC;
        if (need < per) need = per;  // (a single query always fits)
        L.ovf_cap = need > REC_OVF_CAP ? need : REC_OVF_CAP;
    }
    L.off_ovf = off;
    L.ovf_bytes = align_up((size_t)L.ovf_cap * sizeof(OvfEntry), 256);
    off += L.ovf_bytes * (size_t)L.sets;
    L.off_panel = off;
    L.panel_bytes = align_up((size_t)L.rows * ld_items(n_items) * sizeof(float), 256);
    off += L.panel_bytes * (size_t)L.sets;
    L.off_hits = off;
    off += align_up((size_t)L.hit_cap * sizeof(float2), 256);
    L.off_sort = off;
    off += align_up(panel_topn_workspace_bytes(L.rows, n_items, n), 256);
    // class maxima of the panel rows (64 per window) from the accumulating kernel's sweep: the
    // selection kernel takes its threshold from them instead of a first pass over the row
    L.off_pmax = off;
    L.pmax_bytes = align_up((size_t)L.rows * (size_t)nwindows(n_items) * 64 * sizeof(unsigned), 256);
    off += L.pmax_bytes * (size_t)L.sets;
    L.bytes = off;
    return L;
}

}  // namespace rec
}  // namespace lk

#ifdef LK_REC_PHASES
extern "C" int lk_rec_phase_set(unsigned long long *d_buf)
{
    LK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(lk::rec::lk_rec_phase_buf), &d_buf, sizeof(d_buf)));
    return LK_OK;
}
#endif

static int g_rec_last_packed = -1;
// 2 / 1 / 0: the last lk_iknn_recommend call of this process ran the accumulating kernel / the list
// kernel with the packed walk / with the piece-wise walk (-1: none yet) -- test hook, like
// lk_knn_score_last_stats
extern "C" int lk_iknn_recommend_last_packed(void) { return g_rec_last_packed; }

extern "C" size_t lk_iknn_recommend_workspace_bytes(int64_t n_items, int64_t n_queries,
                                                    int64_t max_query_hits, int32_t max_nbrs,
                                                    int32_t n)
{
    if (n_items < 0 || n_queries < 0 || max_nbrs < 1) return 0;
    return lk::rec::layout(n_items, n_queries, max_query_hits, max_nbrs, n).bytes;
}

extern "C" int lk_iknn_recommend(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                                 const float *d_sim_values, int64_t n_items, int64_t n_queries,
                                 const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                                 const float *d_ref_rates, const float *d_item_bias,
                                 int32_t max_nbrs, int32_t min_nbrs, int32_t n, int exclude_refs,
                                 const int64_t *h_query_hits, int64_t max_query_hits, void *d_ws,
                                 int32_t *d_out_idx, float *d_out_score, void *stream)
{
    using namespace lk;
    using namespace lk::rec;
    LK_REQUIRE(max_nbrs >= 1 && min_nbrs >= 1, "lk_iknn_recommend: max_nbrs/min_nbrs must be >= 1");
    LK_REQUIRE(n_items >= 0 && n_queries >= 0, "lk_iknn_recommend: negative size");
    LK_REQUIRE(n_items < ((int64_t)1 << 31) - RW, "lk_iknn_recommend: too many items");
    if (n_queries == 0 || n == 0) return LK_OK;
    const int64_t out_cols = n < 0 ? n_items : n;
    if (out_cols == 0) return LK_OK;
    LK_REQUIRE(d_sim_indptr && d_ref_ptr && h_query_hits && d_ws && d_out_idx,
               "lk_iknn_recommend: null pointer");
    hipStream_t st = as_stream(stream);
    // the per-row window offsets are n_items x (n_items / RW + 1) words: 15 MB at 62 k items, 1 GB
    // at 10^6 -- refused, with the reason, where that table alone would pass 16 GiB (4 x 10^6 items)
    LK_REQUIRE((double)n_items * (double)(nwindows(n_items) + 1) * 4.0 <= 16.0 * (double)(1ull << 30),
               "lk_iknn_recommend: the window table of a %lld-item catalogue exceeds 16 GiB; "
               "score such catalogues through lk_iknn_score_batch with explicit target lists",
               (long long)n_items);
    const Layout L = layout(n_items, n_queries, max_query_hits, max_nbrs, n);
    char *ws = static_cast<char *>(d_ws);
    int *status = reinterpret_cast<int *>(ws + L.off_status);  // [0] NaN similarity, [1] task counter
    unsigned *woff = reinterpret_cast<unsigned *>(ws + L.off_woff);
    int64_t *q_base = reinterpret_cast<int64_t *>(ws + L.off_base);
    auto *q_cursor = reinterpret_cast<unsigned long long *>(ws + L.off_cursor);
    float *heap = reinterpret_cast<float *>(ws + L.off_heap);
    float2 *hits = reinterpret_cast<float2 *>(ws + L.off_hits);
    // heaps of one replay workgroup in LDS; beyond 64 KiB (max_nbrs > 127) up to the CU's 160
    const size_t heap_lds = (size_t)(max_nbrs + 1) * 64 * 2 * sizeof(float);
    const bool lds_replay = heap_lds <= 150 * 1024;
    void *sort_ws = ws + L.off_sort;
    const int nwin = nwindows(n_items);
    const int64_t ld = ld_items(n_items);

    LK_HIP_CHECK(hipMemsetAsync(status, 0, 256, st));
    // the packed walk needs same-address LDS atomics of one instruction served in lane order:
    // probed once per device (status[3] is free until the first batch); LK_REC_PACKED=0: never
    bool packed = false;
    {
        static PerDeviceOnce probed, lane_order;
        const char *e = getenv("LK_REC_PACKED");
        if (!(e && e[0] == '0')) {
            bool &done = probed.flag();
            bool &good = lane_order.flag();
            if (!done) {
                int one = 1, got = 0;
                LK_HIP_CHECK(hipMemcpyAsync(status + 3, &one, sizeof(int), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(lds_atomic_order_probe_kernel, dim3(1), dim3(64), 0, st, status + 3);
                LK_HIP_CHECK(hipMemcpyAsync(&got, status + 3, sizeof(int), hipMemcpyDeviceToHost, st));
                LK_HIP_CHECK(hipStreamSynchronize(st));
                good = got == 1;
                done = true;
                LK_HIP_CHECK(hipMemsetAsync(status + 3, 0, sizeof(int), st));
            }
            packed = good;
        }
    }
    // the accumulating kernel (round 6) needs the packed walk and, besides, same-address ds_add_f32s
    // applied in lane order with the VALU's rounding: probed once per device as well; LK_REC_ACC=0:
    // the list kernel
    bool acc_kernel = false;
    if (packed) {
        static PerDeviceOnce fprobed, fadd_order;
        const char *e = getenv("LK_REC_ACC");
        if (!(e && e[0] == '0')) {
            bool &done = fprobed.flag();
            bool &good = fadd_order.flag();
            if (!done) {
                int one = 1, got = 0;
                LK_HIP_CHECK(hipMemcpyAsync(status + 3, &one, sizeof(int), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(lds_fadd_order_probe_kernel, dim3(1), dim3(64), 0, st, status + 3);
                LK_HIP_CHECK(hipMemcpyAsync(&got, status + 3, sizeof(int), hipMemcpyDeviceToHost, st));
                LK_HIP_CHECK(hipStreamSynchronize(st));
                good = got == 1;
                done = true;
                LK_HIP_CHECK(hipMemsetAsync(status + 3, 0, sizeof(int), st));
            }
            acc_kernel = good;
        }
    }
    g_rec_last_packed = acc_kernel ? 2 : (packed ? 1 : 0);

    // batches: at most L.rows queries and L.hit_cap hits each; a query's hit region starts at the
    // sum of the hits of the batch's queries before it (the list kernel only: the accumulating
    // kernel keeps no lists and leaves the hit region alone).  Accumulating kernel: the
    // queue of targets beyond max_nbrs hits holds every such target a batch can have
    std::vector<int64_t> base((size_t)n_queries);
    std::vector<int64_t> cuts{0};
    std::vector<int64_t> heavy_of_batch;
    {
        int64_t acc = 0, rows = 0, hv = 0;
        for (int64_t q = 0; q < n_queries; ++q) {
            LK_REQUIRE(h_query_hits[q] >= 0, "lk_iknn_recommend: negative hit count");
            // (+ the rounding of every window's share to a 128-byte granule)
            const int64_t h = h_query_hits[q] + 16 * (int64_t)nwin;
            LK_REQUIRE(h <= L.hit_cap,
                       "lk_iknn_recommend: query %lld has %lld hits, more than max_query_hits",
                       (long long)q, (long long)h);
            const int64_t hb = heavy_bound(h_query_hits[q], n_items, max_nbrs);
            if (rows == L.rows || acc + h > L.hit_cap || (acc_kernel && hv + hb > L.ovf_cap)) {
                cuts.push_back(q);
                heavy_of_batch.push_back(hv);
                acc = 0;
                rows = 0;
                hv = 0;
            }
            base[(size_t)q] = acc;
            acc += h;
            hv += hb;
            ++rows;
        }
        cuts.push_back(n_queries);
        heavy_of_batch.push_back(hv);
    }
    if (!acc_kernel) {  // (the hit regions are the list kernel's)
        LK_HIP_CHECK(hipMemcpyAsync(q_base, base.data(), (size_t)n_queries * sizeof(int64_t),
                                    hipMemcpyHostToDevice, st));
        LK_HIP_CHECK(hipStreamSynchronize(st));  // `base` is host memory that dies with this call
        LK_HIP_CHECK(hipMemsetAsync(q_cursor, 0, (size_t)n_queries * sizeof(unsigned long long), st));
    }
    if (n_items > 0) {
        const int64_t cells = n_items * (nwin + 1);
        hipLaunchKernelGGL(row_windows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                           st, d_sim_indptr, d_sim_indices, n_items, nwin, woff);
    }
    // Accumulating kernel with more than one batch: batch b's tail (lists of the queued targets,
    // their replay, the own items struck, the selection) runs on a side stream while the main stream
    // already scores batch b + 1 into the OTHER panel / queue -- the tail is a few small launches and
    // one bandwidth-bound pass, the score kernel is latency-bound: they overlap nearly for free.
    // LK_REC_OVERLAP=0: one stream.
    struct Tail {
        hipStream_t side = nullptr;
        hipEvent_t scored[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
        ~Tail()
        {
            for (int i = 0; i < 2; ++i) {
                if (scored[i]) (void)hipEventDestroy(scored[i]);
                if (done[i]) (void)hipEventDestroy(done[i]);
            }
            if (side) lk::side_stream_release(side);
        }
    } tail;
    bool overlap = acc_kernel && L.sets == 2 && cuts.size() > 2 && n_items > 0;
    if (const char *e = getenv("LK_REC_OVERLAP"))
        if (e[0] == '0') overlap = false;
    if (overlap) {
        tail.side = lk::side_stream_acquire();
        for (int i = 0; i < 2; ++i) {
            LK_HIP_CHECK(hipEventCreateWithFlags(&tail.scored[i], hipEventDisableTiming));
            LK_HIP_CHECK(hipEventCreateWithFlags(&tail.done[i], hipEventDisableTiming));
        }
    }
    bool done_pending[2] = {false, false};
    for (size_t b = 0; b + 1 < cuts.size(); ++b) {
        const int64_t q0 = cuts[b], nq = cuts[b + 1] - cuts[b];
        if (nq <= 0) continue;
        const int set = overlap ? (int)(b & 1) : 0;
        float *panel_b = reinterpret_cast<float *>(ws + L.off_panel + (size_t)set * L.panel_bytes);
        OvfEntry *ovf_b = reinterpret_cast<OvfEntry *>(ws + L.off_ovf + (size_t)set * L.ovf_bytes);
        int *ctr = status + 1 + 8 * set;  // tasks [0], queue length [1]
        unsigned *pmax_b = reinterpret_cast<unsigned *>(ws + L.off_pmax + (size_t)set * L.pmax_bytes);
        const bool use_pmax = acc_kernel && n > 0 && n <= 256 && n_items > 0;
        hipStream_t ts = overlap ? tail.side : st;  // the stream of this batch's tail
        if (n_items > 0) {
            if (overlap && done_pending[set]) {  // the set's previous tail has let go of it
                LK_HIP_CHECK(hipStreamWaitEvent(st, tail.done[set], 0));
                done_pending[set] = false;
            }
            LK_HIP_CHECK(hipMemsetAsync(ctr, 0, 5 * sizeof(int), st));
            const int64_t tasks = nq * nwin;
            int64_t wgs = (tasks + RWAVES - 1) / RWAVES;
            if (wgs > REC_MAX_WGS) wgs = REC_MAX_WGS;
            if (acc_kernel) {
                const int64_t hvb = heavy_of_batch[b] < L.ovf_cap ? heavy_of_batch[b] : L.ovf_cap;
#define LK_REC_ACC_LAUNCH(EXPLV)                                                                   \
    hipLaunchKernelGGL((iknn_score_acc_kernel<EXPLV>), dim3((unsigned)wgs), dim3(RTHREADS), 0, st, \
                       d_sim_indptr, d_sim_indices, d_sim_values, n_items, nwin, woff, q0, nq,      \
                       d_ref_ptr, d_ref_items, d_ref_rates, d_item_bias, max_nbrs, min_nbrs,        \
                       panel_b, ld, ctr, status, ovf_b, (int)L.ovf_cap, ctr + 1, exclude_refs,       \
                       use_pmax ? pmax_b : nullptr)
                if (d_ref_rates)
                    LK_REC_ACC_LAUNCH(true);
                else
                    LK_REC_ACC_LAUNCH(false);
#undef LK_REC_ACC_LAUNCH
                if (overlap) {
                    LK_HIP_CHECK(hipEventRecord(tail.scored[set], st));
                    LK_HIP_CHECK(hipStreamWaitEvent(ts, tail.scored[set], 0));
                }
                if (hvb > 0) {
                    const size_t hl = (size_t)GWAVES * 2 * (size_t)(max_nbrs + 1) * sizeof(float);
                    LK_REQUIRE(hl <= 150 * 1024, "lk_iknn_recommend: max_nbrs = %d is beyond the replay "
                               "kernel's LDS heaps (LK_REC_ACC=0 takes the list kernel)", max_nbrs);
                    if (hl > 64 * 1024) {
                        LK_HIP_CHECK(hipFuncSetAttribute(
                            reinterpret_cast<const void *>(&iknn_heavy_replay_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
                        LK_HIP_CHECK(hipFuncSetAttribute(
                            reinterpret_cast<const void *>(&iknn_heavy_replay_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
                    }
                    int64_t gw = (hvb + GWAVES - 1) / GWAVES;
                    if (gw > 16384) gw = 16384;
                    if (d_ref_rates)
                        hipLaunchKernelGGL((iknn_heavy_replay_kernel<true>), dim3((unsigned)gw),
                                           dim3(GWAVES * 64), hl, ts, ovf_b, ctr + 1, (int)L.ovf_cap,
                                           d_sim_indptr, d_sim_indices, d_sim_values, n_items, nwin,
                                           woff, q0, d_ref_ptr, d_ref_items, d_ref_rates, d_item_bias,
                                           max_nbrs, panel_b, ld, status);
                    else
                        hipLaunchKernelGGL((iknn_heavy_replay_kernel<false>), dim3((unsigned)gw),
                                           dim3(GWAVES * 64), hl, ts, ovf_b, ctr + 1, (int)L.ovf_cap,
                                           d_sim_indptr, d_sim_indices, d_sim_values, n_items, nwin,
                                           woff, q0, d_ref_ptr, d_ref_items, d_ref_rates, d_item_bias,
                                           max_nbrs, panel_b, ld, status);
                }
            } else {
#define LK_REC_LAUNCH(EXPLV)                                                                      \
    do {                                                                                          \
    if (packed)                                                                                   \
    hipLaunchKernelGGL((iknn_score_all_kernel<EXPLV, true>), dim3((unsigned)wgs), dim3(RTHREADS),  \
                       0, st, d_sim_indptr, d_sim_indices, d_sim_values, n_items, nwin, woff, q0,  \
                       nq, d_ref_ptr, d_ref_items, d_ref_rates, d_item_bias, max_nbrs, min_nbrs,   \
                       hits, q_base + q0, q_cursor + q0, panel_b, ld, heap, ctr, status,           \
                       lds_replay ? ovf_b : nullptr, REC_OVF_CAP, ctr + 1);                        \
    else                                                                                          \
    hipLaunchKernelGGL((iknn_score_all_kernel<EXPLV>), dim3((unsigned)wgs), dim3(RTHREADS), 0, st, \
                       d_sim_indptr, d_sim_indices, d_sim_values, n_items, nwin, woff, q0, nq,    \
                       d_ref_ptr, d_ref_items, d_ref_rates, d_item_bias, max_nbrs, min_nbrs, hits, \
                       q_base + q0, q_cursor + q0, panel_b, ld, heap, ctr, status,                 \
                       lds_replay ? ovf_b : nullptr, REC_OVF_CAP, ctr + 1);                        \
    if (lds_replay)                                                                               \
    hipLaunchKernelGGL((iknn_heap_replay_kernel<EXPLV>), dim3(REC_OVF_CAP / 64), dim3(64),         \
                       heap_lds, st, ovf_b, ctr + 1, REC_OVF_CAP, hits, d_item_bias, max_nbrs,     \
                       panel_b, ld);                                                               \
    } while (0)
            if (lds_replay && heap_lds > 64 * 1024) {
                LK_HIP_CHECK(hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&iknn_heap_replay_kernel<true>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)heap_lds));
                LK_HIP_CHECK(hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&iknn_heap_replay_kernel<false>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)heap_lds));
            }
            if (d_ref_rates)
                LK_REC_LAUNCH(true);
            else
                LK_REC_LAUNCH(false);
#undef LK_REC_LAUNCH
            }
            // (the accumulating kernel strikes the own items itself: they never reach the queue)
            if (exclude_refs && !acc_kernel)
                hipLaunchKernelGGL(mask_refs_kernel, dim3((unsigned)nq), dim3(64), 0, ts, d_ref_ptr,
                                   d_ref_items, q0, nq, n_items, panel_b, ld);
        }
        int rc = panel_topn(panel_b, ld, nq, n_items, n, sort_ws, d_out_idx + q0 * out_cols,
                            d_out_score ? d_out_score + q0 * out_cols : nullptr, ts,
                            use_pmax ? pmax_b : nullptr, use_pmax ? nwin * 64 : 0);
        if (rc != LK_OK) return rc;
        if (overlap) {
            LK_HIP_CHECK(hipEventRecord(tail.done[set], ts));
            done_pending[set] = true;
        }
    }
    if (overlap)
        for (int i = 0; i < 2; ++i)
            if (done_pending[i]) LK_HIP_CHECK(hipStreamWaitEvent(st, tail.done[i], 0));
    LK_HIP_CHECK(hipGetLastError());
    int h = 0;
    LK_HIP_CHECK(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    if (h == 2) {
        set_error("lk_iknn_recommend: internal error (queue of heap targets / a gathered list)");
        return LK_E_INVALID;
    }
    if (h != 0) {
        set_error("similarity is null");  // accum.rs:146-151 -> ValueError
        return LK_E_NAN_SIM;
    }
    return LK_OK;
}
