// iknn_build.hip -- item-item similarity build on gfx950.
//
// Stands in for `compute_similarities` / `ItemSimTask::invoke` / `sim_row`
// (src/accel/knn/item_train.rs:33-152) and the order-preserving CSR collection of
// `ArrowCSRConsumer` (src/accel/sparse/consumer.rs:24-142).  For every item i:
//
//     dots[j] = sum over users u of i, ASCENDING u, of  a_ui * a_uj      (j != i)
//     keep dots[j] >= min_sim, rows sorted by column, int64 row offsets.
//
// Bit-exact by construction: for one (i, j) cell the reference adds the rounded
// products round(a_ui*a_uj) in ascending-user order (item_train.rs:112-129, Rust does
// not contract to FMA).  Here ONE WAVE owns a task (row i, column window p) and
// walks the users of i sequentially, applying each user's items in parallel (they
// are distinct columns, so there is no intra-instruction conflict) as a rounded
// v_mul followed by an LDS read / v_add / write; FMA contraction is switched off.
// The accumulation order of every cell is therefore the reference's, independent of
// scheduling.  (LDS float atomics would also be in order, but ds_add_f32 retires
// ~1 lane per 3 cycles per CU on gfx950 -- tools/ub/lds_atomic.hip -- 30x slower.)
//
// Data layout.  Each wave keeps a dense f32 accumulator for a window of W columns in
// its private quarter of the workgroup's LDS (W = 4096 -> 64 KiB per workgroup, two
// workgroups = 8 waves per CU).  A row needs P = ceil(n_items / W) tasks.  A table
// desc[u][p] = {first entry, length} of the slice of user u's row that falls into
// window p (built once by binary search) lets a task touch only that slice.  The
// (user, weight) stream of item i and the descriptors are read 64 users at a time,
// two / one batch ahead.  The slices of a batch are cut into CHUNKS of <= 64 entries
// (empty slices vanish, long slices of heavy users become several chunks) and the
// chunk list is walked with RING chunk loads in flight.
//
// The output size is data dependent.  With room for n_items^2 (index, value) pairs
// in free HBM (the common case on a 288 GB part: 31 GB for 62k items) ONE pass writes
// each task's survivors compacted at staging[row * n_items + p * W] and counts them;
// an exclusive scan turns the counts into offsets (windows of a row are consecutive,
// so rows come out sorted by column) and iknn_unstage_kernel copies them into place.
// Otherwise the same kernel runs twice: COUNT, scan, then WRITE at the offsets.
//
// Roofline: HBM/L2 gather bound; algorithmic bytes per product = 8 (the expanded
// (index, value) stream) -- macs = sum_u n_u^2.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

// never contract mul+add into FMA in this file: bit parity with the reference
#pragma clang fp contract(off)

#define LK_IKNN_W 4096  // columns per LDS window (env LK_IKNN_W: smaller, for experiments)
#ifndef LK_IKNN_RING
#define LK_IKNN_RING 8
#endif

struct lk_iknn_plan {
    int64_t n_users = 0, n_items = 0;
    int64_t row_lo = 0, n_rows = 0;  // output rows [row_lo, row_lo + n_rows) built by this plan
    int32_t is64 = 0;
    int32_t W = 0, P = 0;
    int32_t Q = 0;             // window quads per row: ceil(P / 4)
    int64_t n_tasks = 0;       // n_rows * P   (counts/offsets are indexed by local row*P + p)
    int64_t n_btasks = 0;      // n_rows * Q   (what a workgroup takes: one row, 4 windows)
    int64_t nnz = 0;
    int32_t *d_task = nullptr;  // [n_btasks] local row*Q + quad, heavy rows first
    size_t off_pack = 0, off_seg = 0, off_cnt = 0, off_off = 0, off_rows = 0, ws_bytes = 0;
    // single-pass build through a dense-bound staging area (n_items^2 entries) when it fits
    int32_t staged = 0;
    // symmetric build (staged, whole matrix): only the windows on / right of the diagonal are
    // accumulated, the blocks left of it are MIRRORED from their transposes (iknn_mirror_kernel)
    int32_t symmetric = 0;
    int64_t n_sym_tasks = 0;  // (row, window) tasks actually accumulated
    size_t off_strip = 0;     // symmetric: [n_tasks][W/64 + 1] u16 survivor counts before each
                              // 64-column strip of a task's window (written at extraction)
    size_t off_st_idx = 0, off_st_val = 0;
    lk_task_ctl *ctl = nullptr;  // optional cancel / progress block (lk_iknn_plan_set_ctl)
    // optional timing of the build kernel (HIP events on the launch stream; bench.py roofline)
    bool timing = false;
    mutable int timing_n = 0;
    mutable hipEvent_t ev[4][2] = {};
};

namespace lk {

// desc[u*P + p] = {first entry (absolute index into the packed user rows), length} of the
// slice of user u's row that falls into column window p  (two binary searches per pair)
template <bool IS64>
__global__ void iknn_desc_kernel(const typename IndPtr<IS64>::type *__restrict__ ui_ptr,
                                 const int32_t *__restrict__ ui_idx, int64_t n_users, int P, int W,
                                 int2 *__restrict__ desc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_users * P) return;
    const int64_t u = t / P;
    const int p = (int)(t - u * P);
    const int64_t b = ui_ptr[u], e = ui_ptr[u + 1];
    auto lower = [&](int64_t bound) {
        int64_t lo = b, hi = e;  // first index with column >= bound
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ui_idx[mid] < bound)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    const int64_t s0 = lower((int64_t)p * W), s1 = lower((int64_t)(p + 1) * W);
    desc[t] = make_int2((int)s0, (int)(s1 - s0));
}

// (index, value) of the user rows interleaved as 8-byte pairs: one coalesced 8-byte load
// per lane fetches a slice entry, and a slice is ONE contiguous span
__global__ void iknn_pack_kernel(const int32_t *__restrict__ idx, const float *__restrict__ val,
                                 int64_t nnz, int2 *__restrict__ pack)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz;
         e += (int64_t)gridDim.x * blockDim.x)
        pack[e] = make_int2(idx[e], __builtin_bit_cast(int, val[e]));
    // 64 entries of padding: slice loads are issued for all 64 lanes unconditionally
    if (blockIdx.x == 0 && threadIdx.x < 64) pack[nnz + threadIdx.x] = make_int2(0, 0);
}

typedef int i32x2 __attribute__((ext_vector_type(2)));

// WRITE: store the survivors (column, value); COUNT: store their number per task.
// Two-pass build = (COUNT) then (WRITE at the scanned offsets); staged build = one
// (WRITE + COUNT) pass into a row-strided staging area followed by iknn_unstage_kernel.
// ADDR32: every byte offset into the packed user rows fits 32 bits (nnz < 2^29 - 64): a chunk
// load is then `global_load v, voffset, s[pack]` with voffset = chunk offset + 8 * lane.
template <bool IS64, bool WRITE, bool COUNT, bool ADDR32>
__global__ __launch_bounds__(256) void iknn_build_kernel(
    const typename IndPtr<IS64>::type *__restrict__ ui_ptr, const int2 *__restrict__ ui_pack,
    const typename IndPtr<IS64>::type *__restrict__ iu_ptr, const int32_t *__restrict__ iu_idx,
    const float *__restrict__ iu_val, const int2 *__restrict__ desc,
    const int32_t *__restrict__ tasks, int64_t n_btasks, int64_t n_items, int64_t row_lo, int P,
    int Q, int W, int symmetric, uint16_t *__restrict__ strip_tab,
    float min_sim, int32_t *__restrict__ task_cnt,
    const int64_t *__restrict__ task_off, int32_t *__restrict__ out_idx,
    float *__restrict__ out_val, TaskCtlDev ctl)
{
    extern __shared__ __attribute__((aligned(16))) float lds_acc[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    float *acc = lds_acc + (size_t)wave * W;
    int *scr = reinterpret_cast<int *>(lds_acc + (size_t)4 * W) + wave * 256;  // chunk list scratch
    for (int c = lane; c < W; c += 64) acc[c] = 0.f;

    // a workgroup takes one row and four ADJACENT windows (wave w -> window 4*quad + w):
    // the four waves walk the same users, so the adjacent slices they read share cache
    // lines in L1/L2
    // (`tasks` is dealt to the workgroups serpentine, heavy rows first: see plan_create)
    for (int64_t bt = blockIdx.x; bt < n_btasks; bt += gridDim.x) {
        const int code = tasks[bt];
        const int lrow = code / Q;            // row of this plan's shard
        const int row = (int)row_lo + lrow;  // item id
        const int p = (code - lrow * Q) * 4 + wave;
        if (p >= P) continue;
        // symmetric build: sim(i, j) and sim(j, i) are the same sums of the same products in
        // the same order -- windows left of the row's own are mirrored, not accumulated
        if (symmetric && p < row / W) continue;
        if (ctl.d_cancel) {  // AccelTask.cancel: tasks not started keep their zero count
            int c = 0;
            if (lane == 0) c = ctl_cancelled(ctl, wave == 0 && (bt & 63) == 0) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(c)) break;
        }
        const int task = lrow * P + p;
        const int c_lo = p * W;
        const int wlen = (int)((n_items - c_lo) < W ? (n_items - c_lo) : W);
        const int64_t rb = iu_ptr[row], re = iu_ptr[row + 1];

        // ---- accumulate: users of `row` in ascending order -----------------
        // Per chunk: one coalesced 8-byte-per-lane read (scalar base + lane offset,
        // prefetched RING chunks ahead through a register ring), one v_mul (rounded
        // product) and an LDS read / v_add / write of the lanes' distinct columns; a wave's
        // LDS operations complete in issue order, so every cell receives its terms in
        // ascending-user order, each as round(round(r*v) + acc) -- the reference's
        // arithmetic.  64 chunks are one fully unrolled straight-line body (early exit
        // every RING chunks): with a loop-carried ring the compiler drains the memory
        // queue at every loop head.
        // The (user, weight) stream of the row and the slice descriptors are fetched TWO and
        // ONE batch ahead (user ids -> descriptor gather -> slices is a chain of three
        // dependent memory round trips).
        auto load_users = [&](int64_t b, int &u, float &r) {
            u = 0;
            r = 0.f;
            if (b + lane < re) {
                u = iu_idx[b + lane];
                r = iu_val[b + lane];
            }
        };
        auto load_desc = [&](int64_t b, int u) -> int2 {
            return (b + lane < re) ? desc[(int64_t)u * P + p] : make_int2(0, 0);
        };
        int u1, u2;
        float r0, r1, r2;
        int2 d0, d1;
        {
            int u0;
            load_users(rb, u0, r0);
            load_users(rb + 64, u1, r1);
            d0 = load_desc(rb, u0);
        }
        for (int64_t base = rb; base < re; base += 64) {
            d1 = load_desc(base + 64, u1);       // next batch's descriptors
            load_users(base + 128, u2, r2);      // the batch after that
            // A user's slice is cut into CHUNKS of <= 64 entries and the chunks of the whole
            // batch (empty slices contribute none) become the work list: every load of the
            // pipeline below is one chunk, so long slices (heavy users -- most of the work:
            // the visit-weighted mean slice has 68 entries) are prefetched like short ones.
            // Chunk t belongs to the first user whose inclusive chunk count exceeds t.
            {
                const int ulen = d0.y;
                int incl = (ulen + 63) >> 6;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int y = __shfl_up(incl, d);
                    if (lane >= d) incl += y;
                }
                scr[lane] = incl;
                scr[64 + lane] = d0.x;
                scr[128 + lane] = ulen;
                scr[192 + lane] = __builtin_bit_cast(int, r0);
            }
            const int T = __builtin_amdgcn_readfirstlane(scr[63]);
            for (int t0 = 0; t0 < T; t0 += 64) {
                const int t = t0 + lane;
                int my_beg = 0, my_len = 0, my_l1 = 0;
                float my_r = 0.f;
                if (t < T) {
                    int o = 0;
#pragma unroll
                    for (int step = 32; step; step >>= 1)
                        if (scr[o + step - 1] <= t) o += step;
                    const int ul = scr[128 + o];
                    const int j = t - (scr[o] - ((ul + 63) >> 6));
                    my_beg = scr[64 + o] + 64 * j;
                    my_len = min(64, ul - 64 * j);
                    my_l1 = my_len - 1;  // chunks are never empty
                    my_r = __builtin_bit_cast(float, scr[192 + o]);
                }
                const int nbc = min(64, T - t0);
                constexpr int RING = LK_IKNN_RING;
                int2 ring[RING];
                // lanes past the end of a slice re-read its last entry (same cache line: no
                // extra memory request) and are masked at the update
                const uint32_t my_off8 = (uint32_t)my_beg << 3;  // ADDR32 only
                auto slice_load = [&](int k) -> int2 {
                    const uint32_t li =
                        min((uint32_t)lane, (uint32_t)__builtin_amdgcn_readlane(my_l1, k));
                    if (ADDR32) {
                        typedef const __attribute__((address_space(1))) char *gptr;
                        typedef const __attribute__((address_space(1))) i32x2 *gptr2;
                        const uint32_t voff =
                            (li << 3) + (uint32_t)__builtin_amdgcn_readlane((int)my_off8, k);
                        const i32x2 v = *(gptr2)((gptr)ui_pack + voff);
                        return make_int2(v.x, v.y);
                    }
                    const int2 *sp = ui_pack + __builtin_amdgcn_readlane(my_beg, k);
                    return sp[li];
                };
#pragma unroll
                for (int q = 0; q < RING; ++q) ring[q] = slice_load(q);
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    if ((k % RING) == 0 && k >= nbc) break;  // wave-uniform
                    const float r = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_r), k));
                    const int2 e = ring[k % RING];
                    if (k + RING < 64) ring[k % RING] = slice_load(k + RING);  // past the batch: 0/0
                    // `dots[other] += r * orate` (item_train.rs:128).  No lane mask: the lanes
                    // past the end of the chunk hold copies of its last entry, so they read the
                    // same cell, compute the same sum and store the same bits as the last real
                    // lane.  The self pair (item_train.rs:120-122) is accumulated like any
                    // other and dropped at extraction: no other cell sees it.
                    if (k < nbc) {  // wave-uniform (scalar compare + branch)
                        float prod = r * __builtin_bit_cast(float, e.y);
                        asm volatile("" : "+v"(prod));  // keep the rounded product (no FMA)
                        acc[e.x - c_lo] += prod;
                    }
                }
            }
            d0 = d1;
            r0 = r1;
            u1 = u2;
            r1 = r2;
        }

        // ---- extract: survivors in column order, clear the window -----------
        // staged: window p of (local) row r is compacted at r * n_items + p * W
        int64_t wpos = !WRITE ? 0 : COUNT ? (int64_t)lrow * n_items + c_lo : task_off[task];
        int count = 0;
        uint16_t *strip = (COUNT && strip_tab) ? strip_tab + (size_t)task * (W / 64 + 1) : nullptr;
        for (int c0 = 0; c0 < wlen; c0 += 64) {
            // (symmetric build: where the 64-column strip starts inside the compacted segment --
            // the mirror kernel reads strips of this segment without searching)
            if (strip && lane == 0) strip[c0 >> 6] = (uint16_t)count;
            const int c = c0 + lane;
            float s = 0.f;
            if (c < wlen) {
                s = acc[c];
                acc[c] = 0.f;
            }
            // item_train.rs:120-122 (no self similarity), :135 (threshold)
            const bool keep = (c < wlen) && (c_lo + c != row) && (s >= min_sim);
            const unsigned long long m = __ballot(keep);
            if (WRITE) {
                if (keep) {
                    const int rank = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    out_idx[wpos + rank] = c_lo + c;
                    out_val[wpos + rank] = s;
                }
                wpos += __popcll(m);
            }
            if (COUNT) count += __popcll(m);
        }
        if (strip && lane == 0) strip[(wlen + 63) >> 6] = (uint16_t)count;
        if (COUNT && lane == 0) task_cnt[task] = count;
        if (ctl.d_done && lane == 0) ctl_advance(ctl, 1);  // unit: one (row, window) task
    }
}

// ---- symmetric build: the blocks left of the diagonal, from their transposes -----------------
//
// sim(i, j) = sim(j, i) BIT FOR BIT: both are the sum, over the common users in ascending order,
// of the same rounded products.  So only the windows p >= window(i) of row i are accumulated
// (iknn_build_kernel with `symmetric`), and segment (row j, window q < window(j)) is the
// transpose of what the rows i of window q found in window p = window(j).  One workgroup per
// (block (q, p), strip of 64 columns j): the rows i of window q are taken 256 at a time; every
// thread finds its row's entries inside the strip (binary search in the sorted, compacted
// segment (i, p)) and drops their values into a dense 256 x 64 tile in LDS; the tile is then
// read out column by column -- lane = column j, rows in ascending order, the four waves owning
// consecutive 64-row ranges whose per-column counts are prefixed -- and appended to the segment
// (j, q) of the staging area: compacted, ascending i, exactly what an accumulation would have
// written.  Similarities are >= min_sim > 0, so 0.0f marks an empty cell.
constexpr int MIRROR_ROWS = 256;

__global__ __launch_bounds__(256) void iknn_mirror_kernel(int32_t *__restrict__ cnt,
                                                          const uint16_t *__restrict__ strip_tab,
                                                          int64_t n_items, int P, int W,
                                                          int32_t *__restrict__ st_idx,
                                                          float *__restrict__ st_val)
{
    // tile[column of the strip][row of the group]: the scatter writes (thread = row) and the
    // read-out (lane = row, one column at a time) are both conflict-free
    __shared__ float tile[64][MIRROR_ROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = W / 64;  // strips per window
    int pair = blockIdx.x / S;
    const int strip = blockIdx.x - pair * S;
    int p = 1;
    while (pair >= p) {  // pair index -> (q < p)
        pair -= p;
        ++p;
    }
    const int q = pair;
    const int64_t j0 = (int64_t)p * W + (int64_t)strip * 64;
    if (j0 >= n_items) return;
    const int ncols = (int)((n_items - j0) < 64 ? (n_items - j0) : 64);
    const int64_t i_lo = (int64_t)q * W;
    const int64_t i_hi = (i_lo + W) < n_items ? (i_lo + W) : n_items;
    for (int e = tid; e < 64 * MIRROR_ROWS; e += 256) (&tile[0][0])[e] = 0.f;
    // wave w owns the columns 16 w .. 16 w + 15 of the strip for ALL rows: the write position
    // of a column is a wave-private running count (lane c & 15 of the wave keeps column c's)
    int pos_mine = 0;  // lane l < 16: entries written so far to column 16 * wave + l
    __syncthreads();
    for (int64_t g0 = i_lo; g0 < i_hi; g0 += MIRROR_ROWS) {
        {
            // this row's entries inside the strip: a run of its compacted segment (i, p) whose
            // ends the build kernel recorded at extraction
            const int64_t i = g0 + tid;
            if (i < i_hi) {
                const int64_t base = i * n_items + (int64_t)p * W;
                const uint16_t *tab = strip_tab + (size_t)(i * P + p) * (S + 1) + strip;
                // (a cancelled build leaves tasks without their strip counts: the table is
                // zeroed beforehand when a cancel block is attached, and whatever is read is
                // clamped to the segment and masked into the strip, so a discarded result can
                // never turn into an out-of-bounds access)
                // (the column is masked, not tested: a per-entry branch made this kernel 2.5x
                // slower -- 9.5 ms against 3.9; a masked index cannot leave the tile either)
                int hi = tab[1] < W ? tab[1] : W;
                const int lo = tab[0] < hi ? tab[0] : hi;
                for (int k = lo; k < hi; ++k)
                    tile[(st_idx[base + k] - j0) & 63][tid] = st_val[base + k];
            }
        }
        __syncthreads();
        for (int cl = 0; cl < 16; ++cl) {
            const int c = wave * 16 + cl;
            if (c >= ncols) break;  // wave-uniform
            int pos = __builtin_amdgcn_readlane(pos_mine, cl);
            const int64_t obase = (j0 + c) * n_items + (int64_t)q * W;
#pragma unroll
            for (int ch = 0; ch < MIRROR_ROWS / 64; ++ch) {
                const float v = tile[c][ch * 64 + lane];
                const bool nz = v != 0.f;
                const unsigned long long m = __ballot(nz);
                if (m) {  // wave-uniform
                    if (nz) {
                        const int rank = __builtin_amdgcn_mbcnt_hi(
                            (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        tile[c][ch * 64 + lane] = 0.f;  // leave the tile clean for the next group
                        st_idx[obase + pos + rank] = (int32_t)(g0 + ch * 64 + lane);
                        st_val[obase + pos + rank] = v;
                    }
                    pos += __popcll(m);
                }
            }
            if (lane == cl) pos_mine = pos;
        }
        __syncthreads();
    }
    if (lane < 16 && wave * 16 + lane < ncols) cnt[(j0 + wave * 16 + lane) * P + q] = pos_mine;
}

// Exclusive scan of the int32 task counts into int64 offsets (n_rows*P + 1 outputs) in three
// parallel steps: per-row sums, a one-workgroup scan of the row sums in which every thread
// owns a CONTIGUOUS chunk of rows (two sweeps of independent loads, one block-wide scan of
// 1024 partials), and the per-row expansion -- which also emits the CSR row offsets.
__global__ void iknn_rowsum_kernel(const int32_t *__restrict__ cnt, int64_t n_rows, int P,
                                   int64_t *__restrict__ row_sum)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t v = 0;
    for (int q = 0; q < P; ++q) v += cnt[r * P + q];
    row_sum[r] = v;
}

// in place: row_sum[r] -> exclusive prefix; row_sum[n_rows] = total
__global__ __launch_bounds__(1024) void iknn_scan_kernel(int64_t *__restrict__ row_sum,
                                                        int64_t n_rows)
{
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (n_rows + 1023) / 1024;
    const int64_t lo = std::min<int64_t>(n_rows, (int64_t)tid * per);
    const int64_t hi = std::min<int64_t>(n_rows, lo + per);
    int64_t v = 0;
    for (int64_t i = lo; i < hi; ++i) v += row_sum[i];
    int64_t x = v;  // inclusive scan of the 1024 chunk sums
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int64_t run = x - v;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int64_t i = lo; i < hi; ++i) {
        const int64_t c = row_sum[i];
        row_sum[i] = run;
        run += c;
    }
    if (tid == 1023) row_sum[n_rows] = run;  // the last chunk ends at n_rows
}

__global__ void iknn_taskoff_kernel(const int32_t *__restrict__ cnt,
                                    const int64_t *__restrict__ row_off, int64_t n_rows, int P,
                                    int64_t *__restrict__ task_off,
                                    int64_t *__restrict__ out_indptr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    const int64_t base = row_off[r];
    out_indptr[r] = base;
    if (r == n_rows) {
        task_off[n_rows * P] = base;
        return;
    }
    int64_t run = base;
    for (int q = 0; q < P; ++q) {
        task_off[r * P + q] = run;
        run += cnt[r * P + q];
    }
}

// counts / offsets are indexed by TASK ID = row*P + p (item order, windows of a row
// adjacent), so the scan yields rows in item order sorted by column; `tasks` is only
// the visiting order (heavy rows first).

// staged build: move the survivors of task t from the staging area to their final place
// (one wave per task, coalesced)
__global__ __launch_bounds__(256) void iknn_unstage_kernel(
    const int32_t *__restrict__ task_cnt, const int64_t *__restrict__ task_off, int64_t n_tasks,
    int64_t n_items, int P, int W, const int32_t *__restrict__ st_idx,
    const float *__restrict__ st_val, int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int n = task_cnt[t];
    const int64_t row = t / P;
    const int64_t src = row * n_items + (t - row * P) * W, dst = task_off[t];
    for (int e = lane; e < n; e += 64) {
        out_idx[dst + e] = st_idx[src + e];
        out_val[dst + e] = st_val[src + e];
    }
}

// one resident workgroup per CU at W = 8192 (128 KiB of LDS); more for small windows
static size_t iknn_lds_bytes(int W) { return (size_t)4 * W * sizeof(float) + 4 * 256 * sizeof(int); }
static int64_t iknn_grid(int W)
{
    return 256 * (int64_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / iknn_lds_bytes(W)));
}

}  // namespace lk

extern "C" int lk_iknn_plan_create(lk_iknn_plan **out, const void *h_ui_indptr,
                                   const void *h_iu_indptr, int indptr_is_64, int64_t n_users,
                                   int64_t n_items)
{
    return lk_iknn_plan_create_rows(out, h_ui_indptr, h_iu_indptr, indptr_is_64, n_users,
                                    n_items, 0, n_items);
}

extern "C" int lk_iknn_plan_create_rows(lk_iknn_plan **out, const void *h_ui_indptr,
                                        const void *h_iu_indptr, int indptr_is_64,
                                        int64_t n_users, int64_t n_items, int64_t row_begin,
                                        int64_t row_end)
{
    LK_REQUIRE(out && h_ui_indptr && h_iu_indptr, "lk_iknn_plan_create: null pointer");
    LK_REQUIRE(n_users >= 0 && n_items >= 0 && n_items < (int64_t)INT32_MAX &&
                   n_users < (int64_t)INT32_MAX,
               "lk_iknn_plan_create: bad shape");
    LK_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n_items,
               "lk_iknn_plan_create: bad row range [%lld, %lld) of %lld items",
               (long long)row_begin, (long long)row_end, (long long)n_items);
    const int64_t n_rows = row_end - row_begin;
    auto *p = new lk_iknn_plan();
    p->n_users = n_users;
    p->n_items = n_items;
    p->row_lo = row_begin;
    p->n_rows = n_rows;
    p->is64 = indptr_is_64 ? 1 : 0;
    int64_t W = LK_IKNN_W;
    if (const char *env = getenv("LK_IKNN_W")) {  // tuning knob: columns per LDS window
        const long v = atol(env);
        if (v >= 64 && v <= LK_IKNN_W) W = v / 64 * 64;
    }
    if (n_items < W) W = std::max<int64_t>(64, (n_items + 63) / 64 * 64);
    p->W = (int32_t)W;
    p->P = (int32_t)std::max<int64_t>(1, (n_items + W - 1) / W);
    p->Q = (p->P + 3) / 4;
    p->n_tasks = n_rows * p->P;
    p->n_btasks = n_rows * p->Q;
    p->nnz = indptr_is_64 ? static_cast<const int64_t *>(h_ui_indptr)[n_users]
                          : (int64_t) static_cast<const int32_t *>(h_ui_indptr)[n_users];
    LK_REQUIRE(p->n_tasks < (int64_t)INT32_MAX, "lk_iknn_plan_create: too many tasks");
    LK_REQUIRE(p->nnz < (int64_t)INT32_MAX - 64, "lk_iknn_plan_create: nnz >= 2^31 unsupported");

    auto len = [&](int64_t r) -> int64_t {
        if (indptr_is_64) {
            const int64_t *ip = static_cast<const int64_t *>(h_iu_indptr);
            return ip[r + 1] - ip[r];
        }
        const int32_t *ip = static_cast<const int32_t *>(h_iu_indptr);
        return (int64_t)ip[r + 1] - ip[r];
    };
    // single-pass (staged) build?  288 GB of HBM: when n_items^2 (index, value) pairs fit
    // comfortably, every task writes its survivors straight into a row-strided staging area in
    // ONE pass over the data and a copy kernel compacts them; otherwise count and fill are two
    // full passes.
    const size_t stage = (size_t)n_rows * (size_t)n_items * sizeof(int32_t);
    {
        size_t cap = (size_t)64 << 30, free_b = 0, total_b = 0;
        if (const char *env = getenv("LK_IKNN_STAGE_GB")) cap = (size_t)atol(env) << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap = std::min(cap, free_b / 3);
        if (n_rows > 0 && 2 * stage <= cap) p->staged = 1;
    }
    // symmetric build: staged, the whole matrix, more than one window (LK_IKNN_SYMMETRIC=0: off)
    {
        const char *env = getenv("LK_IKNN_SYMMETRIC");
        p->symmetric = (p->staged && row_begin == 0 && row_end == n_items && p->P > 1 &&
                        p->W % 64 == 0 && !(env && env[0] == '0'))
                           ? 1
                           : 0;
    }
    std::vector<int32_t> rows((size_t)n_rows);  // LOCAL row numbers, heaviest first
    for (int64_t r = 0; r < n_rows; ++r) rows[(size_t)r] = (int32_t)r;
    std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) {
        return len(row_begin + a) > len(row_begin + b);
    });
    // (row, quad of four adjacent windows) tasks of the weight-sorted rows; the symmetric build
    // drops the quads that lie entirely left of the row's own window
    std::vector<int32_t> sorted_tasks;
    sorted_tasks.reserve((size_t)p->n_btasks);
    p->n_sym_tasks = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        const int32_t r = rows[(size_t)i];
        const int qrow = p->symmetric ? (int)((row_begin + r) / W) : 0;
        for (int qd = 0; qd < p->Q; ++qd)
            if (4 * qd + 3 >= qrow) sorted_tasks.push_back(r * p->Q + qd);
        p->n_sym_tasks += p->P - qrow;
    }
    p->n_btasks = (int64_t)sorted_tasks.size();
    // Workgroup b of G takes tasks b, b+G, b+2G, ...: deal the weight-sorted list
    // serpentine (every other round reversed) so the per-workgroup totals stay level.
    std::vector<int32_t> tasks((size_t)p->n_btasks);
    const int64_t G = std::max<int64_t>(1, std::min<int64_t>(p->n_btasks, lk::iknn_grid(p->W)));
    for (int64_t i = 0; i < p->n_btasks; ++i) {
        const int64_t round = i / G, pos = i - round * G;
        const int64_t cnt_in_round = std::min<int64_t>(G, p->n_btasks - round * G);
        const int64_t src = round * G + ((round & 1) ? cnt_in_round - 1 - pos : pos);
        tasks[(size_t)i] = sorted_tasks[(size_t)src];
    }
    size_t bytes = std::max<size_t>(tasks.size(), 1) * sizeof(int32_t);
    if (hipMalloc(reinterpret_cast<void **>(&p->d_task), bytes) != hipSuccess) {
        delete p;
        lk::set_error("lk_iknn_plan_create: hipMalloc failed");
        return LK_E_HIP;
    }
    if (!tasks.empty() &&
        hipMemcpy(p->d_task, tasks.data(), tasks.size() * sizeof(int32_t),
                  hipMemcpyHostToDevice) != hipSuccess) {
        lk_iknn_plan_destroy(p);
        lk::set_error("lk_iknn_plan_create: hipMemcpy failed");
        return LK_E_HIP;
    }
    size_t off = 0;
    p->off_pack = off;
    off += lk::align_up((size_t)(std::max<int64_t>(p->nnz, 1) + 64) * sizeof(int2), 256);
    p->off_seg = off;
    off += lk::align_up((size_t)std::max<int64_t>(n_users, 1) * p->P * sizeof(int2), 256);
    p->off_cnt = off;
    off += lk::align_up((size_t)(p->n_tasks + 1) * sizeof(int32_t), 256);
    p->off_off = off;
    off += lk::align_up((size_t)(p->n_tasks + 1) * sizeof(int64_t), 256);
    p->off_rows = off;
    off += lk::align_up((size_t)(n_rows + 1) * sizeof(int64_t), 256);
    if (p->symmetric) {
        p->off_strip = off;
        off += lk::align_up((size_t)p->n_tasks * (size_t)(p->W / 64 + 1) * sizeof(uint16_t), 256);
    }
    if (p->staged) {
        p->off_st_idx = off;
        off += lk::align_up(stage, 256);
        p->off_st_val = off;
        off += lk::align_up(stage, 256);
    }
    p->ws_bytes = off;
    *out = p;
    return LK_OK;
}

extern "C" int lk_iknn_plan_enable_timing(lk_iknn_plan *p, int enable)
{
    LK_REQUIRE(p != nullptr, "lk_iknn_plan_enable_timing: null plan");
    if (enable && !p->ev[0][0])
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) LK_HIP_CHECK(hipEventCreate(&p->ev[i][j]));
    p->timing = enable != 0;
    p->timing_n = 0;
    return LK_OK;
}

extern "C" int lk_iknn_plan_get_timing(lk_iknn_plan *p, double *ms_build, int32_t *n_launches)
{
    LK_REQUIRE(p && ms_build && n_launches, "lk_iknn_plan_get_timing: null pointer");
    *ms_build = 0.0;
    *n_launches = p->timing_n;
    for (int i = 0; i < p->timing_n; ++i) {
        float a = 0.f;
        LK_HIP_CHECK(hipEventSynchronize(p->ev[i][1]));
        LK_HIP_CHECK(hipEventElapsedTime(&a, p->ev[i][0], p->ev[i][1]));
        *ms_build += a;
    }
    p->timing_n = 0;
    return LK_OK;
}

extern "C" void lk_iknn_plan_destroy(lk_iknn_plan *p)
{
    if (!p) return;
    if (p->ev[0][0])
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) (void)hipEventDestroy(p->ev[i][j]);
    if (p->d_task) (void)hipFree(p->d_task);
    delete p;
}

extern "C" int lk_iknn_plan_set_ctl(lk_iknn_plan *p, lk_task_ctl *ctl)
{
    LK_REQUIRE(p != nullptr, "lk_iknn_plan_set_ctl: null plan");
    p->ctl = ctl;
    return LK_OK;
}

extern "C" size_t lk_iknn_plan_workspace_bytes(const lk_iknn_plan *p)
{
    return p ? p->ws_bytes : 0;
}

namespace lk {

template <bool IS64, bool WRITE, bool COUNT>
static int launch_iknn(const lk_iknn_plan *p, const void *ui_ptr, const int32_t *ui_idx,
                       const float *ui_val, const void *iu_ptr, const int32_t *iu_idx,
                       const float *iu_val, float min_sim, char *ws, int32_t *out_idx,
                       float *out_val, hipStream_t st)
{
    using IT = typename IndPtr<IS64>::type;
    int2 *desc = reinterpret_cast<int2 *>(ws + p->off_seg);
    int2 *pack = reinterpret_cast<int2 *>(ws + p->off_pack);
    int32_t *cnt = reinterpret_cast<int32_t *>(ws + p->off_cnt);
    int64_t *off = reinterpret_cast<int64_t *>(ws + p->off_off);
    if (p->n_tasks == 0) return LK_OK;
    if (COUNT) {
        const int64_t nd = p->n_users * p->P;
        if (nd > 0)
            hipLaunchKernelGGL((iknn_desc_kernel<IS64>), dim3((unsigned)((nd + 255) / 256)),
                               dim3(256), 0, st, static_cast<const IT *>(ui_ptr), ui_idx,
                               p->n_users, p->P, p->W, desc);
        hipLaunchKernelGGL(iknn_pack_kernel, dim3(2048), dim3(256), 0, st, ui_idx, ui_val,
                               p->nnz, pack);
    }
    const size_t lds = iknn_lds_bytes(p->W);
    // (env LK_IKNN_ADDR64 forces the general addressing path: test hook)
    const bool a32 = (p->nnz + 64) < ((int64_t)1 << 29) && !getenv("LK_IKNN_ADDR64");
    auto kern = a32 ? iknn_build_kernel<IS64, WRITE, COUNT, true>
                    : iknn_build_kernel<IS64, WRITE, COUNT, false>;
    LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = std::min<int64_t>(p->n_btasks, iknn_grid(p->W));
    if (p->ctl) {
        if (COUNT) LK_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)p->n_tasks, st));
        // tasks that stop at a cancel never write their strip counts: the mirror kernel below
        // must not read stale workspace for them
        if (COUNT && p->symmetric)
            LK_HIP_CHECK(hipMemsetAsync(ws + p->off_strip, 0,
                                        (size_t)p->n_tasks * (size_t)(p->W / 64 + 1) *
                                            sizeof(uint16_t), st));
        int rc = ctl_begin(p->ctl, p->n_rows, p->symmetric ? p->n_sym_tasks : p->n_tasks, st);
        if (rc != LK_OK) return rc;
    }
    const bool tm = p->timing && p->timing_n < 4;
    if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][0], st));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st,
                       static_cast<const IT *>(ui_ptr), pack, static_cast<const IT *>(iu_ptr),
                       iu_idx, iu_val, desc, p->d_task, p->n_btasks, p->n_items, p->row_lo, p->P, p->Q,
                       p->W, p->symmetric,
                       p->symmetric ? reinterpret_cast<uint16_t *>(ws + p->off_strip) : nullptr,
                       min_sim, cnt, off, out_idx, out_val,
                       p->ctl ? p->ctl->dev() : TaskCtlDev{});
    if (p->symmetric && WRITE && COUNT) {
        // the windows left of the diagonal: transposes of the accumulated blocks
        const int64_t pairs = (int64_t)p->P * (p->P - 1) / 2;
        const int64_t strips = p->W / 64;
        hipLaunchKernelGGL(iknn_mirror_kernel, dim3((unsigned)(pairs * strips)), dim3(256), 0, st,
                           cnt, reinterpret_cast<const uint16_t *>(ws + p->off_strip), p->n_items,
                           p->P, p->W, out_idx, out_val);
    }
    if (tm) {
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][1], st));
        p->timing_n++;
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk

static int check_args(const lk_iknn_plan *plan, const void *a, const void *b, const void *ws,
                      int64_t save_nbrs, const char *who)
{
    LK_REQUIRE(plan && ws, "%s: null plan/workspace", who);
    LK_REQUIRE(plan->n_tasks == 0 || (a && b), "%s: null CSR pointer", who);
    LK_REQUIRE(save_nbrs <= 0,
               "%s: pass save_nbrs <= 0 here and truncate the result with "
               "lk_iknn_truncate_count / lk_iknn_truncate_fill",
               who);
    return LK_OK;
}

extern "C" int lk_iknn_build_count(const lk_iknn_plan *plan, const void *d_ui_indptr,
                                   const int32_t *d_ui_indices, const float *d_ui_values,
                                   const void *d_iu_indptr, const int32_t *d_iu_indices,
                                   const float *d_iu_values, float min_sim, int64_t save_nbrs,
                                   void *d_ws, int64_t *d_out_indptr, int64_t *h_total_nnz,
                                   void *stream)
{
    int rc = check_args(plan, d_ui_indptr, d_iu_indptr, d_ws, save_nbrs, "lk_iknn_build_count");
    if (rc != LK_OK) return rc;
    LK_REQUIRE(d_out_indptr && h_total_nnz, "lk_iknn_build_count: null output");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    int32_t *cnt = reinterpret_cast<int32_t *>(ws + plan->off_cnt);
    int64_t *off = reinterpret_cast<int64_t *>(ws + plan->off_off);
    if (plan->n_tasks == 0) {
        LK_HIP_CHECK(hipMemsetAsync(d_out_indptr, 0, sizeof(int64_t) * (plan->n_rows + 1), st));
        LK_HIP_CHECK(hipStreamSynchronize(st));
        *h_total_nnz = 0;
        return LK_OK;
    }
    {
        const void *up = d_ui_indptr, *ip = d_iu_indptr;
        int32_t *si = plan->staged ? reinterpret_cast<int32_t *>(ws + plan->off_st_idx) : nullptr;
        float *sv = plan->staged ? reinterpret_cast<float *>(ws + plan->off_st_val) : nullptr;
#define LK_IKNN_LAUNCH(IS64, WRITE)                                                          \
    lk::launch_iknn<IS64, WRITE, true>(plan, up, d_ui_indices, d_ui_values, ip, d_iu_indices, \
                                       d_iu_values, min_sim, ws, si, sv, st)
        rc = plan->staged ? (plan->is64 ? LK_IKNN_LAUNCH(true, true) : LK_IKNN_LAUNCH(false, true))
                          : (plan->is64 ? LK_IKNN_LAUNCH(true, false)
                                        : LK_IKNN_LAUNCH(false, false));
#undef LK_IKNN_LAUNCH
    }
    if (rc != LK_OK) return rc;
    {
        int64_t *row_off = reinterpret_cast<int64_t *>(ws + plan->off_rows);
        const unsigned gb = (unsigned)((plan->n_rows + 256) / 256);
        hipLaunchKernelGGL(lk::iknn_rowsum_kernel, dim3(gb), dim3(256), 0, st, cnt, plan->n_rows,
                           plan->P, row_off);
        hipLaunchKernelGGL(lk::iknn_scan_kernel, dim3(1), dim3(1024), 0, st, row_off,
                           plan->n_rows);
        hipLaunchKernelGGL(lk::iknn_taskoff_kernel, dim3(gb), dim3(256), 0, st, cnt, row_off,
                           plan->n_rows, plan->P, off, d_out_indptr);
    }
    LK_HIP_CHECK(hipGetLastError());
    LK_HIP_CHECK(hipMemcpyAsync(h_total_nnz, off + plan->n_tasks, sizeof(int64_t),
                                hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    if (plan->ctl) return lk::ctl_finish(plan->ctl, st);  // LK_E_CANCELLED if interrupted
    return LK_OK;
}

extern "C" int lk_iknn_build_fill(const lk_iknn_plan *plan, const void *d_ui_indptr,
                                  const int32_t *d_ui_indices, const float *d_ui_values,
                                  const void *d_iu_indptr, const int32_t *d_iu_indices,
                                  const float *d_iu_values, float min_sim, int64_t save_nbrs,
                                  void *d_ws, const int64_t *d_out_indptr, int32_t *d_out_indices,
                                  float *d_out_values, void *stream)
{
    int rc = check_args(plan, d_ui_indptr, d_iu_indptr, d_ws, save_nbrs, "lk_iknn_build_fill");
    if (rc != LK_OK) return rc;
    (void)d_out_indptr;  // offsets of the count pass are kept in the workspace
    if (plan->n_tasks == 0) return LK_OK;
    // d_out_indices / d_out_values may be null when the count pass found nothing
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    if (plan->staged) {  // the count call already computed everything: compact the staging area
        if (!d_out_indices || !d_out_values) return LK_OK;  // nothing survived
        hipLaunchKernelGGL(lk::iknn_unstage_kernel, dim3((unsigned)((plan->n_tasks + 3) / 4)),
                           dim3(256), 0, st, reinterpret_cast<const int32_t *>(ws + plan->off_cnt),
                           reinterpret_cast<const int64_t *>(ws + plan->off_off), plan->n_tasks,
                           plan->n_items, plan->P, plan->W,
                           reinterpret_cast<const int32_t *>(ws + plan->off_st_idx),
                           reinterpret_cast<const float *>(ws + plan->off_st_val), d_out_indices,
                           d_out_values);
        LK_HIP_CHECK(hipGetLastError());
        return LK_OK;
    }
    rc = plan->is64
             ? lk::launch_iknn<true, true, false>(plan, d_ui_indptr, d_ui_indices, d_ui_values,
                                                  d_iu_indptr, d_iu_indices, d_iu_values, min_sim,
                                                  ws, d_out_indices, d_out_values, st)
             : lk::launch_iknn<false, true, false>(plan, d_ui_indptr, d_ui_indices, d_ui_values,
                                                   d_iu_indptr, d_iu_indices, d_iu_values,
                                                   min_sim, ws, d_out_indices, d_out_values, st);
    if (rc == LK_OK && plan->ctl) {  // the second full pass is cancellable too (blocking then)
        LK_HIP_CHECK(hipStreamSynchronize(st));
        rc = lk::ctl_finish(plan->ctl, st);
    }
    return rc;
}
