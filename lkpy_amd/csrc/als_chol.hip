// als_chol.hip -- implicit-ALS half-epoch, exact (Cholesky) solver, gfx950.
//
// Stands in for `train_implicit_matrix` / `ImplicitTrainTask::invoke` /
// `train_row_solve` (src/accel/als/implicit.rs:35-125) and `POSV::solve`
// (src/accel/als/solve.rs:65-107).  Per CSR row r with columns c_j, values v_j:
//     A = OtOr + sum_j v_j q_j q_j^T      y = sum_j (v_j + 1) q_j      x = A^-1 y
//     this[r] <- x;   delta_r = ||x - this_old[r]||^2   (empty row: zeros, delta 0)
//
// Work decomposition: ONE WAVE PER ROW (4 independent waves per workgroup, no
// workgroup barriers).  Rows are visited longest-first (plan order); rows longer
// than LK_ALS_LONG_ROW are pre-reduced by a chunk kernel (one wave per chunk of
// <= LK_ALS_CHUNK entries) into partial slabs that the solving wave sums in
// chunk order, so the result is independent of scheduling (bit-reproducible).
//
// Normal-matrix build (the flop carrier, 2*k^2 per CSR entry): f32 MFMA
// v_mfma_f32_16x16x4_f32, K = 4 CSR entries per instruction, upper tiles only
// (A is symmetric): NT(NT+1)/2 MFMAs per 4 entries instead of NT^2.  Features
// are handled in "primed" order p = t*16 + s <-> f = s*NT + t so that a lane's NT
// features of a factor row are one contiguous vector load and 16 lanes fetch the
// whole row of `other` in one coalesced request (the gather is per CSR entry:
// 4*k contiguous bytes).  The CSR (indices, values) stream is read coalesced, 64
// entries per wave-load, and broadcast with ds_bpermute.
//
// Solve: the accumulator tiles are transposed through a wave-private LDS region
// so that lane R holds row R of A' (primed order == a symmetric permutation of A,
// which leaves the solution unchanged); an in-register right-looking Cholesky
// with v_readlane broadcasts (no LDS traffic in the O(k^3) loop), forward
// substitution in registers, back substitution against L^T staged in LDS.
//
// Roofline: f32 MFMA bound for k = 64 (nnz*(2k^2+2k) +
// rows*(k^3/3 + 2k^2) flop per half-epoch); HBM traffic is the CSR stream plus
// the (L2/MALL-resident) gathered factor rows.
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <mutex>
#include <strings.h>
#include <vector>

#include "als_plan.h"
#include "common.h"

#ifndef LK_ALS_RING
#define LK_ALS_RING 4  // gather ring slots (8 costs 20 more registers: no gain at 3 waves/SIMD)
#endif
#ifndef LK_ALS_LOOKAHEAD
#define LK_ALS_LOOKAHEAD 8  // multiplier groups read ahead in chol_step
#endif
#ifndef LK_ALS_GRAM_DMA
#define LK_ALS_GRAM_DMA 1  // k = 64: gathered rows prefetched into LDS (global_load_lds)
#endif
#ifndef LK_ALS_DMA_PIPE
#define LK_ALS_DMA_PIPE 1  // operands of group g+1 fetched from the ring before group g's MFMAs
#endif
#ifndef LK_ALS_GRAM_FENCE
#define LK_ALS_GRAM_FENCE 1
#endif
#ifndef LK_ALS_SLAB_NT
#define LK_ALS_SLAB_NT 1
#endif
#ifndef LK_ALS_SOLVE_PRIO
#define LK_ALS_SOLVE_PRIO 0  // s_setprio level of a wave while it factors / substitutes (0: unchanged)
#endif
#ifndef LK_ALS_PANEL
#define LK_ALS_PANEL 2  // 2: hybrid Cholesky (panels in lane = row layout + MFMA updates);
                        // 0: the round-1 lane = row Cholesky (bit-identical results; A/B timing)
#endif
#ifndef LK_ALS_SOLVE_ATTR
#if LK_ALS_PANEL
// the hybrid solver keeps the matrix in its 40 accumulator registers: 4 waves per SIMD
#define LK_ALS_SOLVE_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#endif
#endif
#ifndef LK_ALS_SOLVE_ATTR
// At least 3 waves per SIMD: the k = 64 kernel then fits 168 registers with 13 dwords of
// scratch instead of 248 registers (2 waves per SIMD): +14 % epochs/s (tools/als_variants.py)
#define LK_ALS_SOLVE_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif

namespace lk {

#ifdef LK_ALS_PHASES
// Diagnostic build only (tools/als_variants.py): shader-clock cycles per phase of the solve
// kernel, one record of 8 words per task (plan order): [0] row set-up (row id, extents), [1]
// normal matrix, [2] transposition, [3] factorisation, [4] substitutions, [5] store + delta,
// [6] row length, [7] whole
__device__ unsigned *lk_als_phase_buf;
#define LK_PHASE_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LK_PHASE_ADD(i, a, b) \
    if (lane_id() == 0 && lk_als_phase_buf) lk_als_phase_buf[t * 8 + (i)] = (unsigned)((b) - (a))
#else
#define LK_PHASE_T(var)
#define LK_PHASE_ADD(i, a, b)
#endif

__host__ __device__ constexpr int als_tiles(int NT) { return NT * (NT + 1) / 2; }
// packed index of upper tile (ti <= tj)
__host__ __device__ constexpr int tidx(int ti, int tj) { return tj * (tj + 1) / 2 + ti; }

template <int NT>
struct Gram {
    f32x4 t[als_tiles(NT)];
    float y[NT];
    float yc = 0.f;  // SEQY: the reference's sequential y chain, lane (sub, slot) = feature sub * NT + slot
};

// (inline asm: chained __builtin_amdgcn_permlane*_swap calls are miscompiled by hipcc 7.2 --
// both results of the later swaps land in one register; the s_nop covers the two wait states
// a VALU write of an operand needs before the swap reads it)
__device__ __forceinline__ void swap32(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float &a, float &b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// SEQY -- the right-hand side in the REFERENCE's order inside the Gram loop (round 5).  The
// reference forms y as ONE sequential float32 chain per feature, product and sum rounded
// separately (src/accel/als/implicit.rs:116-117; als_rhs.hip); the tuned loop kept four partial
// sums per feature (one per entry slot, fused multiply-adds) and combined them at the end.  On
// the CPU that difference alone explains most of the distance between the two arithmetics on
// rows of a few hundred entries (tools: DESIGN.md section 2): y in the reference's order halves
// it.  A lane holds NT features of ONE entry (its slot); after the 4 x 4 (register x slot)
// transposition -- the two v_permlane32_swap + two v_permlane16_swap of the hybrid solver --
// lane (sub, t) holds the products of the group's FOUR entries for feature sub * NT + t and adds
// them in entry order: bit for bit the reference's chain.  +12 instructions per 4-entry group.
template <int NT>
__device__ __forceinline__ void y_chain_step(float &yc, const float (&q)[NT], const float v1)
{
    float p[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) p[t] = t < NT ? __fmul_rn(q[t < NT ? t : 0], v1) : 0.f;
    swap32(p[0], p[2]);
    swap32(p[1], p[3]);
    swap16(p[0], p[1]);
    swap16(p[2], p[3]);
    yc = __fadd_rn(yc, p[0]);
    yc = __fadd_rn(yc, p[1]);
    yc = __fadd_rn(yc, p[2]);
    yc = __fadd_rn(yc, p[3]);
}

template <int NT>
__device__ __forceinline__ void load_q(const float *p, float (&q)[NT])
{
    if constexpr (NT == 1) {
        q[0] = *p;
    } else if constexpr (NT == 2) {
        f32x2 t = *reinterpret_cast<const f32x2 *>(p);
        q[0] = t.x;
        q[1] = t.y;
    } else {
        static_assert(NT == 4, "Cholesky path supports NT in {1,2,4}");
        f32x4 t = *reinterpret_cast<const f32x4 *>(p);
        q[0] = t.x;
        q[1] = t.y;
        q[2] = t.z;
        q[3] = t.w;
    }
}

// Accumulate CSR entries [beg, end) of one row into G (A tiles and y).
//
// Software pipeline: entries are taken in batches of 64 (one coalesced load of
// indices + values per wave, fetched ONE BATCH AHEAD), each batch is 16 groups of
// 4 entries (= one K=4 MFMA step).  The gathered factor rows live in a RING-slot
// register ring: the gather for group g+RING is issued as soon as group g has been
// consumed -- across batch boundaries too -- so RING x ~320 MFMA cycles of this wave's work,
// plus the other two waves of the SIMD, cover every gather and the wave never drains its
// memory queue inside a row.
template <int NT>
struct GatherRing {
    static constexpr int RING = LK_ALS_RING;  // must divide 16 (groups per batch)
    float q[RING][NT];
    float v[RING];
};

// The batch of 64 (column, value) pairs is staged in wave-private LDS (128 words per batch:
// columns, then values): group g's four entries are then ds_read_b32 at an IMMEDIATE offset
// from one base register.  (A ds_bpermute from the loaded registers needs a distinct address
// register for each of the 16 groups: 16 VGPRs the 4-waves-per-SIMD build does not have.)
template <int NT>
__device__ __forceinline__ void ring_issue(GatherRing<NT> &R, const int slot_idx, const int g,
                                           const float *stage_slot,
                                           const float *__restrict__ other)
{
    constexpr int KP = NT * 16;
    const int lane = lane_id();
    const int col = __builtin_bit_cast(int, stage_slot[g * 4]);
    R.v[slot_idx] = stage_slot[64 + g * 4];
    load_q<NT>(other + (int64_t)col * KP + (lane & 15) * NT, R.q[slot_idx]);
}

template <int NT, bool MASKED, bool SEQY = false>
__device__ __forceinline__ void ring_consume(Gram<NT> &G, const GatherRing<NT> &R,
                                             const int slot_idx, const int g, const int nb,
                                             const bool expl)
{
    const int lane = lane_id();
    // MASKED (tail batch only): entries past the row end re-read the row's last entry; kill
    // their q so that neither A (v*q*q) nor y ((v+1)*q) sees them
    const bool live = !MASKED || (g * 4 + (lane >> 4)) < nb;
    const float v = R.v[slot_idx];
    // implicit: A += v q q^T, y += (v + 1) q  (implicit.rs:110-117)
    // explicit: A += q q^T,   y += v q        (explicit.rs:103,109; v = normalised rating)
    const float va = expl ? 1.0f : v;
    float q[NT], a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        q[t] = live ? R.q[slot_idx][t] : 0.f;
        a[t] = q[t] * va;  // `mtl = mt * vals` (implicit.rs:110-111); exact for va == 1
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti)
            G.t[tidx(ti, tj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], q[tj],
                                                                     G.t[tidx(ti, tj)], 0, 0, 0);
    const float v1 = expl ? v : v + 1.0f;  // `vals += 1.0` (implicit.rs:116)
    if constexpr (SEQY) {
        y_chain_step<NT>(G.yc, q, v1);
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) G.y[t] = fmaf(q[t], v1, G.y[t]);
    }
}

constexpr int GRAM_STAGE_WORDS = 256;  // two batches of 64 (column, value) pairs

template <int NT, bool SEQY = false>
__device__ __forceinline__ void gram_accumulate(Gram<NT> &G, const int32_t *__restrict__ cols,
                                                const float *__restrict__ vals, int64_t beg,
                                                int64_t end, const float *__restrict__ other,
                                                int /*ld == 16*NT*/, const bool expl,
                                                float *stage /* GRAM_STAGE_WORDS, wave-private */)
{
    // Every load below is UNCONDITIONAL (out-of-range lanes/groups re-read the row's last
    // entry, which is masked or never consumed): a load inside a branch makes the compiler
    // drain the whole memory queue (s_waitcnt vmcnt(0)) before every MFMA group and the
    // ring would hide nothing.
    const int lane = lane_id();
    constexpr int RING = GatherRing<NT>::RING;
    static_assert(RING >= 1 && RING <= 8 && 16 % RING == 0, "gather ring: 1, 2, 4 or 8 slots");
    GatherRing<NT> R;
    const int64_t last = end - 1;  // end > beg

    // stage_cur / stage_nxt: this lane's view (entry slot = lane >> 4) of the two batch buffers
    float *wr_cur = stage + lane, *wr_nxt = stage + 128 + lane;
    const float *rd_cur = stage + (lane >> 4), *rd_nxt = stage + 128 + (lane >> 4);
    int nxt_col;
    float nxt_val;
    {
        const int64_t e = (beg + lane < end) ? beg + lane : last;
        wr_cur[0] = __builtin_bit_cast(float, cols[e]);
        wr_cur[64] = vals[e];
    }
#pragma unroll
    for (int g = 0; g < RING; ++g) ring_issue<NT>(R, g, g, rd_cur, other);

    int64_t base = beg;
    // full batches: one straight-line body of 16 groups, no branches, no masks
    for (; base + 64 <= end; base += 64) {
        {
            const int64_t e = (base + 64 + lane < end) ? base + 64 + lane : last;
            nxt_col = cols[e];
            nxt_val = vals[e];
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            ring_consume<NT, false, SEQY>(G, R, g % RING, g, 64, expl);
            if (g == 16 - RING - 2) {
                // the next batch goes to LDS two groups before its first entries are needed
                wr_nxt[0] = __builtin_bit_cast(float, nxt_col);
                wr_nxt[64] = nxt_val;
            }
            if (g < 16 - RING)
                ring_issue<NT>(R, g % RING, g + RING, rd_cur, other);
            else
                ring_issue<NT>(R, g % RING, g + RING - 16, rd_nxt, other);
#if LK_ALS_GRAM_FENCE
            // keep the scheduler from hoisting later groups' gathers over this point: the ring
            // depth (and with it the register count) is RING, not whatever fits
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        float *tw = wr_cur;
        wr_cur = wr_nxt;
        wr_nxt = tw;
        const float *tr = rd_cur;
        rd_cur = rd_nxt;
        rd_nxt = tr;
    }
    // tail batch (< 64 entries): wave-uniform branches with no memory operation inside
    if (base < end) {
        const int nb = (int)(end - base);
        const int ngroups = (nb + 3) >> 2;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g < ngroups) ring_consume<NT, true, SEQY>(G, R, g % RING, g, nb, expl);
            if (g < 16 - RING) ring_issue<NT>(R, g % RING, g + RING, rd_cur, other);
        }
    }
}

// ---- the same accumulation with the gathered rows prefetched into LDS (k = 64 only) ---------
//
// `global_load_lds_dwordx4`: every lane names 16 bytes of a factor row and the memory pipe
// writes them to LDS at M0 + 16 * lane -- one instruction moves a group's four rows (1 KiB)
// without touching a VGPR.  The solver's L image (8.4 KiB) is idle while the normal matrix is
// built, so it holds a ring of DMA_RING = 8 groups = 32 CSR entries in flight per wave, twice
// the register ring, for 20 registers less; the operands come back with one ds_read_b128 per
// lane (each lane reads the 16 bytes "its" load brought: conflict-free by construction).
// hipcc does not count these loads, so the waits are explicit: when group g is consumed the
// groups g+1 .. g+DMA_RING-1 (or fewer at the end of the row) were issued after it, and
// `s_waitcnt vmcnt(that many)` is exactly "group g has landed" (loads retire in order; any
// other memory operation in between only makes the wait more conservative).
#ifndef LK_ALS_DMA_RING
#define LK_ALS_DMA_RING 8
#endif
constexpr int DMA_RING = LK_ALS_DMA_RING;
constexpr int GRAM_DMA_WORDS = DMA_RING * 256;  // ring, followed by nothing

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most `n` (wave-uniform, 0 .. DMA_RING-1) loads still in flight
__device__ __forceinline__ void wait_vm_upto(const int n)
{
    if (n >= 7) wait_vm<7>();
    else if (n == 6) wait_vm<6>();
    else if (n == 5) wait_vm<5>();
    else if (n == 4) wait_vm<4>();
    else if (n == 3) wait_vm<3>();
    else if (n == 2) wait_vm<2>();
    else if (n == 1) wait_vm<1>();
    else wait_vm<0>();
}

// group g of the staged batch -> ring slot `slot_idx` (compile-time)
__device__ __forceinline__ void dma_issue(const unsigned ring_lds, const int slot_idx, const int g,
                                          const float *stage_slot,
                                          const float *__restrict__ other)
{
    const int lane = lane_id();
    const unsigned col = __builtin_bit_cast(unsigned, stage_slot[g * 4]);
    // one v_mad_u64_u32: (this lane's 16 bytes of row 0) + col * 256
    const uint64_t lane_base = (uint64_t)(uintptr_t)(other + (lane & 15) * 4);
    const float *src = reinterpret_cast<const float *>(lane_base + (uint64_t)col * 256ull);
    const unsigned dst = ring_lds + slot_idx * 1024;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
}

struct DmaOperand {
    f32x4 q;
    float v;
};
// group g's operands out of the ring (its load must have landed: wait_vm first)
__device__ __forceinline__ DmaOperand dma_fetch(const float *ring, const int slot_idx, const int g,
                                                const float *stage_slot)
{
    DmaOperand o;
    o.q = *reinterpret_cast<const f32x4 *>(ring + slot_idx * 256 + lane_id() * 4);
    o.v = stage_slot[64 + g * 4];
    return o;
}

template <bool MASKED, bool SEQY = false>
__device__ __forceinline__ void dma_apply(Gram<4> &G, const DmaOperand &o, const int g,
                                          const int nb, const bool expl)
{
    const int lane = lane_id();
    const bool live = !MASKED || (g * 4 + (lane >> 4)) < nb;
    const float v = o.v;
    const float va = expl ? 1.0f : v;
    float q[4] = {o.q.x, o.q.y, o.q.z, o.q.w}, a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        q[t] = live ? q[t] : 0.f;
        a[t] = q[t] * va;
    }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti)
            G.t[tidx(ti, tj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], q[tj],
                                                                     G.t[tidx(ti, tj)], 0, 0, 0);
    const float v1 = expl ? v : v + 1.0f;
    if constexpr (SEQY) {
        y_chain_step<4>(G.yc, q, v1);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) G.y[t] = fmaf(q[t], v1, G.y[t]);
    }
}

template <bool MASKED, bool SEQY = false>
__device__ __forceinline__ void dma_consume(Gram<4> &G, const float *ring, const int slot_idx,
                                            const int g, const int nb, const float *stage_slot,
                                            const bool expl)
{
    dma_apply<MASKED, SEQY>(G, dma_fetch(ring, slot_idx, g, stage_slot), g, nb, expl);
}

template <int NT>
__host__ __device__ constexpr int slab_floats();
template <int NT>
__device__ __forceinline__ void slab_store(const Gram<NT> &G, float *__restrict__ slab);
template <int NT>
__device__ __forceinline__ void gram_zero(Gram<NT> &G);

// ring: GRAM_DMA_WORDS floats, stage: GRAM_STAGE_WORDS floats, both wave-private LDS
// `slab` (optional; reference-order work units, als_plan.h): at every 256-entry boundary that is
// followed by more entries the accumulators are stored to *slab (the next slab follows it) and
// start again from zero -- matrixmultiply's KC = 256 blocks, each an fma chain of its own -- while
// the gather ring keeps running: the wave never drains its memory queue inside a unit.  (The
// slab stores count in vmcnt like the loads: the first waits after a boundary are a little more
// conservative than needed, never less.)  The last block of the range is left in G.
template <bool SEQY = false>
__device__ __forceinline__ void gram_accumulate_dma(Gram<4> &G, const int32_t *__restrict__ cols,
                                                    const float *__restrict__ vals, int64_t beg,
                                                    int64_t end, const float *__restrict__ other,
                                                    const bool expl, float *ring, float *stage,
                                                    float *__restrict__ slab = nullptr)
{
    constexpr int RING = DMA_RING;
    static_assert(RING == 8 || RING == 4, "DMA ring: 4 or 8 groups");
    const int lane = lane_id();
    const int64_t last = end - 1;  // end > beg
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t) reinterpret_cast<void *>(ring));

    float *wr_cur = stage + lane, *wr_nxt = stage + 128 + lane;
    const float *rd_cur = stage + (lane >> 4), *rd_nxt = stage + 128 + (lane >> 4);
    {
        const int64_t e = (beg + lane < end) ? beg + lane : last;
        wr_cur[0] = __builtin_bit_cast(float, cols[e]);
        wr_cur[64] = vals[e];
    }
    // groups of the whole row; group gg lives in batch gg / 16
    const int total = (int)((end - beg + 3) >> 2);
#pragma unroll
    for (int g = 0; g < RING; ++g)
        if (g < total) dma_issue(ring_lds, g, g, rd_cur, other);

    int64_t base = beg;
    int g0 = 0;  // first group of the current batch
    // batches that are followed by at least RING more groups: no guards, constant waits
    for (; g0 + 16 + RING <= total; base += 64, g0 += 16) {
        int nxt_col;
        float nxt_val;
        {
            const int64_t e = (base + 64 + lane < end) ? base + 64 + lane : last;
            nxt_col = cols[e];
            nxt_val = vals[e];
        }
#if LK_ALS_DMA_PIPE
        // operands one group ahead: group g+1 is read out of the ring (its load is the next
        // one to land) before the 10 MFMAs of group g are issued, so the ds_read latency
        // hides behind them.  (Group 0's operands of the NEXT batch are fetched by that
        // batch's first iteration: the pipeline restarts at batch boundaries.)
        wait_vm<RING - 1>();
        DmaOperand cur = dma_fetch(ring, 0, 0, rd_cur);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            DmaOperand nxt = cur;
            if (g + 1 < 16) {
                wait_vm<RING - 2>();
                nxt = dma_fetch(ring, (g + 1) % RING, g + 1, rd_cur);
            }
            dma_apply<false, SEQY>(G, cur, g, 64, expl);
            if (g == 16 - RING - 2) {
                wr_nxt[0] = __builtin_bit_cast(float, nxt_col);
                wr_nxt[64] = nxt_val;
            }
            if (g < 16 - RING)
                dma_issue(ring_lds, g % RING, g + RING, rd_cur, other);
            else
                dma_issue(ring_lds, g % RING, g + RING - 16, rd_nxt, other);
            cur = nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            wait_vm<RING - 1>();
            dma_consume<false, SEQY>(G, ring, g % RING, g, 64, rd_cur, expl);
            if (g == 16 - RING - 2) {
                wr_nxt[0] = __builtin_bit_cast(float, nxt_col);
                wr_nxt[64] = nxt_val;
            }
            if (g < 16 - RING)
                dma_issue(ring_lds, g % RING, g + RING, rd_cur, other);
            else
                dma_issue(ring_lds, g % RING, g + RING - 16, rd_nxt, other);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (slab && ((g0 + 16) & 63) == 0) {  // (wave-uniform) a 256-entry block is complete
            slab_store<4>(G, slab);
            slab += slab_floats<4>();
            gram_zero<4>(G);
        }
        float *tw = wr_cur;
        wr_cur = wr_nxt;
        wr_nxt = tw;
        const float *tr = rd_cur;
        rd_cur = rd_nxt;
        rd_nxt = tr;
    }
    // the last batches: wave-uniform guards, waits that shrink towards the end of the row
    for (; g0 < total; base += 64, g0 += 16) {
        const int nb = (int)((end - base < 64) ? end - base : 64);
        const bool more = g0 + 16 < total;  // another batch follows
        int nxt_col = 0;
        float nxt_val = 0.f;
        if (more) {
            const int64_t e = (base + 64 + lane < end) ? base + 64 + lane : last;
            nxt_col = cols[e];
            nxt_val = vals[e];
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int gg = g0 + g;
            if (gg < total) {
                const int later = total - 1 - gg;  // groups issued after this one
                wait_vm_upto(later < RING - 1 ? later : RING - 1);
                dma_consume<true, SEQY>(G, ring, g % RING, g, nb, rd_cur, expl);
            }
            if (g == 16 - RING - 2 && more) {
                wr_nxt[0] = __builtin_bit_cast(float, nxt_col);
                wr_nxt[64] = nxt_val;
            }
            if (gg + RING < total) {
                if (g < 16 - RING)
                    dma_issue(ring_lds, g % RING, g + RING, rd_cur, other);
                else
                    dma_issue(ring_lds, g % RING, g + RING - 16, rd_nxt, other);
            }
        }
        if (slab && ((g0 + 16) & 63) == 0 && g0 + 16 < total) {  // a block boundary, more follows
            slab_store<4>(G, slab);
            slab += slab_floats<4>();
            gram_zero<4>(G);
        }
        float *tw = wr_cur;
        wr_cur = wr_nxt;
        wr_nxt = tw;
        const float *tr = rd_cur;
        rd_cur = rd_nxt;
        rd_nxt = tr;
    }
}

// ---- a FULL 256-entry chunk (k = 64) ---------------------------------------------------------
// Reference-order rows are cut into matrixmultiply's KC = 256 blocks (als_plan.h): every chunk of
// such a row but its last has exactly 256 entries = 64 groups = 4 batches.  The general routine
// above sends the last batch of any range down its guarded path (wave-uniform tests, waits that
// shrink); for a 256-entry chunk that is a quarter of the work.  Here the batch loop is unrolled
// over the four batches, so every guard and every wait count is a compile-time constant.  Same
// operations in the same order: bit-identical to gram_accumulate_dma on the same range.
#ifndef LK_ALS_CHUNK256_FAST
#define LK_ALS_CHUNK256_FAST 1
#endif
__device__ __forceinline__ void gram_accumulate_dma_256(Gram<4> &G, const int32_t *__restrict__ cols,
                                                        const float *__restrict__ vals, int64_t beg,
                                                        const float *__restrict__ other,
                                                        const bool expl, float *ring, float *stage)
{
    constexpr int RING = DMA_RING, TOTAL = 64;
    const int lane = lane_id();
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t) reinterpret_cast<void *>(ring));
    float *const wr0 = stage + lane, *const wr1 = stage + 128 + lane;
    const float *const rd0 = stage + (lane >> 4), *const rd1 = stage + 128 + (lane >> 4);
    wr0[0] = __builtin_bit_cast(float, cols[beg + lane]);
    wr0[64] = vals[beg + lane];
#pragma unroll
    for (int g = 0; g < RING; ++g) dma_issue(ring_lds, g, g, rd0, other);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float *const wr_nxt = (b & 1) ? wr0 : wr1;
        const float *const rd_cur = (b & 1) ? rd1 : rd0, *const rd_nxt = (b & 1) ? rd0 : rd1;
        int nxt_col = 0;
        float nxt_val = 0.f;
        if (b < 3) {
            nxt_col = cols[beg + 64 * (b + 1) + lane];
            nxt_val = vals[beg + 64 * (b + 1) + lane];
        }
        {
            const int later = TOTAL - 1 - 16 * b;
            wait_vm_upto(later < RING - 1 ? later : RING - 1);
        }
        DmaOperand cur = dma_fetch(ring, 0, 0, rd_cur);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int gg = 16 * b + g;
            DmaOperand nxt = cur;
            if (g + 1 < 16) {
                // group gg + 1 must have landed: the groups issued after it are still in flight
                const int later = TOTAL - 1 - (gg + 1);
                const int issued_after = (gg + RING < TOTAL ? gg + RING - 1 : TOTAL - 1) - (gg + 1);
                wait_vm_upto(issued_after < later ? issued_after : later);
                nxt = dma_fetch(ring, (g + 1) % RING, g + 1, rd_cur);
            }
            dma_apply<false>(G, cur, g, 64, expl);
            if (g == 16 - RING - 2 && b < 3) {
                wr_nxt[0] = __builtin_bit_cast(float, nxt_col);
                wr_nxt[64] = nxt_val;
            }
            if (gg + RING < TOTAL) {
                if (g < 16 - RING)
                    dma_issue(ring_lds, g % RING, g + RING, rd_cur, other);
                else
                    dma_issue(ring_lds, g % RING, g + RING - 16, rd_nxt, other);
            }
            cur = nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// slab layout: tile e as [64 lanes][4 registers] (one 16-byte store / load per lane and tile:
// a slab is written in als_tiles + 1 instructions where the register-major layout took 4 x as
// many -- it matters since round 5, when a wave stores a slab every 256 entries), then y as
// [64 lanes][NT]
template <int NT>
__host__ __device__ constexpr int slab_floats()
{
    return (als_tiles(NT) * 4 + NT) * 64;
}

template <int NT>
__device__ __forceinline__ void slab_store(const Gram<NT> &G, float *__restrict__ slab)
{
    const int lane = lane_id();
    // (LK_ALS_SLAB_NT: streaming stores -- a slab is written once and read once by another
    // kernel; kept out of the L2 it would otherwise share with the gathered factor rows)
#pragma unroll
    for (int e = 0; e < als_tiles(NT); ++e) {
#if LK_ALS_SLAB_NT
        __builtin_nontemporal_store(G.t[e], reinterpret_cast<f32x4 *>(slab + (e * 64 + lane) * 4));
#else
        *reinterpret_cast<f32x4 *>(slab + (e * 64 + lane) * 4) = G.t[e];
#endif
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) slab[als_tiles(NT) * 256 + lane * NT + t] = G.y[t];
}

template <int NT>
__device__ __forceinline__ void slab_add(Gram<NT> &G, const float *__restrict__ slab)
{
    const int lane = lane_id();
#pragma unroll
    for (int e = 0; e < als_tiles(NT); ++e)
        G.t[e] += *reinterpret_cast<const f32x4 *>(slab + (e * 64 + lane) * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) G.y[t] += slab[als_tiles(NT) * 256 + lane * NT + t];
}

template <int NT>
__device__ __forceinline__ void gram_zero(Gram<NT> &G)
{
#pragma unroll
    for (int e = 0; e < als_tiles(NT); ++e) G.t[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) G.y[t] = 0.f;
}

// ---- slab groups: slab[head] += slab[head + 1] + ... + slab[head + cnt - 1] ------------------
// (grid: groups x ceil(slab_floats / 1024); float4 per thread; chunk order inside the group)
__global__ __launch_bounds__(256) void slab_group_reduce_kernel(float *__restrict__ slabs,
                                                                int64_t slab_floats,
                                                                const int32_t *__restrict__ grp_head,
                                                                const int32_t *__restrict__ grp_cnt,
                                                                int parts)
{
    const int g = blockIdx.x / parts;
    const int64_t e = ((int64_t)(blockIdx.x - g * parts) * 256 + threadIdx.x) * 4;
    if (e >= slab_floats) return;
    float *head = slabs + (size_t)grp_head[g] * slab_floats + e;
    const int cnt = grp_cnt[g];
    f32x4 acc = *reinterpret_cast<const f32x4 *>(head);
    int c = 1;
    // (reference-order rows are ONE group of up to thousands of slabs: eight loads in flight,
    // the additions strictly in chunk order)
    for (; c + 8 <= cnt; c += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4 *>(head + (size_t)(c + u) * slab_floats));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; c < cnt; ++c)
        acc += *reinterpret_cast<const f32x4 *>(head + (size_t)c * slab_floats);
    *reinterpret_cast<f32x4 *>(head) = acc;
}

int launch_slab_group_reduce(const lk_als_plan *p, float *slabs, size_t slab_floats, hipStream_t st)
{
    if (p->n_groups <= 0) return LK_OK;
    const int parts = (int)((slab_floats / 4 + 255) / 256);
    hipLaunchKernelGGL(slab_group_reduce_kernel, dim3((unsigned)(p->n_groups * parts)), dim3(256),
                       0, st, slabs, (int64_t)slab_floats, p->d_grp_head, p->d_grp_cnt, parts);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ---- chunk kernel: one wave per chunk of a long row ------------------------
template <int NT>
__host__ __device__ constexpr int chunk_lds_floats()
{
    return GRAM_STAGE_WORDS + ((LK_ALS_GRAM_DMA && NT == 4) ? GRAM_DMA_WORDS : 0);
}

// (body + thin __global__ wrapper: the fused kernel below runs chunk blocks and solve blocks in
// ONE launch)
template <int NT, bool EXPL>
__device__ __forceinline__ void als_chunk_body(
    const int32_t *__restrict__ indices, const float *__restrict__ values,
    const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_len,
    int64_t n_chunks, const float *__restrict__ other, int ld, float *__restrict__ slabs,
    const int64_t blk, float *__restrict__ lds_flat,
    const int32_t *__restrict__ chunk_slab = nullptr, const int block_len = 0)
{
    constexpr bool DMA = LK_ALS_GRAM_DMA && NT == 4;
    float(*stage_all)[chunk_lds_floats<NT>()] =
        reinterpret_cast<float(*)[chunk_lds_floats<NT>()]>(lds_flat);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c = blk * 4 + wave;
    if (c >= n_chunks) return;
    Gram<NT> G;
    gram_zero<NT>(G);
    const int64_t beg = chunk_beg[c];
    // the unit's first slab (a unit of several 256-entry blocks stores one slab per block)
    float *slab = slabs + (size_t)(chunk_slab ? chunk_slab[c] : c) * slab_floats<NT>();
    if constexpr (DMA) {
        const int len = chunk_len[c];
        if (block_len == 256 && len > 256) {  // (wave-uniform) a reference-order work unit
            gram_accumulate_dma(G, indices, values, beg, beg + len, other, EXPL,
                                stage_all[wave] + GRAM_STAGE_WORDS, stage_all[wave], slab);
            slab += (size_t)((len - 1) >> 8) * slab_floats<NT>();  // the last block is still in G
        } else if (LK_ALS_CHUNK256_FAST && len == 256)  // a full reference-order block
            gram_accumulate_dma_256(G, indices, values, beg, other, EXPL,
                                    stage_all[wave] + GRAM_STAGE_WORDS, stage_all[wave]);
        else
            gram_accumulate_dma(G, indices, values, beg, beg + len, other, EXPL,
                                stage_all[wave] + GRAM_STAGE_WORDS, stage_all[wave]);
    } else
        gram_accumulate<NT>(G, indices, values, beg, beg + chunk_len[c], other, ld, EXPL,
                            stage_all[wave]);
    slab_store<NT>(G, slab);
}

template <int NT, bool EXPL>
__global__ __launch_bounds__(256) void als_chunk_kernel(
    const int32_t *__restrict__ indices, const float *__restrict__ values,
    const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_len,
    int64_t n_chunks, const float *__restrict__ other, int ld, float *__restrict__ slabs,
    const int32_t *__restrict__ chunk_slab, int block_len)
{
    __shared__ __attribute__((aligned(16))) float lds_flat[4 * chunk_lds_floats<NT>()];
    als_chunk_body<NT, EXPL>(indices, values, chunk_beg, chunk_len, n_chunks, other, ld, slabs,
                             (int64_t)blockIdx.x, lds_flat, chunk_slab, block_len);
}

// ---- solve: lane R owns row R of the (primed) normal matrix -----------------
//
// Right-looking Cholesky with the matrix rows in registers (lane i: a[c] = A'[i][c]).
// Step j: pivot broadcast (v_readlane), rinv = rsq(pivot), column j of L
// (strictly lower: zero on and above the diagonal, so later updates need no lane
// masks) is written to a packed LDS image; the NEXT pivot's update uses a
// v_readlane fast path, the bulk of the trailing update reads L_cj back as
// wave-uniform ds_read_b128 broadcasts (4 multipliers per LDS instruction), which
// keeps the O(k^3/3) loop at one v_fma per element.  Forward substitution runs on
// the register rows, back substitution on the LDS image (column access).
//
// Packed strictly-lower image: column j holds rows c in [c0(j), KP), c0 = (j+1)&~3
// (16-byte aligned segments); off(j) = 4*KP*m - 8m^2 + 4m + r*(KP - 4m), j = 4m + r.
template <int KP>
struct LPack {
    __host__ __device__ static constexpr int c0(int j) { return (j + 1) & ~3; }
    __host__ __device__ static constexpr int off(int j)
    {
        const int m = j >> 2, r = j & 3;
        return 4 * KP * m - 8 * m * m + 4 * m + r * (KP - 4 * m);
    }
    static constexpr int SIZE = KP * KP / 2 + KP;
};

// Row storage: KP floats as KP/2 register PAIRS so the trailing update can use
// v_pk_fma_f32 (two FMAs per VALU issue).  AT(a, c) is element c.
#define AT(a, c) ((a)[(c) >> 1][(c) & 1])

// One pipelined factorisation step (J compile-time): with column J of L in `lj`,
// (1) finish column J+1 through the v_readlane fast path and start ITS pivot chain
// (readlane -> rsq -> scale -> LDS write) while (2) the bulk of step J's trailing
// update (c >= J+2) streams its multipliers back from LDS.  The two are independent,
// so the v_pk_fma stream covers the pivot latency.
template <int KP, int J>
__device__ __forceinline__ void chol_step(f32x2 (&a)[KP / 2], float &lj, float &dinv,
                                          float &minpiv, float *__restrict__ lds)
{
    using P = LPack<KP>;
    const int lane = lane_id();
    const float ln = bcast(lj, J + 1);
    AT(a, J + 1) = fmaf(-lj, ln, AT(a, J + 1));
    const float ajj = bcast(AT(a, J + 1), J + 1);
    minpiv = fminf(minpiv, ajj);
    const float rinv = __builtin_amdgcn_rsqf(ajj);
    // 1/L_jj goes to a small LDS array (read back once, per lane, after the factorisation):
    // a per-step `dinv = lane == j ? rinv : dinv` select is sunk by the compiler to the end,
    // which keeps all k rinv values alive in registers
    if (lane == 0) lds[P::SIZE + J + 1] = rinv;
    const float lnext = (lane > J + 1) ? AT(a, J + 1) * rinv : 0.f;
    AT(a, J + 1) = lnext;
    if constexpr (J + 2 < KP) {
        if (lane >= P::c0(J + 1) && lane < KP) lds[P::off(J + 1) + lane - P::c0(J + 1)] = lnext;
    }
    // multipliers L_cJ, c >= J+2, as wave-uniform ds_read_b128 broadcasts; the reads run
    // LOOKAHEAD groups ahead of the FMAs that use them
    constexpr int C0 = (J + 2) & ~3;
    constexpr int NG = (KP - C0) / 4;
    constexpr int LOOKAHEAD = LK_ALS_LOOKAHEAD;
    if constexpr (NG > 0) {
        const float *col = lds + P::off(J) - P::c0(J);
        const f32x2 nl = f32x2{-lj, -lj};
        f32x4 lq[NG];
#pragma unroll
        for (int g = 0; g < NG && g < LOOKAHEAD; ++g)
            lq[g] = *reinterpret_cast<const f32x4 *>(col + C0 + 4 * g);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + LOOKAHEAD < NG)
                lq[g + LOOKAHEAD] =
                    *reinterpret_cast<const f32x4 *>(col + C0 + 4 * (g + LOOKAHEAD));
            const int c4 = C0 + 4 * g;
            // a_ic -= L_iJ * L_cJ for c = c4 .. c4+3, c >= J+2
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c4 + 2 * h;
                const f32x2 m = f32x2{lq[g][2 * h], lq[g][2 * h + 1]};
                if (c >= J + 2)
                    a[c >> 1] = __builtin_elementwise_fma(nl, m, a[c >> 1]);
                else if (c + 1 >= J + 2)
                    AT(a, c + 1) = fmaf(-lj, m[1], AT(a, c + 1));
            }
            // pin the updates here: without it the compiler sinks every FMA chain down to
            // the step that first reads a[c] (a left-looking schedule that keeps all
            // multipliers alive and spills hundreds of registers)
            asm volatile("" : "+v"(a[c4 >> 1]), "+v"(a[(c4 >> 1) + 1]));
        }
    }
    lj = lnext;
    __builtin_amdgcn_sched_barrier(0);
}

template <int KP, int... Js>
__device__ __forceinline__ void chol_steps(f32x2 (&a)[KP / 2], float &lj, float &dinv,
                                           float &minpiv, float *__restrict__ lds,
                                           std::integer_sequence<int, Js...>)
{
    (chol_step<KP, Js>(a, lj, dinv, minpiv, lds), ...);
}

// a: row `lane` of A' (entries c <= lane valid, anything above), b = rhs.  On return
// b = solution for primed row `lane`; returns the smallest pivot seen (<= 0 or a
// non-finite solution => not SPD).
template <int KP>
__device__ __forceinline__ float chol_solve(f32x2 (&a)[KP / 2], float &b,
                                            float *__restrict__ lds
#ifdef LK_ALS_PHASES
                                            ,
                                            unsigned long long *tmid
#endif
)
{
    using P = LPack<KP>;
    const int lane = lane_id();
    float minpiv = 3.0e38f;
    float dinv = 0.f;  // lane j keeps 1 / L_jj

    // column 0
    float lj;
    {
        const float ajj = bcast(AT(a, 0), 0);
        minpiv = fminf(minpiv, ajj);
        const float rinv = __builtin_amdgcn_rsqf(ajj);
        if (lane == 0) lds[P::SIZE] = rinv;
        lj = (lane > 0) ? AT(a, 0) * rinv : 0.f;  // strictly-lower column 0
        AT(a, 0) = lj;
        if (lane < KP) lds[P::off(0) + lane - P::c0(0)] = lj;
    }
    chol_steps<KP>(a, lj, dinv, minpiv, lds, std::make_integer_sequence<int, KP - 1>{});
#ifdef LK_ALS_PHASES
    asm volatile("" : "+v"(lj), "+v"(b));
    *tmid = __builtin_amdgcn_s_memtime();
#endif
    dinv = (lane < KP) ? lds[P::SIZE + lane] : 0.f;
    // forward: L z = y.  a[j] is zero for lanes <= j, so no masks: lane i only
    // receives the terms j < i; z_i = b_i * dinv_i.
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const float zj = bcast(b * dinv, j);
        b = fmaf(-AT(a, j), zj, b);
    }
    b *= dinv;
    // backward: L^T x = z.  Lane i needs L[j][i] (j > i) = column i of the LDS image,
    // read four rows at a time (ds_read_b128; rows <= i inside the column are stored
    // zeros, rows below c0(i) are outside it).
    const int my_c0 = (lane + 1) & ~3;
    int my_off = P::off(lane) - my_c0;
    // tie the column address to the finished forward pass: otherwise all 16 ds_read_b128 of
    // the back substitution are hoisted above it and sit on 64 VGPRs next to the 64 of `a`
    asm volatile("" : "+v"(my_off), "+v"(b));
    const float *mycol = lds + my_off;
#pragma unroll
    for (int j4 = KP / 4 - 1; j4 >= 0; --j4) {
        f32x4 l4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane < KP - 1 && 4 * j4 >= my_c0) l4 = *reinterpret_cast<const f32x4 *>(mycol + 4 * j4);
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const int j = 4 * j4 + u;
            if (j >= 1) {
                const float xj = bcast(b * dinv, j);
                b = fmaf(-l4[u], xj, b);
            }
        }
    }
    b *= dinv;
    return minpiv;
}

// ---- hybrid Cholesky (LK_ALS_PANEL == 2) -------------------------------------------------------
//
// The matrix stays in the accumulator tiles; panels of FOUR columns J .. J+3 are
//   E. extracted into the lane = row layout (lane i: A'[i][J .. J+3]; by symmetry these are the
//      four registers of row group MG of the tiles (TJ, t): one masked ds_write_b128 per tile,
//      one ds_read_b128 per lane),
//   F. factored there with registers and v_readlane only (no LDS in the dependent chain); the
//      forward substitution rides along and every finished column goes to the strictly-lower
//      L image of the back substitution,
//   C. turned into MFMA operands by a 4 x 4 (register x row group) transposition made of two
//      v_permlane32_swap and two v_permlane16_swap: q[t], lane (s, c) = L[16 t + c][J + s],
//   U. applied to everything right of the panel: ONE v_mfma_f32_16x16x4_f32 per tile with
//      -q[ti] as the A operand and q[tj] as the B operand.
// tools/emul/hybrid_chol.py is the lane-level NumPy model of this; tools/ub/permlane_swap.hip
// checks the swap semantics on the device.
template <int NT>
__host__ __device__ constexpr int hybrid_lds_floats()
{
    // L image | KP reciprocal pivots | extraction scratch (64 lanes x 4)
    return LPack<NT * 16>::SIZE + NT * 16 + 256;
}


#ifndef LK_ALS_NOBRANCH
#define LK_ALS_NOBRANCH 1
#endif
template <int NT, int M>
__device__ __forceinline__ void hybrid_step(Gram<NT> &G, float &b, float &minpiv,
                                            float *__restrict__ lds, const int lane)
{
    constexpr int KP = NT * 16, J = 4 * M, TJ = M >> 2, MG = M & 3;
    using P = LPack<KP>;
    const int slot = lane >> 4;
    float *rinvarr = lds + P::SIZE;
    float *scr = rinvarr + KP;

    // E. extraction
#pragma unroll
    for (int t = TJ; t < NT; ++t)
        if (slot == MG)
            *reinterpret_cast<f32x4 *>(&scr[(t * 16 + (lane & 15)) * 4]) = G.t[tidx(TJ, t)];
    f32x4 pv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (slot >= TJ && lane < KP) pv = *reinterpret_cast<const f32x4 *>(&scr[lane * 4]);
    float pr[4] = {pv.x, pv.y, pv.z, pv.w};

    // F. the four columns
    float lp[4];
#pragma unroll
    for (int s0 = 0; s0 < 4; ++s0) {
        const int j = J + s0;
        const float piv = bcast(pr[s0], j);
        minpiv = fminf(minpiv, piv);
        const float rinv = __builtin_amdgcn_rsqf(piv);
#if LK_ALS_NOBRANCH
        // (no exec-mask round trips in the column loop: rinv is wave-uniform, every lane stores
        // the same value to the same word; lanes outside the stored segment of column j aim at
        // their own word of the extraction scratch, which is dead until the next panel)
        rinvarr[j] = rinv;
        const float lj = (lane > j) ? pr[s0] * rinv : 0.f;  // strictly-lower column j
        if (j + 1 < KP) {
            const bool in = lane >= P::c0(j) && lane < KP;
            float *dst = in ? &lds[P::off(j) + lane - P::c0(j)] : &scr[lane];
            *dst = lj;
        }
#else
        if (lane == 0) rinvarr[j] = rinv;
        const float lj = (lane > j) ? pr[s0] * rinv : 0.f;  // strictly-lower column j
        if (j + 1 < KP) {
            if (lane >= P::c0(j) && lane < KP) lds[P::off(j) + lane - P::c0(j)] = lj;
        }
#endif
        // forward substitution: z_j = y_j / L_jj, y -= L[:, j] z_j
        const float zj = bcast(b, j) * rinv;
        b = fmaf(-lj, zj, b);
#pragma unroll
        for (int s = s0 + 1; s < 4; ++s) pr[s] = fmaf(-lj, bcast(lj, J + s), pr[s]);
        lp[s0] = lj;
    }
    if constexpr (J + 4 < KP) {
        // C. register x row-group transposition: lp[s] @ group t  ->  q[t] @ group s
        swap32(lp[0], lp[2]);
        swap32(lp[1], lp[3]);
        swap16(lp[0], lp[1]);
        swap16(lp[2], lp[3]);
        // U. rank-4 update (tile row TJ first: the next panel is extracted from it)
        float nq[NT];
#pragma unroll
        for (int t = TJ; t < NT; ++t) nq[t] = -lp[t];
#pragma unroll
        for (int ti = TJ; ti < NT; ++ti)
#pragma unroll
            for (int t2 = ti; t2 < NT; ++t2)
                G.t[tidx(ti, t2)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    nq[ti], lp[t2], G.t[tidx(ti, t2)], 0, 0, 0);
    }
}

template <int NT, int... Ms>
__device__ __forceinline__ void hybrid_steps(Gram<NT> &G, float &b, float &minpiv,
                                             float *__restrict__ lds, const int lane,
                                             std::integer_sequence<int, Ms...>)
{
    (hybrid_step<NT, Ms>(G, b, minpiv, lds, lane), ...);
}

// G: accumulator tiles of A' and the right-hand side (every lane: y'[16 t + sub]).  Returns
// the smallest pivot; b = solution for primed row `lane`.
template <int NT>
__device__ __forceinline__ float hybrid_solve(Gram<NT> &G, float &b, float *__restrict__ lds
#ifdef LK_ALS_PHASES
                                              ,
                                              unsigned long long *tmid
#endif
)
{
    constexpr int KP = NT * 16;
    using P = LPack<KP>;
    // the lane number is made opaque here so that nothing derived from it for the solver
    // (row-group masks, LDS addresses) is kept alive across the normal-matrix loop
    int lane = lane_id();
    asm volatile("" : "+v"(lane));
    // rhs for primed row `lane`: tile lane >> 4, sub lane & 15 -> G.y[lane >> 4] of this lane
    b = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) b = ((lane >> 4) == tt) ? G.y[tt] : b;
    float minpiv = 3.0e38f;
    hybrid_steps<NT>(G, b, minpiv, lds, lane, std::make_integer_sequence<int, KP / 4>{});
#ifdef LK_ALS_PHASES
    asm volatile("" : "+v"(b));
    *tmid = __builtin_amdgcn_s_memtime();
#endif
    // backward: L^T x = z, lane = primed row (z = b / L_jj after the folded forward pass)
    const float dinv = (lane < KP) ? lds[P::SIZE + lane] : 0.f;
    b *= dinv;
    const int my_c0 = (lane + 1) & ~3;
    const float *mycol = lds + P::off(lane) - my_c0;
#pragma unroll
    for (int j4 = KP / 4 - 1; j4 >= 0; --j4) {
        f32x4 l4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane < KP - 1 && 4 * j4 >= my_c0) l4 = *reinterpret_cast<const f32x4 *>(mycol + 4 * j4);
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const int j = 4 * j4 + u;
            if (j >= 1) {
                const float xj = bcast(b * dinv, j);
                b = fmaf(-l4[u], xj, b);
            }
        }
    }
    b *= dinv;
    return minpiv;
}

// transposition buffer: row R' (tile row tr = R' >> 4) keeps its (tr+1)*16 lower
// entries; stride inside tile row tr is (tr+1)*16 + 4 floats (16-byte aligned).  For NT = 4
// the transposition runs in TWO passes (tile rows 0..2, then tile row 3 alone, both from
// offset 0): 6.9 KiB instead of 11 KiB, so that the wave's LDS is the 8.7 KiB of the L image
// and FOUR workgroups fit a CU.
template <int NT>
struct TPack {
    static constexpr int SPLIT = NT == 4 ? 3 : NT;  // tile rows of the first pass
    __host__ __device__ static constexpr int stride(int tr) { return (tr + 1) * 16 + 4; }
    __host__ __device__ static constexpr int base(int tr)
    {
        int b = 0;
        for (int t = (tr >= SPLIT ? SPLIT : 0); t < tr; ++t) b += 16 * stride(t);
        return b;
    }
    static constexpr int SIZE = base(SPLIT) > 16 * stride(NT - 1) ? base(SPLIT)
                                                                  : (NT > SPLIT ? 16 * stride(NT - 1) : base(NT));
};

template <int NT>
__host__ __device__ constexpr int solve_lds_floats()
{
#if LK_ALS_PANEL
    return hybrid_lds_floats<NT>();
#endif
    // the L image is followed by the k reciprocal pivots
    return TPack<NT>::SIZE > LPack<NT * 16>::SIZE + NT * 16 ? TPack<NT>::SIZE
                                                            : LPack<NT * 16>::SIZE + NT * 16;
}

// EXPL: explicit-feedback model (explicit.rs) instead of the implicit one (implicit.rs); a
// template parameter so that the implicit instantiation carries nothing of it
// CTL: poll the task-control block (cancel) before the row and count it when done; a
// template parameter so that the uncontrolled instantiation -- the training engine's -- is
// instruction for instruction the tuned kernel
// YREF: take the right-hand side from `y_ref` ([tasks x KP], indexed by the TASK t of this launch's
// order, natural feature order; als_rhs.hip:
// the reference's summation order) instead of the accumulated one; a template parameter so that
// the default instantiation is instruction for instruction the tuned kernel
// SEQY: rows that form their own right-hand side do it in the reference's order (y_chain_step); a
// template parameter so that LK_ALS_RHS_ORDER=accurate keeps round 4's kernel instruction for
// instruction
template <int NT, bool IS64, bool EXPL, bool CTL, bool YREF = false, bool SEQY = false>
__device__ __forceinline__ void als_solve_body(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_rows,
    const int32_t *__restrict__ row_slab, const float *__restrict__ other, int ld_other,
    float *__restrict__ this_, int ld_this, const float *__restrict__ otor_p,
    const float *__restrict__ slabs, float *__restrict__ row_delta, int *__restrict__ status,
    int k, float reg, TaskCtlDev ctl, const float *__restrict__ y_ref, int chunk_rt,
    const int64_t blk, float *__restrict__ lds_flat)
{
    constexpr int KP = NT * 16;
    float(*lds_all)[solve_lds_floats<NT>()] =
        reinterpret_cast<float(*)[solve_lds_floats<NT>()]>(lds_flat);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;
    const int64_t t = blk * 4 + wave;
    if (t >= n_rows) return;
    LK_PHASE_T(ph0);
    if constexpr (CTL) {
        // AccelTask.cancel (src/accel/tasks/mod.rs:88-95): rows not started yet are skipped
        int c = 0;
        if (lane == 0) c = ctl_cancelled(ctl, (blk & 63) == 0 && wave == 0) ? 1 : 0;
        if (__builtin_amdgcn_readfirstlane(c)) return;
    }
    const int row = order[t];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    float *lds = lds_all[wave];

    // feature owned by this lane in the lane==row phase
    const int my_f = (lane & 15) * NT + (lane >> 4);
    const bool my_valid = (lane < KP) && (my_f < k);
    float *xrow = this_ + (int64_t)row * ld_this;

    if (end == beg) {  // implicit.rs:98-101
        if (lane < KP) xrow[lane] = 0.f;
        if (lane == 0) row_delta[row] = 0.f;
        if constexpr (CTL)
            if (lane == 0) ctl_advance(ctl, 1);
        return;
    }

    LK_PHASE_T(ph1);
    Gram<NT> G;
    // start from OtOr (primed, padded with identity on the pad features)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                G.t[tidx(ti, tj)][r] = otor_p[(ti * 16 + slot * 4 + r) * KP + tj * 16 + sub];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) G.y[tt] = 0.f;

    const int first_slab = row_slab[row];
    if (first_slab >= 0) {
        const int64_t n = end - beg;
        // (reference-order plans, YREF only: 256-entry chunks whose slabs were summed one after
        // the other into the row's first slab -- that one is added, after OtOr: a = otor + mtm)
        const bool refo = YREF && chunk_rt > 0;
        const int ch = refo ? chunk_rt : LK_ALS_CHUNK;
        const int ns = (int)((n + ch - 1) / ch);
        // many chunks: the groups were pre-summed into their heads (slab_group_reduce_kernel)
        const int step = refo ? ns : (ns > LK_ALS_SLAB_GROUP ? LK_ALS_SLAB_GROUP : 1);
        for (int s = 0; s < ns; s += step)
            slab_add<NT>(G, slabs + (size_t)(first_slab + s) * slab_floats<NT>());
    } else {
        // (the solver's LDS is idle while the normal matrix is built: it stages the CSR
        // batches and, for k = 64, holds the ring of prefetched factor rows)
#if LK_ALS_GRAM_DMA && LK_ALS_PANEL == 2
        if constexpr (NT == 4)
            gram_accumulate_dma<SEQY && !YREF>(G, indices, values, beg, end, other, EXPL, lds,
                                               lds + LPack<KP>::SIZE + KP);
        else
#endif
            gram_accumulate<NT, SEQY && !YREF>(G, indices, values, beg, end, other, ld_other,
                                               EXPL, lds);
    }
    if (EXPL) {
        // explicit.rs:104-107: mtm[i][i] += reg * n, AFTER the product, real features only
        // (the pad features keep the identity they got from otor_p)
        const float dg = reg * (float)(end - beg);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (slot * 4 + r == sub && (slot * 4 + r) * NT + ti < k) G.t[tidx(ti, ti)][r] += dg;
    }

#ifdef LK_ALS_PHASES
    asm volatile("" : "+v"(G.t[0]), "+v"(G.t[als_tiles(NT) - 1]));
#endif
    LK_PHASE_T(ph2);
    if (SEQY && !YREF && first_slab < 0) {
        // the chain of feature sub * NT + tt lives in lane (sub, slot = tt): hand it to every slot
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) G.y[tt] = __shfl(G.yc, sub + 16 * tt, 64);
    } else {
        // y: combine the 4 entry slots -> every lane has the full y for feature (t, sub)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            G.y[tt] += __shfl_xor(G.y[tt], 16, 64);
            G.y[tt] += __shfl_xor(G.y[tt], 32, 64);
        }
    }
    if constexpr (YREF) {
        // primed (tt, sub) <-> feature sub * NT + tt; pad features carry y = 0 (their factor
        // columns are zero, so the reference-order sum over them is exactly 0 as well)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) G.y[tt] = y_ref[(int64_t)t * KP + sub * NT + tt];
    }
#if LK_ALS_PANEL
    const float old = my_valid ? xrow[my_f] : 0.f;
    float b;
#ifdef LK_ALS_PHASES
    LK_PHASE_T(ph3);
    unsigned long long ph4 = 0;
    const float minpiv = hybrid_solve<NT>(G, b, lds, &ph4);
    asm volatile("" : "+v"(b));
    LK_PHASE_T(ph5);
#else
    // The factorisation is one long dependent chain (v_readlane -> rsq -> mul -> fma per column):
    // a wave in it has ONE ready instruction at a time, while the co-resident waves in their
    // normal-matrix loops always have several.  LK_ALS_SOLVE_PRIO > 0 lets the chain win the
    // SIMD's issue arbitration for its duration.
#if LK_ALS_SOLVE_PRIO
    __builtin_amdgcn_s_setprio(LK_ALS_SOLVE_PRIO);
#endif
    const float minpiv = hybrid_solve<NT>(G, b, lds);
#if LK_ALS_SOLVE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
#else
    // tile (ti,tj): lane holds D[i = slot*4+r][j = sub] = A'[ti*16+i][tj*16+j]
    //             = A'[row' = tj*16+sub][col' = ti*16 + slot*4 + r]  (symmetry)
    // lane = primed row: tile row slot (= lane >> 4), row-in-tile sub
    f32x2 a[KP / 2];
    float b = 0.f;
    {
        using T = TPack<NT>;
        int rowoff = 0;
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
            rowoff = (slot == tr) ? T::base(tr) + sub * T::stride(tr) : rowoff;
#pragma unroll
        for (int pass = 0; pass < (T::SPLIT < NT ? 2 : 1); ++pass) {
            const int t0 = pass == 0 ? 0 : T::SPLIT, t1 = pass == 0 ? T::SPLIT : NT;
#pragma unroll
            for (int tj = t0; tj < t1; ++tj)
#pragma unroll
                for (int ti = 0; ti <= tj; ++ti)
                    *reinterpret_cast<f32x4 *>(
                        &lds[T::base(tj) + sub * T::stride(tj) + ti * 16 + slot * 4]) =
                        G.t[tidx(ti, tj)];
            // the wave's LDS operations complete in order: no barrier between the passes
            const bool mine = lane < KP && slot >= t0 && slot < t1;
#pragma unroll
            for (int c4 = 0; c4 < KP / 4; ++c4) {
                if ((c4 >> 2) >= t1) continue;  // beyond this pass's widest row
                f32x4 v = pass == 0 ? f32x4{0.f, 0.f, 0.f, 0.f}
                                    : f32x4{a[c4 * 2][0], a[c4 * 2][1], a[c4 * 2 + 1][0],
                                            a[c4 * 2 + 1][1]};
                if (mine && (c4 >> 2) <= slot)
                    v = *reinterpret_cast<const f32x4 *>(&lds[rowoff + c4 * 4]);
                a[c4 * 2 + 0] = f32x2{v.x, v.y};
                a[c4 * 2 + 1] = f32x2{v.z, v.w};
            }
            if (pass == 0 && T::SPLIT < NT) {
                // columns only the last tile row has: defined (zero) before pass 2 merges
#pragma unroll
                for (int c4 = T::SPLIT * 4; c4 < KP / 4; ++c4) {
                    a[c4 * 2 + 0] = f32x2{0.f, 0.f};
                    a[c4 * 2 + 1] = f32x2{0.f, 0.f};
                }
            }
        }
    }
    // rhs for primed row `lane`: tile lane>>4, sub lane&15 -> G.y[lane>>4] of this lane
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) b = (slot == tt) ? G.y[tt] : b;
    if (lane >= KP) b = 0.f;

    const float old = my_valid ? xrow[my_f] : 0.f;
#ifdef LK_ALS_PHASES
    asm volatile("" : "+v"(a[0]), "+v"(a[KP / 2 - 1]), "+v"(b));
    LK_PHASE_T(ph3);
    unsigned long long ph4 = 0;
    const float minpiv = chol_solve<KP>(a, b, lds, &ph4);
    asm volatile("" : "+v"(b));
    LK_PHASE_T(ph5);
#else
    const float minpiv = chol_solve<KP>(a, b, lds);
#endif
#endif
    // not SPD: a non-positive pivot, or NaN/Inf anywhere in the solution
    const bool bad = !(minpiv > 0.f) || (my_valid && !(fabsf(b) <= 3.0e38f));
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);

    float d = 0.f;
    if (my_valid) {
        xrow[my_f] = b;
        d = b - old;
    }
    const float d2 = wave_sum(d * d);
    if (lane == 0) row_delta[row] = d2;
    if constexpr (CTL)
        if (lane == 0) ctl_advance(ctl, 1);  // progress unit = rows (tasks/mod.rs:97-105)
#ifdef LK_ALS_PHASES
    LK_PHASE_T(ph6);
    LK_PHASE_ADD(0, ph0, ph1);
    LK_PHASE_ADD(1, ph1, ph2);
    LK_PHASE_ADD(2, ph2, ph3);
    LK_PHASE_ADD(3, ph3, ph4);
    LK_PHASE_ADD(4, ph4, ph5);
    LK_PHASE_ADD(5, ph5, ph6);
    LK_PHASE_ADD(6, (unsigned long long)beg, (unsigned long long)end);
    LK_PHASE_ADD(7, ph0, ph6);
#endif
}

template <int NT, bool IS64, bool EXPL, bool CTL, bool YREF = false, bool SEQY = false>
__global__ __launch_bounds__(256) LK_ALS_SOLVE_ATTR void als_solve_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_rows,
    const int32_t *__restrict__ row_slab, const float *__restrict__ other, int ld_other,
    float *__restrict__ this_, int ld_this, const float *__restrict__ otor_p,
    const float *__restrict__ slabs, float *__restrict__ row_delta, int *__restrict__ status,
    int k, float reg, TaskCtlDev ctl, const float *__restrict__ y_ref = nullptr,
    int chunk_rt = 0)
{
    __shared__ __attribute__((aligned(16))) float lds_flat[4 * solve_lds_floats<NT>()];
    als_solve_body<NT, IS64, EXPL, CTL, YREF, SEQY>(indptr, indices, values, order, n_rows,
                                                    row_slab, other, ld_other, this_, ld_this,
                                                    otor_p, slabs, row_delta, status, k, reg, ctl,
                                                    y_ref, chunk_rt, (int64_t)blockIdx.x, lds_flat);
}

// ---- chunk blocks and short-row solve blocks in ONE launch (round 4; an experiment, off by
// default: see als_fused_enabled) -----------------------------------------------------------------
// The chunk kernel is pure matrix-core work (one wave per 1024-entry chunk, no factorisation); the
// solve kernel alternates matrix-core work with a latency-bound v_readlane chain and keeps the
// matrix cores 58-62 % busy.  Launched one after the other they never overlap -- and a second
// stream does not help, the first launch fills every wave slot.  Here the chunk blocks are
// INTERLEAVED with the solve blocks of the rows that need no chunks (block b is a chunk block
// when b % stride == 0, until the chunks run out), so at any time the resident waves are a mix of
// both: the chunk waves' MFMAs fill the issue slots the factorisation chains leave.  No
// dependency between the two kinds of blocks: the rows that DO consume slabs (a prefix of the
// longest-first order: rows [0, n_long)) are solved by a small launch afterwards, behind the
// slab-group reduction -- same arithmetic, same bits as the separate launches.
template <int NT>
__host__ __device__ constexpr int fused_lds_floats()
{
    return 4 * (solve_lds_floats<NT>() > chunk_lds_floats<NT>() ? solve_lds_floats<NT>()
                                                                : chunk_lds_floats<NT>());
}

template <int NT, bool IS64, bool EXPL>
__global__ __launch_bounds__(256) LK_ALS_SOLVE_ATTR void als_fused_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_short,
    const int32_t *__restrict__ row_slab, const float *__restrict__ other, int ld_other,
    float *__restrict__ this_, int ld_this, const float *__restrict__ otor_p,
    float *__restrict__ slabs, float *__restrict__ row_delta, int *__restrict__ status, int k,
    float reg, const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_len,
    int64_t n_chunks, int n_cb, int stride)
{
    __shared__ __attribute__((aligned(16))) float lds_flat[fused_lds_floats<NT>()];
    const int b = blockIdx.x;
    const int q = b / stride;
    const bool slot0 = (b - q * stride) == 0;  // wave-uniform: the whole block takes one role
    if (slot0 && q < n_cb) {
        als_chunk_body<NT, EXPL>(indices, values, chunk_beg, chunk_len, n_chunks, other, ld_other,
                                 slabs, (int64_t)q, lds_flat);
    } else {
        const int before = slot0 ? n_cb : (q + 1 < n_cb ? q + 1 : n_cb);  // chunk blocks below b
        als_solve_body<NT, IS64, EXPL, false, false>(
            indptr, indices, values, order, n_short, row_slab, other, ld_other, this_, ld_this,
            otor_p, slabs, row_delta, status, k, reg, TaskCtlDev{}, nullptr, 0,
            (int64_t)(b - before), lds_flat);
    }
}

// ---- Woodbury row solve for rows with 17..64 entries at padded k = 128 / 256 -----------------
//
// Same identity as csrc/als_wb.hip (see there): x = sum_j (w_j - sqrt(v_j) u_j) z_j with
// S u = sqrt(v) o (S0 w), S = I + diag(sqrt v) S0 diag(sqrt v), S0 = [q_i . z_j] -- but S is up
// to 64 x 64 now: exactly the system the k = 64 hybrid solver above factors in accumulator
// tiles.  One wave per row; entry e = 16 t + c, lane (s, c) = (feature quarter s, slot c) holds
// the entries t = 0..3 of its slot.  Pass 1 streams the quarter's features four at a time and
// accumulates ALL 16 tiles of S0 (the lower ones only feed the row sums S0 w); `hybrid_solve<4>`
// solves; pass 2 re-reads z (L1/L2 hits) for x = sum_e g_e z_e with a 16-lane DPP butterfly.
template <int CTRL>
__device__ __forceinline__ float wb_dpp_add(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float wb_row16_sum(float x)
{
    x = wb_dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
    x = wb_dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
    x = wb_dpp_add<0x141>(x);  // row_half_mirror
    x = wb_dpp_add<0x140>(x);  // row_mirror
    return x;
}

template <int KP, bool IS64>
__global__ __launch_bounds__(256) void als_wb64_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_tasks,
    const float *__restrict__ other, const float *__restrict__ z, float *__restrict__ this_,
    float *__restrict__ row_delta, int *__restrict__ status)
{
    constexpr int QF = KP / 4;  // features per quarter
    constexpr int NQ = QF / 4;  // chunks of 4 features
    __shared__ __attribute__((aligned(16))) float lds_all[4][hybrid_lds_floats<4>()];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int s = lane >> 4, c = lane & 15;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    if (status[1] != 0) return;  // Z unavailable (OtOr not positive definite): dense fallback
    const int row = order[task];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const int n = (int)(end - beg);  // 17 .. 64 (any 1 .. 64 is handled)
    float *xrow = this_ + (int64_t)row * KP;
    float *lds = lds_all[wave];
    const int nte = __builtin_amdgcn_readfirstlane((n + 15) >> 4);  // entry tiles in use

    // this lane's entries: slot c of every tile (slots past the row end: zero weights)
    const float *mrow[4], *zrow[4];
    float w[4], sv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int el = 16 * t + c;
        const int64_t e = beg + (el < n ? el : n - 1);
        const int col = indices[e];
        const float v = el < n ? values[e] : 0.f;
        w[t] = el < n ? v + 1.0f : 0.f;
        sv[t] = __builtin_sqrtf(v);
        // Feature interleave (round 4): step q covers features 16 q .. 16 q + 15 and lane (s, c)
        // takes 4 s .. 4 s + 3 of them, so the four lanes of an entry slot read 64 CONTIGUOUS
        // bytes of the gathered row per step.  (Rounds 2-3 gave lane (s, c) the quarter
        // [64 s, 64 s + 64): 16 bytes per 128-byte line and step, each line revisited on 8
        // steps -- with 230 MB of rows in flight they did not survive in L2: FETCH_SIZE 164 GB per
        // cfg5 item half for 32 GB of rows, the kernel ran at 6.6 TB/s of HBM traffic.)  MFMA
        // step el then contracts features {16 q + el, + 4, + 8, + 12} on BOTH operands.
        mrow[t] = other + (int64_t)col * KP + 4 * s;
        zrow[t] = z + (int64_t)col * KP + 4 * s;
    }
    // pass 1: S0 tiles, acc[ti][tj] lane (s', c') register r = S0[16 ti + 4 s' + r][16 tj + c']
    f32x4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < NQ; ++q) {
        f32x4 mq[4], zq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            mq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            zq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < nte) {
                mq[t] = *reinterpret_cast<const f32x4 *>(mrow[t] + 16 * q);
                zq[t] = *reinterpret_cast<const f32x4 *>(zrow[t] + 16 * q);
            }
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                if (ti < nte && tj < nte) {
#pragma unroll
                    for (int el = 0; el < 4; ++el)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            mq[ti][el], zq[tj][el], acc[ti][tj], 0, 0, 0);
                }
    }
    // S = I + diag(sv) S0 diag(sv) (upper tiles) and the right-hand side sv o (S0 w); rows with
    // at most 32 entries (round 4: most of the 17 .. 64 range) are a 32 x 32 system -- the k = 32
    // instance of the hybrid solver, a quarter of the factorisation work
    float b;
    float minpiv;
    auto build = [&](auto &G, auto ntc) {
        constexpr int NTS = decltype(ntc)::value;
#pragma unroll
        for (int ti = 0; ti < NTS; ++ti) {
            float svr[4], rhs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                svr[r] = __shfl(sv[ti], 4 * s + r, 64);  // sqrt(v) of row 16 ti + 4 s + r
                float r0 = 0.f;
#pragma unroll
                for (int tj = 0; tj < NTS; ++tj) r0 += wb_row16_sum(acc[ti][tj][r] * w[tj]);
                rhs[r] = svr[r] * r0;
            }
#pragma unroll
            for (int tj = ti; tj < NTS; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    G.t[tidx(ti, tj)][r] = svr[r] * sv[tj] * acc[ti][tj][r] +
                                           ((ti == tj && (4 * s + r) == c) ? 1.0f : 0.f);
            // rhs of row 16 ti + c for every lane: it sits in row group c >> 2, register c & 3
            float sel = rhs[0];
            sel = (c & 3) == 1 ? rhs[1] : sel;
            sel = (c & 3) == 2 ? rhs[2] : sel;
            sel = (c & 3) == 3 ? rhs[3] : sel;
            G.y[ti] = __shfl(sel, (c >> 2) * 16 + c, 64);
        }
    };
#ifdef LK_ALS_PHASES
    unsigned long long tmid_unused = 0;
#endif
    if (nte <= 2) {  // wave-uniform
        Gram<2> G;
        build(G, std::integral_constant<int, 2>{});
#ifdef LK_ALS_PHASES
        minpiv = hybrid_solve<2>(G, b, lds, &tmid_unused);
#else
        minpiv = hybrid_solve<2>(G, b, lds);
#endif
    } else {
        Gram<4> G;
        build(G, std::integral_constant<int, 4>{});
#ifdef LK_ALS_PHASES
        minpiv = hybrid_solve<4>(G, b, lds, &tmid_unused);
#else
        minpiv = hybrid_solve<4>(G, b, lds);
#endif
    }
    // b = u' of entry `lane`;  g_e = w_e - sv_e u'_e for this lane's four entries
    float g[4];
    bool bad = !(minpiv > 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        // (tiles past the row's entries -- and, for a 32 x 32 system, lanes the solver did not
        // define -- carry no weight)
        g[t] = t < nte ? w[t] - sv[t] * __shfl(b, 16 * t + c, 64) : 0.f;
        bad = bad || !(fabsf(g[t]) <= 3.0e38f);
    }
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);
    // pass 2: x = sum_e g_e z_e
    float d2 = 0.f;
    for (int q = 0; q < NQ; ++q) {
        f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < nte) {
                const f32x4 zq = *reinterpret_cast<const f32x4 *>(zrow[t] + 16 * q);
                a4.x = fmaf(g[t], zq.x, a4.x);
                a4.y = fmaf(g[t], zq.y, a4.y);
                a4.z = fmaf(g[t], zq.z, a4.z);
                a4.w = fmaf(g[t], zq.w, a4.w);
            }
        f32x4 xs;
        xs.x = wb_row16_sum(a4.x);
        xs.y = wb_row16_sum(a4.y);
        xs.z = wb_row16_sum(a4.z);
        xs.w = wb_row16_sum(a4.w);
        if (c == 0) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(xrow + 4 * s + 16 * q);
            const f32x4 old = *dst;
            *dst = xs;
            const f32x4 d = xs - old;
            d2 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
    d2 = wave_sum(d2);
    if (lane == 0) row_delta[row] = d2;
}

// rows [t0, t1) of the plan order (17 .. 64 entries each)
int als_wb64_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                    const float *values, int64_t t0, int64_t t1, float *this_,
                    const float *other, const float *z, float *row_delta, int *status,
                    hipStream_t st)
{
    const int64_t n = t1 - t0;
    if (n <= 0) return LK_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define LK_WB64(KPV, IS)                                                                        \
    hipLaunchKernelGGL((als_wb64_kernel<KPV, IS>), grid, block, 0, st,                          \
                       static_cast<const typename IndPtr<IS>::type *>(indptr), indices, values, \
                       p->d_order + t0, n, other, z, this_, row_delta, status)
    if (p->KP == 256) {
        if (is64)
            LK_WB64(256, true);
        else
            LK_WB64(256, false);
    } else if (p->KP == 128) {
        if (is64)
            LK_WB64(128, true);
        else
            LK_WB64(128, false);
    } else {
        set_error("Woodbury row solve: unsupported padded embedding size %d", p->KP);
        return LK_E_INVALID;
    }
#undef LK_WB64
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// OtOr [k x k] -> primed [KP x KP] with identity on the pad features.
template <int NT>
__global__ void als_prep_otor_kernel(const float *__restrict__ otor, int ld_otor, int k,
                                     float *__restrict__ otor_p)
{
    constexpr int KP = NT * 16;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KP * KP) return;
    int pr = idx / KP, pc = idx % KP;
    int fr = (pr & 15) * NT + (pr >> 4), fc = (pc & 15) * NT + (pc >> 4);
    float v;
    if (fr < k && fc < k)
        v = otor ? otor[fr * ld_otor + fc] : 0.f;  // explicit mode: no OtOr term
    else
        v = (pr == pc) ? 1.0f : 0.0f;
    otor_p[idx] = v;
}

// deterministic two-stage sum of row deltas -> sqrt
__global__ void delta_partial_kernel(const float *__restrict__ row_delta, int64_t n,
                                     float *__restrict__ partial)
{
    __shared__ float sm[256];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    int64_t b = (int64_t)blockIdx.x * per, e = b + per;
    if (e > n) e = n;
    float s = 0.f;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) s += row_delta[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ void delta_final_kernel(const float *__restrict__ partial, int n,
                                   float *__restrict__ out)
{
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sqrtf(sm[0]);
}

constexpr int DELTA_BLOCKS = LK_DELTA_BLOCKS;

int launch_delta_reduce(const float *row_delta, int64_t n_rows, float *partial, float *out_frob,
                        hipStream_t st)
{
    hipLaunchKernelGGL(delta_partial_kernel, dim3(DELTA_BLOCKS), dim3(256), 0, st, row_delta,
                       n_rows, partial);
    hipLaunchKernelGGL(delta_final_kernel, dim3(1), dim3(256), 0, st, partial, DELTA_BLOCKS,
                       out_frob);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// LK_ALS_FUSED=1: chunk blocks and short-row solve blocks in one launch (als_fused_kernel).
// OFF by default -- measured (tools/fused_ab.py, ML-25M shape, same bits): k = 64 3.35 vs 3.12
// ms/epoch, k = 32 1.28 vs 1.19: the chunk waves' MFMAs do not fill "idle" matrix-core time of the
// factorisation chains, they compete with them for the SIMD's issue slots (user half 1.89 vs
// 1.70 ms, item half 1.38 vs 1.37).  Kept as a knob so that the experiment can be repeated.
static bool als_fused_enabled()
{
    const char *e = getenv("LK_ALS_FUSED");
    return e && e[0] == '1';
}

static bool seqy_enabled()
{
    const char *e = getenv("LK_ALS_SEQY");
    return !(e && e[0] == '0');
}

static bool reduce_on_side()
{
    const char *e = getenv("LK_ALS_REDUCE_SIDE");
    return !(e && e[0] == '0');
}

template <int NT, bool IS64, bool EXPL = false>
static int launch_chol(const lk_als_plan *p, const void *indptr, const int32_t *indices,
                       const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                       const float *other, int ld_other, const float *otor, int ld_otor,
                       char *ws, float *out_frob, hipStream_t st, float reg = 0.f)
{
    constexpr int KP = NT * 16;
    int *status = reinterpret_cast<int *>(ws + p->off_status);
    float *otor_p = reinterpret_cast<float *>(ws + p->off_otor);
    float *row_delta = reinterpret_cast<float *>(ws + p->off_delta);
    float *partial = reinterpret_cast<float *>(ws + p->off_partial);
    float *slabs = reinterpret_cast<float *>(ws + p->off_slabs);

    LK_REQUIRE(!p->ref_order || (p->d_yref && !p->ctl),
               "a reference-order ALS plan needs its rhs workspace (lk_als_plan_set_rhs_workspace) "
               "and no task-control block");
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 64, st));
    if (p->ctl) {
        // a cancelled half-epoch leaves the rows not yet started untouched; their deltas
        // must not be garbage in the (discarded) sum
        LK_HIP_CHECK(hipMemsetAsync(row_delta, 0, (size_t)n_rows * sizeof(float), st));
        int rc = ctl_begin(p->ctl, n_rows, n_rows, st);
        if (rc != LK_OK) return rc;
    }
    hipLaunchKernelGGL(als_prep_otor_kernel<NT>, dim3((KP * KP + 255) / 256), dim3(256), 0, st,
                       otor, ld_otor, k, otor_p);
    const bool tm = p->timing && p->timing_n < lk_als_plan::TIMING_RING;
    if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][0], st));
    // (CG hybrid: only the chunked rows, the first dense_limit tasks of the order)
    const int64_t n_solve = p->dense_limit >= 0 && p->dense_limit < n_rows ? p->dense_limit : n_rows;
    // rows whose right-hand side comes from the reference-order chain (als_rhs.hip): the long
    // rows (the first n_long tasks) of a hybrid plan, every row of a strict reference-order plan
    float *yref = p->hybrid ? reinterpret_cast<float *>(ws + p->off_yref)
                            : (p->ctl ? nullptr : p->d_yref);
    const int64_t n_y = !yref ? 0 : (p->hybrid ? std::min<int64_t>(p->n_long, n_solve) : n_solve);
    const int ref_chunk = (p->ref_order || p->hybrid) ? p->chunk : 0;
    // one launch for the chunks AND the rows that need none (als_fused_kernel): the plain exact
    // half-epoch only -- no task control, no reference order, not the CG hybrid's prefix
    const bool fused = als_fused_enabled() && p->n_chunks > 0 && !p->ctl && !yref &&
                       !ref_chunk && p->dense_limit < 0 && p->n_long < n_rows;
    using IT = typename IndPtr<IS64>::type;
    if (fused) {
        const int64_t n_short = n_rows - p->n_long;  // tasks [n_long, n_rows) of the order
        const int64_t n_cb = (p->n_chunks + 3) / 4, n_sb = (n_short + 3) / 4;
        int64_t stride = (n_cb + n_sb) / n_cb;
        if (stride < 1) stride = 1;
        hipLaunchKernelGGL((als_fused_kernel<NT, IS64, EXPL>), dim3((unsigned)(n_cb + n_sb)),
                           dim3(256), 0, st, static_cast<const IT *>(indptr), indices, values,
                           p->d_order + p->n_long, n_short, p->d_row_slab, other, ld_other, this_,
                           ld_this, otor_p, slabs, row_delta, status, k, reg, p->d_chunk_beg,
                           p->d_chunk_len, p->n_chunks, (int)n_cb, (int)stride);
        int rc = launch_slab_group_reduce(p, slabs, slab_floats<NT>(), st);
        if (rc != LK_OK) return rc;
        if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][1], st));
        if (p->n_long > 0)  // the rows that consume the slabs
            hipLaunchKernelGGL((als_solve_kernel<NT, IS64, EXPL, false>),
                               dim3((unsigned)((p->n_long + 3) / 4)), dim3(256), 0, st,
                               static_cast<const IT *>(indptr), indices, values, p->d_order,
                               p->n_long, p->d_row_slab, other, ld_other, this_, ld_this, otor_p,
                               slabs, row_delta, status, k, reg, TaskCtlDev{});
    } else {
        // The chains run on the plan's second stream, beside the chunk kernel and the solve of
        // the other rows; only the (small) solve launch of the long rows waits for them.  They are
        // ENQUEUED BEHIND the chunk kernel (the fork point is in front of it): the chain kernel is
        // LDS-heavy and latency-bound; enqueued first it takes every CU's LDS for its first round
        // of workgroups and the MFMA-bound chunk kernel waits (measured: +0.25 ms per cfg2 item
        // half); enqueued second it trickles in as chunk workgroups retire and does most of its
        // work under the solve of the short rows.  LK_ALS_CHAIN_FIRST=1: the old order.
        hipStream_t sr = st;
        const char *cf = getenv("LK_ALS_CHAIN_FIRST");
        const bool chain_first = cf && cf[0] == '1';
        auto launch_chains = [&]() -> int {
            return launch_rhs_reference(p, indptr, IS64 ? 1 : 0, indices, values, p->d_order, n_y,
                                        other, EXPL, yref, sr);
        };
        if (n_y > 0) {
            int rc = plan_fork_rhs(p, st, &sr);
            if (rc != LK_OK) return rc;
            if (chain_first && (rc = launch_chains()) != LK_OK) return rc;
        }
        if (p->n_chunks > 0) {
            hipLaunchKernelGGL((als_chunk_kernel<NT, EXPL>),
                               dim3((unsigned)((p->n_chunks + 3) / 4)), dim3(256), 0, st, indices,
                               values, p->d_chunk_beg, p->d_chunk_len, p->n_chunks, other,
                               ld_other, slabs, p->d_chunk_slab,
                               p->unit > p->chunk ? (int)p->chunk : 0);
        }
        if (n_y > 0 && !chain_first) {
            int rc = launch_chains();
            if (rc != LK_OK) return rc;
        }
        if (p->n_chunks > 0) {
            // the ordered slab sums of reference-order rows are pure HBM streaming: on the second
            // stream (behind the chains) they run under the solve of the rows that need no slabs
            // (LK_ALS_REDUCE_SIDE=0: launch stream)
            hipStream_t sg = st;
            if (n_y > 0 && sr != st && reduce_on_side()) {
                int rc = plan_rhs_wait_main(p, st);
                if (rc != LK_OK) return rc;
                sg = sr;
            }
            int rc = launch_slab_group_reduce(p, slabs, slab_floats<NT>(), sg);
            if (rc != LK_OK) return rc;
        }
        if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][1], st));
        const IT *ip = static_cast<const IT *>(indptr);
#define LK_SOLVE_LAUNCH(CTLV, YREFV, T0, NTASKS, YPTR)                                            \
    LK_SOLVE_LAUNCH_S(CTLV, YREFV, false, T0, NTASKS, YPTR)
#define LK_SOLVE_LAUNCH_S(CTLV, YREFV, SEQV, T0, NTASKS, YPTR)                                    \
    hipLaunchKernelGGL((als_solve_kernel<NT, IS64, EXPL, CTLV, YREFV, SEQV>),                      \
                       dim3((unsigned)(((NTASKS) + 3) / 4)), dim3(256), 0, st, ip, indices,        \
                       values, p->d_order + (T0), (NTASKS), p->d_row_slab, other, ld_other, this_, \
                       ld_this, otor_p, slabs, row_delta, status, k, reg,                          \
                       (CTLV) ? p->ctl->dev() : TaskCtlDev{}, (YPTR), ref_chunk)
        // the rows that take their own right-hand side first (longest-first inside the launch) ...
        // (hybrid / reference-order plans: these rows form y in the reference's order inside
        // their Gram loop -- SEQY; LK_ALS_SEQY=0: the four-slot sums of the accurate mode)
        const bool seqy = (p->hybrid || p->ref_order) && seqy_enabled();
        if (n_solve > n_y) {
            if (p->ctl) {
                if (seqy)
                    LK_SOLVE_LAUNCH_S(true, false, true, n_y, n_solve - n_y, nullptr);
                else
                    LK_SOLVE_LAUNCH(true, false, n_y, n_solve - n_y, nullptr);
            } else {
                if (seqy)
                    LK_SOLVE_LAUNCH_S(false, false, true, n_y, n_solve - n_y, nullptr);
                else
                    LK_SOLVE_LAUNCH(false, false, n_y, n_solve - n_y, nullptr);
            }
        }
        // ... then the rows of the chains (hybrid plans: the long rows -- one slab + one solve each)
        if (n_y > 0) {
            int rc = plan_join_rhs(p, st);
            if (rc != LK_OK) return rc;
            if (p->ctl)
                LK_SOLVE_LAUNCH(true, true, 0, n_y, yref);
            else
                LK_SOLVE_LAUNCH(false, true, 0, n_y, yref);
        }
#undef LK_SOLVE_LAUNCH
#undef LK_SOLVE_LAUNCH_S
    }
    if (tm) {
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][2], st));
        p->timing_n++;
    }
    hipLaunchKernelGGL(delta_partial_kernel, dim3(DELTA_BLOCKS), dim3(256), 0, st, row_delta,
                       n_rows, partial);
    hipLaunchKernelGGL(delta_final_kernel, dim3(1), dim3(256), 0, st, partial, DELTA_BLOCKS,
                       out_frob);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

int als_cg_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                      const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                      const float *other, int ld_other, const float *otor, int ld_otor, char *ws,
                      float *out_frob, hipStream_t st);

int als_chol_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                        const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                        const float *other, int ld_other, const float *otor, int ld_otor, char *ws,
                        float *out_frob, hipStream_t st)
{
#define LK_CHOL_CASE(NT)                                                                         \
    return is64 ? launch_chol<NT, true>(p, indptr, indices, values, n_rows, k, this_, ld_this,  \
                                        other, ld_other, otor, ld_otor, ws, out_frob, st)       \
                : launch_chol<NT, false>(p, indptr, indices, values, n_rows, k, this_, ld_this, \
                                         other, ld_other, otor, ld_otor, ws, out_frob, st)
    switch (p->NT) {
        case 1: LK_CHOL_CASE(1);
        case 2: LK_CHOL_CASE(2);
        case 4: LK_CHOL_CASE(4);
    }
#undef LK_CHOL_CASE
    set_error("lk_als_implicit_half_epoch: no Cholesky kernel for padded k=%d", p->KP);
    return LK_E_INVALID;
}

}  // namespace lk

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

#ifdef LK_ALS_PHASES
extern "C" int lk_als_phase_set(void *d_buf)
{
    LK_HIP_CHECK(hipDeviceSynchronize());
    LK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(lk::lk_als_phase_buf), &d_buf, sizeof(void *)));
    return LK_OK;
}
#endif

extern "C" int32_t lk_padded_dim(int32_t k)
{
    if (k < 1) return 0;
    if (k <= 16) return 16;
    if (k <= 32) return 32;
    if (k <= 64) return 64;
    if (k <= 128) return 128;
    if (k <= 256) return 256;
    // above 256: multiples of 64 up to 1024, served by the HBM-tile solver of als_big.hip
    if (k <= 1024) return (k + 63) / 64 * 64;
    return 0;
}

// ---- pool of schedule buffers (per device; plans of a few thousand rows come and go per call) ----
namespace {
struct PackPool {
    static constexpr int SLOTS = 8;
    static constexpr size_t MAX_BYTES = (size_t)8 << 20;  // larger buffers are not pooled
    std::mutex mu;
    struct Slot {
        char *ptr = nullptr;
        size_t cap = 0;
        int dev = -1;
    } slot[SLOTS];
};
PackPool &pack_pool()
{
    static PackPool pool;
    return pool;
}
char *pack_pool_take(size_t bytes, size_t *cap)
{
    if (bytes > PackPool::MAX_BYTES) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    PackPool &pl = pack_pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    int best = -1;
    for (int i = 0; i < PackPool::SLOTS; ++i)
        if (pl.slot[i].ptr && pl.slot[i].dev == dev && pl.slot[i].cap >= bytes &&
            (best < 0 || pl.slot[i].cap < pl.slot[best].cap))
            best = i;
    if (best < 0) return nullptr;
    char *ptr = pl.slot[best].ptr;
    *cap = pl.slot[best].cap;
    pl.slot[best].ptr = nullptr;
    return ptr;
}
bool pack_pool_give(char *ptr, size_t cap)
{
    if (cap > PackPool::MAX_BYTES) return false;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    PackPool &pl = pack_pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    for (int i = 0; i < PackPool::SLOTS; ++i)
        if (!pl.slot[i].ptr) {
            pl.slot[i].ptr = ptr;
            pl.slot[i].cap = cap;
            pl.slot[i].dev = dev;
            return true;
        }
    return false;  // pool full: the caller frees
}
}  // namespace

template <typename T>
static int upload(T **dst, const std::vector<T> &src)
{
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (!src.empty())
        LK_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LK_OK;
}

extern "C" int lk_als_plan_create_ex(lk_als_plan **out, const void *h_indptr, int indptr_is_64,
                                     int64_t n_rows, int32_t k, int32_t solver, int32_t flags);

extern "C" int lk_als_plan_create(lk_als_plan **out, const void *h_indptr, int indptr_is_64,
                                  int64_t n_rows, int32_t k, int32_t solver)
{
    // the default is the hybrid order (include/lkamd.h); LK_ALS_RHS_ORDER=accurate: the tuned
    // kernels' own summation on every row (round 4's default)
    // (case-insensitive, unknown values refused -- the rules of lkpy_amd._device.als_order_mode.
    // `reference` = the STRICT mode needs the caller's right-hand-side workspace
    // (lk_als_plan_set_rhs_workspace): a plan made here gets the flag, the caller attaches the
    // buffer; until then its rows > 256 entries sum the normal matrix in the reference's blocks
    // and the right-hand side in the kernels' own order)
    const char *e = getenv("LK_ALS_RHS_ORDER");
    int32_t flags = LK_ALS_PLAN_HYBRID_ORDER;
    if (e && e[0]) {
        if (!strcasecmp(e, "accurate")) flags = 0;
        else if (!strcasecmp(e, "reference")) flags = LK_ALS_PLAN_REFERENCE_ORDER;
        else
            LK_REQUIRE(!strcasecmp(e, "auto") || !strcasecmp(e, "hybrid") || !strcasecmp(e, "default"),
                       "lk_als_plan_create: unknown LK_ALS_RHS_ORDER '%s' (auto / reference / accurate)",
                       e);
    }
    return lk_als_plan_create_ex(out, h_indptr, indptr_is_64, n_rows, k, solver, flags);
}

extern "C" int lk_als_plan_create_ex(lk_als_plan **out, const void *h_indptr, int indptr_is_64,
                                     int64_t n_rows, int32_t k, int32_t solver, int32_t flags)
{
    LK_REQUIRE(out != nullptr && h_indptr != nullptr, "lk_als_plan_create: null pointer");
    LK_REQUIRE((flags & ~(LK_ALS_PLAN_REFERENCE_ORDER | LK_ALS_PLAN_HYBRID_ORDER)) == 0,
               "lk_als_plan_create_ex: unknown flags");
    LK_REQUIRE((flags & (LK_ALS_PLAN_REFERENCE_ORDER | LK_ALS_PLAN_HYBRID_ORDER)) !=
                   (LK_ALS_PLAN_REFERENCE_ORDER | LK_ALS_PLAN_HYBRID_ORDER),
               "lk_als_plan_create_ex: reference order is either strict or hybrid");
    LK_REQUIRE(n_rows >= 0 && n_rows < (int64_t)INT32_MAX, "lk_als_plan_create: bad n_rows");
    int KP = lk_padded_dim(k);
    LK_REQUIRE(KP > 0, "lk_als_plan_create: unsupported embedding size k=%d (1..1024)", k);
    // the reference solves every row exactly (sposv): so does AUTO, at every k
    if (solver == LK_SOLVER_AUTO) solver = LK_SOLVER_CHOLESKY;
    LK_REQUIRE(solver == LK_SOLVER_CHOLESKY || solver == LK_SOLVER_CG,
               "lk_als_plan_create: unknown solver %d", solver);
    LK_REQUIRE(!(solver == LK_SOLVER_CG && (KP < 64 || KP > 256)),
               "lk_als_plan_create: the CG solver serves 32 < k <= 256 (got %d)", k);
    LK_REQUIRE(!((flags & LK_ALS_PLAN_REFERENCE_ORDER) && KP > 256),
               "lk_als_plan_create_ex: reference-order plans stop at k = 256 (got %d)", k);

    auto *p = new lk_als_plan();
    p->n_rows = n_rows;
    p->k = k;
    p->KP = KP;
    p->NT = KP / 16;
    p->solver = solver;
    p->is64 = indptr_is_64 ? 1 : 0;
    p->cg_max_iter = 0;
    if (flags & LK_ALS_PLAN_REFERENCE_ORDER) {
        if (solver != LK_SOLVER_CHOLESKY) {
            delete p;
            lk::set_error("lk_als_plan_create_ex: reference order belongs to the exact solver");
            return LK_E_INVALID;
        }
        p->ref_order = true;
        p->chunk = 256;     // matrixmultiply's KC (oracle/lk_oracle.c: LKO_SGEMM_KC)
        p->long_row = 256;  // every row the reference sums in more than one block
    }
    if ((flags & LK_ALS_PLAN_HYBRID_ORDER) && solver == LK_SOLVER_CHOLESKY && KP <= 256) {
        // (the CG option and k > 256 have no slab path: the flag does not apply to them)
        p->hybrid = true;
        p->chunk = 256;
        const char *e = getenv("LK_ALS_REF_LEN");
        int rl = e ? atoi(e) : LK_ALS_LONG_ROW;
        if (rl < 256) rl = 256;                          // (one block: nothing to reorder)
        if (rl > LK_ALS_LONG_ROW) rl = LK_ALS_LONG_ROW;  // longer rows must be chunked anyway
        p->long_row = rl;
    }
    // (k > 256: no chunk slabs -- als_big.hip spreads a long row over its Gram grid)
    const int64_t CHUNK = p->chunk, LONG_ROW = KP > 256 ? INT64_MAX : (int64_t)p->long_row;

    auto len = [&](int64_t r) -> int64_t {
        if (indptr_is_64) {
            const int64_t *ip = static_cast<const int64_t *>(h_indptr);
            return ip[r + 1] - ip[r];
        }
        const int32_t *ip = static_cast<const int32_t *>(h_indptr);
        return (int64_t)ip[r + 1] - ip[r];
    };
    auto start = [&](int64_t r) -> int64_t {
        return indptr_is_64 ? static_cast<const int64_t *>(h_indptr)[r]
                            : (int64_t) static_cast<const int32_t *>(h_indptr)[r];
    };

    // rows by descending length, ties in row order.  Lengths are small integers: a counting sort
    // (histogram of the lengths, offsets from the longest down, rows placed in row order) does in
    // O(rows + longest) what the stable comparison sort did in O(rows log rows) -- a fold-in plan
    // is built per call (10 000 rows: 0.3 of the call's 3.3 ms went into the sort), cfg5's user
    // plan orders 10^7 rows.  Only a matrix whose longest row dwarfs its row count sorts keys.
    std::vector<int32_t> order((size_t)n_rows);
    {
        int64_t longest = 0;
        bool sane = true;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t n = len(r);
            if (n < 0) sane = false;
            if (n > longest) longest = n;
        }
        if (sane && longest <= 4 * n_rows + 65536) {
            std::vector<int64_t> at((size_t)longest + 2, 0);
            for (int64_t r = 0; r < n_rows; ++r) ++at[(size_t)len(r)];
            int64_t run = 0;  // at[n] = first position of the rows of length n (longest first)
            for (int64_t n = longest; n >= 0; --n) {
                const int64_t c = at[(size_t)n];
                at[(size_t)n] = run;
                run += c;
            }
            for (int64_t r = 0; r < n_rows; ++r) order[(size_t)(at[(size_t)len(r)]++)] = (int32_t)r;
        } else {
            for (int64_t r = 0; r < n_rows; ++r) order[(size_t)r] = (int32_t)r;
            std::stable_sort(order.begin(), order.end(),
                             [&](int32_t x, int32_t y) { return len(x) > len(y); });
        }
    }

    {
        // first task whose row has <= 16 entries (the order is longest first)
        int64_t lo = 0, hi = n_rows;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 16)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_short = lo;
        hi = n_rows;
        while (lo < hi) {  // first task whose row has <= 8 entries
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 8)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_8 = lo;
        hi = n_rows;
        while (lo < hi) {  // first task whose row has <= 4 entries
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 4)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_4 = lo;
        lo = 0;
        hi = p->t_short;
        while (lo < hi) {  // first task whose row has <= 64 entries
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 64)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_mid = lo;
        hi = p->t_short;
        while (lo < hi) {  // first task whose row has <= 32 entries
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 32)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_32 = lo;
        lo = 0;
        hi = p->t_mid;
        while (lo < hi) {  // first task whose row has <= 128 entries
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > 128)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_128 = lo;
        lo = 0;
        hi = n_rows;
        const int64_t cg_len = 16384 / KP;
        while (lo < hi) {  // first task whose row the CG kernel holds in registers
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > cg_len)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_cg = lo;
        hi = n_rows;
        while (lo < hi) {  // ... and that ONE wave holds
            const int64_t mid = (lo + hi) >> 1;
            if (len(order[(size_t)mid]) > cg_len / 4)
                lo = mid + 1;
            else
                hi = mid;
        }
        p->t_cg1 = lo;
    }
    std::vector<int32_t> row_slab((size_t)n_rows, -1);
    std::vector<int32_t> chunk_row, chunk_slab;
    std::vector<int64_t> chunk_beg;
    std::vector<int32_t> chunk_len;
    // work units (als_plan.h): hybrid plans at padded k = 64 keep 1024-entry units, one slab per
    // 256-entry block (LK_ALS_REF_UNIT: entries per unit, a multiple of 256; 256 = a unit per block)
    p->unit = p->chunk;
    // (padded k = 256: the LDS-staged chunk kernel of als_blk.hip takes units as well; with
    // LK_BLK_CHUNK_DMA=0 -- the register-ring kernel, which has no block boundaries -- a unit is
    // a chunk)
    const char *dma_off = getenv("LK_BLK_CHUNK_DMA");
    const bool units256 = KP == 256 && !(dma_off && dma_off[0] == '0');
    // (padded k = 64: only the LDS-DMA Gram accumulation flushes a slab per 256-entry block; a
    // -DLK_ALS_GRAM_DMA=0 build stores one slab per unit, so there a unit must be a chunk)
    if (p->hybrid && ((KP == 64 && LK_ALS_GRAM_DMA) || units256)) {
        const char *e = getenv("LK_ALS_REF_UNIT");
        int u = e ? atoi(e) : LK_ALS_CHUNK;
        if (u < p->chunk) u = p->chunk;
        p->unit = u / p->chunk * p->chunk;
    }
    const int64_t UNIT = p->unit;
    {  // (CG plans too: their chunked rows are solved by the exact kernels, als_cg.hip)
        for (int64_t r = 0; r < n_rows; ++r) {
            int64_t n = len(r);
            if (n > LONG_ROW) {
                row_slab[(size_t)r] = (int32_t)p->n_slabs;
                for (int64_t o = 0; o < n; o += UNIT) {
                    chunk_row.push_back((int32_t)r);
                    chunk_beg.push_back(start(r) + o);
                    chunk_len.push_back((int32_t)std::min<int64_t>(UNIT, n - o));
                    chunk_slab.push_back((int32_t)(p->n_slabs + o / CHUNK));
                }
                p->n_slabs += (n + CHUNK - 1) / CHUNK;
                p->n_long++;
            }
        }
    }
    if (p->n_slabs >= (int64_t)INT32_MAX) {
        delete p;
        lk::set_error("lk_als_plan_create: too many slabs");
        return LK_E_INVALID;
    }
    p->n_chunks = (int64_t)chunk_row.size();
    // slab groups of the rows with many chunks (LK_ALS_SLAB_GROUP, als_plan.h)
    std::vector<int32_t> grp_head, grp_cnt;
    for (int64_t r = 0; r < n_rows; ++r) {
        if (row_slab[(size_t)r] < 0) continue;
        const int64_t ns = (len(r) + CHUNK - 1) / CHUNK;
        if (p->ref_order || p->hybrid) {  // ONE group per row: head += every other slab, in chunk order
            grp_head.push_back(row_slab[(size_t)r]);
            grp_cnt.push_back((int32_t)ns);
            continue;
        }
        if (ns <= LK_ALS_SLAB_GROUP) continue;
        for (int64_t s0 = 0; s0 < ns; s0 += LK_ALS_SLAB_GROUP) {
            const int64_t c = std::min<int64_t>(LK_ALS_SLAB_GROUP, ns - s0);
            if (c >= 2) {
                grp_head.push_back((int32_t)(row_slab[(size_t)r] + s0));
                grp_cnt.push_back((int32_t)c);
            }
        }
    }
    p->n_groups = (int64_t)grp_head.size();

    // the schedule arrays: ONE device allocation and ONE copy (a fold-in plan of a batch of queries
    // is built per call -- eight allocations and blocking copies were 0.4 ms of a 4 ms call)
    {
        auto padded = [](size_t bytes) { return lk::align_up(std::max<size_t>(bytes, 8), 256); };
        const size_t b_order = padded(order.size() * 4), b_rslab = padded(row_slab.size() * 4),
                     b_crow = padded(chunk_row.size() * 4), b_cbeg = padded(chunk_beg.size() * 8),
                     b_cslab = padded(chunk_slab.size() * 4), b_ghead = padded(grp_head.size() * 4),
                     b_gcnt = padded(grp_cnt.size() * 4), b_clen = padded(chunk_len.size() * 4);
        const size_t total = b_order + b_rslab + b_crow + b_cbeg + b_cslab + b_ghead + b_gcnt + b_clen;
        std::vector<char> host(total, 0);
        size_t o = 0;
        auto put = [&](const void *src, size_t bytes, size_t slot) {
            if (bytes) memcpy(host.data() + o, src, bytes);
            const size_t at = o;
            o += slot;
            return at;
        };
        const size_t o_order = put(order.data(), order.size() * 4, b_order);
        const size_t o_rslab = put(row_slab.data(), row_slab.size() * 4, b_rslab);
        const size_t o_cbeg = put(chunk_beg.data(), chunk_beg.size() * 8, b_cbeg);
        const size_t o_crow = put(chunk_row.data(), chunk_row.size() * 4, b_crow);
        const size_t o_cslab = put(chunk_slab.data(), chunk_slab.size() * 4, b_cslab);
        const size_t o_ghead = put(grp_head.data(), grp_head.size() * 4, b_ghead);
        const size_t o_gcnt = put(grp_cnt.data(), grp_cnt.size() * 4, b_gcnt);
        const size_t o_clen = put(chunk_len.data(), chunk_len.size() * 4, b_clen);
        hipError_t e = hipSuccess;
        p->d_pack = pack_pool_take(total, &p->pack_cap);
        if (!p->d_pack) {
            p->pack_cap = total;
            e = hipMalloc(reinterpret_cast<void **>(&p->d_pack), total);
        }
        if (e == hipSuccess) e = hipMemcpy(p->d_pack, host.data(), total, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            lk::set_error("lk_als_plan_create: %s", hipGetErrorString(e));
            lk_als_plan_destroy(p);
            return LK_E_HIP;
        }
        p->d_order = reinterpret_cast<int32_t *>(p->d_pack + o_order);
        p->d_row_slab = reinterpret_cast<int32_t *>(p->d_pack + o_rslab);
        p->d_chunk_beg = reinterpret_cast<int64_t *>(p->d_pack + o_cbeg);
        p->d_chunk_row = reinterpret_cast<int32_t *>(p->d_pack + o_crow);
        p->d_chunk_slab = reinterpret_cast<int32_t *>(p->d_pack + o_cslab);
        p->d_grp_head = reinterpret_cast<int32_t *>(p->d_pack + o_ghead);
        p->d_grp_cnt = reinterpret_cast<int32_t *>(p->d_pack + o_gcnt);
        p->d_chunk_len = reinterpret_cast<int32_t *>(p->d_pack + o_clen);
    }

    size_t off = 0;
    p->off_status = off;
    off += 256;
    p->off_otor = off;
    off += lk::align_up((size_t)KP * KP * sizeof(float), 256);
    p->off_delta = off;
    off += lk::align_up((size_t)std::max<int64_t>(n_rows, 1) * sizeof(float), 256);
    p->off_partial = off;
    off += lk::align_up((size_t)lk::DELTA_BLOCKS * sizeof(float), 256);
    p->off_slabs = off;
    // one-wave slabs (als_chol.hip, k <= 64) or four-wave slabs (als_blk.hip, k = 128 / 256)
    if (KP > 256) {
        // the tile scratch of a batch of rows (als_big.hip) takes the slabs' place
        off += lk::align_up(lk::als_big_scratch_bytes(KP, n_rows), 256);
    } else {
        size_t slab_f = KP > 64 ? lk::als_blk_slab_floats(p->NT)
                                : (size_t)(lk::als_tiles(p->NT) * 4 + p->NT) * 64;
        off += lk::align_up((size_t)std::max<int64_t>(p->n_slabs, 1) * slab_f * sizeof(float),
                            256);
    }
    if (p->hybrid) {  // y of the long rows in the reference's order, one row of KP floats per task
        p->off_yref = off;
        off += lk::align_up((size_t)std::max<int64_t>(p->n_long, 1) * KP * sizeof(float), 256);
    }
    if (KP > 64 && KP <= 256) {  // OtOr^-1 for the Woodbury rows (lk_als_plan_set_z_workspace)
        p->off_ginv = off;
        off += lk::align_up((size_t)KP * KP * sizeof(float), 256);
        p->off_invws = off;
        off += lk::align_up(lk::spd_inverse_workspace_bytes(KP), 256);
    }
    p->ws_bytes = off;
    *out = p;
    return LK_OK;
}

extern "C" int lk_als_plan_enable_timing(lk_als_plan *p, int enable)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_enable_timing: null plan");
    if (enable && !p->ev[0][0]) {
        for (int i = 0; i < lk_als_plan::TIMING_RING; ++i)
            for (int j = 0; j < 3; ++j) LK_HIP_CHECK(hipEventCreate(&p->ev[i][j]));
    }
    p->timing = enable != 0;
    p->timing_n = 0;
    return LK_OK;
}

extern "C" int lk_als_plan_get_timing(lk_als_plan *p, double *ms_chunk, double *ms_solve,
                                      int32_t *n_launches)
{
    LK_REQUIRE(p && ms_chunk && ms_solve && n_launches, "lk_als_plan_get_timing: null pointer");
    *ms_chunk = 0.0;
    *ms_solve = 0.0;
    *n_launches = p->timing_n;
    for (int i = 0; i < p->timing_n; ++i) {
        float a = 0.f, b = 0.f;
        LK_HIP_CHECK(hipEventSynchronize(p->ev[i][2]));
        LK_HIP_CHECK(hipEventElapsedTime(&a, p->ev[i][0], p->ev[i][1]));
        LK_HIP_CHECK(hipEventElapsedTime(&b, p->ev[i][1], p->ev[i][2]));
        *ms_chunk += a;
        *ms_solve += b;
    }
    p->timing_n = 0;
    return LK_OK;
}

extern "C" void lk_als_plan_destroy(lk_als_plan *p)
{
    if (!p) return;
    if (p->ev[0][0])
        for (int i = 0; i < lk_als_plan::TIMING_RING; ++i)
            for (int j = 0; j < 3; ++j) (void)hipEventDestroy(p->ev[i][j]);
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        lk::side_stream_release(p->side);
        (void)hipEventDestroy(p->ev_fork);
        (void)hipEventDestroy(p->ev_join);
    }
    if (p->side_rhs) {
        (void)hipStreamSynchronize(p->side_rhs);
        lk::side_stream_release(p->side_rhs);
        (void)hipEventDestroy(p->ev_fork_rhs);
        (void)hipEventDestroy(p->ev_join_rhs);
        (void)hipEventDestroy(p->ev_mid_rhs);
    }
    // (d_order ... d_chunk_len point into d_pack.)  Small schedule buffers go back to a per-device
    // pool instead of hipFree: a fold-in plan lives for one call, and hipMalloc + hipFree were a
    // quarter of a millisecond of it.  hipFree waits for the device; so does this.
    if (p->d_pack) {
        (void)hipDeviceSynchronize();
        if (!pack_pool_give(p->d_pack, p->pack_cap)) (void)hipFree(p->d_pack);
    }
    delete p;
}

extern "C" size_t lk_als_plan_workspace_bytes(const lk_als_plan *p) { return p ? p->ws_bytes : 0; }
extern "C" int32_t lk_als_plan_solver(const lk_als_plan *p) { return p ? p->solver : -1; }

extern "C" int lk_als_plan_set_ctl(lk_als_plan *p, lk_task_ctl *ctl)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_ctl: null plan");
    p->ctl = ctl;
    return LK_OK;
}

extern "C" int64_t lk_als_plan_short_rows(const lk_als_plan *p)
{
    return p ? p->n_rows - p->t_short : 0;
}

extern "C" int64_t lk_als_plan_long_rows(const lk_als_plan *p) { return p ? p->n_long : 0; }

extern "C" const float *lk_als_plan_yref(const lk_als_plan *p, const void *d_ws)
{
    if (!p || !d_ws || !p->hybrid) return nullptr;
    return reinterpret_cast<const float *>(static_cast<const char *>(d_ws) + p->off_yref);
}

extern "C" int64_t lk_als_plan_woodbury_rows(const lk_als_plan *p)
{
    if (!p) return 0;
    if (p->KP == 256) return p->n_rows - p->t_mid;
    const int lim = wb64_k128_limit();  // padded k = 128
    return p->n_rows - (lim >= 64 ? p->t_mid : (lim >= 32 ? p->t_32 : p->t_short));
}

extern "C" int lk_als_plan_set_z(lk_als_plan *p, const float *d_z)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_z: null plan");
    p->d_z = d_z;
    return LK_OK;
}

extern "C" int lk_als_plan_set_z_shared(lk_als_plan *p, const float *d_z, const void *d_flag)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_z_shared: null plan");
    LK_REQUIRE((d_z == nullptr) == (d_flag == nullptr),
               "lk_als_plan_set_z_shared: Z and its flag word go together");
    LK_REQUIRE(d_z == nullptr || (p->KP > 64 && p->KP <= 256),
               "lk_als_plan_set_z_shared: the Woodbury kernels serve padded k = 128 / 256 only");
    p->d_z = d_z;
    p->d_zflag_src = static_cast<const int *>(d_flag);
    if (d_z) p->d_zbuf = nullptr;
    return LK_OK;
}

extern "C" int lk_als_plan_set_z_leader(lk_als_plan *p, int on)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_z_leader: null plan");
    p->z_for_others = on != 0;
    return LK_OK;
}

extern "C" const void *lk_als_plan_z_flag(const lk_als_plan *p, const void *d_ws)
{
    if (!p || !d_ws) return nullptr;
    return static_cast<const char *>(d_ws) + p->off_status + sizeof(int);
}

extern "C" int lk_als_plan_set_z_workspace(lk_als_plan *p, float *d_zbuf)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_z_workspace: null plan");
    LK_REQUIRE(d_zbuf == nullptr || (p->KP > 64 && p->KP <= 256),
               "lk_als_plan_set_z_workspace: the Woodbury kernels serve padded k = 128 / 256 only");
    p->d_zbuf = d_zbuf;
    if (d_zbuf) {
        p->d_z = nullptr;
        p->d_zflag_src = nullptr;
    }
    return LK_OK;
}

extern "C" int lk_als_plan_set_cg(lk_als_plan *p, float tol, int32_t max_iter)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_cg: null plan");
    LK_REQUIRE(tol > 0.f, "lk_als_plan_set_cg: tol must be positive");
    p->cg_tol = tol;
    p->cg_max_iter = max_iter;
    return LK_OK;
}

extern "C" int lk_als_implicit_half_epoch(const lk_als_plan *plan, const void *d_indptr,
                                          const int32_t *d_indices, const float *d_values,
                                          int64_t n_rows, int64_t n_cols, int32_t k,
                                          float *d_this, int32_t ld_this, const float *d_other,
                                          int32_t ld_other, const float *d_otor, int32_t ld_otor,
                                          void *d_ws, float *d_out_frob, void *stream)
{
    LK_REQUIRE(plan != nullptr, "lk_als_implicit_half_epoch: null plan");
    LK_REQUIRE(n_rows == plan->n_rows && k == plan->k,
               "lk_als_implicit_half_epoch: plan built for %lld rows, k=%d; got %lld rows, k=%d",
               (long long)plan->n_rows, plan->k, (long long)n_rows, k);
    LK_REQUIRE(ld_this == plan->KP && ld_other == plan->KP,
               "lk_als_implicit_half_epoch: factor leading dimensions (%d, %d) must equal "
               "lk_padded_dim(k)=%d",
               ld_this, ld_other, plan->KP);
    LK_REQUIRE(ld_otor >= k, "lk_als_implicit_half_epoch: ld_otor < k");
    LK_REQUIRE(d_indptr && d_this && d_otor && d_ws && d_out_frob,
               "lk_als_implicit_half_epoch: null pointer");
    LK_REQUIRE(n_cols >= 0 && (n_cols == 0 || d_other), "lk_als_implicit_half_epoch: null other");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    if (plan->solver == LK_SOLVER_CG)
        return lk::als_cg_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows, k,
                                     d_this, ld_this, d_other, ld_other, d_otor, ld_otor, ws,
                                     d_out_frob, st);
    if (plan->KP > 256)
        return lk::als_big_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows, k,
                                      d_this, d_other, d_otor, ld_otor, ws, d_out_frob, st, false,
                                      0.f);
    if (plan->KP > 64)
        return lk::als_blk_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows,
                                      n_cols, k, d_this, d_other, d_otor, ld_otor, ws, d_out_frob,
                                      st, false, 0.f);
    return lk::als_chol_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows, k,
                                   d_this, ld_this, d_other, ld_other, d_otor, ld_otor, ws,
                                   d_out_frob, st);
}

extern "C" int lk_als_explicit_half_epoch(const lk_als_plan *plan, const void *d_indptr,
                                          const int32_t *d_indices, const float *d_values,
                                          int64_t n_rows, int64_t n_cols, int32_t k,
                                          float *d_this, int32_t ld_this, const float *d_other,
                                          int32_t ld_other, float reg, void *d_ws,
                                          float *d_out_frob, void *stream)
{
    LK_REQUIRE(plan != nullptr, "lk_als_explicit_half_epoch: null plan");
    LK_REQUIRE(n_rows == plan->n_rows && k == plan->k,
               "lk_als_explicit_half_epoch: plan built for %lld rows, k=%d; got %lld rows, k=%d",
               (long long)plan->n_rows, plan->k, (long long)n_rows, k);
    LK_REQUIRE(ld_this == plan->KP && ld_other == plan->KP,
               "lk_als_explicit_half_epoch: factor leading dimensions (%d, %d) must equal "
               "lk_padded_dim(k)=%d",
               ld_this, ld_other, plan->KP);
    LK_REQUIRE(d_indptr && d_this && d_ws && d_out_frob,
               "lk_als_explicit_half_epoch: null pointer");
    LK_REQUIRE(n_cols >= 0 && (n_cols == 0 || d_other), "lk_als_explicit_half_epoch: null other");
    LK_REQUIRE(plan->solver != LK_SOLVER_CG,
               "lk_als_explicit_half_epoch: only the exact (Cholesky) solver is built for the "
               "explicit model");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    if (plan->KP > 256)
        return lk::als_big_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows, k,
                                      d_this, d_other, nullptr, 0, ws, d_out_frob, st, true, reg);
    if (plan->KP > 64)
        return lk::als_blk_half_epoch(plan, d_indptr, plan->is64, d_indices, d_values, n_rows,
                                      n_cols, k, d_this, d_other, nullptr, 0, ws, d_out_frob, st,
                                      true, reg);
#define LK_CHOL_CASE(NT)                                                                         \
    return plan->is64                                                                            \
               ? lk::launch_chol<NT, true, true>(plan, d_indptr, d_indices, d_values, n_rows, k, \
                                                 d_this, ld_this, d_other, ld_other, nullptr, 0, \
                                                 ws, d_out_frob, st, reg)                        \
               : lk::launch_chol<NT, false, true>(plan, d_indptr, d_indices, d_values, n_rows,   \
                                                  k, d_this, ld_this, d_other, ld_other,         \
                                                  nullptr, 0, ws, d_out_frob, st, reg)
    switch (plan->NT) {
        case 1: LK_CHOL_CASE(1);
        case 2: LK_CHOL_CASE(2);
        case 4: LK_CHOL_CASE(4);
    }
#undef LK_CHOL_CASE
    lk::set_error("lk_als_explicit_half_epoch: no Cholesky kernel for padded k=%d", plan->KP);
    return LK_E_INVALID;
}

extern "C" int lk_als_check_status(const lk_als_plan *plan, void *d_ws, void *stream)
{
    LK_REQUIRE(plan && d_ws, "lk_als_check_status: null pointer");
    int status[2] = {0, 0};
    LK_HIP_CHECK(hipMemcpyAsync(status, static_cast<char *>(d_ws) + plan->off_status,
                                sizeof(status), hipMemcpyDeviceToHost, lk::as_stream(stream)));
    LK_HIP_CHECK(hipStreamSynchronize(lk::as_stream(stream)));
    if (plan->ctl) {
        // AccelTask protocol: a cancelled task reports the interruption, not a result
        int rc = lk::ctl_finish(plan->ctl, lk::as_stream(stream));
        if (rc != LK_OK) return rc;
    }
    if (status[0] != 0) {
        // reference: RuntimeError("ALS solve error: ...") (src/accel/als/implicit.rs:79)
        lk::set_error("ALS solve error: normal matrix of row %d is not positive definite",
                      status[0] - 1);
        return LK_E_NOT_SPD;
    }
    return LK_OK;
}

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes)
    {
        LK_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
        return LK_OK;
    }
};
}  // namespace

extern "C" int lk_als_implicit_half_epoch_host_ctl(const void *h_indptr, int indptr_is_64,
                                                   const int32_t *h_indices,
                                                   const float *h_values, int64_t n_rows,
                                                   int64_t n_cols, int32_t k, float *h_this,
                                                   const float *h_other, const float *h_otor,
                                                   int32_t solver, float *h_out_frob,
                                                   lk_task_ctl *ctl)
{
    LK_REQUIRE(h_indptr && h_this && h_otor && h_out_frob, "half_epoch_host: null pointer");
    LK_REQUIRE(n_rows >= 0 && n_cols >= 0, "half_epoch_host: negative size");
    const int KP = lk_padded_dim(k);
    LK_REQUIRE(KP > 0, "half_epoch_host: unsupported k=%d", k);
    const int64_t nnz = indptr_is_64 ? static_cast<const int64_t *>(h_indptr)[n_rows]
                                     : static_cast<const int32_t *>(h_indptr)[n_rows];
    LK_REQUIRE(nnz >= 0 && (nnz == 0 || (h_indices && h_values && h_other)),
               "half_epoch_host: null pointer");
    lk_als_plan *plan = nullptr;
    int rc = lk_als_plan_create(&plan, h_indptr, indptr_is_64, n_rows, k, solver);
    if (rc != LK_OK) return rc;
    if (ctl && (rc = lk_als_plan_set_ctl(plan, ctl)) != LK_OK) {
        lk_als_plan_destroy(plan);
        return rc;
    }
    const size_t ipb = (size_t)(n_rows + 1) * (indptr_is_64 ? 8 : 4);
    DevBuf ip, idx, val, th, thp, ot, otp, oo, ws, fr, zb;
    auto fail = [&](int c) {
        lk_als_plan_destroy(plan);
        return c;
    };
    if ((rc = ip.alloc(ipb)) || (rc = idx.alloc((size_t)nnz * 4)) ||
        (rc = val.alloc((size_t)nnz * 4)) || (rc = th.alloc((size_t)n_rows * k * 4)) ||
        (rc = thp.alloc((size_t)n_rows * KP * 4)) || (rc = ot.alloc((size_t)n_cols * k * 4)) ||
        (rc = otp.alloc((size_t)n_cols * KP * 4)) || (rc = oo.alloc((size_t)k * k * 4)) ||
        (rc = ws.alloc(lk_als_plan_workspace_bytes(plan))) || (rc = fr.alloc(4)))
        return fail(rc);
    // padded k = 128 / 256, no task control: the short rows take the Woodbury kernels when there
    // are enough of them to pay for Z = other * OtOr^-1 (the rule of lkpy_amd/_device.py::ALSPlan:
    // LK_ALS_WB_MIN_ROWS, default 4096) and no confidence value is negative (they take sqrt(v))
    if (!ctl && KP > 64 && KP <= 256 && plan->solver == LK_SOLVER_CHOLESKY && n_cols > 0) {
        const char *e = getenv("LK_ALS_WB_MIN_ROWS");
        const int64_t wb_min = e ? atoll(e) : 4096;
        bool neg = false;
        for (int64_t i = 0; i < nnz && !neg; ++i) neg = h_values[i] < 0.f;
        if (wb_min > 0 && lk_als_plan_woodbury_rows(plan) >= wb_min && !neg) {
            if ((rc = zb.alloc((size_t)n_cols * KP * 4))) return fail(rc);
            if ((rc = lk_als_plan_set_z_workspace(plan, static_cast<float *>(zb.p)))) return fail(rc);
        }
    }
#define LK_H(expr)                                                              \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            lk::set_error("%s failed: %s", #expr, hipGetErrorString(_e));       \
            return fail(LK_E_HIP);                                              \
        }                                                                       \
    } while (0)
    LK_H(hipMemcpy(ip.p, h_indptr, ipb, hipMemcpyHostToDevice));
    if (nnz > 0) {
        LK_H(hipMemcpy(idx.p, h_indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
        LK_H(hipMemcpy(val.p, h_values, (size_t)nnz * 4, hipMemcpyHostToDevice));
    }
    if (n_rows > 0) LK_H(hipMemcpy(th.p, h_this, (size_t)n_rows * k * 4, hipMemcpyHostToDevice));
    if (n_cols > 0) LK_H(hipMemcpy(ot.p, h_other, (size_t)n_cols * k * 4, hipMemcpyHostToDevice));
    LK_H(hipMemcpy(oo.p, h_otor, (size_t)k * k * 4, hipMemcpyHostToDevice));
    if ((rc = lk_pad_rows((const float *)th.p, n_rows, k, k, (float *)thp.p, KP, nullptr)) ||
        (rc = lk_pad_rows((const float *)ot.p, n_cols, k, k, (float *)otp.p, KP, nullptr)))
        return fail(rc);
    rc = lk_als_implicit_half_epoch(plan, ip.p, (const int32_t *)idx.p, (const float *)val.p,
                                    n_rows, n_cols, k, (float *)thp.p, KP, (const float *)otp.p,
                                    KP, (const float *)oo.p, k, ws.p, (float *)fr.p, nullptr);
    if (rc != LK_OK) return fail(rc);
    // (LK_E_CANCELLED when the task-control block was cancelled: the rows solved so far are
    // still copied back below -- `this` is updated in place row by row in the reference too)
    const int rc_status = lk_als_check_status(plan, ws.p, nullptr);
    if (rc_status != LK_OK && rc_status != LK_E_CANCELLED) return fail(rc_status);
    // (a failed status leaves the message in lk_last_error: set_error below must not run)
    if ((rc = lk_unpad_rows((const float *)thp.p, n_rows, k, KP, (float *)th.p, k, nullptr)))
        return fail(rc);
    LK_H(hipDeviceSynchronize());
    if (n_rows > 0) LK_H(hipMemcpy(h_this, th.p, (size_t)n_rows * k * 4, hipMemcpyDeviceToHost));
    LK_H(hipMemcpy(h_out_frob, fr.p, 4, hipMemcpyDeviceToHost));
#undef LK_H
    lk_als_plan_destroy(plan);
    return rc_status;
}

extern "C" int lk_als_implicit_half_epoch_host(const void *h_indptr, int indptr_is_64,
                                               const int32_t *h_indices, const float *h_values,
                                               int64_t n_rows, int64_t n_cols, int32_t k,
                                               float *h_this, const float *h_other,
                                               const float *h_otor, int32_t solver,
                                               float *h_out_frob)
{
    return lk_als_implicit_half_epoch_host_ctl(h_indptr, indptr_is_64, h_indices, h_values, n_rows,
                                               n_cols, k, h_this, h_other, h_otor, solver,
                                               h_out_frob, nullptr);
}
