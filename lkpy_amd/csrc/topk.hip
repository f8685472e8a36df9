// topk.hip -- batched dense scoring (f32 MFMA) + per-row top-N selection, gfx950.
//
// Stands in, for a BATCH of users, for
//   * `ALSBase.__call__` scoring, scores = Q u  (src/lenskit/als/_common.py:159-170; a
//     NumPy GEMV per query in the reference), and
//   * `TopNRanker` -> `ItemList.top_n` -> `_accel.data.argtopn`
//     (src/lenskit/basic/topn.py:45-69, src/lenskit/data/_items.py:942-998,
//     src/accel/data/sorting.rs:132-172 with the heap of src/accel/indirect/heap.rs).
//
// Scoring: S[b][i] = fma(U[b][k-1], Q[i][k-1], ... fma(U[b][0], Q[i][0], 0)) -- the f32
// MFMA (v_mfma_f32_32x32x2_f32) is bit-for-bit this k-ordered fmaf chain, so the scores
// equal the oracle's fixed-order scores exactly and integer top-N index lists can be
// compared bit-exactly.  The block stages a 64-user x 64-feature and a 256-item x
// 64-feature panel in LDS (row stride 65 floats: conflict-free ds_read_b32 operand
// fetches), 4 waves each own 32 users x 128 items (4 accumulator tiles).
//
// Selection (one workgroup per user row, scores L2-resident): NaN and excluded items are
// skipped (sorting.rs:143; candidates = training items minus the query's items,
// src/lenskit/basic/candidates.py:77-94), MSB-first 8-bit radix select finds the n-th
// largest key, equal keys are taken by LOWEST index, the <= n winners are bitonic-sorted in
// LDS by (score desc, index asc).  Deterministic.
//
// Roofline: scoring is f32-MFMA bound (2*B*I*k flop).  Large calls take the FUSED path (no
// score matrix): stage 1 scores a strided sample of the catalogue and takes a per-row threshold
// from it (sample_tau_kernel), stage 2 is the full GEMM whose epilogue keeps only the scores
// that reach the threshold (score_filter_kernel), stage 3 sorts each row's few hundred
// candidates exactly (cand_select_kernel); rows the threshold fails on are redone through the
// panel path (GEMM panel written once, row_topn_kernel), which also serves small calls.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace lk {

constexpr int SC_UB = 64;   // users per block tile
constexpr int SC_IB = 256;  // items per block tile
#ifndef LK_TOPK_KC
#define LK_TOPK_KC 32
#endif
constexpr int SC_KC = LK_TOPK_KC;   // features staged per pass (42 KiB of LDS: 3 workgroups per CU)
constexpr int SC_LD = SC_KC + 1;    // conflict-free ds_read_b32 operand fetches

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef LK_TOPK_PHASES
// Diagnostic build only (tools/topk_variants.py): shader-clock cycles of the filter kernel per
// wave, 8 words: [0] operand wait + staging + barrier, [1] MFMA loop, [2] barrier after the loop,
// [3] flags + records, [4] flush, [5] barrier after the epilogue, [6] tiles, [7] whole
__device__ unsigned long long *lk_topk_phase_buf;
#define LK_TP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LK_TP_ADD(i, a, b) ph[i] += (b) - (a)
#else
#define LK_TP_T(var)
#define LK_TP_ADD(i, a, b)
#endif

// MODE 0 (panel): write the score tile to `scores`.  MODE 1 (fused selection, stage 2): nothing
// is written but the entries that reach the row's threshold tau[u] (a lower bound of its n-th
// largest candidate score, from stage 1): they are appended to the row's candidate list as
// (key << 32 | ~index) -- the B x I score matrix never exists in memory.  MODE 2 (stage 1, round
// 5): the workgroup walks every tile of the SAMPLE items and keeps a running maximum per
// accumulator cell -- 256 class maxima per row (class = sample column mod 256), written once to
// scores[u * ld_s + class] at the end: 1 KiB per row instead of the row's whole sample panel.
// UT: 32-user sub-tiles per wave.  Workgroup tile = (64 UT) users x 256 items, wave tile =
// (32 UT) users x 128 items.  UT = 2 (the fused path: no C tile to write, so the accumulators
// may fill the registers) halves the operand bytes per flop -- at k = 64 the 64 x 256 tile is
// bound by L2 -> LDS traffic, not by the matrix cores.
template <int MODE, int UT>
__device__ __forceinline__ void score_panel_body(
    const float *__restrict__ users, int ld_u, int64_t n_users, const float *__restrict__ items,
    int ld_i, int64_t n_items, int kp, float *__restrict__ scores, int64_t ld_s,
    const float *__restrict__ tau, unsigned long long *__restrict__ cand,
    unsigned *__restrict__ cand_cnt, int cand_cap, float *lds_all)
{
    constexpr bool FILTER = MODE == 1;  // candidates appended in the epilogue
    constexpr bool WALK = MODE != 0;    // the workgroup owns its rows and walks the item tiles
    constexpr int UB = SC_UB * UT;
    float *lu = lds_all;
    float *li = lds_all + UB * SC_LD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // panel writer: grid (item tiles, user tiles), one tile per workgroup.  Fused filter: grid
    // (user tiles): the workgroup OWNS its rows and walks every item tile, so the per-row
    // candidate counters live in LDS and no global atomic is ever issued.
    const int64_t u0 = (int64_t)(WALK ? blockIdx.x : blockIdx.y) * UB;
    const int wu = (wave & 1) * 32 * UT;  // wave's user offset inside the tile
    const int wi = (wave >> 1) * 128;     // wave's item offset inside the tile
    const int64_t n_itiles = (n_items + SC_IB - 1) / SC_IB;

    // per-row state of the fused filter, behind the operand slabs
    float *s_tau = lds_all + (UB + SC_IB) * SC_LD;
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_tau + UB);
    if constexpr (FILTER) {
        for (int r = tid; r < UB; r += 256) {
            s_tau[r] = (u0 + r < n_users) ? tau[u0 + r] : __builtin_inff();
            s_cnt[r] = 0u;
        }
    }
#ifdef LK_TOPK_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LK_TP_T(tp_begin);
#endif

    constexpr int UV = UB * (SC_KC / 4) / 256;     // float4 per thread, user panel
    constexpr int IV = SC_IB * (SC_KC / 4) / 256;  // float4 per thread, item panel
    f32x4 ru[UV], ri[IV];

    const int64_t it_begin = WALK ? 0 : (int64_t)blockIdx.x;
    const int64_t it_end = WALK ? n_itiles : it_begin + 1;
    int64_t itile = it_begin;
    // MODE 2: running class maxima, NaN = nothing yet (v_max_f32 returns the other operand)
    f32x16 mx[UT][4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[ut][t][r] = __builtin_nanf("");
    }
    do {
        const int64_t i0 = itile * SC_IB;
        f32x16 acc[UT][4];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ut][t][r] = 0.f;

        // Software pipeline: the next 32-feature slab of both panels is fetched into registers
        // (coalesced float4 reads) while the MFMAs of the current slab run out of LDS.
        auto fetch = [&](int64_t t0, int kc) {  // t0: first item of the tile
            if constexpr (WALK) {
                // the fused path runs with kp % SC_KC == 0 (use_fused): no feature predicate; rows
                // past the end are clamped to the last one (their scores are never taken: tau =
                // +inf for such users, the `in` test for such items) -- unpredicated loads, one
                // 32-bit lane offset against a scalar tile base instead of a 64-bit address each
                constexpr int RPQ = 256 / (SC_KC / 4);  // rows per q step
                const int r0 = tid / (SC_KC / 4);
                const unsigned cb = (unsigned)(tid % (SC_KC / 4)) * 16u;
                const char *ubase = reinterpret_cast<const char *>(users + u0 * ld_u + kc);
                const char *ibase = reinterpret_cast<const char *>(items + t0 * ld_i + kc);
                const int64_t nu64 = n_users - u0, ni64 = n_items - t0;
                const int nu = (int)(nu64 < UB ? nu64 : UB) - 1;
                const int ni = (int)(ni64 < SC_IB ? ni64 : SC_IB) - 1;
#pragma unroll
                for (int q = 0; q < UV; ++q) {
                    const int r = min(r0 + q * RPQ, nu);
                    ru[q] = *reinterpret_cast<const f32x4 *>(
                        ubase + ((unsigned)r * (unsigned)ld_u * 4u + cb));
                }
#pragma unroll
                for (int q = 0; q < IV; ++q) {
                    const int r = min(r0 + q * RPQ, ni);
                    ri[q] = *reinterpret_cast<const f32x4 *>(
                        ibase + ((unsigned)r * (unsigned)ld_i * 4u + cb));
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < UV; ++q) {
                const int e = tid + q * 256, r = e / (SC_KC / 4), c4 = e % (SC_KC / 4);
                ru[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u0 + r < n_users && kc + c4 * 4 < kp)
                    ru[q] = *reinterpret_cast<const f32x4 *>(users + (u0 + r) * ld_u + kc + c4 * 4);
            }
#pragma unroll
            for (int q = 0; q < IV; ++q) {
                const int e = tid + q * 256, r = e / (SC_KC / 4), c4 = e % (SC_KC / 4);
                ri[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t0 + r < n_items && kc + c4 * 4 < kp)
                    ri[q] = *reinterpret_cast<const f32x4 *>(items + (t0 + r) * ld_i + kc + c4 * 4);
            }
        };
        auto stage = [&]() {
#pragma unroll
            for (int q = 0; q < UV; ++q) {
                const int e = tid + q * 256, r = e / (SC_KC / 4), c4 = e % (SC_KC / 4);
                float *d = &lu[r * SC_LD + c4 * 4];
                d[0] = ru[q].x; d[1] = ru[q].y; d[2] = ru[q].z; d[3] = ru[q].w;
            }
#pragma unroll
            for (int q = 0; q < IV; ++q) {
                const int e = tid + q * 256, r = e / (SC_KC / 4), c4 = e % (SC_KC / 4);
                float *d = &li[r * SC_LD + c4 * 4];
                d[0] = ri[q].x; d[1] = ri[q].y; d[2] = ri[q].z; d[3] = ri[q].w;
            }
        };
        LK_TP_T(tp0);
        // filter: the first slab of every tile but the first was requested in the previous
        // tile's epilogue
        if (!WALK || itile == it_begin) fetch(i0, 0);
        for (int kc = 0; kc < kp; kc += SC_KC) {
            LK_TP_T(tp1);
            stage();
            __syncthreads();
            if (kc + SC_KC < kp) fetch(i0, kc + SC_KC);
            LK_TP_T(tp2);
            // v_mfma_f32_32x32x2_f32: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
            for (int kk = 0; kk < SC_KC; kk += 2) {
                float a[UT];
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) a[ut] = lu[(wu + ut * 32 + r) * SC_LD + kk + h];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float b = li[(wi + t * 32 + r) * SC_LD + kk + h];
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut)
                        acc[ut][t] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(a[ut], b, acc[ut][t], 0, 0, 0);
                }
            }
#ifdef LK_TOPK_PHASES
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[UT - 1][3]));
#endif
            LK_TP_T(tp3);
            __syncthreads();  // everyone is done reading before the next slab is staged
            LK_TP_T(tp4);
            LK_TP_ADD(0, kc == 0 ? tp0 : tp1, tp2);
            LK_TP_ADD(1, tp2, tp3);
            LK_TP_ADD(2, tp3, tp4);
        }
        // C/D: col (item) = lane&31, row (user) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        if constexpr (MODE == 2) {
            // the next tile's first slab, in flight behind the maxima
            if (itile + 1 < it_end) fetch(i0 + SC_IB, 0);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // columns past the end (clamped operand rows) never count
                    const bool in = i0 + wi + t * 32 + (lane & 31) < n_items;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg)
                        mx[ut][t][rg] = __builtin_fmaxf(
                            mx[ut][t][rg], in ? acc[ut][t][rg] : __builtin_nanf(""));
                }
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int64_t it = i0 + wi + t * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int64_t u =
                            u0 + wu + ut * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
                        if (u < n_users && it < n_items) scores[u * ld_s + it] = acc[ut][t][rg];
                    }
                }
        } else {
            // Hits (score >= tau of its row; NaN fails) are rare after stage 1: 0.5 % of the
            // scores, 30 % of the accumulator registers of a wave hold one.  A branch per
            // register (taken when nothing hit) and a handler per hit cost a wave as much as
            // the tile's MFMAs (measured: 14.6 k cycles of a 41 k-cycle tile), so the epilogue is
            // straight-line code per 32 x 32 tile instead: 16 flag bits per lane (compare +
            // add-with-carry on the VALU alone), and a lane with any flag stores its 16
            // accumulators as one RECORD (4 ds_write_b128 + an id word with the flags) in this
            // wave's share of the operand LDS, which is idle until the next tile is staged.  The
            // flush takes a lane per record, walks its flagged values (1.1 on average), tests
            // x >= tau itself, takes the slot from the row's LDS counter and writes the candidate.
            constexpr int REC_CAP = 128;  // records per wave
            static_assert((4 * REC_CAP * 16 + 4 * REC_CAP) * 4 <= (UB + SC_IB) * SC_LD * 4,
                          "records must fit the operand LDS");
            float *rvals = lds_all + wave * (REC_CAP * 16);
            unsigned *rids = reinterpret_cast<unsigned *>(lds_all + 4 * REC_CAP * 16) + wave * REC_CAP;
            const unsigned long long lt_mask = (1ull << lane) - 1ull;
            int base = 0;  // records held, wave-uniform
            LK_TP_T(tp5);
            auto flush = [&]() {
                for (int r0 = 0; r0 < base; r0 += 64) {
                    const int rec = r0 + lane;
                    const unsigned id = rec < base ? rids[rec] : 0u;
                    unsigned lm = id >> 16;
                    const unsigned ls = id & 63u;
                    const unsigned rowb = wu + ((id >> 8) & 0xffu) * 32 + 4 * (ls >> 5);
                    const unsigned it = (unsigned)(i0 + wi + ((id >> 6) & 3u) * 32 + (ls & 31u));
                    while (lm) {
                        const int b = 31 - __clz(lm);
                        lm &= ~(1u << b);
                        const int rg = 15 - b;
                        const float x = rvals[rec * 16 + rg];
                        const unsigned row = rowb + (rg & 3) + 8 * (rg >> 2);
                        // the flag is "not below": NaN ends here; so does a +inf score of a
                        // row past the end (tau = +inf, clamped operands): no list to write to
                        if (x >= s_tau[row] && (int64_t)(u0 + row) < n_users) {
                            const unsigned pos = atomicAdd(&s_cnt[row], 1u);  // LDS
                            if (pos < (unsigned)cand_cap)
                                cand[(u0 + row) * cand_cap + pos] =
                                    ((unsigned long long)f2key(x) << 32) | (0xffffffffu - it);
                        }
                    }
                }
                base = 0;
            };
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                float th[16];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg)
                    th[rg] = s_tau[wu + ut * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int col = wi + t * 32 + (lane & 31);
                    const bool in = i0 + col < n_items;
                    // lm: bit 15 - rg = !(x[rg] < th[rg]), shifted in through the carry; two
                    // chains, a scalar mask pair each
                    unsigned lma = 0u, lmb = 0u;
#pragma unroll
                    for (int rg = 0; rg < 8; ++rg) {
                        unsigned long long ca, cb;
                        asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                            : "+v"(lma), "=&s"(ca) : "v"(acc[ut][t][rg]), "v"(th[rg]));
                        asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                            : "+v"(lmb), "=&s"(cb) : "v"(acc[ut][t][rg + 8]), "v"(th[rg + 8]));
                    }
                    const unsigned lm = (lma << 8) | lmb;
                    const bool h = in && lm != 0u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(h);
                    if (base + __popcll(m) > REC_CAP) flush();  // wave-uniform
                    if (h) {
                        const int slot = base + __popcll(m & lt_mask);
                        f32x4 *dst = reinterpret_cast<f32x4 *>(rvals + slot * 16);
                        const f32x16 a = acc[ut][t];
                        dst[0] = f32x4{a[0], a[1], a[2], a[3]};
                        dst[1] = f32x4{a[4], a[5], a[6], a[7]};
                        dst[2] = f32x4{a[8], a[9], a[10], a[11]};
                        dst[3] = f32x4{a[12], a[13], a[14], a[15]};
                        rids[slot] = (lm << 16) | (unsigned)((ut << 8) | (t << 6)) | (unsigned)lane;
                    }
                    base += __popcll(m);
                }
                // the next tile's first slab: requested once the first half of the accumulators
                // is dead (its registers take the data), in flight behind the rest of the epilogue
                if (ut == 0 && itile + 1 < it_end) fetch(i0 + SC_IB, 0);
            }
            LK_TP_T(tp6);
            flush();
            LK_TP_T(tp7);
            __syncthreads();  // the records lie in the next tile's operand slab
            LK_TP_T(tp8);
            LK_TP_ADD(3, tp5, tp6);
            LK_TP_ADD(4, tp6, tp7);
            LK_TP_ADD(5, tp7, tp8);
            LK_TP_ADD(6, 0, 1);
        }
        if constexpr (!WALK) break;  // one tile per workgroup: no loop at all for the compiler
        ++itile;
    } while (itile < it_end);
    if constexpr (MODE == 2) {
        // class = column of the 256-wide tile; C/D layout as above
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cls = wi + t * 32 + (lane & 31);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int64_t u = u0 + wu + ut * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
                    if (u < n_users) scores[u * ld_s + cls] = mx[ut][t][rg];
                }
            }
    }
    if constexpr (FILTER) {
        __syncthreads();
        for (int r = tid; r < UB; r += 256)
            if (u0 + r < n_users) cand_cnt[u0 + r] = s_cnt[r];
#ifdef LK_TOPK_PHASES
        LK_TP_T(tp_end);
        ph[7] = tp_end - tp_begin;
        if (lane == 0 && lk_topk_phase_buf)
            for (int i = 0; i < 8; ++i)
                lk_topk_phase_buf[((size_t)blockIdx.x * 4 + wave) * 8 + i] = ph[i];
#endif
    }
}

#define LK_SCORE_ARGS                                                                            \
    const float *__restrict__ users, int ld_u, int64_t n_users, const float *__restrict__ items, \
        int ld_i, int64_t n_items, int kp, float *__restrict__ scores, int64_t ld_s,             \
        const float *__restrict__ tau, unsigned long long *__restrict__ cand,                    \
        unsigned *__restrict__ cand_cnt, int cand_cap

// the panel writer: 64 x 256 tile, 3 workgroups per CU
__global__ __launch_bounds__(256) void score_panel_kernel(LK_SCORE_ARGS)
{
    __shared__ __attribute__((aligned(16))) float lds_all[(SC_UB + SC_IB) * SC_LD];
    score_panel_body<0, 1>(users, ld_u, n_users, items, ld_i, n_items, kp, scores, ld_s, tau,
                               cand, cand_cnt, cand_cap, lds_all);
}

// stage 1 of the fused selection (round 5): a workgroup owns 64 rows, walks the sample tiles and
// writes 256 class maxima per row -- `scores` is the [rows x 256] class table, ld_s = 256
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void sample_cmax_kernel(
    LK_SCORE_ARGS)
{
    __shared__ __attribute__((aligned(16))) float lds_all[(SC_UB + SC_IB) * SC_LD];
    score_panel_body<2, 1>(users, ld_u, n_users, items, ld_i, n_items, kp, scores, ld_s, tau, cand,
                           cand_cnt, cand_cap, lds_all);
}

// the fused filter: 128 x 256 tile, 128 accumulator registers, held to 2 workgroups per CU
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void score_filter_kernel(
    LK_SCORE_ARGS)
{
    __shared__ __attribute__((aligned(16))) float lds_all[(2 * SC_UB + SC_IB) * SC_LD + 4 * SC_UB];
    score_panel_body<1, 2>(users, ld_u, n_users, items, ld_i, n_items, kp, scores, ld_s, tau,
                              cand, cand_cnt, cand_cap, lds_all);
}
#undef LK_SCORE_ARGS

#ifndef LK_TOPK_DMA
#define LK_TOPK_DMA 2  // operands by global_load_lds: 1 = k 64 only (score_filter64_kernel), 2 = also
                       // k 32 / 128 / 256 (score_filter_slab_kernel); 0: never (register-staged)
#endif
// ---- the fused filter for 64 features: operands straight into LDS ------------------------------
//
// Same tile (128 users x 256 items per workgroup, 64 x 128 per wave, 128 accumulators), same
// arithmetic (every score the k-ordered fmaf chain) and the same record epilogue as
// score_filter_kernel, but no operand passes through a register on its way to LDS:
//   * the workgroup's 128 x 64 user panel is loaded ONCE (32 KiB, resident for all item tiles),
//   * the item tiles arrive as 16-feature slabs (256 x 16 floats = 16 KiB) in two buffers:
//     slab g + 1 is requested (`global_load_lds_dwordx4`: a wave's instruction moves 1 KiB, lane l
//     -> LDS bytes [16 l, 16 l + 16) of the block M0 names) before the MFMAs of slab g start, and
//     one `s_waitcnt vmcnt(0)` + barrier per slab publishes it -- no staging registers, no
//     ds_write, one barrier per slab instead of two.
// The LDS side of such a load is linear in the lane, so the bank swizzle sits on the GLOBAL side:
// position p of row r holds chunk p ^ f(r) of the row (f = r & 15 for the 16-chunk user rows,
// (r >> 2) & 3 for the 4-chunk slab rows); the 32 rows of an MFMA operand fetch then hit 16
// different 4-bank groups, two rows each (the unpadded pitch allows no better).
// The records of the epilogue live in item buffer 1 (idle between a tile's last slab and the
// next tile's first prefetch into it) -- wave w's 64 records are exactly the 4 KiB its own DMA
// instructions write, so no barrier separates the flush from the next tile.
// Item-split launches (small batches: fewer than one round of 2 x 256 workgroups of 128 users each
// would leave most of the chip idle -- 10 000 users are 79 workgroups): blockIdx.y = part y of
// S = gridDim.y walks the item tiles y, y + S, y + 2 S, ... (interleaved: item numbers often follow
// popularity, a contiguous range would put most candidates into part 0) and keeps ITS candidates of
// a row in a sub-list of its own -- sub_cap entries at ((row * S + y) * sub_cap), count in
// cand_cnt[row * S + y], positions from the workgroup's LDS counter as in the unsplit launch --
// which cand_merge_kernel then packs into the row's list.  (A first version appended to the
// row's list through a global counter: an atomic with return crosses the XCDs' L2s, and one to
// three such round trips per tile and wave doubled the kernel's time.)
#define LK_FILTER_SPLIT_RANGE                                                                    \
    const bool split = tiles_per_wg > 0;                                                         \
    const int64_t t_step = split ? (int64_t)gridDim.y : 1;                                       \
    const int64_t t_lo = split ? (int64_t)blockIdx.y : 0;                                        \
    const int64_t t_hi = n_itiles;                                                               \
    const int64_t l_stride = split ? t_step * (int64_t)tiles_per_wg : (int64_t)cand_cap;         \
    const int64_t l_off = split ? (int64_t)blockIdx.y * tiles_per_wg : 0;                        \
    const unsigned l_cap = split ? (unsigned)tiles_per_wg : (unsigned)cand_cap;                  \
    if (t_lo >= t_hi) {                                                                          \
        for (int rr = threadIdx.x; rr < 128; rr += 256)                                          \
            if ((int64_t)blockIdx.x * 128 + rr < n_users)                                        \
                cand_cnt[((int64_t)blockIdx.x * 128 + rr) * t_step + t_lo] = 0u;                 \
        return;                                                                                  \
    }

constexpr int F64_U_FLOATS = 128 * 64;
constexpr int F64_I_FLOATS = 256 * 16;  // one slab buffer
constexpr int F64_LDS_FLOATS = F64_U_FLOATS + 2 * F64_I_FLOATS + 4 * 64 /* record ids */ + 2 * 128;

__device__ __forceinline__ void lds_dma16(const void *src, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_byte_addr)
        : "memory");
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void score_filter64_kernel(
    const float *__restrict__ users, int64_t n_users, const float *__restrict__ items,
    int64_t n_items, const float *__restrict__ tau, unsigned long long *__restrict__ cand,
    unsigned *__restrict__ cand_cnt, int cand_cap, int tiles_per_wg)
{
    constexpr int UT = 2, UB = 128;
    __shared__ __attribute__((aligned(1024))) float lds_all[F64_LDS_FLOATS];
    float *lu = lds_all;
    float *li = lds_all + F64_U_FLOATS;
    unsigned *rids_all = reinterpret_cast<unsigned *>(li + 2 * F64_I_FLOATS);
    float *s_tau = reinterpret_cast<float *>(rids_all + 4 * 64);
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_tau + UB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t u0 = (int64_t)blockIdx.x * UB;
    const int wu = (wave & 1) * 64, wi = (wave >> 1) * 128;
    const int64_t n_itiles = (n_items + SC_IB - 1) / SC_IB;
    LK_FILTER_SPLIT_RANGE
    const unsigned lds_u = (unsigned)(uintptr_t)lu, lds_i = (unsigned)(uintptr_t)li;

    for (int r = tid; r < UB; r += 256) {
        s_tau[r] = (u0 + r < n_users) ? tau[u0 + r] : __builtin_inff();
        s_cnt[r] = 0u;
    }
    // user panel: instruction n = 8 wave + q moves rows 4 n .. 4 n + 3 (lane: row l >> 4,
    // position l & 15 <- chunk (l & 15) ^ (row & 15))
    {
        const int64_t nu64 = n_users - u0;
        const int nu = (int)(nu64 < UB ? nu64 : UB) - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = wave * 8 + q;
            const int row = n * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            const float *src = users + (u0 + min(row, nu)) * 64 + c * 4;
            lds_dma16(src, lds_u + (unsigned)n * 1024u);
        }
    }
    // item slab (tile origin t0, features 16 s ..) -> buffer s & 1: instruction n = 4 wave + q
    // moves rows 16 n .. 16 n + 15 (lane: row l >> 2, position l & 3 <- chunk (l & 3) ^ ((row >> 2) & 3))
    auto slab_dma = [&](int64_t t0, int s) {
        const int64_t ni64 = n_items - t0;
        const int ni = (int)(ni64 < SC_IB ? ni64 : SC_IB) - 1;
        const float *base = items + t0 * 64 + 16 * s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = wave * 4 + q;
            const int row = n * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((lane >> 4) & 3);  // (row >> 2) & 3 == (lane >> 4) & 3
            const float *src = base + (int64_t)min(row, ni) * 64 + c * 4;
            lds_dma16(src, lds_i + (unsigned)(s & 1) * (F64_I_FLOATS * 4u) + (unsigned)n * 1024u);
        }
    };
    slab_dma(t_lo * SC_IB, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // operand addresses (floats).  A: row wu + 32 ut + r, feature k: chunk (k >> 2) ^ (r & 15);
    // B: row wi + 32 t + r of the slab, feature k' < 16: chunk (k' >> 2) ^ ((r >> 2) & 3)
    const int r = lane & 31, h = lane >> 5;
    int offa[16], offb[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) offa[c] = (wu + r) * 64 + ((c ^ (r & 15)) << 2) + h;
#pragma unroll
    for (int c = 0; c < 4; ++c) offb[c] = (wi + r) * 16 + ((c ^ ((r >> 2) & 3)) << 2) + h;

    float *rvals = li + F64_I_FLOATS + wave * (64 * 16);  // 64 records of 16 floats: this wave's 4 KiB of buffer 1
    unsigned *rids = rids_all + wave * 64;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int64_t itile = t_lo; itile < t_hi; itile += t_step) {
        const int64_t i0 = itile * SC_IB;
        f32x16 acc[UT][4];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ut][t][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // request the next slab (of this tile, or the first of the next one)
            if (s < 3) slab_dma(i0, s + 1);
            else if (itile + t_step < t_hi) slab_dma(i0 + t_step * SC_IB, 0);
            const float *ib = li + (s & 1) * F64_I_FLOATS;
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                float a[UT], b[4];
#pragma unroll
                for (int ut = 0; ut < UT; ++ut)
                    a[ut] = lu[offa[4 * s + (kk >> 2)] + ut * (32 * 64) + (kk & 2)];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = ib[offb[kk >> 2] + t * (32 * 16) + (kk & 2)];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut)
                        acc[ut][t] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(a[ut], b[t], acc[ut][t], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next slab
            __syncthreads();  // everyone's share; everyone done reading this slab
        }
        // epilogue: see score_panel_body (records; here in this wave's 4 KiB of item buffer 1)
        int base = 0;
        auto flush = [&]() {
            const int rec = lane;  // base <= 64
            const unsigned id = rec < base ? rids[rec] : 0u;
            unsigned lm = id >> 16;
            const unsigned ls = id & 63u;
            const unsigned rowb = wu + ((id >> 8) & 0xffu) * 32 + 4 * (ls >> 5);
            const unsigned it = (unsigned)(i0 + wi + ((id >> 6) & 3u) * 32 + (ls & 31u));
            while (lm) {
                const int bt = 31 - __clz(lm);
                lm &= ~(1u << bt);
                const int rg = 15 - bt;
                const float x = rvals[rec * 16 + rg];
                const unsigned row = rowb + (rg & 3) + 8 * (rg >> 2);
                // the flag is "not below": NaN ends here, and so does a +inf score of a row past
                // the end (tau = +inf, clamped operands)
                if (x >= s_tau[row] && (int64_t)(u0 + row) < n_users) {
                    const unsigned pos = atomicAdd(&s_cnt[row], 1u);  // LDS
                    if (pos < l_cap)  // (item-split launch: this part's sub-list of the row)
                        cand[(u0 + row) * l_stride + l_off + pos] =
                            ((unsigned long long)f2key(x) << 32) | (0xffffffffu - it);
                }
            }
            base = 0;
        };
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            float th[16];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
                th[rg] = s_tau[wu + ut * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = wi + t * 32 + (lane & 31);
                const bool in = i0 + col < n_items;
                unsigned lma = 0u, lmb = 0u;
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    unsigned long long ca, cb;
                    asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                        : "+v"(lma), "=&s"(ca) : "v"(acc[ut][t][rg]), "v"(th[rg]));
                    asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                        : "+v"(lmb), "=&s"(cb) : "v"(acc[ut][t][rg + 8]), "v"(th[rg + 8]));
                }
                const unsigned lm = (lma << 8) | lmb;
                const bool hh = in && lm != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hh);
                if (base + __popcll(m) > 64) flush();  // wave-uniform
                if (hh) {
                    const int slot = base + __popcll(m & lt_mask);
                    f32x4 *dst = reinterpret_cast<f32x4 *>(rvals + slot * 16);
                    const f32x16 a = acc[ut][t];
                    dst[0] = f32x4{a[0], a[1], a[2], a[3]};
                    dst[1] = f32x4{a[4], a[5], a[6], a[7]};
                    dst[2] = f32x4{a[8], a[9], a[10], a[11]};
                    dst[3] = f32x4{a[12], a[13], a[14], a[15]};
                    rids[slot] = (lm << 16) | (unsigned)((ut << 8) | (t << 6)) | (unsigned)lane;
                }
                base += __popcll(m);
            }
        }
        flush();
    }
    __syncthreads();
    for (int rr = tid; rr < UB; rr += 256)
        if (u0 + rr < n_users) cand_cnt[(u0 + rr) * t_step + t_lo] = s_cnt[rr];
}

#if LK_TOPK_DMA >= 2
// ---- the same DMA staging for the other feature counts (LK_TOPK_DMA >= 2, the default) ----------
// Round 3, first run on a GPU (tools/topk_variants.py, 162 541 x 62 423, n = 100, list checksums
// over all rows identical to the register-staged kernel): k = 32: 9.11 -> 9.00 ms, k = 128:
// 25.08 -> 24.44 ms (106 TF), k = 256: 46.23 -> 45.02 ms (115 TF = 0.73 of the MFMA peak).
// The same DMA staging for the other feature counts (KP = 32, 128, 256): the user panel does not
// stay resident (64 / 128 KiB at KP = 128 / 256), so a slab buffer holds the 16-feature slab of
// BOTH operands (128 x 16 user floats + 256 x 16 item floats = 24 KiB; two buffers), six DMA
// instructions per wave and slab.  KP / 16 slabs per tile (even, so slab s sits in buffer
// s & 1), walked two per loop iteration; records in the item half of buffer 1 as in
// score_filter64_kernel.  Layout model: tools/emul/filter64_layout.py (slab rows are placed the
// same way for both operands).
template <int KP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void score_filter_slab_kernel(
    const float *__restrict__ users, int64_t n_users, const float *__restrict__ items,
    int64_t n_items, const float *__restrict__ tau, unsigned long long *__restrict__ cand,
    unsigned *__restrict__ cand_cnt, int cand_cap, int tiles_per_wg)
{
    constexpr int UT = 2, UB = 128, NS = KP / 16;
    static_assert(NS >= 2 && NS % 2 == 0, "an even number of 16-feature slabs");
    constexpr int US = UB * 16, IS = SC_IB * 16, BUF = US + IS;  // floats
    __shared__ __attribute__((aligned(1024))) float lds_all[2 * BUF + 4 * 64 + 2 * UB];
    unsigned *rids_all = reinterpret_cast<unsigned *>(lds_all + 2 * BUF);
    float *s_tau = reinterpret_cast<float *>(rids_all + 4 * 64);
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_tau + UB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t u0 = (int64_t)blockIdx.x * UB;
    const int wu = (wave & 1) * 64, wi = (wave >> 1) * 128;
    const int64_t n_itiles = (n_items + SC_IB - 1) / SC_IB;
    LK_FILTER_SPLIT_RANGE
    const unsigned lds0 = (unsigned)(uintptr_t)lds_all;

    for (int r = tid; r < UB; r += 256) {
        s_tau[r] = (u0 + r < n_users) ? tau[u0 + r] : __builtin_inff();
        s_cnt[r] = 0u;
    }
    const int64_t nu64 = n_users - u0;
    const int nu = (int)(nu64 < UB ? nu64 : UB) - 1;
    // slab s of both operands (item tile origin t0) -> buffer s & 1.  A DMA instruction moves 16
    // rows of 16 floats: lane -> row l >> 2, position l & 3 <- chunk (l & 3) ^ ((row >> 2) & 3).
    auto slab_dma = [&](int64_t t0, int s) {
        const int64_t ni64 = n_items - t0;
        const int ni = (int)(ni64 < SC_IB ? ni64 : SC_IB) - 1;
        const int c = (lane & 3) ^ ((lane >> 4) & 3);
        const unsigned buf = lds0 + (unsigned)(s & 1) * (BUF * 4u);
#pragma unroll
        for (int q = 0; q < 2; ++q) {  // user rows 16 n .., n = 2 wave + q
            const int n = wave * 2 + q;
            const int row = n * 16 + (lane >> 2);
            const float *src = users + (u0 + min(row, nu)) * KP + 16 * s + c * 4;
            lds_dma16(src, buf + (unsigned)n * 1024u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // item rows 16 n .., n = 4 wave + q
            const int n = wave * 4 + q;
            const int row = n * 16 + (lane >> 2);
            const float *src = items + (t0 + min(row, ni)) * KP + 16 * s + c * 4;
            lds_dma16(src, buf + (unsigned)(US * 4) + (unsigned)n * 1024u);
        }
    };
    slab_dma(t_lo * SC_IB, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int r = lane & 31, h = lane >> 5;
    int offa[4], offb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        offa[c] = (wu + r) * 16 + ((c ^ ((r >> 2) & 3)) << 2) + h;
        offb[c] = US + (wi + r) * 16 + ((c ^ ((r >> 2) & 3)) << 2) + h;
    }
    float *rvals = lds_all + BUF + US + wave * (64 * 16);  // this wave's 4 KiB of buffer 1's item half
    unsigned *rids = rids_all + wave * 64;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int64_t itile = t_lo; itile < t_hi; itile += t_step) {
        const int64_t i0 = itile * SC_IB;
        f32x16 acc[UT][4];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ut][t][e] = 0.f;
        for (int sp = 0; sp < NS / 2; ++sp) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int s = 2 * sp + par;
                if (s + 1 < NS) slab_dma(i0, s + 1);
                else if (itile + t_step < t_hi) slab_dma(i0 + t_step * SC_IB, 0);
                const float *bb = lds_all + par * BUF;
#pragma unroll
                for (int kk = 0; kk < 16; kk += 2) {
                    float a[UT], b[4];
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut)
                        a[ut] = bb[offa[kk >> 2] + ut * (32 * 16) + (kk & 2)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[t] = bb[offb[kk >> 2] + t * (32 * 16) + (kk & 2)];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ut = 0; ut < UT; ++ut)
                            acc[ut][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ut], b[t], acc[ut][t],
                                                                              0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        int base = 0;
        auto flush = [&]() {
            const int rec = lane;  // base <= 64
            const unsigned id = rec < base ? rids[rec] : 0u;
            unsigned lm = id >> 16;
            const unsigned ls = id & 63u;
            const unsigned rowb = wu + ((id >> 8) & 0xffu) * 32 + 4 * (ls >> 5);
            const unsigned it = (unsigned)(i0 + wi + ((id >> 6) & 3u) * 32 + (ls & 31u));
            while (lm) {
                const int bt = 31 - __clz(lm);
                lm &= ~(1u << bt);
                const int rg = 15 - bt;
                const float x = rvals[rec * 16 + rg];
                const unsigned row = rowb + (rg & 3) + 8 * (rg >> 2);
                if (x >= s_tau[row] && (int64_t)(u0 + row) < n_users) {
                    const unsigned pos = atomicAdd(&s_cnt[row], 1u);  // LDS
                    if (pos < l_cap)  // (item-split launch: this part's sub-list of the row)
                        cand[(u0 + row) * l_stride + l_off + pos] =
                            ((unsigned long long)f2key(x) << 32) | (0xffffffffu - it);
                }
            }
            base = 0;
        };
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            float th[16];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
                th[rg] = s_tau[wu + ut * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = wi + t * 32 + (lane & 31);
                const bool in = i0 + col < n_items;
                unsigned lma = 0u, lmb = 0u;
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) {
                    unsigned long long ca, cb;
                    asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                        : "+v"(lma), "=&s"(ca) : "v"(acc[ut][t][rg]), "v"(th[rg]));
                    asm("v_cmp_nlt_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                        : "+v"(lmb), "=&s"(cb) : "v"(acc[ut][t][rg + 8]), "v"(th[rg + 8]));
                }
                const unsigned lm = (lma << 8) | lmb;
                const bool hh = in && lm != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hh);
                if (base + __popcll(m) > 64) flush();  // wave-uniform
                if (hh) {
                    const int slot = base + __popcll(m & lt_mask);
                    f32x4 *dst = reinterpret_cast<f32x4 *>(rvals + slot * 16);
                    const f32x16 a = acc[ut][t];
                    dst[0] = f32x4{a[0], a[1], a[2], a[3]};
                    dst[1] = f32x4{a[4], a[5], a[6], a[7]};
                    dst[2] = f32x4{a[8], a[9], a[10], a[11]};
                    dst[3] = f32x4{a[12], a[13], a[14], a[15]};
                    rids[slot] = (lm << 16) | (unsigned)((ut << 8) | (t << 6)) | (unsigned)lane;
                }
                base += __popcll(m);
            }
        }
        flush();
    }
    __syncthreads();
    for (int rr = tid; rr < UB; rr += 256)
        if (u0 + rr < n_users) cand_cnt[(u0 + rr) * t_step + t_lo] = s_cnt[rr];
}
#endif  // LK_TOPK_DMA >= 2



// Bitonic sort of p2 (a power of two >= 2) LDS elements, descending, by the 256 threads of a
// workgroup.  Thread q of a stage owns the PAIR (i, i | j), i = q with a zero inserted at bit
// log2 j -- no idle half -- and the 64 pairs of a wave then cover exactly 128 consecutive
// elements whenever j <= 64: those stages need no workgroup barrier (a wave's LDS operations
// execute in order), so a 512-element sort takes 6 barriers instead of 45.  Callers
// synchronise before (data in place) and may read after the final barrier.
template <typename T>
__device__ __forceinline__ void bitonic_desc_lds(T *a, unsigned p2, int tid)
{
    const unsigned npair = p2 >> 1;
    auto cx = [&](unsigned q, unsigned k, unsigned j) {
        const unsigned i = ((q & ~(j - 1u)) << 1) | (q & (j - 1u)), p = i | j;
        const T x = a[i], y = a[p];
        const bool desc = (i & k) == 0;
        if (desc ? (x < y) : (x > y)) {
            a[i] = y;
            a[p] = x;
        }
    };
    const unsigned klocal = p2 < 128u ? p2 : 128u;
    for (unsigned q = tid; q < npair; q += 256)
        for (unsigned k = 2; k <= klocal; k <<= 1)
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                cx(q, k, j);
                __builtin_amdgcn_wave_barrier();
            }
    __syncthreads();
    for (unsigned k = 256; k <= p2; k <<= 1) {
        for (unsigned j = k >> 1; j >= 128; j >>= 1) {
            for (unsigned q = tid; q < npair; q += 256) cx(q, k, j);
            __syncthreads();
        }
        for (unsigned q = tid; q < npair; q += 256)
            for (unsigned j = 64; j > 0; j >>= 1) {
                cx(q, k, j);
                __builtin_amdgcn_wave_barrier();
            }
        __syncthreads();
    }
}

// stage 3 of the fused selection: one workgroup per row.  The candidates (every entry >= tau)
// are loaded, the row's excluded items are struck out through a small LDS hash of the
// candidates' item numbers (the exclusion list may be in any order and of any length: it is
// only walked once), and the survivors are bitonic-sorted by (score desc, index asc).  Rows
// whose candidate list overflowed are flagged for the unfused path.
// Two tiers: the kernel is latency-bound (a chain of dependent global loads, LDS atomics and
// barriers per row), so what counts is how many rows a CU holds at once.  The first launch keeps
// LCAP = 1024 keys in LDS (16 KiB: 8 workgroups per CU instead of 5 with room for 2048) and
// hands the few rows with more candidates to a second launch with LCAP = STRIDE = 2048.
// (Measured and dropped, lists identical in both: (i) striking exclusions out by comparing every
// candidate with every entry read as LDS broadcasts -- no hash, no atomics, but
// O(candidates x exclusions / 64): 20.7 ms per call against 17.0 with the hash; (ii) running
// this selection as an epilogue of score_filter64_kernel, each workgroup sorting its own 128
// rows while its co-resident partner multiplies: 20.7 ms against 15.4 -- a row is a ~20 us chain
// of dependent loads, LDS atomics and barriers, tolerable only with 8 rows in flight per CU; two
// 256-register workgroups per CU give it one; (iii) a first tier with ONE WAVE per row (512 keys
// + 1024 hash slots = 8 KiB: 20 rows in flight per CU, no workgroup barrier) in front of these
// two: 18.0 ms against 15.4 -- more than a quarter of the rows have over 512 candidates and fall
// through to the next tier, and a wave alone walks the 45 sort steps four pairs per lane.)
template <int LCAP, int STRIDE>
__global__ __launch_bounds__(256) void cand_select_kernel(
    const unsigned long long *__restrict__ cand, const unsigned *__restrict__ cand_cnt,
    const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_items,
    int64_t user_base, int n, int32_t *__restrict__ out_idx, float *__restrict__ out_score,
    int64_t out_ld, int *__restrict__ redo /* [0] = count, [1 ..] = user rows */, int redo_cap,
    int *__restrict__ big /* [0] = count, [1 ..] = batch rows with more than LCAP candidates */,
    int big_cap, const int *__restrict__ row_list /* second tier: the rows to take, or null */,
    int64_t row0 = 0 /* first tier over a row range: batch row of workgroup 0 */)
{
    __shared__ unsigned long long key[LCAP];
    __shared__ int hslot[2 * LCAP];  // open addressing: candidate position + 1, 0 = empty
    const int tid = threadIdx.x;
    int64_t b = blockIdx.x + row0;
    if (row_list) {  // second tier: one workgroup per listed row
        if ((int)blockIdx.x >= min(row_list[0], big_cap)) return;
        b = row_list[1 + blockIdx.x];
    }
    // the row's three scalars are requested together, before anything waits on them
    const unsigned m = cand_cnt[b];
    int64_t eb = 0, ee = 0;
    if (excl_ptr) {
        eb = excl_ptr[user_base + b];
        ee = excl_ptr[user_base + b + 1];
    }
    int32_t *oidx = out_idx + b * out_ld;
    float *osc = out_score ? out_score + b * out_ld : nullptr;
    // rows whose candidate list overflowed, or that end up with fewer than n valid candidates
    // (the threshold is a rank of the sample chosen so that this is a 1e-5 event per row -- or
    // the row simply has fewer than n valid items), are listed and redone exactly by the caller
    auto flag = [&]() {
        const int pos = atomicAdd(&redo[0], 1);
        if (pos < redo_cap) redo[1 + pos] = (int)(user_base + b);
    };
    if (m > (unsigned)STRIDE) {
        if (tid == 0) flag();
        return;
    }
    if (m > (unsigned)LCAP) {
        // more candidates than this tier's LDS holds (a few rows in a hundred): the second-tier
        // launch (LCAP = STRIDE) takes the row; beyond its list: the exact redo path
        if (tid == 0) {
            const int pos = big ? atomicAdd(&big[0], 1) : big_cap;
            if (pos < big_cap)
                big[1 + pos] = (int)b;
            else
                flag();
        }
        return;
    }
    unsigned p2 = 1;
    while (p2 < m) p2 <<= 1;
    if (p2 < 2) p2 = 2;
    for (unsigned i = tid; i < p2; i += 256) key[i] = i < m ? cand[b * STRIDE + i] : 0ull;
    if (ee > eb) {
        const unsigned hmask = 2 * p2 - 1;  // table of 2 p2 >= 2 m slots
        for (unsigned i = tid; i <= hmask; i += 256) hslot[i] = 0;
        __syncthreads();
        for (unsigned i = tid; i < m; i += 256) {
            const unsigned it = 0xffffffffu - (unsigned)(key[i] & 0xffffffffu);
            unsigned h = (it * 2654435761u) & hmask;
            while (atomicCAS(&hslot[h], 0, (int)i + 1) != 0) h = (h + 1) & hmask;
        }
        __syncthreads();
        for (int64_t e = eb + tid; e < ee; e += 256) {
            const unsigned it = (unsigned)excl_items[e];
            unsigned h = (it * 2654435761u) & hmask;
            for (;;) {
                const int s = hslot[h];
                if (s == 0) break;
                const unsigned ci = 0xffffffffu - (unsigned)(key[s - 1] & 0xffffffffu);
                if (ci == it) {
                    key[s - 1] = 0ull;  // excluded: sorts to the end, never emitted
                    break;
                }
                h = (h + 1) & hmask;
            }
        }
    }
    __syncthreads();
    bitonic_desc_lds(key, p2, tid);
    for (int i = tid; i < n; i += 256) {
        const unsigned long long c = (unsigned)i < p2 ? key[i] : 0ull;
        if (c != 0ull) {
            oidx[i] = (int32_t)(0xffffffffu - (unsigned)(c & 0xffffffffu));
            if (osc) osc[i] = key2f((unsigned)(c >> 32));
        } else {
            oidx[i] = -1;
            if (osc) osc[i] = __builtin_nanf("");
        }
    }
    // sorted descending, zeros (excluded / padding) last: fewer than n valid candidates?
    if (tid == 0 && ((unsigned)(n - 1) >= p2 || key[n - 1] == 0ull)) flag();
}

#ifdef LK_WSEL_PHASES
// Diagnostic build only (tools/wsel_phases.py): s_memtime ticks of cand_select_wave_kernel summed
// over all rows, 10 words: [0] row scalars, [1] candidate loads, [2] hash clear + insert,
// [3] exclusion walk, [4] look-up, [5] count + threshold search, [6] compaction, [7] sort,
// [8] output, [9] rows.  Every stamp waits for the memory operations before it.
__device__ unsigned long long *lk_wsel_phase_buf;
#define LK_WP_T(var)                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LK_WP_ADD(i, a, b) wp[i] += (b) - (a)
#else
#define LK_WP_T(var)
#define LK_WP_ADD(i, a, b)
#endif
// Stage 3, first tier, a WAVE per row (round 5; `cand_select_kernel` above stays as the second
// tier and as LK_TOPK_SELECT=sort).  The row's candidates never sort as a whole: they sit in
// registers (16 keys per lane: up to 1024), the excluded ones are struck out through a per-wave
// LDS hash of the candidates' item numbers (a slot holds item + 1; an exclusion that finds its
// item sets the slot's top bit; every candidate then looks its own slot up again), the threshold
// key comes from a bitwise search with wave ballots that stops as soon as between n and 128 keys
// reach it (keys are distinct: (score key << 32 | ~item)), and only those <= 128 keys are
// compacted through LDS and sorted -- a 128-key bitonic network IN REGISTERS (two keys per lane,
// partners by DPP / ds_bpermute), no workgroup barrier anywhere.  5.25 KiB of LDS and <= 64
// registers per row: 30 rows in flight per CU against 8 (16 beside a filter workgroup):
//  * the first 512 candidates are requested before the row's count is known (the list has room
//    for 2048: what lies beyond the count is dropped after the load), the first 192 exclusions
//    before the hash is built, the next 192 while the current ones are looked up;
//  * table probes go out together (four key registers of a lane, the three exclusions of a
//    batch) and only the collisions walk on one by one;
//  * loops over the key registers branch once per group of eight (rows of up to 512 candidates
//    use one group), not once per register: a branch costs a wave more than eight compares;
//  * the search runs on the score halves of the keys with 32-bit compares; the index halves are
//    searched (64-bit compares) only when more than 128 keys share the threshold score.
constexpr int WSEL_CAP = 1024;  // candidates a wave keeps in registers
// hash slots: 21 / 16 of the capacity (load <= 0.76, 0.26 at the usual 350 candidates)
constexpr int WSEL_GRP = 8;     // key registers per group: the first group is always processed
                                // (and loaded before the count is known), the second if m > 512
constexpr int WSEL_LS = 4;      // table probes of a lane that go out together
constexpr int EXCL_UNROLL = 3;  // exclusion entries a lane has in flight (twice that: the next
                                // batch is requested before the current one is looked up)

// v[lane ^ J] for all 64 lanes: DPP inside a row of 16 (quad_perm for 1 and 2, row_half_mirror
// then a reversed quad for 4, row_ror:8 for 8), ds_bpermute across rows
template <int J>
__device__ __forceinline__ unsigned xor_lane(unsigned v)
{
    const int x = (int)v;
    if constexpr (J == 1)
        return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
    else if constexpr (J == 2)
        return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
    else if constexpr (J == 4)
        return (unsigned)__builtin_amdgcn_update_dpp(
            0, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false), 0x1B, 0xf, 0xf, false);
    else if constexpr (J == 8)
        return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);
    else
        return (unsigned)__shfl_xor(x, J, 64);
}

// Bitonic network over 128 keys held two per lane (register H of lane l = element l + 64 H),
// descending.  Element i is compared with i ^ J; the pair sorts descending iff (i & KK) == 0; the
// lower element of the pair (bit J clear) keeps the larger key then, the upper the smaller.
template <int KK, int J>
__device__ __forceinline__ void bitonic128_steps(unsigned long long (&r)[2], int lane)
{
    if constexpr (J == 64) {  // the two registers of a lane (KK = 128: descending)
        const unsigned long long x = r[0], y = r[1];
        r[0] = x > y ? x : y;
        r[1] = x > y ? y : x;
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned xl = (unsigned)r[h], xh = (unsigned)(r[h] >> 32);
            const unsigned long long y =
                ((unsigned long long)xor_lane<J>(xh) << 32) | xor_lane<J>(xl);
            const bool keep_max = ((lane & J) == 0) == (((lane + 64 * h) & KK) == 0);
            if ((r[h] > y) != keep_max) r[h] = y;
        }
    }
    if constexpr (J > 1) bitonic128_steps<KK, J / 2>(r, lane);
}
template <int KK>
__device__ __forceinline__ void bitonic128_desc(unsigned long long (&r)[2], int lane)
{
    bitonic128_steps<KK, KK / 2>(r, lane);
    if constexpr (KK < 128) bitonic128_desc<KK * 2>(r, lane);
}

constexpr int WSEL_LONG_EXCL = 1024;  // exclusion entries beyond which a row goes to the second tier
template <int STRIDE, int CAP>
__device__ __forceinline__ void wave_select_row(
    const int64_t b /* batch row, wave-uniform */, unsigned *tab /* LDS, CAP / 16 * 21 words */,
    const unsigned long long *__restrict__ cand, const unsigned *__restrict__ cand_cnt,
    const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_items,
    int64_t user_base, int n, int32_t *__restrict__ out_idx, float *__restrict__ out_score,
    int64_t out_ld, int *__restrict__ redo /* [0] = count, [1 ..] = user rows */, int redo_cap,
    int *__restrict__ big /* [0] = count, [1 ..] = batch rows with more than CAP candidates */,
    int big_cap, int long_excl = WSEL_LONG_EXCL)
{
    constexpr int NJ = CAP / 64, NG = NJ / WSEL_GRP, WSEL_TAB = CAP / 16 * 21;
    static_assert(STRIDE >= 64 * WSEL_GRP, "eager loads stay inside the row's list");
    static_assert(WSEL_TAB % 4 == 0 && WSEL_TAB * 4 >= 128 * 8, "table holds the 128 selected keys");
    // open addressing: item + 1, top bit = excluded; the 128 selected keys reuse its space
    unsigned long long *sbuf = reinterpret_cast<unsigned long long *>(tab);
    const int lane = threadIdx.x;
#ifdef LK_WSEL_PHASES
    unsigned long long wp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
#endif
    LK_WP_T(wp0);
    unsigned long long k[NJ];
#pragma unroll
    for (int j = 0; j < WSEL_GRP; ++j) k[j] = cand[b * STRIDE + j * 64 + lane];
    const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)cand_cnt[b]);
    int64_t eb = 0, ee = 0;
    if (excl_ptr) {
        eb = excl_ptr[user_base + b];
        ee = excl_ptr[user_base + b + 1];
    }
#ifdef LK_WSEL_PHASES
    asm volatile("" ::"s"(m), "s"(eb), "s"(ee));
#endif
    LK_WP_T(wp1);
    LK_WP_ADD(0, wp0, wp1);
    auto flag = [&]() {
        const int pos = atomicAdd(&redo[0], 1);
        if (pos < redo_cap) redo[1 + pos] = (int)(user_base + b);
    };
    if (m > (unsigned)STRIDE) {  // the list overflowed: the exact redo path
        if (lane == 0) flag();
        return;
    }
    // second tier (cand_select_kernel: a workgroup per row with room for STRIDE keys): rows with
    // more candidates than this tier holds -- and rows with a very long exclusion list, which 256
    // threads walk four times as fast as one wave: a launch is as long as its longest row, and in
    // a SMALL batch (10 000 sampled users: 0.5 ms of the 2.1 ms call) nothing hides that walk
    // (long_excl: 1024 entries, 256 in a small batch -- lk::long_excl)
    if (m > (unsigned)CAP || ee - eb > (int64_t)long_excl) {
        if (lane == 0) {
            const int pos = big ? atomicAdd(&big[0], 1) : big_cap;
            if (pos < big_cap)
                big[1 + pos] = (int)b;
            else
                flag();
        }
        return;
    }
    // key registers in use, in groups of eight: group 0 always (empty slots hold 0 and never
    // count), the next per 512 candidates -- ONE wave-uniform branch per group and loop, not one
    // per register
    const int ng = m > (unsigned)(64 * WSEL_GRP) ? (int)((m + 64 * WSEL_GRP - 1) / (64 * WSEL_GRP)) : 1;
    // the first batch of the exclusion list, in flight while the table is built
    int its[EXCL_UNROLL];
#pragma unroll
    for (int u = 0; u < EXCL_UNROLL; ++u) {
        const int64_t e = eb + u * 64 + lane;
        its[u] = e < ee ? excl_items[e] : -1;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned i = (unsigned)(j * 64 + lane);
        if (j < WSEL_GRP) {
            if (i >= m) k[j] = 0ull;
        } else {
            k[j] = 0ull;
        }
    }
    if (ng > 1) {
#pragma unroll
        for (int j = WSEL_GRP; j < NJ; ++j) {
            const unsigned i = (unsigned)(j * 64 + lane);
            if (i < m) k[j] = cand[b * STRIDE + i];
        }
    }
    LK_WP_T(wp2);
    LK_WP_ADD(1, wp1, wp2);
    auto slot_of = [](unsigned it) { return __umulhi(it * 2654435761u, (unsigned)WSEL_TAB); };
    auto next = [](unsigned h) { return h + 1u == (unsigned)WSEL_TAB ? 0u : h + 1u; };
    if (ee > eb) {
        {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int i = lane; i < WSEL_TAB / 4; i += 64) reinterpret_cast<f32x4 *>(tab)[i] = z;
        }
        wave_lds_sync();
        // insert: a group's first slots are tried together, collisions walk on one by one
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng) break;  // wave-uniform
#pragma unroll
            for (int q = 0; q < WSEL_GRP; q += WSEL_LS) {
                unsigned old[WSEL_LS];
#pragma unroll
                for (int jj = 0; jj < WSEL_LS; ++jj) {
                    const unsigned long long key = k[g * WSEL_GRP + q + jj];
                    const unsigned it = 0xffffffffu - (unsigned)(key & 0xffffffffu);
                    old[jj] = key != 0ull ? atomicCAS(&tab[slot_of(it)], 0u, it + 1u) : 0u;
                }
#pragma unroll
                for (int jj = 0; jj < WSEL_LS; ++jj)
                    if (old[jj] != 0u) {
                        const unsigned it =
                            0xffffffffu - (unsigned)(k[g * WSEL_GRP + q + jj] & 0xffffffffu);
                        unsigned h = next(slot_of(it));
                        while (atomicCAS(&tab[h], 0u, it + 1u) != 0u) h = next(h);
                    }
            }
        }
        wave_lds_sync();
        LK_WP_T(wp3);
        LK_WP_ADD(2, wp2, wp3);
        // the exclusion list, 3 x 64 entries at a time, the next batch requested before this one
        // is looked up (a launch is at least as long as its longest row -- ML-25M: 32 202 entries)
        for (int64_t e0 = eb; e0 < ee; e0 += 64 * EXCL_UNROLL) {
            int nxt[EXCL_UNROLL];
            const bool more = e0 + 64 * EXCL_UNROLL < ee;  // wave-uniform
#pragma unroll
            for (int u = 0; u < EXCL_UNROLL; ++u) {
                const int64_t e = e0 + 64 * EXCL_UNROLL + u * 64 + lane;
                nxt[u] = (more && e < ee) ? excl_items[e] : -1;
            }
            unsigned sl[EXCL_UNROLL];
#pragma unroll
            for (int u = 0; u < EXCL_UNROLL; ++u)
                sl[u] = its[u] >= 0 ? tab[slot_of((unsigned)its[u])] : 0u;
#pragma unroll
            for (int u = 0; u < EXCL_UNROLL; ++u) {
                unsigned s = sl[u];
                if (s == 0u) continue;  // not a candidate (or no entry)
                const unsigned it1 = (unsigned)its[u] + 1u;
                unsigned h = slot_of((unsigned)its[u]);
                for (;;) {
                    if ((s & 0x7fffffffu) == it1) {
                        tab[h] = s | 0x80000000u;  // excluded
                        break;
                    }
                    h = next(h);
                    s = tab[h];
                    if (s == 0u) break;
                }
            }
#pragma unroll
            for (int u = 0; u < EXCL_UNROLL; ++u) its[u] = nxt[u];
        }
        wave_lds_sync();
        LK_WP_T(wp4);
        LK_WP_ADD(3, wp3, wp4);
        // every candidate finds its slot again: struck?
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng) break;  // wave-uniform
#pragma unroll
            for (int q = 0; q < WSEL_GRP; q += WSEL_LS) {
                unsigned sl[WSEL_LS];
#pragma unroll
                for (int jj = 0; jj < WSEL_LS; ++jj) {
                    const unsigned long long key = k[g * WSEL_GRP + q + jj];
                    sl[jj] = key != 0ull
                                 ? tab[slot_of(0xffffffffu - (unsigned)(key & 0xffffffffu))]
                                 : 0u;
                }
#pragma unroll
                for (int jj = 0; jj < WSEL_LS; ++jj) {
                    const int j = g * WSEL_GRP + q + jj;
                    if (k[j] == 0ull) continue;
                    const unsigned it = 0xffffffffu - (unsigned)(k[j] & 0xffffffffu);
                    unsigned s = sl[jj], h = slot_of(it);
                    while ((s & 0x7fffffffu) != it + 1u && s != 0u) {  // (s == 0: cannot happen)
                        h = next(h);
                        s = tab[h];
                    }
                    if (s >> 31) k[j] = 0ull;
                }
            }
        }
        LK_WP_T(wp5);
        LK_WP_ADD(4, wp4, wp5);
        wave_lds_sync();  // the table is dead: the selected keys take its place
    }
    LK_WP_T(wp6);
    // ballots over the key registers in use
    auto count = [&](auto pred) {
        int c = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng) break;  // wave-uniform
#pragma unroll
            for (int jj = 0; jj < WSEL_GRP; ++jj)
                c += __popcll(__builtin_amdgcn_ballot_w64(pred(k[g * WSEL_GRP + jj])));
        }
        return c;
    };
    const int valid = count([](unsigned long long x) { return x != 0ull; });
    // cur: count(key >= cur) >= n throughout; done once it is also <= 128 (or cur is the n-th key)
    unsigned long long cur = 1ull;  // <= 128 valid keys: all of them
    if (valid > 128) {
        // score halves first (a valid key's score half is > 0: f2key(-inf) = 0x007fffff)
        unsigned ch = 0u;
        bool done = false;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned trial = ch | (1u << bit);
            const int cnt =
                count([trial](unsigned long long x) { return (unsigned)(x >> 32) >= trial; });
            if (cnt >= n) {  // wave-uniform
                ch = trial;
                if (cnt <= 128) {
                    done = true;
                    break;
                }
            }
        }
        cur = (unsigned long long)ch << 32;
        if (!done) {
            // more than 128 keys reach the n-th best score: its ties are cut by the index halves
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned long long trial = cur | (1ull << bit);
                const int cnt = count([trial](unsigned long long x) { return x >= trial; });
                if (cnt >= n) {
                    cur = trial;
                    if (cnt <= 128) break;
                }
            }
        }
    }
    LK_WP_T(wp7);
    LK_WP_ADD(5, wp6, wp7);
    int base = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g >= ng) break;  // wave-uniform
#pragma unroll
        for (int jj = 0; jj < WSEL_GRP; ++jj) {
            const unsigned long long key = k[g * WSEL_GRP + jj];
            const bool take = key >= cur;  // cur >= 1: never a struck or empty slot
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(take);
            // takers in the lanes below this one
            const unsigned below = __builtin_amdgcn_mbcnt_hi(
                (unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            if (take) sbuf[base + (int)below] = key;
            base += __popcll(mk);
        }
    }
    wave_lds_sync();
    // two keys per lane (element lane and element 64 + lane; nothing beyond `base`), sorted
    // descending by a bitonic network in registers: no LDS round trip per step
    unsigned long long r[2];
    r[0] = lane < base ? sbuf[lane] : 0ull;
    r[1] = lane + 64 < base ? sbuf[lane + 64] : 0ull;
    LK_WP_T(wp8);
    LK_WP_ADD(6, wp7, wp8);
    bitonic128_desc<2>(r, lane);
    LK_WP_T(wp9);
    LK_WP_ADD(7, wp8, wp9);
    int32_t *oidx = out_idx + b * out_ld;
    float *osc = out_score ? out_score + b * out_ld : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) {
            const unsigned long long c = r[h];
            if (c != 0ull) {
                oidx[i] = (int32_t)(0xffffffffu - (unsigned)(c & 0xffffffffu));
                if (osc) osc[i] = key2f((unsigned)(c >> 32));
            } else {
                oidx[i] = -1;
                if (osc) osc[i] = __builtin_nanf("");
            }
        }
    }
    // fewer than n valid candidates: listed and redone exactly by the caller
    if (lane == 0 && valid < n) flag();
#ifdef LK_WSEL_PHASES
    LK_WP_T(wp10);
    LK_WP_ADD(8, wp9, wp10);
    if (lane == 0 && lk_wsel_phase_buf) {
        // a record per row ([2][262144][16] words: this kernel's, then cmax_tau_kernel's) -- no
        // shared counters: 10^6 atomics on one line would be the only thing measured
        unsigned long long *dst = lk_wsel_phase_buf + (size_t)b * 16;
        for (int i = 0; i < 9; ++i) dst[i] = wp[i];
        dst[9] = wp0;
        dst[10] = wp10;
        dst[11] = m | ((unsigned long long)(ee - eb) << 32);
    }
#endif
}

// first tier: a row per workgroup of one wave, up to WSEL_CAP candidates in registers
template <int STRIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void cand_select_wave_kernel(
    const unsigned long long *__restrict__ cand, const unsigned *__restrict__ cand_cnt,
    const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_items,
    int64_t user_base, int n, int32_t *__restrict__ out_idx, float *__restrict__ out_score,
    int64_t out_ld, int *__restrict__ redo, int redo_cap, int *__restrict__ big, int big_cap,
    int64_t row0 /* batch row of workgroup 0 */, int long_excl)
{
    __shared__ __attribute__((aligned(16))) unsigned tab[WSEL_CAP / 16 * 21];
    wave_select_row<STRIDE, WSEL_CAP>((int64_t)blockIdx.x + row0, tab, cand, cand_cnt, excl_ptr,
                                      excl_items, user_base, n, out_idx, out_score, out_ld, redo,
                                      redo_cap, big, big_cap, long_excl);
}

// Stage 1 of the fused selection, a wave per row: tau[b] = the r-th largest of 256 class maxima
// of the row's sample scores (class = float4 index mod 256; NaN -- excluded items -- skipped).
// At least r sample scores reach it, so it is a lower bound of the r-th largest sample score
// (the two coincide unless two of the r best share a class: tau is then the next class
// maximum down, a few per cent more candidates) -- one sweep of the row, no sort: the r-th
// largest of the 256 keys, four per lane, comes from a bitwise search with wave ballots.
// Fewer than r classes with a valid score: tau = -inf (everything passes, the row overflows and
// is redone through the panel path).
__global__ __launch_bounds__(256) void sample_tau_kernel(const float *__restrict__ sub, int64_t ld,
                                                         int64_t n_sub, int r, int64_t n_rows,
                                                         float *__restrict__ tau,
                                                         unsigned *__restrict__ cand_cnt,
                                                         const int64_t *__restrict__ excl_ptr,
                                                         const int32_t *__restrict__ excl_items,
                                                         int64_t user_base, int stride, int words)
{
    // `words` > 0: the row's excluded SAMPLE columns are struck out here, through a per-wave LDS
    // bitmap (n_sub bits) filled from the exclusion list -- the sample panel is then never
    // rewritten by score_mask_kernel (a launch and a read-modify-write pass less per call)
    extern __shared__ unsigned bm_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= n_rows) return;  // wave-uniform
    unsigned *bm = bm_all + (size_t)wave * words;
    const bool masked = words > 0 && excl_ptr != nullptr;
    if (masked) {
        for (int w = lane; w < words; w += 64) bm[w] = 0u;
        __builtin_amdgcn_wave_barrier();
        const int64_t eb = excl_ptr[user_base + b], ee = excl_ptr[user_base + b + 1];
        for (int64_t e = eb + lane; e < ee; e += 64) {
            const int it = excl_items[e];
            if (it >= 0 && it % stride == 0) {
                const int j = it / stride;
                if (j < n_sub) atomicOr(&bm[j >> 5], 1u << (j & 31));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float *row = sub + b * ld;
    const f32x4 *row4 = reinterpret_cast<const f32x4 *>(row);  // ld % 64 == 0, 256-byte aligned base
    const int64_t n4 = n_sub / 4;
    unsigned v[4] = {0u, 0u, 0u, 0u};  // valid keys are >= f2key(-inf) > 0
    for (int64_t j0 = 0; j0 < n4; j0 += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t i = j0 + 64 * c + lane;
            if (i < n4) {
                const f32x4 x = row4[i];
                // bits of columns 4 i .. 4 i + 3 (4 i is a multiple of 4: never straddles a word)
                const unsigned ex = masked ? (bm[(4 * i) >> 5] >> ((4 * i) & 31)) & 15u : 0u;
                if (x.x == x.x && !(ex & 1u)) v[c] = max(v[c], f2key(x.x));
                if (x.y == x.y && !(ex & 2u)) v[c] = max(v[c], f2key(x.y));
                if (x.z == x.z && !(ex & 4u)) v[c] = max(v[c], f2key(x.z));
                if (x.w == x.w && !(ex & 8u)) v[c] = max(v[c], f2key(x.w));
            }
        }
    }
    for (int64_t i = n4 * 4 + lane; i < n_sub; i += 64) {
        const float x = row[i];
        const bool ex = masked && ((bm[i >> 5] >> (i & 31)) & 1u);
        if (x == x && !ex) v[0] = max(v[0], f2key(x));
    }
    // largest t with #{v >= t} >= r
    unsigned cur = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = cur | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) cnt += __popcll(__builtin_amdgcn_ballot_w64(v[c] >= trial));
        if (cnt >= r) cur = trial;  // wave-uniform
    }
    if (lane == 0) {
        tau[b] = cur ? key2f(cur) : -__builtin_inff();
        cand_cnt[b] = 0u;
    }
}

// Stage 1 with the class maxima from the sample GEMM's epilogue (sample_cmax_kernel; round 5), a
// wave per row.  cmax[b][c] = the maximum over the sample columns j = c (mod 256) of the row's
// scores -- taken WITHOUT the row's exclusions, and a user's own items are exactly its high scores.
// So the classes that hold an excluded sample column ("dirty": about ten per row at a sample of
// 1/15) cannot be used as they are.  Up to `drop_max` of them are simply DROPPED: the threshold is
// then a rank of the remaining classes -- a sample of (256 - d) / 256 of the sample, which holds
// none of the row's exclusions -- and the rank is the one fused_tau_rank gives for that smaller
// sampling fraction (ranks.r[d]: the same 1e-6 bound on "fewer than n items reach tau").  Rows with
// more dirty classes (ML-25M: users with more than ~3000 items) keep too few classes that way:
// `repair_max` of their dirty classes are REPAIRED (the rest dropped): the class maximum is
// taken again over the columns that are not excluded, each score recomputed as the GEMM computes
// it (fmaf over the padded features in order: the same bits), a lane per (class, column), merged
// by an LDS atomic max on the ordered key.  (Repairing every row was the first version: 0.73 ms
// per call against 0.52 for the kernel it replaced -- 170 gathered 256-byte rows per user, each
// lane its own row, is 1024 cache-line requests per 64 scores.)  tau[b] = the r-th largest class
// key, as in sample_tau_kernel.
// LDS per wave: bitmap of the excluded sample columns [words] | dirty-class bits [8] | count [8]
// | dirty-class list [256] | class keys [256].
constexpr int CMAX_LDS_EXTRA = 16 + 256 + 256;  // words per wave beside the bitmap
constexpr int CMAX_DROP_MAX = 128;              // up to here every dirty class is dropped
constexpr int CMAX_REPAIR_MAX = 64;             // beyond: this many are repaired, the rest dropped
constexpr int CMAX_RANKS = 256 - CMAX_REPAIR_MAX + 1;
constexpr int TAU_UNROLL = 8;                   // exclusion entries a lane has in flight
struct TauRanks {
    unsigned char r[CMAX_RANKS];  // rank for d dropped classes (r[0]: none dropped)
};
__global__ __launch_bounds__(256) void cmax_tau_kernel(
    const float *__restrict__ cmax, const float *__restrict__ users, int ld_u,
    const float *__restrict__ qs, int kp, int64_t n_sub, TauRanks ranks, int drop_max,
    int repair_max, int64_t n_rows, float *__restrict__ tau, unsigned *__restrict__ cand_cnt,
    const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_items,
    int64_t user_base, int stride, int words)
{
    int r = ranks.r[0];
    extern __shared__ unsigned bm_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= n_rows) return;  // wave-uniform
    unsigned *bm = bm_all + (size_t)wave * (words + CMAX_LDS_EXTRA);
    unsigned *dirty = bm + words;
    unsigned *nd_p = dirty + 8;
    unsigned *dlist = dirty + 16;
    unsigned *ckey = dlist + 256;
#ifdef LK_WSEL_PHASES
    // [32 ..]: [0] row loads, [1] clear + exclusion walk, [2] class keys to LDS, [3] repairs,
    // [4] search + store, [9] rows, [10] the longest row, [11] its dirty classes
    unsigned long long wp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
    unsigned long long nd_seen = 0;
#endif
    LK_WP_T(wp0);
    const f32x4 x = reinterpret_cast<const f32x4 *>(cmax + b * 256)[lane];
    unsigned v[4];  // classes 4 lane .. 4 lane + 3; 0 = no valid score (keys are >= f2key(-inf) > 0)
    v[0] = x.x == x.x ? f2key(x.x) : 0u;
    v[1] = x.y == x.y ? f2key(x.y) : 0u;
    v[2] = x.z == x.z ? f2key(x.z) : 0u;
    v[3] = x.w == x.w ? f2key(x.w) : 0u;
    int64_t eb = 0, ee = 0;
    if (excl_ptr) {
        eb = excl_ptr[user_base + b];
        ee = excl_ptr[user_base + b + 1];
    }
#ifdef LK_WSEL_PHASES
    asm volatile("" ::"s"(eb), "s"(ee), "v"(v[0]), "v"(v[3]));
#endif
    LK_WP_T(wp1);
    LK_WP_ADD(0, wp0, wp1);
    if (ee > eb) {
        for (int w = lane; w < words + 16; w += 64) bm[w] = 0u;  // bitmap, dirty bits, count
        wave_lds_sync();
        // eight loads in flight per lane, the next eight requested before these are marked: a
        // launch is at least as long as its longest list
        int its[TAU_UNROLL];
#pragma unroll
        for (int u = 0; u < TAU_UNROLL; ++u) {
            const int64_t e = eb + u * 64 + lane;
            its[u] = e < ee ? excl_items[e] : -1;
        }
        for (int64_t e0 = eb; e0 < ee; e0 += 64 * TAU_UNROLL) {
            int nxt[TAU_UNROLL];
            const bool more = e0 + 64 * TAU_UNROLL < ee;  // wave-uniform
#pragma unroll
            for (int u = 0; u < TAU_UNROLL; ++u) {
                const int64_t e = e0 + 64 * TAU_UNROLL + u * 64 + lane;
                nxt[u] = (more && e < ee) ? excl_items[e] : -1;
            }
#pragma unroll
            for (int u = 0; u < TAU_UNROLL; ++u) {
                const int it = its[u];
                if (it >= 0 && it % stride == 0) {
                    const int j = it / stride;
                    if (j < n_sub) {
                        atomicOr(&bm[j >> 5], 1u << (j & 31));
                        const unsigned cls = (unsigned)j & 255u, bit = 1u << (cls & 31u);
                        const unsigned old = atomicOr(&dirty[cls >> 5], bit);
                        if (!(old & bit)) dlist[atomicAdd(nd_p, 1u)] = cls;  // first to mark it
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < TAU_UNROLL; ++u) its[u] = nxt[u];
        }
        wave_lds_sync();
        const int nd = __builtin_amdgcn_readfirstlane((int)*nd_p);
        LK_WP_T(wp2);
        LK_WP_ADD(1, wp1, wp2);
#ifdef LK_WSEL_PHASES
        nd_seen = (unsigned long long)nd;
#endif
        if (nd > 0 && nd <= drop_max) {
            // classes 4 lane .. 4 lane + 3: bits (4 lane) & 31 .. + 3 of their word
            const unsigned bits = (dirty[(4 * lane) >> 5] >> ((4 * lane) & 31)) & 15u;
            if (bits & 1u) v[0] = 0u;
            if (bits & 2u) v[1] = 0u;
            if (bits & 4u) v[2] = 0u;
            if (bits & 8u) v[3] = 0u;
            r = ranks.r[nd];
        } else if (nd > 0) {
            ckey[4 * lane + 0] = v[0];
            ckey[4 * lane + 1] = v[1];
            ckey[4 * lane + 2] = v[2];
            ckey[4 * lane + 3] = v[3];
            wave_lds_sync();
            for (int d = lane; d < nd; d += 64) ckey[dlist[d]] = 0u;
            wave_lds_sync();
            LK_WP_T(wp3);
            LK_WP_ADD(2, wp2, wp3);
            // at most repair_max classes are repaired (the first of the list: any will do); the
            // others stay at "no score" -- dropped, like the rows with few dirty classes
            const int n_rep = nd < repair_max ? nd : repair_max;
            r = ranks.r[nd - n_rep];
            const int cpc = (int)((n_sub + 255) / 256);  // columns per class, at most
            const int total = n_rep * cpc;
            const float *urow = users + b * ld_u;  // wave-uniform: scalar loads
            for (int p0 = 0; p0 < total; p0 += 64) {
                const int p = p0 + lane;
                const int d = p / cpc, s = p - d * cpc;
                if (p < total) {
                    const unsigned cls = dlist[d];
                    const int64_t j = (int64_t)cls + 256 * (int64_t)s;
                    if (j < n_sub && !((bm[j >> 5] >> (j & 31)) & 1u)) {
                        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(qs + j * kp);
                        float acc = 0.f;
#pragma unroll 8
                        for (int kk = 0; kk < kp; kk += 4) {
                            const f32x4 q = q4[kk >> 2];
                            acc = __builtin_fmaf(urow[kk + 0], q.x, acc);
                            acc = __builtin_fmaf(urow[kk + 1], q.y, acc);
                            acc = __builtin_fmaf(urow[kk + 2], q.z, acc);
                            acc = __builtin_fmaf(urow[kk + 3], q.w, acc);
                        }
                        if (acc == acc) atomicMax(&ckey[cls], f2key(acc));
                    }
                }
            }
            wave_lds_sync();
            v[0] = ckey[4 * lane + 0];
            v[1] = ckey[4 * lane + 1];
            v[2] = ckey[4 * lane + 2];
            v[3] = ckey[4 * lane + 3];
            LK_WP_T(wp4);
            LK_WP_ADD(3, wp3, wp4);
        }
    }
    LK_WP_T(wp5);
    // largest t with #{v >= t} >= r
    unsigned cur = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = cur | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) cnt += __popcll(__builtin_amdgcn_ballot_w64(v[c] >= trial));
        if (cnt >= r) cur = trial;  // wave-uniform
    }
    if (lane == 0) {
        tau[b] = cur ? key2f(cur) : -__builtin_inff();
        cand_cnt[b] = 0u;
    }
#ifdef LK_WSEL_PHASES
    LK_WP_T(wp6);
    LK_WP_ADD(4, wp5, wp6);
    if (lane == 0 && lk_wsel_phase_buf) {
        unsigned long long *dst = lk_wsel_phase_buf + ((size_t)262144 + (size_t)b) * 16;
        for (int i = 0; i < 5; ++i) dst[i] = wp[i];
        dst[9] = wp0;
        dst[10] = wp6;
        dst[11] = nd_seen | ((unsigned long long)(ee - eb) << 32);
    }
#endif
}

// scores[b][excluded item] = NaN  (NaN is skipped by the selection, like the reference).
// `stride` > 1: the panel holds the SAMPLE items 0, stride, 2 stride, ... (column = item / stride)
__global__ void score_mask_kernel(const int64_t *__restrict__ excl_ptr,
                                  const int32_t *__restrict__ excl_items, int64_t user_base,
                                  int64_t n_rows, int64_t n_items, float *__restrict__ scores,
                                  int64_t ld_s, int stride = 1)
{
    const int64_t b = blockIdx.x;
    if (b >= n_rows) return;
    const int64_t s = excl_ptr[user_base + b], e = excl_ptr[user_base + b + 1];
    for (int64_t q = s + threadIdx.x; q < e; q += blockDim.x) {
        const int it = excl_items[q];
        if (it < 0 || it >= n_items) continue;
        if (stride == 1)
            scores[b * ld_s + it] = __builtin_nanf("");
        else if (it % stride == 0)
            scores[b * ld_s + it / stride] = __builtin_nanf("");
    }
}

// sample of the item factors: rows 0, stride, 2 stride, ... (one float4 per thread)
__global__ void sample_rows_kernel(const float *__restrict__ items, int ld, int64_t n_sample,
                                   int stride, float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = ld / 4;
    if (e >= n_sample * per_row) return;
    const int64_t r = e / per_row;
    const int c4 = (int)(e - r * per_row);
    reinterpret_cast<f32x4 *>(out)[e] =
        reinterpret_cast<const f32x4 *>(items + r * stride * (int64_t)ld)[c4];
}

// Radix-select path (any n <= MAXN, any row): MSB-first 8-bit radix select of the n-th
// largest key, ties by lowest index, winners left UNSORTED in cand[0 .. count).
__device__ __forceinline__ unsigned row_topn_radix(const float *__restrict__ row, int64_t row_len,
                                                   int n, unsigned long long *cand)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need, s_count, s_valid;
    const int tid = threadIdx.x;

    if (tid == 0) {
        s_prefix = 0;
        s_need = (unsigned)n;
        s_count = 0;
        s_valid = 0;
    }
    __syncthreads();
    // count valid entries
    {
        unsigned v = 0;
        for (int64_t i = tid; i < row_len; i += 256) v += (row[i] == row[i]) ? 1u : 0u;
        atomicAdd(&s_valid, v);
    }
    __syncthreads();
    const unsigned valid = s_valid;
    const unsigned take = valid < (unsigned)n ? valid : (unsigned)n;
    if (tid == 0) s_need = take;
    __syncthreads();

    unsigned kth = 0;
    if (take > 0 && take < valid) {
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned hmask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
            for (int64_t i = tid; i < row_len; i += 256) {
                const float x = row[i];
                if (x == x) {
                    const unsigned k = f2key(x);
                    if ((k & hmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            if (tid == 0) {
                unsigned need = s_need, cum = 0;
                int b = 255;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= need) break;
                    cum += hist[b];
                }
                s_need = need - cum;  // how many to take from bin b (and below bits)
                s_prefix = prefix | ((unsigned)b << shift);
            }
            __syncthreads();
        }
        kth = s_prefix;
    }
    const unsigned need_eq = (take > 0 && take < valid) ? s_need : 0u;
    __syncthreads();
    if (tid == 0) s_count = 0;
    __syncthreads();
    // (a) everything strictly above the k-th key (or every valid entry when take == valid)
    for (int64_t i0 = 0; i0 < row_len; i0 += 256) {
        const int64_t i = i0 + tid;
        bool keep = false;
        unsigned k = 0;
        if (i < row_len) {
            const float x = row[i];
            if (x == x) {
                k = f2key(x);
                keep = (take == valid) ? true : (k > kth);
            }
        }
        if (keep && take > 0) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            cand[pos] = ((unsigned long long)k << 32) | (0xffffffffu - (unsigned)i);
        }
    }
    __syncthreads();
    // (b) ties on the k-th key: the need_eq LOWEST indices, walked in index order
    if (need_eq > 0) {
        __shared__ unsigned wcnt[4];
        __shared__ unsigned s_taken;
        if (tid == 0) s_taken = 0;
        __syncthreads();
        for (int64_t i0 = 0; i0 < row_len; i0 += 256) {
            const int64_t i = i0 + tid;
            bool eq = false;
            if (i < row_len) {
                const float x = row[i];
                eq = (x == x) && (f2key(x) == kth);
            }
            const unsigned long long m = __ballot(eq);
            const int lane = tid & 63, w = tid >> 6;
            if (lane == 0) wcnt[w] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned before = s_taken;
            for (int q = 0; q < w; ++q) before += wcnt[q];
            const unsigned rank =
                before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (eq && rank < need_eq) {
                const unsigned pos = atomicAdd(&s_count, 1u);
                cand[pos] = ((unsigned long long)kth << 32) | (0xffffffffu - (unsigned)i);
            }
            __syncthreads();
            if (tid == 0) s_taken += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
            if (s_taken >= need_eq) break;
        }
        __syncthreads();
    }
    return s_count;  // == take
}

// One workgroup per row: indices of the n largest non-NaN scores, descending, ties by
// lower index; rows with fewer than n candidates are padded with -1 / NaN.
//
// Fast path (n <= 256): every thread keeps the largest key of its strided share of the
// row; the n-th largest of those 256 maxima is a lower bound tau of the n-th largest
// entry (at least n entries are >= tau), typically passed by only a small multiple of n
// entries.  A second sweep collects the entries >= tau (wave ballot + one LDS counter
// update per wave), and the exact order (score desc, index asc) is settled by the same
// bitonic sort as before.  No per-element LDS atomics: the radix histograms cost ~3 cycles
// per element per CU.  Rows the bound cannot handle (fewer than n threads saw a valid
// entry, or more than MAXN entries pass) take the radix path; results are identical.
// `class_max` (optional; the item-kNN recommend kernel's sweep provides it): per row
// `classes` order-preserving keys, each the largest valid key of a disjoint class of the row's
// entries (0: none) -- the first sweep is then not needed.
template <int MAXN>
__global__ __launch_bounds__(256) void row_topn_kernel(const float *__restrict__ scores,
                                                       int64_t ld_s, int64_t row_len, int n,
                                                       int32_t *__restrict__ out_idx,
                                                       float *__restrict__ out_score,
                                                       int64_t out_ld,
                                                       const unsigned *__restrict__ class_max,
                                                       int classes)
{
    __shared__ unsigned long long cand[MAXN];
    __shared__ unsigned tmax[256];
    __shared__ unsigned f_count;
    const int tid = threadIdx.x;
    const float *row = scores + (int64_t)blockIdx.x * ld_s;
    int32_t *oidx = out_idx + (int64_t)blockIdx.x * out_ld;
    float *osc = out_score ? out_score + (int64_t)blockIdx.x * out_ld : nullptr;

    unsigned m = 0;
    bool done = false;
    if (n <= 256) {
        unsigned best = 0;  // valid keys are >= 0x007fffff (f2key(-inf)); 0 = nothing seen
        // 16-byte loads when the row allows it (score panels always do)
        const bool vec = (reinterpret_cast<uintptr_t>(row) & 15u) == 0;
        const int64_t n4 = vec ? row_len / 4 : 0;
        const f32x4 *row4 = reinterpret_cast<const f32x4 *>(row);
        if (class_max) {
            const unsigned *cm = class_max + (int64_t)blockIdx.x * classes;
            for (int j = tid; j < classes; j += 256) best = max(best, cm[j]);
        } else {
            for (int64_t i = tid; i < n4; i += 256) {
                const f32x4 v = row4[i];
                if (v.x == v.x) best = max(best, f2key(v.x));
                if (v.y == v.y) best = max(best, f2key(v.y));
                if (v.z == v.z) best = max(best, f2key(v.z));
                if (v.w == v.w) best = max(best, f2key(v.w));
            }
            for (int64_t i = n4 * 4 + tid; i < row_len; i += 256) {
                const float x = row[i];
                if (x == x) best = max(best, f2key(x));
            }
        }
        tmax[tid] = best;
        if (tid == 0) f_count = 0;
        __syncthreads();
        bitonic_desc_lds(tmax, 256u, tid);
        const unsigned tau = tmax[n - 1];
        if (tau > 0) {
            const int lane = tid & 63;
            // collect (wave-uniform control flow: every lane of a wave calls it together)
            auto offer = [&](bool keep, unsigned k, int64_t i) {
                const unsigned long long mask = __ballot(keep);
                if (mask) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&f_count, (unsigned)__popcll(mask));
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    const unsigned pos = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                    if (keep && pos < (unsigned)MAXN)
                        cand[pos] = ((unsigned long long)k << 32) | (0xffffffffu - (unsigned)i);
                }
            };
            // (x >= tau_f <=> f2key(x) >= tau for every non-NaN x, and false for NaN: the keys
            // preserve the order, -0 ranks with +0 on both sides.  Few entries pass, so a wave first
            // asks whether ANY of its 256 does: one ballot per 16-byte load instead of four offers)
            const float tau_f = key2f(tau);
            // (four 16-byte loads in flight per thread: with one, a workgroup waits a memory latency
            // per 4 KB and the sweep runs at 3.4 TB/s)
            constexpr int TU = 4;
            for (int64_t i0 = 0; i0 < n4; i0 += 256 * TU) {
                f32x4 vv[TU];
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const int64_t i = i0 + 256 * u + tid;
                    vv[u] = row4[i < n4 ? i : n4 - 1];
                }
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const int64_t i = i0 + 256 * u + tid;
                    const bool in = i < n4;
                    const f32x4 v = vv[u];
                    const bool hit =
                        in && (v.x >= tau_f || v.y >= tau_f || v.z >= tau_f || v.w >= tau_f);
                    if (__ballot(hit) == 0ull) continue;  // (wave-uniform)
                    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned k = f2key(xs[c]);
                        offer(in && xs[c] == xs[c] && k >= tau, k, i * 4 + c);
                    }
                }
            }
            for (int64_t i0 = n4 * 4; i0 < row_len; i0 += 256) {
                const int64_t i = i0 + tid;
                float x = 0.f;
                const bool in = i < row_len;
                if (in) x = row[i];
                const unsigned k = f2key(x);
                offer(in && x == x && k >= tau, k, i);
            }
            __syncthreads();
            if (f_count <= (unsigned)MAXN) {
                m = f_count;
                done = true;
            }
        }
    }
    if (!done) {
        __syncthreads();
        m = row_topn_radix(row, row_len, n, cand);
    }
    // bitonic sort, descending, on the padded power of two
    unsigned p2 = 1;
    while (p2 < m) p2 <<= 1;
    for (unsigned i = m + tid; i < p2; i += 256) cand[i] = 0ull;
    __syncthreads();
    if (p2 >= 2) bitonic_desc_lds(cand, p2, tid);
    for (int i = tid; i < n; i += 256) {
        if ((unsigned)i < m) {
            const unsigned long long c = cand[i];
            oidx[i] = (int32_t)(0xffffffffu - (unsigned)(c & 0xffffffffu));
            if (osc) osc[i] = key2f((unsigned)(c >> 32));
        } else {
            oidx[i] = -1;
            if (osc) osc[i] = __builtin_nanf("");
        }
    }
}

constexpr int TOPN_MAX = 4096;
constexpr int64_t SCORE_BATCH_DEFAULT = 2048;  // user rows scored per panel

// rows per score panel (env LK_SCORE_BATCH: tuning knob, multiple of 64)
static int64_t score_batch()
{
    static const int64_t v = [] {
        const char *e = getenv("LK_SCORE_BATCH");
        const long x = e ? atol(e) : 0;
        return (x >= 64 && x <= 65536) ? (int64_t)(x / 64 * 64) : SCORE_BATCH_DEFAULT;
    }();
    return v;
}

static int64_t padded_items(int64_t n_items) { return (n_items + 63) / 64 * 64; }

// ---- fused scoring + selection (never materialises the B x I score matrix) -----------------
// Stage 1: a strided SAMPLE of the items (every stride-th, about 1 / 24 of the catalogue) is
// scored -- class maxima kept in the GEMM's epilogue (round 5), or a small panel (the fallback);
// tau_b = the r-th best of the row's valid sample classes.  r = n would make tau_b a certain
// lower bound of the row's n-th best overall (24 n candidates per row); instead r is the
// smallest rank for which "fewer than n items of the whole catalogue reach tau_b" is a 1e-6
// event per row under the sampling (binomial tail, fused_tau_rank): r = 18 for n = 100 at
// ML-25M's 1 / 22 -- ~400 candidates per row to append, hash and select from.  Stage 2:
// the full GEMM, whose epilogue appends the entries >= tau_b to the row's candidate list.
// Stage 3: exclusions struck out, exact (score desc, index asc) order among the candidates;
// a row that ends up with fewer than n valid candidates (the rare event above, an overflowing
// list, or a row that simply has fewer than n valid items) is listed and redone EXACTLY
// through the panel path -- so the results are those of the panel path, bit for bit, always.
#ifndef LK_TOPK_SAMPLE_DIV_DEFAULT
#define LK_TOPK_SAMPLE_DIV_DEFAULT 24  // cfg2, k = 64, round 5: 12 -> 12.72 ms, 16 -> 12.34, 20 -> 12.20, 24 -> 12.15, 32 -> 12.16
                                       // (round 1, sample panel + whole-list sorts: 8 -> 23.5, 16 -> 22.5, 32 -> 23.6)
#endif
constexpr int FUSED_CAP = 2048;        // candidates per row
// Rows per batch of the fused path.  One launch over ALL rows lets the hardware deal the
// 128-user workgroups to the CUs as they finish: 162 541 users are 1270 workgroups = 2.48 waves
// of 512 resident workgroups, against 3 launches (512 + 512 + 246, the last one at one
// workgroup per CU) when batches were 65 536 rows.  Bounded by the candidate lists (rows x
// 16 KiB) and, when stage 1 writes the sample panel (`panel`: LK_TOPK_STAGE1=panel or a sample too
// large for cmax_tau_kernel's bitmaps), by that panel (rows x sample items floats: 16 GiB at
// most).  A batch of several rounds is a WHOLE number of rounds (512 workgroups = 65 536 rows): at
// k = 256 a partial round takes as long as a full one (a filter workgroup alone on its CU runs
// no faster: 252 ms for 259 workgroups, 255 ms for 512), and the panel bound used to cut batches
// wherever it fell -- a million-item catalogue at a sample of 1/24 got batches of 771 workgroups,
// two rounds of time for one and a half of work (1.19 s against 0.96 s for 205 824 users).
// LK_TOPK_FUSED_ROWS overrides (tuning knob / A-B runs).
static int64_t fused_rows(int64_t n_sub_padded, bool panel)
{
    const char *e = getenv("LK_TOPK_FUSED_ROWS");
    int64_t r = e ? (int64_t)atol(e) : (int64_t)262144;
    const int64_t by_panel = panel ? ((int64_t)16 << 30) / (n_sub_padded * 4) : r;
    if (r > by_panel) r = by_panel;
    r = r / 128 * 128;
    const int64_t round = 512 * 128;
    if (r > round) r = r / round * round;
    // floor: a batch is at least 8192 rows -- unless the catalogue is so large (sample > 512 Ki
    // items) that 8192 sample rows would not fit the 16 GiB panel bound: the bound wins
    int64_t lo = 8192;
    if (lo > by_panel / 128 * 128) lo = by_panel / 128 * 128;
    if (lo < 128) lo = 128;
    return r < lo ? lo : r;
}
constexpr int FUSED_MAX_N = 128;       // expected candidates <= 8 n <= FUSED_CAP / 2

// Item-split filter launches (LK_FILTER_SPLIT_RANGE): the S sub-lists of a row -> the row's list.
// One wave per row: lane s holds part s's count, a wave scan gives the offsets, the parts are
// copied one after the other (64 keys per step).  A part that overflowed its sub-list makes the
// row an overflowed row (count STRIDE + 1: the exact redo path takes it).
__global__ __launch_bounds__(256) void cand_merge_kernel(
    const unsigned long long *__restrict__ sub, const unsigned *__restrict__ sub_cnt, int parts,
    int sub_cap, int64_t rows, unsigned long long *__restrict__ cand, unsigned *__restrict__ cnt,
    int stride)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= rows) return;
    const int lane = threadIdx.x & 63;
    const unsigned c = lane < parts ? sub_cnt[b * parts + lane] : 0u;
    const bool over = __any(c > (unsigned)sub_cap);
    unsigned incl = c;  // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    const unsigned total = __shfl(incl, 63, 64);
    if (over || total > (unsigned)stride) {
        if (lane == 0) cnt[b] = (unsigned)stride + 1u;
        return;
    }
    for (int p = 0; p < parts; ++p) {
        const unsigned n = __shfl(c, p, 64), base = __shfl(incl, p, 64) - n;
        const unsigned long long *src = sub + (b * parts + p) * (int64_t)sub_cap;
        for (unsigned i = lane; i < n; i += 64) cand[b * stride + base + i] = src[i];
    }
    if (lane == 0) cnt[b] = total;
}

static bool tau_mask()
{
    const char *e = getenv("LK_TOPK_TAU_MASK");  // A/B knob (read per call)
    return !(e && e[0] == '0');
}

// LK_TOPK_STAGE1=panel: the round-4 stage 1 (sample panel written, sample_tau_kernel sweeps it);
// default: class maxima from the sample GEMM's epilogue (sample_cmax_kernel + cmax_tau_kernel)
static bool stage1_cmax()
{
    const char *e = getenv("LK_TOPK_STAGE1");  // A/B knob (read per call)
    return !(e && e[0] == 'p');
}
// LK_TOPK_SELECT=sort: the round-4 first selection tier (a workgroup per row, whole-list sort);
// default: a wave per row, threshold search + 128-key sort (cand_select_wave_kernel)
static bool select_wave()
{
    const char *e = getenv("LK_TOPK_SELECT");  // A/B knob (read per call)
    return !(e && e[0] == 's');
}

static int64_t fused_min_items()
{
    const char *e = getenv("LK_TOPK_FUSED_MIN_ITEMS");  // test hook / tuning knob (read per call)
    return e ? (int64_t)atol(e) : (int64_t)16384;
}

static int64_t fused_min_users()
{
    const char *e = getenv("LK_TOPK_FUSED_MIN_USERS");  // test hook / tuning knob
    return e ? (int64_t)atol(e) : (int64_t)8192;  // fewer rows: too few workgroups, panel path
}

static bool use_fused(int64_t n_users, int64_t n_items, int32_t n, int kp = SC_KC)
{
    // kp: whole SC_KC-feature slabs only (the filter kernel loads them without a feature predicate)
    // (kp <= 256: the fused layout reserves 256 floats per sample row; larger embeddings --
    // als_big.hip -- take the panel path)
    return n >= 1 && n <= FUSED_MAX_N && n_items >= fused_min_items() &&
           n_items >= 64 * (int64_t)n && n_users >= fused_min_users() && kp % SC_KC == 0 &&
           kp <= 256;
}

// target size of the stage-1 sample: catalogue / 24 (LK_TOPK_SAMPLE_DIV), at least 16 n, a multiple of 256
static int64_t fused_sub_items(int64_t n_items, int32_t n)
{
    const char *e = getenv("LK_TOPK_SAMPLE_DIV");  // tuning knob: catalogue / sample size
    const long div = e ? atol(e) : 0;
    int64_t s = n_items / ((div >= 2 && div <= 64) ? div : LK_TOPK_SAMPLE_DIV_DEFAULT);
    if (s < 16 * (int64_t)n) s = 16 * (int64_t)n;
    s = (s + 255) / 256 * 256;
    return s < n_items ? s : n_items;
}
// the sample is every stride-th item (item ids often follow popularity or age: a prefix of the
// catalogue would not be representative)
// LK_TOPK_OVERLAP=0: no side stream (see stage 2 of the fused path)
static bool topk_overlap()
{
    const char *e = getenv("LK_TOPK_OVERLAP");
    return !(e && e[0] == '0');
}
// Measured (ML-25M shape, trained factors, n = 100): the cfg2 call over all 162 541 users takes
// 12.0 ms with the boundary at 4096 or 1024 and 12.5 at 256 (its selection hides beside the filter's
// partial round either way); a 10 000-user batch -- where the selection launch stands alone and is
// as long as its slowest row -- 2.55 / 2.13 / 2.01 / 2.10 ms at 4096 / 1024 / 256 / 64.
static int long_excl(bool small_batch)
{
    const char *e = getenv("LK_TOPK_LONG_EXCL");  // tuning knob
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : (small_batch ? 256 : WSEL_LONG_EXCL);
}
static bool topk_split()
{
    const char *e = getenv("LK_TOPK_SPLIT");  // 0: never split the item tiles of a small batch
    return !(e && e[0] == '0');
}
struct TopkSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
// one per host thread (the sharded tests run ranks as threads) AND per device (a thread
// that scores on a second device must not record events / launch on the first device's stream)
static TopkSide &topk_side()
{
    constexpr int MAX_DEV = 64;
    static thread_local TopkSide s[MAX_DEV];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return s[(dev >= 0 && dev < MAX_DEV) ? dev : 0];
}

static int fused_stride(int64_t n_items, int32_t n)
{
    const int64_t st = n_items / fused_sub_items(n_items, n);
    return (int)(st < 1 ? 1 : st);
}
static int64_t fused_sample_items(int64_t n_items, int32_t n)
{
    const int st = fused_stride(n_items, n);
    return (n_items + st - 1) / st;
}
// rank of the sample score used as the threshold: smallest r with P[Binomial(n-1, f) >= r] <= 1e-6,
// f = sample fraction (at least r of the row's n-1 best items would have to be in the sample
// for fewer than n items to reach the threshold).  LK_TOPK_TAU_EXACT=1: r = n (a certain bound).
static bool tau_exact()
{
    const char *e = getenv("LK_TOPK_TAU_EXACT");
    return e && e[0] == '1';
}
// `keep`: the part of the sample the threshold is taken from (cmax_tau_kernel drops the classes
// that hold an exclusion)
static int fused_tau_rank(int64_t n_items, int32_t n, double keep = 1.0)
{
    if (tau_exact()) return n;
    const double f = keep * (double)fused_sample_items(n_items, n) / (double)n_items;
    const int m = n - 1;
    // tail[r] = P[X >= r], X ~ Binomial(m, f), from the pmf
    double pmf = 1.0;
    for (int i = 0; i < m; ++i) pmf *= (1.0 - f);  // P[X = 0]
    double cdf = 0.0;
    for (int r = 0; r <= m; ++r) {
        // tail of rank r = 1 - P[X <= r-1]
        if (r >= 1 && 1.0 - cdf <= 1.0e-6) return r;
        cdf += pmf;
        pmf = pmf * (double)(m - r) / (double)(r + 1) * f / (1.0 - f);
    }
    return n;
}
constexpr int FUSED_REDO_CAP = 4096;  // rows redone one by one; more: everything through the panel
constexpr int FUSED_LCAP = 1024;      // candidates the first selection tier holds in LDS
constexpr int FUSED_BIG_CAP = 32768;  // rows per batch the second tier takes (more: redo path)

struct FusedLayout {
    size_t off_sub, off_tau, off_cnt, off_cand, off_flags, off_qs, off_parts, off_pcnt, bytes;
    int parts, part_cap;  // item-split launches of a small batch (parts <= 1: none)
    int tail_parts;       // ... of the partial last round of a large batch
};
// candidates per (row, part) sub-list: twice a row's whole list divided by the parts (the tiles are
// interleaved, so a part holds ~1/S of a row's candidates; a part that overflows sends the row to
// the exact redo path), at least 512: S = 2 -> 2048, 4 -> 1024, 6 -> 768, >= 8 -> 512
static int filter_part_cap(int parts)
{
    if (parts < 2) return FUSED_CAP;
    int c = (2 * FUSED_CAP / parts + 127) / 128 * 128;
    if (c < 512) c = 512;
    return c > FUSED_CAP ? FUSED_CAP : c;
}

// how many parts the item tiles of a batch of `rows` users are split into: as many as keep the
// launch inside ONE round of 2 x 256 workgroups, at least 4 tiles each; 1 = no split
static int filter_parts(int64_t rows, int64_t n_items, int kp)
{
    const int64_t wgs = (rows + 2 * SC_UB - 1) / (2 * SC_UB);
    const int64_t n_itiles = (n_items + SC_IB - 1) / SC_IB;
    const bool dma_kernel = (LK_TOPK_DMA && kp == 64) ||
                            (LK_TOPK_DMA >= 2 && (kp == 32 || kp == 128 || kp == 256));
    if (!dma_kernel || !topk_split() || wgs >= 2 * 256 || n_itiles < 8) return 1;
    int64_t parts = (2 * 256) / wgs;
    if (parts > n_itiles / 4) parts = n_itiles / 4;
    if (parts > 64) parts = 64;  // (cand_merge_kernel: a lane per part)
    return parts < 2 ? 1 : (int)parts;
}

// rows of the partial LAST round of a batch of several rounds of filter workgroups (0: the batch
// is whole rounds, or a single round)
static int64_t tail_rows(int64_t rows)
{
    const int64_t wgs = (rows + 2 * SC_UB - 1) / (2 * SC_UB), round = 2 * 256;
    if (!topk_overlap() || wgs <= round || wgs % round == 0) return 0;
    return rows - (wgs / round) * round * (2 * SC_UB);
}
// LK_TOPK_SPLIT_TAIL=1 (default 0): the partial last round of a large batch is item-split as a
// small batch is (its workgroups otherwise sit one per CU).  Measured twice on the cfg2 call (1270
// workgroups = two rounds + 246): 13.1 against 12.3 ms with global counters, 12.1 against 11.9 ms
// with per-part sub-lists -- the lone workgroups of the last round run at 0.55 of a paired one's
// time, which splitting them does not beat.  Kept as a knob, off.
static bool topk_split_tail()
{
    const char *e = getenv("LK_TOPK_SPLIT_TAIL");
    return e && e[0] == '1';
}

// stage 1 as class maxima (sample_cmax_kernel + cmax_tau_kernel): unless switched off, and as long
// as four waves' bitmaps of the sample fit 64 KiB of LDS (catalogues up to ~3 M items at 1/24)
static bool cmax_usable(int64_t n_items, int32_t n)
{
    const int64_t words = (fused_sample_items(n_items, n) + 31) / 32;
    return stage1_cmax() && (size_t)(words + CMAX_LDS_EXTRA) * 16 <= 65536;
}

static FusedLayout fused_layout(int64_t n_users, int64_t n_items, int32_t n, int kp = SC_KC)
{
    const int64_t nsub = padded_items(fused_sample_items(n_items, n));
    const bool cmax = cmax_usable(n_items, n);
    const int64_t FUSED_ROWS = fused_rows(nsub, !cmax);
    const int64_t rows = n_users < FUSED_ROWS ? n_users : FUSED_ROWS;
    FusedLayout L;
    size_t off = 0;
    L.off_sub = off;  // the sample panel, or 256 class maxima per row
    off += align_up((size_t)rows * (cmax ? 256 : nsub) * 4, 256);
    L.off_tau = off;
    off += align_up((size_t)rows * 4, 256);
    L.off_cnt = off;
    off += align_up((size_t)rows * 4, 256);
    L.off_cand = off;
    off += align_up((size_t)rows * FUSED_CAP * 8, 256);
    L.off_flags = off;  // redo list: count + rows; the second tier's two lists likewise
    off += align_up((size_t)(1 + FUSED_REDO_CAP + 2 * (1 + FUSED_BIG_CAP)) * 4, 256);
    L.off_qs = off;     // sample of the item factors, [n_sample x 256 floats at most]
    off += align_up((size_t)nsub * 256 * 4, 256);
    // sub-lists of an item-split launch (only a batch below one round of workgroups has them:
    // rows * parts <= 64 Ki and parts * capacity <= 2 lists: at most 1 GiB, at 32 Ki rows).  Sized for the largest split any feature
    // count would take, so that the workspace does not depend on k.
    L.parts = filter_parts(rows, n_items, kp);
    L.part_cap = filter_part_cap(L.parts);
    L.off_parts = L.off_pcnt = off;
    int maxp = filter_parts(rows, n_items, 64);
    int64_t sub_rows = rows;
    // ... or the partial last round of a batch of several rounds (its rows only)
    L.tail_parts = 1;
    const int64_t tr = tail_rows(rows);
    if (maxp <= 1 && tr > 0 && topk_split_tail()) {
        L.tail_parts = filter_parts(tr, n_items, kp);
        maxp = filter_parts(tr, n_items, 64);
        sub_rows = tr;
        L.part_cap = filter_part_cap(L.tail_parts);
    }
    if (maxp > 1) {
        off += align_up((size_t)sub_rows * maxp * filter_part_cap(maxp) * 8, 256);
        L.off_pcnt = off;
        off += align_up((size_t)sub_rows * maxp * 4, 256);
    }
    L.bytes = off;
    return L;
}

}  // namespace lk

#ifdef LK_WSEL_PHASES
extern "C" int lk_wsel_phase_set(unsigned long long *d_buf)
{
    LK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(lk::lk_wsel_phase_buf), &d_buf, sizeof(d_buf)));
    return LK_OK;
}
#endif

#ifdef LK_TOPK_PHASES
extern "C" int lk_topk_phase_set(unsigned long long *d_buf)
{
    LK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(lk::lk_topk_phase_buf), &d_buf, sizeof(d_buf)));
    return LK_OK;
}
#endif

extern "C" size_t lk_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int32_t n)
{
    int64_t rows = n_users < lk::score_batch() ? n_users : lk::score_batch();
    if (rows < 1) rows = 1;
    size_t bytes = lk::align_up((size_t)rows * (size_t)lk::padded_items(n_items) * sizeof(float), 256) + 256;
    // n < 0 (rank everything) or n beyond the selection kernel's capacity: full sort of the panel
    if (n < 0 || n > lk::TOPN_MAX) bytes += lk::topn_sort_workspace_bytes(rows, n_items);
    // fused path: its buffers sit behind the panel (the panel serves overflowing batches)
    if (lk::use_fused(n_users, n_items, n)) bytes += lk::fused_layout(n_users, n_items, n).bytes;
    return bytes;
}

namespace lk {
// Selection over a score panel that somebody else filled (iknn_recommend.hip): the n largest
// non-NaN entries of every row, descending, ties by lower index; n < 0 or n > TOPN_MAX: full sort.
size_t panel_topn_workspace_bytes(int64_t rows, int64_t n_items, int32_t n)
{
    return (n < 0 || n > TOPN_MAX) ? topn_sort_workspace_bytes(rows, n_items) : 0;
}
int panel_topn(const float *panel, int64_t ld_s, int64_t rows, int64_t n_items, int32_t n,
               void *sort_ws, int32_t *out_idx, float *out_score, hipStream_t st,
               const unsigned *class_max = nullptr, int classes_per_row = 0)
{
    if (rows <= 0 || n == 0) return LK_OK;
    if (n < 0 || n > TOPN_MAX) {
        const int64_t cols = n < 0 ? n_items : n;
        return topn_sort(panel, ld_s, rows, n_items, cols, sort_ws, out_idx, out_score, cols, st);
    }
    hipLaunchKernelGGL(row_topn_kernel<TOPN_MAX>, dim3((unsigned)rows), dim3(256), 0, st, panel,
                       ld_s, n_items, n, out_idx, out_score, (int64_t)n, class_max, classes_per_row);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
}  // namespace lk

extern "C" size_t lk_argtopn_workspace_bytes(int64_t n_rows, int64_t row_len, int32_t n)
{
    if (n >= 0 && n <= lk::TOPN_MAX) return 0;
    return lk::topn_sort_workspace_bytes(n_rows, row_len);
}

extern "C" int lk_score_dense(const float *d_users, int32_t ld_users, int64_t n_users,
                              const float *d_items, int32_t ld_items, int64_t n_items, int32_t k,
                              float *d_out, int64_t ld_out, void *stream)
{
    const int KP = lk_padded_dim(k);
    LK_REQUIRE(KP > 0, "lk_score_dense: unsupported k=%d", k);
    LK_REQUIRE(ld_users == KP && ld_items == KP,
               "lk_score_dense: leading dimensions (%d, %d) must equal lk_padded_dim(k)=%d",
               ld_users, ld_items, KP);
    LK_REQUIRE(n_users >= 0 && n_items >= 0 && ld_out >= n_items, "lk_score_dense: bad shape");
    if (n_users == 0 || n_items == 0) return LK_OK;
    LK_REQUIRE(d_users && d_items && d_out, "lk_score_dense: null pointer");
    dim3 grid((unsigned)((n_items + lk::SC_IB - 1) / lk::SC_IB),
              (unsigned)((n_users + lk::SC_UB - 1) / lk::SC_UB));
    hipLaunchKernelGGL(lk::score_panel_kernel, grid, dim3(256), 0, lk::as_stream(stream),
                       d_users, ld_users, n_users, d_items, ld_items, n_items, KP, d_out, ld_out,
                       (const float *)nullptr, (unsigned long long *)nullptr, (unsigned *)nullptr,
                       0);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_argtopn(const float *d_scores, int64_t n_rows, int64_t row_len, int32_t n,
                          void *d_ws, int32_t *d_out_idx, void *stream)
{
    LK_REQUIRE(n_rows >= 0 && row_len >= 0, "lk_argtopn: negative size");
    if (n == 0 || n_rows == 0) return LK_OK;
    LK_REQUIRE(d_out_idx && (row_len == 0 || d_scores), "lk_argtopn: null pointer");
    if (n < 0 || n > lk::TOPN_MAX) {
        // `argsort_descending` (sorting.rs:69-103) / a list longer than the selection kernel
        // holds: full stable sort; the output has min(n, row_len) columns (n < 0: row_len)
        const int64_t cols = (n < 0 || n > row_len) ? row_len : n;
        if (cols == 0) return LK_OK;
        LK_REQUIRE(d_ws, "lk_argtopn: n=%d needs the workspace of lk_argtopn_workspace_bytes", n);
        return lk::topn_sort(d_scores, row_len, n_rows, row_len, cols, d_ws, d_out_idx, nullptr,
                             cols, lk::as_stream(stream));
    }
    hipLaunchKernelGGL(lk::row_topn_kernel<lk::TOPN_MAX>, dim3((unsigned)n_rows), dim3(256), 0,
                       lk::as_stream(stream), d_scores, row_len, row_len, n, d_out_idx,
                       (float *)nullptr, (int64_t)n, (const unsigned *)nullptr, 0);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_score_topk(const float *d_users, int32_t ld_users, int64_t n_users,
                             const float *d_items, int32_t ld_items, int64_t n_items, int32_t k,
                             int32_t n, const int64_t *d_excl_ptr, const int32_t *d_excl_items,
                             void *d_ws, int32_t *d_out_idx, float *d_out_score, void *stream)
{
    const int KP = lk_padded_dim(k);
    LK_REQUIRE(KP > 0, "lk_score_topk: unsupported k=%d", k);
    LK_REQUIRE(ld_users == KP && ld_items == KP,
               "lk_score_topk: leading dimensions (%d, %d) must equal lk_padded_dim(k)=%d",
               ld_users, ld_items, KP);
    LK_REQUIRE(n_users >= 0 && n_items >= 0, "lk_score_topk: negative size");
    if (n_users == 0 || n == 0) return LK_OK;
    // n < 0: rank every candidate (TopNRanker without n, basic/topn.py:61-69); the output then
    // has n_items columns.  Lists beyond the selection kernel's capacity are fully sorted.
    const bool full = n < 0 || n > lk::TOPN_MAX;
    const int64_t out_cols = n < 0 ? n_items : n;
    if (out_cols == 0) return LK_OK;
    LK_REQUIRE(d_users && d_ws && d_out_idx && (n_items == 0 || d_items),
               "lk_score_topk: null pointer");
    hipStream_t st = lk::as_stream(stream);
    float *panel = static_cast<float *>(d_ws);
    const int64_t ld_s = lk::padded_items(n_items);
    const int64_t pb = n_users < lk::score_batch() ? n_users : lk::score_batch();
    const size_t panel_bytes = lk::align_up((size_t)pb * (size_t)ld_s * sizeof(float), 256) + 256;

    // the panel path for rows [ub, ub + rows): GEMM, exclusion mask, selection (or full sort)
    auto run_panel = [&](int64_t ub, int64_t rows) -> int {
        if (n_items > 0) {
            dim3 grid((unsigned)((n_items + lk::SC_IB - 1) / lk::SC_IB),
                      (unsigned)((rows + lk::SC_UB - 1) / lk::SC_UB));
            hipLaunchKernelGGL(lk::score_panel_kernel, grid, dim3(256), 0, st,
                               d_users + ub * ld_users, ld_users, rows, d_items, ld_items,
                               n_items, KP, panel, ld_s, (const float *)nullptr,
                               (unsigned long long *)nullptr, (unsigned *)nullptr, 0);
            if (d_excl_ptr)
                hipLaunchKernelGGL(lk::score_mask_kernel, dim3((unsigned)rows), dim3(64), 0, st,
                                   d_excl_ptr, d_excl_items, ub, rows, n_items, panel, ld_s);
        }
        if (full) {
            char *sort_ws = static_cast<char *>(d_ws) + panel_bytes;
            return lk::topn_sort(panel, ld_s, rows, n_items, out_cols, sort_ws,
                                 d_out_idx + ub * out_cols,
                                 d_out_score ? d_out_score + ub * out_cols : nullptr, out_cols, st);
        }
        hipLaunchKernelGGL(lk::row_topn_kernel<lk::TOPN_MAX>, dim3((unsigned)rows), dim3(256), 0,
                           st, panel, ld_s, n_items, n, d_out_idx + ub * n,
                           d_out_score ? d_out_score + ub * n : nullptr, (int64_t)n,
                           (const unsigned *)nullptr, 0);
        return LK_OK;
    };

    if (!full && lk::use_fused(n_users, n_items, n, KP)) {
        const lk::FusedLayout L = lk::fused_layout(n_users, n_items, n, KP);
        char *fw = static_cast<char *>(d_ws) + panel_bytes;
        auto *pcand = reinterpret_cast<unsigned long long *>(fw + L.off_parts);
        unsigned *pcnt = reinterpret_cast<unsigned *>(fw + L.off_pcnt);
        float *sub = reinterpret_cast<float *>(fw + L.off_sub);
        float *tau = reinterpret_cast<float *>(fw + L.off_tau);
        unsigned *cnt = reinterpret_cast<unsigned *>(fw + L.off_cnt);
        auto *cand = reinterpret_cast<unsigned long long *>(fw + L.off_cand);
        int *redo = reinterpret_cast<int *>(fw + L.off_flags);
        float *qs = reinterpret_cast<float *>(fw + L.off_qs);
        const int stride = lk::fused_stride(n_items, n);
        const int64_t n_sub = lk::fused_sample_items(n_items, n);
        const int64_t ld_sub = lk::padded_items(n_sub);
        const int r_tau = lk::fused_tau_rank(n_items, n);
        const bool use_cmax = lk::cmax_usable(n_items, n);
        const int64_t FUSED_ROWS = lk::fused_rows(ld_sub, !use_cmax);
        const int64_t batches = (n_users + FUSED_ROWS - 1) / FUSED_ROWS;
        LK_HIP_CHECK(hipMemsetAsync(redo, 0, sizeof(int), st));
        // the sample rows of the item factors, contiguous
        {
            const int64_t vec = n_sub * (KP / 4);
            hipLaunchKernelGGL(lk::sample_rows_kernel, dim3((unsigned)((vec + 255) / 256)),
                               dim3(256), 0, st, d_items, KP, n_sub, stride, qs);
        }
        for (int64_t bi = 0; bi < batches; ++bi) {
            const int64_t ub = bi * FUSED_ROWS;
            const int64_t rows = (n_users - ub) < FUSED_ROWS ? (n_users - ub) : FUSED_ROWS;
            const float *ub_users = d_users + ub * ld_users;
            // stage 1 for rows [r0, r0 + nr) of the batch: threshold from the sample
            const int words = (int)((n_sub + 31) / 32);
            const size_t cmax_lds = (size_t)(words + lk::CMAX_LDS_EXTRA) * 16;  // 4 waves
            auto stage1 = [&](int64_t r0, int64_t nr, hipStream_t s) {
                const float *uu = ub_users + r0 * ld_users;
                if (use_cmax) {
                    // 256 class maxima per row from the sample GEMM's epilogue ([rows x 256] floats
                    // at the head of the sample panel's space); exclusions in cmax_tau_kernel
                    float *cm = sub + r0 * 256;
                    hipLaunchKernelGGL(lk::sample_cmax_kernel,
                                       dim3((unsigned)((nr + lk::SC_UB - 1) / lk::SC_UB)), dim3(256), 0,
                                       s, uu, ld_users, nr, qs, KP, n_sub, KP, cm, (int64_t)256,
                                       (const float *)nullptr, (unsigned long long *)nullptr,
                                       (unsigned *)nullptr, 0);
                    // LK_TOPK_TAU_EXACT=1 (r = n, a certain bound): every dirty class is repaired --
                    // a row must keep n classes.  LK_TOPK_DROP_MAX: A/B knob.
                    lk::TauRanks ranks;
                    for (int d = 0; d <= lk::CMAX_RANKS - 1; ++d)
                        ranks.r[d] =
                            (unsigned char)lk::fused_tau_rank(n_items, n, (256 - d) / 256.0);
                    int drop_max = lk::tau_exact() ? 0 : lk::CMAX_DROP_MAX;
                    if (const char *e = getenv("LK_TOPK_DROP_MAX")) {
                        const int v = atoi(e);
                        if (v >= 0 && v < drop_max) drop_max = v;
                    }
                    const int repair_max = lk::tau_exact() ? 256 : lk::CMAX_REPAIR_MAX;
                    hipLaunchKernelGGL(lk::cmax_tau_kernel, dim3((unsigned)((nr + 3) / 4)),
                                       dim3(256), cmax_lds, s, cm, uu, ld_users, qs, KP, n_sub, ranks,
                                       drop_max, repair_max, nr, tau + r0, cnt + r0, d_excl_ptr,
                                       d_excl_items, ub + r0, stride, words);
                    return;
                }
                float *sp = sub + r0 * ld_sub;
                const dim3 ugrid_sub((unsigned)((n_sub + lk::SC_IB - 1) / lk::SC_IB),
                                     (unsigned)((nr + lk::SC_UB - 1) / lk::SC_UB));
                hipLaunchKernelGGL(lk::score_panel_kernel, ugrid_sub, dim3(256), 0, s, uu, ld_users,
                                   nr, qs, KP, n_sub, KP, sp, ld_sub, (const float *)nullptr,
                                   (unsigned long long *)nullptr, (unsigned *)nullptr, 0);
                // exclusions are struck out of the sample inside sample_tau_kernel (per-wave LDS
                // bitmap) when 4 bitmaps fit 64 KiB of LDS; LK_TOPK_TAU_MASK=0 or a huge sample:
                // the separate score_mask_kernel pass over the panel
                const bool tau_mask = d_excl_ptr && lk::tau_mask() && (size_t)words * 16 <= 65536;
                if (d_excl_ptr && !tau_mask)
                    hipLaunchKernelGGL(lk::score_mask_kernel, dim3((unsigned)nr), dim3(64), 0, s,
                                       d_excl_ptr, d_excl_items, ub + r0, nr, n_items, sp, ld_sub,
                                       stride);
                hipLaunchKernelGGL(lk::sample_tau_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256),
                                   tau_mask ? (size_t)words * 16 : 0, s, sp, ld_sub, n_sub, r_tau, nr,
                                   tau + r0, cnt + r0, d_excl_ptr, d_excl_items, ub + r0, stride,
                                   tau_mask ? words : 0);
            };
            // stage 2: the full contraction, candidates only
            // one workgroup per 128 users, walking all item tiles (no global atomics).
            // The workgroups all take the same time and two fit a CU: a grid that is not a multiple
            // of 2 x 256 ends in a partial round with ONE workgroup per CU (cfg2: 1270 workgroups
            // = 2 full rounds + 246), half of every CU's registers and LDS idle for ~2.5 ms.  The
            // rows of the full rounds (range A) are therefore filtered by a launch of their own and
            // their selection -- latency-bound, 16 KiB of LDS per row -- runs on a side stream in
            // the shadow of the last round (range B); LK_TOPK_OVERLAP=0: one launch each, in order
            // rows for the second selection tier, a list per range: [0] = count, [1 ..] = rows
            int *big = redo + 1 + lk::FUSED_REDO_CAP;
            int *big_b = big + 1 + lk::FUSED_BIG_CAP;
            LK_HIP_CHECK(hipMemsetAsync(big, 0, sizeof(int), st));
            LK_HIP_CHECK(hipMemsetAsync(big_b, 0, sizeof(int), st));
            const int64_t wgs = (rows + 2 * lk::SC_UB - 1) / (2 * lk::SC_UB);
            const int64_t round = 2 * 256;
            int64_t rows_a = rows;  // rows of range A (all of them: no split)
            if (lk::topk_overlap() && wgs > round && wgs % round != 0)
                rows_a = (wgs / round) * round * (2 * lk::SC_UB);
            // item-split launches: only when the WHOLE batch is less than one round of workgroups
            // (measured on the 1270-workgroup cfg2 call: splitting its partial last round of 246
            // in two made the call slower, 13.1 against 12.3 ms)
            const int parts = batches == 1 ? L.parts : 1;
            // (LK_TOPK_SPLIT_TAIL=1: the partial last round split as well -- measured slower, off)
            const int tail_parts = batches == 1 ? L.tail_parts : 1;
            auto filter = [&](int64_t r0, int64_t nr, hipStream_t s, int parts = 1) {
                dim3 ugrid((unsigned)((nr + 2 * lk::SC_UB - 1) / (2 * lk::SC_UB)));
                const float *uu = ub_users + r0 * ld_users;
                float *tau_r = tau + r0;
                unsigned *cnt_r = cnt + r0;
                unsigned long long *cand_r = cand + r0 * lk::FUSED_CAP;
                // (split launches: the kernels' `tiles_per_wg` argument carries the sub-list
                // capacity, their lists and counters are the parts' -- LK_FILTER_SPLIT_RANGE)
                int tiles_per_wg = 0;
                int cap_arg = lk::FUSED_CAP;
                if (parts > 1) {  // (the sub-list buffers hold THIS launch's rows from their start)
                    ugrid.y = (unsigned)parts;
                    tiles_per_wg = L.part_cap;
                    cand_r = pcand;
                    cnt_r = pcnt;
                }
                if (LK_TOPK_DMA && KP == 64)
                    hipLaunchKernelGGL(lk::score_filter64_kernel, ugrid, dim3(256), 0, s, uu, nr,
                                       d_items, n_items, tau_r, cand_r, cnt_r, cap_arg,
                                       tiles_per_wg);
#if LK_TOPK_DMA >= 2
                else if (KP == 32)
                    hipLaunchKernelGGL(lk::score_filter_slab_kernel<32>, ugrid, dim3(256), 0, s, uu,
                                       nr, d_items, n_items, tau_r, cand_r, cnt_r, cap_arg,
                                       tiles_per_wg);
                else if (KP == 128)
                    hipLaunchKernelGGL(lk::score_filter_slab_kernel<128>, ugrid, dim3(256), 0, s, uu,
                                       nr, d_items, n_items, tau_r, cand_r, cnt_r, cap_arg,
                                       tiles_per_wg);
                else if (KP == 256)
                    hipLaunchKernelGGL(lk::score_filter_slab_kernel<256>, ugrid, dim3(256), 0, s, uu,
                                       nr, d_items, n_items, tau_r, cand_r, cnt_r, cap_arg,
                                       tiles_per_wg);
#endif
                else
                    hipLaunchKernelGGL(lk::score_filter_kernel, ugrid, dim3(256), 0, s, uu, ld_users,
                                       nr, d_items, ld_items, n_items, KP, (float *)nullptr,
                                       (int64_t)0, tau_r, cand_r, cnt_r, lk::FUSED_CAP);
                if (parts > 1)
                    hipLaunchKernelGGL(lk::cand_merge_kernel, dim3((unsigned)((nr + 3) / 4)),
                                       dim3(256), 0, s, cand_r, cnt_r, parts, L.part_cap, nr,
                                       cand + r0 * lk::FUSED_CAP, cnt + r0, lk::FUSED_CAP);
            };
            // stage 3, first tier: exclusions, exact order
            const bool wave_sel = lk::select_wave();
            auto select = [&](int64_t r0, int64_t nr, hipStream_t s, int *big) {
                if (wave_sel) {
                    hipLaunchKernelGGL((lk::cand_select_wave_kernel<lk::FUSED_CAP>),
                                       dim3((unsigned)nr), dim3(64), 0, s, cand, cnt, d_excl_ptr,
                                       d_excl_items, ub, n, d_out_idx + ub * n,
                                       d_out_score ? d_out_score + ub * n : nullptr, (int64_t)n,
                                       redo, lk::FUSED_REDO_CAP, big, lk::FUSED_BIG_CAP, r0,
                                       lk::long_excl(parts > 1));
                    return;
                }
                hipLaunchKernelGGL((lk::cand_select_kernel<lk::FUSED_LCAP, lk::FUSED_CAP>),
                                   dim3((unsigned)nr), dim3(256), 0, s, cand, cnt, d_excl_ptr,
                                   d_excl_items, ub, n, d_out_idx + ub * n,
                                   d_out_score ? d_out_score + ub * n : nullptr, (int64_t)n, redo,
                                   lk::FUSED_REDO_CAP, big, lk::FUSED_BIG_CAP, (const int *)nullptr,
                                   r0);
            };
            // second tier: the rows the first listed (more candidates than it holds; mostly the
            // users with hundreds of items of their own among the candidates: 5 % of ML-25M's).
            // A workgroup per listed row, whole-list sort: measured against a wave per row with 32
            // keys per lane (0.34 against 1.34 ms beside the partial round) -- these rows are all
            // hash walks and a 2048-key sort, work for 256 threads.  Range A's rows are listed by
            // the side stream's launch and taken there, beside the partial round, too.
            auto tier2 = [&](int64_t nr, hipStream_t s, const int *list) {
                const dim3 grid2((unsigned)(nr < lk::FUSED_BIG_CAP ? nr : lk::FUSED_BIG_CAP));
                hipLaunchKernelGGL((lk::cand_select_kernel<lk::FUSED_CAP, lk::FUSED_CAP>), grid2,
                                   dim3(256), 0, s, cand, cnt, d_excl_ptr, d_excl_items, ub, n,
                                   d_out_idx + ub * n, d_out_score ? d_out_score + ub * n : nullptr,
                                   (int64_t)n, redo, lk::FUSED_REDO_CAP, (int *)nullptr,
                                   lk::FUSED_BIG_CAP, list);
            };
            if (rows_a < rows) {
                lk::TopkSide &sd = lk::topk_side();
                if (!sd.stream) {
                    LK_HIP_CHECK(hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking));
                    LK_HIP_CHECK(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
                    LK_HIP_CHECK(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming));
                }
                // (Measured and dropped in round 5: the partial round FIRST with range A's stage 1
                // beside it on the side stream -- a seventh of the call's matrix-core work where
                // half of every CU idles -- and range B's selection beside range A's filter:
                // 16.6 ms against 12.3.  The lone filter workgroups slow from 2.5 to 4.2-5.4 ms
                // next to a sample_cmax workgroup, the full rounds from 8.7 to 10.4 ms next to the
                // selection: whatever shares a SIMD with the filter costs it more than it saves.)
                stage1(0, rows, st);
                filter(0, rows_a, st);
                LK_HIP_CHECK(hipEventRecord(sd.fork, st));  // range A filtered
                LK_HIP_CHECK(hipStreamWaitEvent(sd.stream, sd.fork, 0));
                filter(rows_a, rows - rows_a, st, tail_parts);
                // both tiers of range A's selection beside the partial round
                select(0, rows_a, sd.stream, big);
                tier2(rows_a, sd.stream, big);
                LK_HIP_CHECK(hipEventRecord(sd.join, sd.stream));
                select(rows_a, rows - rows_a, st, big_b);
                tier2(rows - rows_a, st, big_b);
                LK_HIP_CHECK(hipStreamWaitEvent(st, sd.join, 0));
            } else {
                stage1(0, rows, st);
                filter(0, rows, st, parts);
                select(0, rows, st, big);
                tier2(rows, st, big);
            }
        }
        LK_HIP_CHECK(hipGetLastError());
        // rows that did not end up with n valid candidates are redone exactly, one by one
        // (a handful per call); more than the list holds: everything through the panel path
        std::vector<int> h_redo((size_t)1 + lk::FUSED_REDO_CAP, 0);
        LK_HIP_CHECK(hipMemcpyAsync(h_redo.data(), redo, sizeof(int), hipMemcpyDeviceToHost, st));
        LK_HIP_CHECK(hipStreamSynchronize(st));
        const int n_redo = h_redo[0];
        if (n_redo > lk::FUSED_REDO_CAP) {
            for (int64_t ub = 0; ub < n_users; ub += lk::score_batch()) {
                const int64_t rows = (n_users - ub) < lk::score_batch() ? (n_users - ub)
                                                                        : lk::score_batch();
                int rc = run_panel(ub, rows);
                if (rc != LK_OK) return rc;
            }
        } else if (n_redo > 0) {
            LK_HIP_CHECK(hipMemcpyAsync(h_redo.data() + 1, redo + 1, sizeof(int) * (size_t)n_redo,
                                        hipMemcpyDeviceToHost, st));
            LK_HIP_CHECK(hipStreamSynchronize(st));
            for (int i = 0; i < n_redo; ++i) {
                int rc = run_panel((int64_t)h_redo[(size_t)1 + i], 1);
                if (rc != LK_OK) return rc;
            }
        }
        LK_HIP_CHECK(hipGetLastError());
        return LK_OK;
    }

    for (int64_t ub = 0; ub < n_users; ub += lk::score_batch()) {
        const int64_t rows = (n_users - ub) < lk::score_batch() ? (n_users - ub) : lk::score_batch();
        int rc = run_panel(ub, rows);
        if (rc != LK_OK) return rc;
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
