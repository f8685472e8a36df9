// als_blk.hip -- exact (Cholesky) ALS row solve for 64 < k <= 256 on gfx950: ONE WORKGROUP
// (4 waves) PER ROW, the normal matrix resident in MFMA accumulators from the first CSR entry
// to the last pivot.
//
// Stands in, for embedding sizes the one-wave-per-row kernel of als_chol.hip cannot hold, for
// `train_row_solve` (src/accel/als/implicit.rs:87-125) / `train_explicit_row`
// (src/accel/als/explicit.rs:80-119) with `POSV::solve` = LAPACK sposv
// (src/accel/als/solve.rs:65-107): per CSR row
//     A = OtOr + sum_j v_j q_j q_j^T,   y = sum_j (v_j + 1) q_j,   x = A^-1 y   (exactly solved).
//
// Data distribution.  A' (= A in "primed" feature order, a symmetric permutation that leaves
// the solution unchanged) is cut into 16x16 tiles; the upper tiles (ti <= tj) are dealt 2x2
// block-cyclic to the four waves: wave (wr, wc) owns tile (ti, tj) iff ti = wr, tj = wc (mod 2).
// Every wave then holds an (NT/2)x(NT/2) upper-triangular grid of LOCAL tiles with static
// register indices (one code path for all waves; the wave coordinates enter only through
// addresses and a few wave-uniform branches), NT(NT+1)/2 tiles of 4 accumulator registers in
// total: 36 registers per wave at k = 128, 144 at k = 256.  The accumulators hold -A', so
// that both the Gram update (-v q q^T) and the Cholesky trailing update (+L L^T) are plain
// MFMA accumulations.
//
// Phases (v_mfma_f32_16x16x4_f32 throughout; f32 MFMA is bit-for-bit an fmaf chain):
//  1. Gram: CSR entries four at a time (one K = 4 MFMA step); a lane loads exactly the NT/2
//     features of the gathered factor row that its wave's A operands need and the NT/2 its B
//     operands need (primed order makes both contiguous), so the four waves together fetch a
//     row twice, from L1/L2.
//  2. Blocked right-looking Cholesky, block size 16.  Step b: the owners of block row b write
//     -acc to an LDS panel (k-group-major layout: every panel access below is conflict-free);
//     every wave factors the 16x16 diagonal block in registers (lane = row, v_readlane
//     multipliers) WHILE solving its own panel rows against it (same multipliers: the
//     triangular solve rides along for free), the right-hand side being one more row; the
//     finished panel L(:, b) goes back to LDS in MFMA-operand layout and the trailing update
//     acc(ti, tj) += L(ti, b) L(tj, b)^T runs on the matrix cores, 4 MFMAs per tile.  The
//     owners of block row b reload their L tiles into the accumulators they vacated: L ends up
//     REGISTER RESIDENT (the packed factor of a 256 x 256 matrix is 131 KB -- it would take
//     a whole CU's LDS and leave one row per CU; like this two rows per CU overlap their
//     latency-bound pivot chains with each other's MFMA phases).
//  3. Back substitution, block row by block row from the bottom: tile-times-vector partial
//     sums straight from the accumulator registers (4 FMAs + a 16-lane reduction per tile),
//     the 16x16 diagonal solves in one wave against the strictly-lower diagonal blocks kept
//     in LDS (16 KB at k = 256).
//
// Rows longer than LK_ALS_LONG_ROW entries are pre-reduced by the chunk kernel (one workgroup
// per chunk, same Gram phase) into slabs that the solving workgroup sums in chunk order:
// bit-reproducible, independent of scheduling.
//
// Roofline: f32 MFMA (157.3 TFLOP/s).  Flops per row: n(2k^2 + 2k) +
// k^3/3 + 2k^2; executed MFMA work: n/4 * NT(NT+1)/2 (Gram, upper tiles only) +
// 4 * sum_{b<NT} (NT-1-b)(NT-b)/2 (trailing updates) instructions of 2048 flop.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "als_plan.h"
#include "common.h"

namespace lk {
namespace blk {

#ifdef LK_BLK_PHASES
// Diagnostic build only (tools/blk_phases.py): shader-clock cycles per row and phase, as seen by
// wave 0 -- 8 words per row: [0] Gram (or slab sum), [1] panel publish + barrier, [2] diagonal
// block + panel rows (the v_readlane chain), [3] panel write-back + barrier, [4] forward step +
// trailing MFMA update, [5] back substitution, [6] row length, [7] whole row
__device__ unsigned *lk_blk_phase_buf;
#define LK_BP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LK_BP_ADD(i, a, b) ph[i] += (unsigned)((b) - (a))
#define LK_BP_ARG , unsigned(&ph)[8]
#define LK_BP_PASS , ph
#else
#define LK_BP_T(var)
#define LK_BP_ADD(i, a, b)
#define LK_BP_ARG
#define LK_BP_PASS
#endif

template <int CTRL>
__device__ __forceinline__ float blk_dpp_add(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}
// sum over the 16 lanes of a row group, in every lane
__device__ __forceinline__ float blk_row16_sum(float x)
{
    x = blk_dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
    x = blk_dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
    x = blk_dpp_add<0x141>(x);  // row_half_mirror
    x = blk_dpp_add<0x140>(x);  // row_mirror
    return x;
}

template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

__host__ __device__ constexpr int lt(int I, int J) { return J * (J + 1) / 2 + I; }

template <int NT>
struct Cfg {
    static constexpr int KP = NT * 16;
    static constexpr int NL = NT / 2;               // local tile grid of a wave: NL x NL, upper
    static constexpr int T = NL * (NL + 1) / 2;     // accumulator tiles per wave
    // primed order: p = t*16 + s  <->  feature f = s*NT + pos(t); even blocks first, then odd,
    // so that the blocks a wave needs as A (t = wr mod 2) or B (t = wc mod 2) operands are NL
    // CONTIGUOUS floats of the gathered row
    __host__ __device__ static constexpr int pos(int t) { return (t & 1) * NL + (t >> 1); }
    // LDS layout (floats)
    static constexpr int P_SUB = KP * 4;            // one k-group sub-panel: [row][4]
    static constexpr int P_SIZE = 4 * P_SUB;        // panel: [k-group g][row][4], 16 KB at KP = 256
    static constexpr int OFF_P0 = 0;
    static constexpr int OFF_P1 = P_SIZE;           // double buffered: one barrier less per step
    static constexpr int OFF_LD = 2 * P_SIZE;       // NT strictly-lower diagonal blocks [16][16]
    static constexpr int OFF_RINV = OFF_LD + NT * 256;  // 1 / L_jj
    static constexpr int OFF_Y = OFF_RINV + KP;     // rhs, forward-substituted in place
    static constexpr int OFF_Z = OFF_Y + KP;        // z = L^-1 y
    static constexpr int OFF_X = OFF_Z + KP;        // solution (primed)
    static constexpr int OFF_ZB = OFF_X + KP;       // z of the current block, double buffered
    static constexpr int OFF_SP = OFF_ZB + 32;      // back substitution: partial sums of 2 waves
    static constexpr int OFF_RED = OFF_SP + 32;     // delta reduction
    // LK_BLK_TWIN: the leader wave's multipliers, column by column, for the other panel waves
    static constexpr int OFF_COL = OFF_RED + 8;     // [16 columns][16]: L_bb[c][j] at j * 16 + c
    static constexpr int OFF_CRINV = OFF_COL + 256; // 1 / L_jj of the current block
    static constexpr int OFF_FLAG = OFF_CRINV + 16; // int: 16 b + (columns of block b published)
    static constexpr int LDS_FLOATS = OFF_FLAG + 16;
    // slab of one chunk: [wave][T*4 + NL][64]
    static constexpr int SLAB_WAVE = (T * 4 + NL) * 64;
    static constexpr int SLAB = 4 * SLAB_WAVE;
};

template <int N>
__device__ __forceinline__ void load_vec(const float *p, float (&q)[N])
{
    static_assert(N == 2 || N % 4 == 0, "vector loads of 2 or multiples of 4 floats");
    if constexpr (N == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2 *>(p);
        q[0] = t.x;
        q[1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < N / 4; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(p + 4 * c);
            q[4 * c + 0] = t.x;
            q[4 * c + 1] = t.y;
            q[4 * c + 2] = t.z;
            q[4 * c + 3] = t.w;
        }
    }
}

// operands of one K = 4 MFMA step for this wave: entry e = lane >> 4 of the group
template <int NT>
struct Ops {
    float qa[NT / 2];  // blocks t = 2I + wr, element s = lane & 15
    float qb[NT / 2];  // blocks t = 2J + wc
    float v;
};

template <int NT>
__device__ __forceinline__ void ops_issue(Ops<NT> &o, int g, int col_reg, float val_reg,
                                          const float *__restrict__ other, int lane, int wr, int wc)
{
    constexpr int NL = NT / 2;
    const int s = (g * 4 + (lane >> 4)) & 63;
    const int col = __shfl(col_reg, s, 64);
    o.v = __shfl(val_reg, s, 64);
    const float *base = other + (int64_t)col * (NT * 16) + (lane & 15) * NT;
    load_vec<NL>(base + wr * NL, o.qa);
    load_vec<NL>(base + wc * NL, o.qb);
}

// ---- donated tiles ---------------------------------------------------------------------------
// The 2 x 2 block-cyclic deal gives wave (wr, wc) = (1, 0) NL local diagonal tiles that are LOWER
// tiles of the matrix ("phantom": computed like the others, never read): 4 of its 10 Gram MFMAs
// per entry group at k = 128, 8 of 36 at k = 256, while the other three waves have no slack --
// every SIMD issues T MFMAs per group for 36 (136) useful tiles out of 40 (144).  With
// LK_BLK_DONATE the three other waves each leave NL / 4 of their tiles -- local (d, NL - 1),
// d < NL / 4 -- to the phantom wave, which accumulates them in its phantom slots (it holds every
// feature block as one of its two operands: odd blocks as qa, even blocks as qb): T - NL / 4
// MFMAs per group for every wave (9 instead of 10, 34 instead of 36).  The solve kernel hands the
// tiles over through LDS after the Gram phase; the chunk kernels store them straight into the
// owner's part of the slab.  Same products, same accumulation chains: bit-identical results.
#ifndef LK_BLK_DONATE
#define LK_BLK_DONATE 1
#endif
template <int NT>
struct Don {
    static constexpr int NL = NT / 2;
    static constexpr int PER = LK_BLK_DONATE ? NL / 4 : 0;  // tiles each of the 3 donors leaves
    static constexpr int ND = 3 * PER;                      // phantom slots in use: lt(p, p), p < ND
    // donor of phantom slot p: 0 = wave (0,0), 1 = wave (0,1), 2 = wave (1,1)
    __host__ __device__ static constexpr int donor(int p) { return p / (PER ? PER : 1); }
    __host__ __device__ static constexpr int dwave(int p) { return donor(p) == 0 ? 0 : (donor(p) == 1 ? 2 : 3); }
    __host__ __device__ static constexpr int dr(int p) { return donor(p) == 2 ? 1 : 0; }
    __host__ __device__ static constexpr int dc(int p) { return donor(p) == 0 ? 0 : 1; }
    __host__ __device__ static constexpr int di(int p) { return p % (PER ? PER : 1); }  // local row
    __host__ __device__ static constexpr bool donated(int I, int J) { return J == NL - 1 && I < PER; }
};

// acc -= v q q^T (implicit; explicit: acc -= q q^T), y += (v + 1) q (explicit: v q).
// ONE straight-line MFMA sequence per wave ROLE (PH: the phantom wave (1, 0), see above) and
// every group (entries past the end of the row are zeroed by a select): any branch around an
// MFMA inside the loop makes the compiler keep two copies of the accumulators and shuffle them
// at the loop head -- the role is a template parameter of the whole loop instead.
template <int NT, bool EXPL, bool PH = false>
__device__ __forceinline__ void ops_consume(f32x4 (&acc)[Cfg<NT>::T], float (&yacc)[NT / 2],
                                            const Ops<NT> &o, bool live)
{
    constexpr int NL = NT / 2;
    using DN = Don<NT>;
    // `mtl = mt * vals` (implicit.rs:110-111), negated (exact); `vals += 1.0` (implicit.rs:116)
    const float va = EXPL ? -1.0f : -o.v;
    const float v1 = EXPL ? o.v : o.v + 1.0f;
    float qa[NL], na[NL], qb[NL];
#pragma unroll
    for (int I = 0; I < NL; ++I) {
        qa[I] = live ? o.qa[I] : 0.f;
        na[I] = qa[I] * va;
        yacc[I] = fmaf(qa[I], v1, yacc[I]);
    }
#pragma unroll
    for (int J = 0; J < NL; ++J) qb[J] = live ? o.qb[J] : 0.f;
    sfor<0, NL>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        sfor<0, J + 1>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            if constexpr (PH && I == J) {
                // phantom slot p = I: the donated tile (2 di + dr, 2 (NL - 1) + dc) of its donor
                if constexpr (I < DN::ND) {
                    constexpr int p = I, i = DN::di(p);
                    const float a = DN::dr(p) ? na[i] : qb[i] * va;
                    const float b = DN::dc(p) ? qa[NL - 1] : qb[NL - 1];
                    acc[lt(I, J)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[lt(I, J)], 0, 0, 0);
                }
            } else if constexpr (!PH && DN::donated(I, J)) {
                // left to the phantom wave
            } else {
                acc[lt(I, J)] = __builtin_amdgcn_mfma_f32_16x16x4f32(na[I], qb[J], acc[lt(I, J)], 0, 0, 0);
            }
        });
    });
}

// CSR entries [beg, end) of one row (or chunk) into acc / yacc.  Entries are taken in batches
// of 64 (one coalesced load of indices + values per wave), four per MFMA step.  Every load is
// unconditional (lanes / groups past the end re-read the last entry and are masked at use).
//
// The gathered factor rows come from L2 / MALL / HBM with 1-2 k cycles of latency, a step's
// MFMAs take 0.3 k (k = 128) .. 1.2 k (k = 256): with the operands of ONE step in flight (rounds
// 1-2) the phase ran at the gather latency -- 2.0-2.6 k cycles per step whatever the row length
// (LK_BLK_PHASES: 48 % of a k = 128 row, 33 % at k = 256).  LK_BLK_RING = D keeps the operands
// of D steps in flight in a register ring that runs on across batch boundaries (the next
// batch's indices / values are loaded one batch ahead); D = 0 is the old single-lookahead loop.
// Measured per ML-25M-shaped epoch (tools/blk_variants.py): k = 256: D = 0 81.8 ms, 1: 80.8,
// 2: 75.5, 4: 80.9 (spills); k = 128: D = 0 19.15, 2: 18.80, 4: 19.29, 8 at 3 waves/SIMD: 20.4
// -- at k = 128 the four resident workgroups already cover most of each other's gather latency.
#ifndef LK_BLK_SOLVE_PRIO
#define LK_BLK_SOLVE_PRIO 0
#endif
#ifndef LK_BLK_RING8
#define LK_BLK_RING8 2   // k = 128: 9 registers per step (4 spills 150 B more and gains nothing)
#endif
#ifndef LK_BLK_RING16
#define LK_BLK_RING16 2  // k = 256: 17 registers per step, the register file is full
#endif
#ifndef LK_BLK_RING16_CHUNK
#define LK_BLK_RING16_CHUNK LK_BLK_RING16  // the chunk kernel (no factorisation state to keep)
#endif
template <int NT, bool EXPL, int D = (NT == 16 ? LK_BLK_RING16 : LK_BLK_RING8), bool PH = false>
__device__ __forceinline__ void gram_accumulate(f32x4 (&acc)[Cfg<NT>::T], float (&yacc)[NT / 2],
                                                const int32_t *__restrict__ cols,
                                                const float *__restrict__ vals, int64_t beg,
                                                int64_t end, const float *__restrict__ other,
                                                int lane, int wr, int wc)
{
    const int64_t last = end - 1;  // end > beg
    if constexpr (D == 0) {
        for (int64_t base = beg; base < end; base += 64) {
            const int64_t e = (base + lane < end) ? base + lane : last;
            const int col_reg = cols[e];
            const float val_reg = vals[e];
            const int nb = (end - base) < 64 ? (int)(end - base) : 64;
            const int ng = (nb + 3) >> 2;
            Ops<NT> cur, nxt;
            ops_issue<NT>(cur, 0, col_reg, val_reg, other, lane, wr, wc);
            for (int g = 0; g < ng; ++g) {
                const int gn = (g + 1 < ng) ? g + 1 : g;
                ops_issue<NT>(nxt, gn, col_reg, val_reg, other, lane, wr, wc);
                ops_consume<NT, EXPL, PH>(acc, yacc, cur, (g * 4 + (lane >> 4)) < nb);
                cur = nxt;
            }
        }
    } else {
        static_assert(D == 0 || 16 % (D ? D : 1) == 0, "the ring depth divides the 16 steps of a batch");
        auto load_batch = [&](int64_t base, int &c, float &v) {
            const int64_t e = (base + lane < end) ? base + lane : last;
            c = cols[e];
            v = vals[e];
        };
        int colA, colB;
        float valA, valB;
        load_batch(beg, colA, valA);
        load_batch(beg + 64, colB, valB);
        Ops<NT> ring[D];
        sfor<0, D>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            ops_issue<NT>(ring[d], d, colA, valA, other, lane, wr, wc);
        });
        for (int64_t base = beg; base < end; base += 64) {
            const int nb = (end - base) < 64 ? (int)(end - base) : 64;
            const int ng = (nb + 3) >> 2;
            int colC;
            float valC;
            load_batch(base + 128, colC, valC);  // two batches ahead
            for (int g0 = 0; g0 < ng; g0 += D) {
                sfor<0, D>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    ops_consume<NT, EXPL, PH>(acc, yacc, ring[d], ((g0 + d) * 4 + (lane >> 4)) < nb);
                    // step g0 + d + D: of this batch, or already of the next one
                    const int sn = g0 + d + D;
                    const bool nx = sn >= 16;  // wave-uniform
                    ops_issue<NT>(ring[d], nx ? sn - 16 : sn, nx ? colB : colA, nx ? valB : valA,
                                  other, lane, wr, wc);
                });
            }
            colA = colB;
            valA = valB;
            colB = colC;
            valB = valC;
        }
    }
}

// acc / yacc of one chunk -> its slab ([wave][T * 4 + NL][64]); the phantom wave's donated tiles go
// to their owner's part (the owners skip those tiles), its own phantom positions are zeroed
template <int NT>
__device__ __forceinline__ void store_chunk_slab(const f32x4 (&acc)[Cfg<NT>::T],
                                                 const float (&yacc)[NT / 2], float *slab_chunk,
                                                 int wave, int lane)
{
    using C = Cfg<NT>;
    using DN = Don<NT>;
    constexpr int NL = C::NL;
    const bool ph = wave == 1;
    float *slab = slab_chunk + (size_t)wave * C::SLAB_WAVE;
    sfor<0, NL>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        sfor<0, J + 1>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = lt(I, J);
            if constexpr (I == J && I < DN::ND) {
                constexpr int p = I;
                float *own = slab_chunk + (size_t)DN::dwave(p) * C::SLAB_WAVE;
                constexpr int td = lt(DN::di(p), NL - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ph) own[(td * 4 + r) * 64 + lane] = acc[t][r];
                    slab[(t * 4 + r) * 64 + lane] = ph ? 0.f : acc[t][r];
                }
            } else if constexpr (DN::donated(I, J)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ph) slab[(t * 4 + r) * 64 + lane] = acc[t][r];  // (its own real tile)
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(t * 4 + r) * 64 + lane] = acc[t][r];
            }
        });
    });
#pragma unroll
    for (int i = 0; i < NL; ++i) slab[(C::T * 4 + i) * 64 + lane] = yacc[i];
}

// ---- chunk kernel: one workgroup per chunk of a long row -----------------------------------
template <int NT, bool EXPL>
__global__ __launch_bounds__(256) void als_blk_chunk_kernel(
    const int32_t *__restrict__ indices, const float *__restrict__ values,
    const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_len,
    const float *__restrict__ other, float *__restrict__ slabs)
{
    using C = Cfg<NT>;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int wr = wave & 1, wc = wave >> 1;
    const int64_t c = blockIdx.x;
    f32x4 acc[C::T];
    float yacc[C::NL];
#pragma unroll
    for (int t = 0; t < C::T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C::NL; ++i) yacc[i] = 0.f;
    const int64_t beg = chunk_beg[c];
    constexpr int DR = NT == 16 ? LK_BLK_RING16_CHUNK : LK_BLK_RING8;
    if (wave == 1)  // the phantom wave's role (donated tiles), see Don
        gram_accumulate<NT, EXPL, DR, true>(acc, yacc, indices, values, beg, beg + chunk_len[c],
                                            other, lane, wr, wc);
    else
        gram_accumulate<NT, EXPL, DR, false>(acc, yacc, indices, values, beg, beg + chunk_len[c],
                                             other, lane, wr, wc);
    store_chunk_slab<NT>(acc, yacc, slabs + (size_t)c * C::SLAB, wave, lane);
}

// ---- the chunk kernel at k = 256 with the gathered rows staged through LDS --------------------
//
// At cfg5 the chunked rows of the item half gather 6 x 10^7 user rows of 1 KiB out of a 10 GB
// table: no reuse (TCC hit rate 23 %), 57 GB from HBM in 36 ms = 1.6 TB/s.  als_blk_chunk_kernel
// keeps the operands of two MFMA steps per wave in flight -- 16 KiB per CU, 4 MB over the chip,
// a quarter of what 8 TB/s x 2 us of latency asks for -- and every row is requested four times
// (once per wave, L1-served).  Here a row is requested ONCE, by `global_load_lds_dwordx4` (a
// wave's instruction moves the whole 1 KiB row, no VGPR involved), into a ring of 4 stages of 16
// rows (64 KiB of LDS, two workgroups per CU): three stages = 48 KiB per workgroup are in flight
// while one is consumed, six times the bytes in flight of the register ring.  One barrier per
// stage publishes everyone's rows and frees the slot consumed before.
// The LDS side of such a load is linear in the lane, so the bank swizzle sits on the GLOBAL side:
// position p = 4 s + i of a row holds its 16-byte chunk 4 s + (i ^ (s >> 2)); the operand fetch of
// the 16 lanes of an entry (chunk 4 s + m, s = 0..15, m fixed per wave role and half) then covers
// 16 different 4-bank groups.  Same MFMA sequence per entry group as ops_consume above: the
// slabs are bit-identical to als_blk_chunk_kernel's.
#ifndef LK_BLK_CHUNK_DMA_STAGES
#define LK_BLK_CHUNK_DMA_STAGES 4
#endif
// lane l: 16 (4) bytes from `src` to LDS byte address lds + 16 l (4 l)
__device__ __forceinline__ void blk_dma16(const void *src, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_byte_addr)
        : "memory");
}
__device__ __forceinline__ void blk_dma4(const void *src, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_byte_addr)
        : "memory");
}
// LDS (dynamic): the row ring, then per wave 4 batch slots of [64 columns | 64 values]
constexpr int CHUNK_DMA_RING_FLOATS = LK_BLK_CHUNK_DMA_STAGES * 16 * 256;
constexpr int CHUNK_DMA_META_FLOATS = 4 /* waves */ * 4 /* slots */ * 128;
constexpr size_t CHUNK_DMA_LDS_BYTES = (size_t)(CHUNK_DMA_RING_FLOATS + CHUNK_DMA_META_FLOATS) * 4;

// Every memory operation of the main loop is a DMA issued by inline asm, invisible to hipcc's
// s_waitcnt insertion -- on purpose: a compiler-counted load among them would be waited for with
// a count that ignores the younger DMAs, i.e. by draining the ring.  So the column indices and
// values of a 64-entry batch also arrive by DMA (`global_load_lds_dword`, wave-private slots: no
// cross-wave ordering to think about), two batches ahead, and all vmcnt waits are explicit:
// per wave and batch the queue holds, in issue order,
//     [barrier of stage 4b]   meta(b + 2) x 2, rows(4b + 3) x 4
//     [barrier of stage 4b+1] rows(4b + 4) x 4   ... +2: rows(4b + 5), +3: rows(4b + 6)
// so "the rows of stage k have landed" is vmcnt <= 8 (the two younger stages), + 2 where the
// meta pair was issued in between (stages 4b + 1 and 4b + 2).
template <bool EXPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void als_blk_chunk_dma_kernel(
    const int32_t *__restrict__ indices, const float *__restrict__ values,
    const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_len,
    const float *__restrict__ other, float *__restrict__ slabs,
    const int32_t *__restrict__ chunk_slab, int block_len)
{
    constexpr int NT = 16;
    using C = Cfg<NT>;
    static_assert(LK_BLK_CHUNK_DMA_STAGES == 4, "the slot of a stage is its index within the 64-entry batch");
    extern __shared__ __attribute__((aligned(1024))) float chunk_dma_lds[];
    float *ring = chunk_dma_lds;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int wr = wave & 1, wc = wave >> 1;
    float *meta = chunk_dma_lds + CHUNK_DMA_RING_FLOATS + wave * 512;  // this wave's 4 slots
    const int64_t c = blockIdx.x;
    f32x4 acc[C::T];
    float yacc[C::NL];
#pragma unroll
    for (int t = 0; t < C::T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C::NL; ++i) yacc[i] = 0.f;
    const int64_t beg = chunk_beg[c];
    const int len = chunk_len[c];  // >= 1
    const int64_t end = beg + len, last = end - 1;
    const int n_stage = (len + 15) >> 4;
    // Reference-order WORK UNITS (als_plan.h, round 5): a unit of several 256-entry blocks keeps
    // the row ring running across its blocks and stores one slab per block -- the accumulators
    // are written out and start again from zero at every block boundary that is followed by more
    // entries (matrixmultiply's KC = 256 blocks, each an fma chain of its own).  The slab stores
    // count in vmcnt like the DMAs: the explicit waits below are then a little more conservative
    // than needed (never less).  block_len == 0: one slab for the whole range.
    float *slab = slabs + (size_t)(chunk_slab ? chunk_slab[c] : c) * C::SLAB;
    const int flush_stages = block_len > 0 ? block_len >> 4 : 0;  // stages per block (16)
    const unsigned ring_lds = (unsigned)(uintptr_t) reinterpret_cast<void *>(ring);
    const unsigned meta_lds = (unsigned)(uintptr_t) reinterpret_cast<void *>(meta);

    // this lane's 16 bytes of a row on the global side (see the swizzle above) ...
    const int ds_ = lane >> 2, di_ = lane & 3;
    const uint64_t lane_src =
        (uint64_t)(uintptr_t)other + (uint64_t)(16 * (4 * ds_ + (di_ ^ (ds_ >> 2))));
    // ... and the operand positions of lane (entry e = lane >> 4, element s = lane & 15)
    const int s = lane & 15, e4 = lane >> 4;
    const float *rd = ring + e4 * 256 + 16 * s;
    const int xa0 = 4 * ((2 * wr) ^ (s >> 2)), xa1 = 4 * ((2 * wr + 1) ^ (s >> 2));
    const int xb0 = 4 * ((2 * wc) ^ (s >> 2)), xb1 = 4 * ((2 * wc + 1) ^ (s >> 2));

    // columns / values of batch b (entries beg + 64 b ..; past the end: the last entry) -> slot b & 3
    auto issue_meta = [&](int b) {
        const int64_t e0 = beg + 64 * (int64_t)b + lane;
        const int64_t e = e0 < end ? e0 : last;
        const unsigned dst = meta_lds + (unsigned)(b & 3) * 512u;
        blk_dma4(indices + e, dst);
        blk_dma4(values + e, dst + 256u);
    };
    // wave w brings row 4 g + w of the stage (g = 0..3); `j` = the stage's index in batch b
    auto issue_stage = [&](int b, int j) {
        const int *cols = reinterpret_cast<const int *>(meta + (b & 3) * 128);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned col = (unsigned)__builtin_amdgcn_readfirstlane(cols[16 * j + 4 * g + wave]);
            blk_dma16(reinterpret_cast<const void *>(lane_src + (uint64_t)col * 1024ull),
                      ring_lds + (unsigned)(j * 16 + 4 * g + wave) * 1024u);
        }
    };
    auto fetch = [&](Ops<NT> &o, int b, int j, int g) {
        const float *p = rd + (j * 16 + 4 * g) * 256;
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(p + xa0);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(p + xa1);
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(p + xb0);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(p + xb1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o.qa[i] = a0[i];
            o.qa[4 + i] = a1[i];
            o.qb[i] = b0[i];
            o.qb[4 + i] = b1[i];
        }
        o.v = meta[(b & 3) * 128 + 64 + 16 * j + 4 * g + e4];
    };

    issue_meta(0);
    issue_meta(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j < n_stage) issue_stage(0, j);

    auto stages = [&](auto phc) {
    constexpr bool PH = decltype(phc)::value;
    for (int k0 = 0, b = 0; k0 < n_stage; k0 += 4, ++b) {
        sfor<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int k = k0 + j;
            if (k < n_stage) {  // workgroup-uniform
                const int ahead = n_stage - 1 - k;
                if (ahead >= 2) {
                    if (j == 1 || j == 2)
                        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else if (ahead == 1) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // everyone's rows of stage k; everyone done with stage k - 1 (whose slot the
                // next request overwrites)
                asm volatile("s_barrier" ::: "memory");
                if (j == 0) issue_meta(b + 2);  // (slot (b + 2) & 3: batch b - 2's, long consumed)
                if (k + 3 < n_stage) {
                    if (j == 0)
                        issue_stage(b, 3);
                    else
                        issue_stage(b + 1, j - 1);
                }
                const int nb = len - 16 * k;  // live entries of the stage (may exceed 16)
                Ops<NT> cur, nxt;
                fetch(cur, b, j, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g + 1 < 4) fetch(nxt, b, j, g + 1);
                    ops_consume<NT, EXPL, PH>(acc, yacc, cur, (4 * g + e4) < nb);
                    cur = nxt;
                }
                if (j == 3 && flush_stages > 0 && ((k + 1) % flush_stages) == 0 &&
                    k + 1 < n_stage) {  // (workgroup-uniform) a block is complete, more follow
                    store_chunk_slab<NT>(acc, yacc, slab, wave, lane);
                    slab += C::SLAB;
#pragma unroll
                    for (int t = 0; t < C::T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < C::NL; ++i) yacc[i] = 0.f;
                }
            }
        });
    }
    };
    if (wave == 1)  // the phantom wave's role (donated tiles), see Don
        stages(std::integral_constant<bool, true>{});
    else
        stages(std::integral_constant<bool, false>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a trailing meta request still targets our LDS)
    store_chunk_slab<NT>(acc, yacc, slab, wave, lane);
}

// LK_BLK_CHUNK_DMA=0: the register-ring chunk kernel at k = 256 too (A/B timing)
static bool chunk_dma_enabled()
{
    const char *e = getenv("LK_BLK_CHUNK_DMA");
    return !(e && e[0] == '0');
}

// ---- one step of the blocked factorisation ---------------------------------------------------
// `rot` (wave-uniform, 0..3): which wave of the workgroup plays "panel wave 0" for this row.  The
// panel rows of a step are dealt to the waves in order (vtid = panel row), so at k = 128 only two
// waves ever have rows (R <= 128) and at the late steps only one: a wave WITHOUT panel rows used
// to run the whole 16-step v_readlane chain anyway (424 instructions per step, for nothing --
// 62 % of the chain instructions at k = 128, 37 % at k = 256); it now skips it, and the roles
// rotate with the row number so that the chain-carrying waves of the resident workgroups sit on
// different SIMDs.
template <int NT, int b>
__device__ __forceinline__ void chol_step(f32x4 (&acc)[Cfg<NT>::T], float *__restrict__ lds,
                                          int tid, int lane, int wave, int wr, int wc, int rot,
                                          float &minpiv LK_BP_ARG)
{
    LK_BP_T(bp0);
    using C = Cfg<NT>;
    constexpr int NL = C::NL;
    constexpr int R = C::KP - 16 * b;  // panel rows: block row b and everything below
    constexpr int Ib = b >> 1;         // local tile row of block row b (in its owners)
    float *P = lds + ((b & 1) ? C::OFF_P1 : C::OFF_P0);
    const int sub = lane & 15, slot = lane >> 4;
    const bool phantom = wr > wc;
    const int vwave = (wave + rot) & 3;  // role in the panel phase
    const int vtid = vwave * 64 + lane;
#ifndef LK_BLK_TWIN
#define LK_BLK_TWIN 0  // 1: the diagonal block is factored by the leader wave only (see phase 2)
#endif
#ifndef LK_BLK_SKIP
#define LK_BLK_SKIP 1  // 0: every wave runs the chain (rounds 1-3; A/B timing, tools/blk_variants.py)
#endif
    const bool has_rows = !LK_BLK_SKIP || vwave * 64 < R;  // wave-uniform

    // (1) the owners of block row b publish A'(b-block rows, columns >= b) = -acc, transposed
    // by symmetry into panel rows: tile (b, tj), lane (j = sub, slot) holds
    // D[i = 4 slot + r][j] = A'[16 tj + j][16 b + 4 slot + r] -> panel row 16 (tj - b) + j,
    // columns 4 slot .. 4 slot + 3 = k-group `slot`
    if constexpr (LK_BLK_TWIN && b == 0) {  // the twins' column counter starts at 0 for every row
        if (tid == 0) *reinterpret_cast<volatile int *>(&lds[C::OFF_FLAG]) = 0;
    }
    if (wr == (b & 1)) {
        sfor<Ib, NL>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if (J > Ib || !phantom) {  // J == Ib: tile (b, b - wr + wc) exists iff wc >= wr
                const int prow = 16 * (2 * J + wc - b) + sub;
                const f32x4 v = acc[lt(Ib, J)];
                *reinterpret_cast<f32x4 *>(&P[slot * C::P_SUB + prow * 4]) =
                    f32x4{-v.x, -v.y, -v.z, -v.w};
            }
        });
    }
    __syncthreads();
    LK_BP_T(bp1);
    LK_BP_ADD(1, bp0, bp1);

    // (2) lane = panel row.  a: this thread's own row (16 columns of the block); d: diagonal
    // block row (lane & 15), replicated in every wave so that the multipliers are v_readlane
    // broadcasts and no wave waits for another.  Thread 0 carries the right-hand side instead
    // of a matrix row (rows 0..15 of the panel ARE the diagonal block: their `a` is redundant).
    float a[16], d[16];
    float myrinv = 0.f;
#ifdef LK_BLK_PHASES
    unsigned long long bp2 = bp1;
#endif
    // LK_BLK_TWIN: the waves with panel rows other than the leader (vwave 0) do not factor the
    // diagonal block again (136 v_readlane + 136 FMAs each, a third of the chain): the leader
    // publishes column j of L_bb and 1 / L_jj in LDS as it gets them and raises a counter; a
    // twin follows one column behind, its multipliers LDS broadcasts.  (A wave's LDS operations
    // are performed in order, so data before counter needs only a compiler barrier.)  Same
    // multipliers, same FMAs in the same order: bit-identical.
    constexpr bool TWINS = LK_BLK_TWIN && R > 64;  // some wave besides the leader has panel rows
    const bool twin = TWINS && vwave != 0;         // wave-uniform
    if (has_rows) {
    {
        const int prow = vtid < R ? vtid : R - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(&P[g * C::P_SUB + prow * 4]);
            a[4 * g + 0] = t.x;
            a[4 * g + 1] = t.y;
            a[4 * g + 2] = t.z;
            a[4 * g + 3] = t.w;
            if (!twin) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(&P[g * C::P_SUB + sub * 4]);
                d[4 * g + 0] = u.x;
                d[4 * g + 1] = u.y;
                d[4 * g + 2] = u.z;
                d[4 * g + 3] = u.w;
            }
        }
        if (vtid == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = lds[C::OFF_Y + 16 * b + c];
        }
    }
    if (!twin) {
        sfor<0, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float piv = bcast(d[j], j);
            minpiv = fminf(minpiv, piv);
            const float rinv = __builtin_amdgcn_rsqf(piv);
            myrinv = (sub == j) ? rinv : myrinv;
            d[j] *= rinv;  // lanes >= j: L[lane][j] (lane j: sqrt(pivot))
            if constexpr (TWINS) {
                // every row group holds the same d: four lanes store the same value
                lds[C::OFF_COL + j * 16 + sub] = d[j];
                lds[C::OFF_CRINV + j] = rinv;
                asm volatile("" ::: "memory");
                *reinterpret_cast<volatile int *>(&lds[C::OFF_FLAG]) = 16 * b + j + 1;
            }
            a[j] *= rinv;  // own row: x_j = (a_j - sum_{c<j} x_c L[j][c]) / L[j][j]
            sfor<j + 1, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const float m = bcast(d[j], c);  // L[c][j]
                d[c] = fmaf(-d[j], m, d[c]);
                a[c] = fmaf(-a[j], m, a[c]);
            });
        });
    } else {
        const unsigned flag_addr = (unsigned)(uintptr_t) reinterpret_cast<void *>(&lds[C::OFF_FLAG]);
        sfor<0, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            // wait for column j.  ONE asm statement, not a C loop: a loop in the middle of this
            // straight-line code makes the register allocator spill the accumulators around it
            // (scratch 20 -> 816 B at k = 128).  Bounded: a wait that gives up reports the row.
            int seen, sval, spins;
            asm volatile(
                "s_mov_b32 %2, 0\n"
                "1:\n\t"
                "ds_read_b32 %0, %3\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_readfirstlane_b32 %1, %0\n\t"
                "s_cmp_ge_i32 %1, %4\n\t"
                "s_cbranch_scc1 2f\n\t"
                "s_add_u32 %2, %2, 1\n\t"
                "s_cmp_lt_u32 %2, 0x10000\n\t"
                "s_cbranch_scc1 1b\n"
                "2:"
                : "=&v"(seen), "=&s"(sval), "=&s"(spins)
                : "v"(flag_addr), "s"(16 * b + j + 1)
                : "memory", "scc");
            if (spins >= 0x10000) minpiv = -1.0f;  // (never: reported as a failed solve)
            const float rinv = lds[C::OFF_CRINV + j];
            float m[16];
#pragma unroll
            for (int q = (j + 1) / 4; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(&lds[C::OFF_COL + j * 16 + 4 * q]);
                m[4 * q + 0] = t.x;
                m[4 * q + 1] = t.y;
                m[4 * q + 2] = t.z;
                m[4 * q + 3] = t.w;
            }
            a[j] *= rinv;
            sfor<j + 1, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                a[c] = fmaf(-a[j], m[c], a[c]);
            });
        });
    }

#ifdef LK_BLK_PHASES
    bp2 = __builtin_amdgcn_s_memtime();
#endif
    LK_BP_ADD(2, bp1, bp2);
    // (3) L panel rows back in place (MFMA-operand layout), diagonal block + 1/L_jj to their
    // permanent home, z_b = L_bb^-1 (y_b - ...) from thread 0
    if (vtid >= 16 && vtid < R) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(&P[g * C::P_SUB + vtid * 4]) =
                f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    }
    if (vwave == 0 && lane < 16) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(&lds[C::OFF_LD + b * 256 + lane * 16 + 4 * g]) =
                f32x4{4 * g + 0 < lane ? d[4 * g + 0] : 0.f, 4 * g + 1 < lane ? d[4 * g + 1] : 0.f,
                      4 * g + 2 < lane ? d[4 * g + 2] : 0.f, 4 * g + 3 < lane ? d[4 * g + 3] : 0.f};
        lds[C::OFF_RINV + 16 * b + lane] = myrinv;
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                lds[C::OFF_ZB + (b & 1) * 16 + c] = a[c];
                lds[C::OFF_Z + 16 * b + c] = a[c];
            }
        }
    }
    }  // has_rows
    __syncthreads();
    LK_BP_T(bp3);
    LK_BP_ADD(3, bp2, bp3);

    if constexpr (b + 1 < NT) {
        // (4) forward substitution of the rows below: y_r -= L[r][b-block] . z_b
        if (has_rows && vtid >= 16 && vtid < R) {
            float s = lds[C::OFF_Y + 16 * b + vtid];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 z =
                    *reinterpret_cast<const f32x4 *>(&lds[C::OFF_ZB + (b & 1) * 16 + 4 * g]);
                s = fmaf(-a[4 * g + 0], z.x, s);
                s = fmaf(-a[4 * g + 1], z.y, s);
                s = fmaf(-a[4 * g + 2], z.z, s);
                s = fmaf(-a[4 * g + 3], z.w, s);
            }
            lds[C::OFF_Y + 16 * b + vtid] = s;
        }
        // keep the operand loads of (5) below this point: hoisted above, they would be live
        // together with a[] and push the kernel past its 256-register budget (scratch spills
        // cost a memory round trip each -- they were half of the kernel's time)
        __builtin_amdgcn_sched_barrier(0);
        // (5) trailing update acc(ti, tj) += L(ti, b) L(tj, b)^T for ti, tj > b.  Operand for
        // MFMA step kk: lane (m = sub, kg = slot) supplies L[16 (t - b) + m][4 kg + kk] -- the
        // contraction index is enumerated as c = 4 kg + kk on BOTH operands, so one
        // ds_read_b128 per block feeds four MFMAs.
        // Branch-free: a block that is not below the panel for THIS wave gets a zero operand
        // (its MFMAs add nothing; the other waves need the same instructions at that point
        // anyway), so that every accumulator is updated in place by one instruction stream.
        f32x4 lb[NL];
        sfor<0, NL>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if constexpr (2 * J + 1 > b) {
                const bool on = 2 * J + wc > b;
                const int prow = on ? 16 * (2 * J + wc - b) + sub : sub;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&P[slot * C::P_SUB + prow * 4]);
                lb[J] = on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        });
        sfor<0, NL>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            if constexpr (2 * I + 1 > b) {
                const bool on = 2 * I + wr > b;
                const int prow = on ? 16 * (2 * I + wr - b) + sub : sub;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&P[slot * C::P_SUB + prow * 4]);
                const f32x4 la = on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                sfor<I, NL>([&](auto Jc) {
                    constexpr int J = decltype(Jc)::value;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[lt(I, J)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            la[kk], lb[J][kk], acc[lt(I, J)], 0, 0, 0);
                });
                // one A operand in flight ahead of the MFMAs that use it, not all NL of them
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // (6) the owners of block row b take their L tiles back into the registers they
        // vacated: tile (b, tj) in accumulator layout is exactly the B operand of block tj
        // (a select, not a branch: see above)
        sfor<Ib, NL>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if constexpr (2 * J + 1 > b) {
                const bool mine = (wr == (b & 1)) && (2 * J + wc > b);
                acc[lt(Ib, J)] = mine ? lb[J] : acc[lt(Ib, J)];
            }
        });
    }
    LK_BP_T(bp4);
    LK_BP_ADD(4, bp3, bp4);
}

// ---- one block of the back substitution  L^T x = z ---------------------------------------
template <int NT, int b>
__device__ __forceinline__ void back_step(const f32x4 (&acc)[Cfg<NT>::T], float *__restrict__ lds,
                                          int lane, int wave, int wr, int wc, int rot)
{
    using C = Cfg<NT>;
    constexpr int NL = C::NL;
    constexpr int Ib = b >> 1;
    const int sub = lane & 15, slot = lane >> 4;
    // s_b[c] = sum_{r > b} sum_j L[16 r + j][16 b + c] x[16 r + j]: tile (b, r) holds
    // L[16 r + sub][16 b + 4 slot + rr] in register rr
    if (wr == (b & 1)) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        sfor<Ib, NL>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if constexpr (2 * J + 1 > b) {
                if (2 * J + wc > b) {
                    const float xv = lds[C::OFF_X + 16 * (2 * J + wc) + sub];
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = fmaf(acc[lt(Ib, J)][r], xv, t[r]);
                }
            }
        });
        // 16-lane butterfly on DPP row operations (VALU, no LDS-crossbar round trip per step):
        // xor 1, xor 2, then half-mirror / mirror -- the quads (halves) hold equal sums by then,
        // so these pair the same values as xor 4 / xor 8: bit-identical to the __shfl_xor form
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = blk_row16_sum(t[r]);
        if (sub == 0)
            *reinterpret_cast<f32x4 *>(&lds[C::OFF_SP + wc * 16 + slot * 4]) =
                f32x4{t[0], t[1], t[2], t[3]};
    }
    __syncthreads();
    // diagonal block: lane = column c; x_j for j = 15 .. 0, each broadcast to the lanes c < j
    if (((wave + rot) & 3) == 0) {
        const int c = lane & 15;
        float dc = lds[C::OFF_Z + 16 * b + c] - lds[C::OFF_SP + c] - lds[C::OFF_SP + 16 + c];
        const float ri = lds[C::OFF_RINV + 16 * b + c];
        float lc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) lc[j] = lds[C::OFF_LD + b * 256 + j * 16 + c];  // L[j][c], j > c
        float xc = 0.f;
        sfor<0, 16>([&](auto jj) {
            constexpr int j = 15 - decltype(jj)::value;
            const float xj = bcast(dc * ri, j);
            xc = (c == j) ? xj : xc;
            dc = fmaf(-lc[j], xj, dc);  // strictly lower: touches the lanes c < j only
        });
        if (lane < 16) lds[C::OFF_X + 16 * b + c] = xc;
    }
    __syncthreads();
}

template <int NT, int... Bs>
__device__ __forceinline__ void chol_all(f32x4 (&acc)[Cfg<NT>::T], float *lds, int tid, int lane,
                                         int wave, int wr, int wc, int rot, float &minpiv LK_BP_ARG,
                                         std::integer_sequence<int, Bs...>)
{
    (chol_step<NT, Bs>(acc, lds, tid, lane, wave, wr, wc, rot, minpiv LK_BP_PASS), ...);
}

template <int NT, int... Bs>
__device__ __forceinline__ void back_all(const f32x4 (&acc)[Cfg<NT>::T], float *lds, int lane,
                                         int wave, int wr, int wc, int rot,
                                         std::integer_sequence<int, Bs...>)
{
    (back_step<NT, NT - 1 - Bs>(acc, lds, lane, wave, wr, wc, rot), ...);
}

#ifndef LK_ALS_BLK_ATTR16
#define LK_ALS_BLK_ATTR16 __attribute__((amdgpu_waves_per_eu(2)))
#endif
#ifndef LK_ALS_BLK_ATTR8
#define LK_ALS_BLK_ATTR8 __attribute__((amdgpu_waves_per_eu(4)))
#endif

// ---- solve kernel: one workgroup per row ------------------------------------------------------
template <int NT, bool IS64, bool EXPL, bool CTL>
__device__ __forceinline__ void als_blk_solve_body(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_rows,
    const int32_t *__restrict__ row_slab, const float *__restrict__ other,
    float *__restrict__ this_, const float *__restrict__ notor_p,
    const float *__restrict__ slabs, float *__restrict__ row_delta, int *__restrict__ status,
    int k, float reg, TaskCtlDev ctl, float *lds, const int64_t t,
    const float *__restrict__ y_ref = nullptr, int chunk_rt = 0, int64_t n_yref = 0)
{
    using C = Cfg<NT>;
    constexpr int KP = C::KP, NL = C::NL;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = lane_id();
    const int wr = wave & 1, wc = wave >> 1;
    const int sub = lane & 15, slot = lane >> 4;
    const bool phantom = wr > wc;
    if (t >= n_rows) return;
    if constexpr (CTL) {  // AccelTask.cancel: rows not started yet are skipped
        // one decision per WORKGROUP (the waves meet at barriers later: they must agree)
        if (tid == 0) lds[C::OFF_RED + 4] = ctl_cancelled(ctl, (blockIdx.x & 63) == 0) ? 1.f : 0.f;
        __syncthreads();
        const bool c = lds[C::OFF_RED + 4] != 0.f;
        __syncthreads();
        if (c) return;
    }
    const int row = order[t];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    float *xrow = this_ + (int64_t)row * KP;
    if (end == beg) {  // implicit.rs:98-101
        if (tid < KP) xrow[tid] = 0.f;
        if (tid == 0) {
            row_delta[row] = 0.f;
            if constexpr (CTL) ctl_advance(ctl, 1);
        }
        return;
    }

#ifdef LK_BLK_PHASES
    unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    LK_BP_T(bp_begin);
    // -- phase 1: -A' = -OtOr' - sum v q q^T in the accumulators ------------------------------
    f32x4 acc[C::T];
    float yacc[NL];
    sfor<0, NL>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        sfor<0, J + 1>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            int ti = 2 * I + wr, tj = 2 * J + wc;
            bool zero = I == J && phantom;
            if constexpr (I == J && I < Don<NT>::ND) {
                // phantom slot I of wave (1, 0): the tile its donor left to it starts from the
                // donor's values, so that the accumulation chain is the donor's own
                using DN = Don<NT>;
                if (phantom) {
                    ti = 2 * DN::di(I) + DN::dr(I);
                    tj = 2 * (NL - 1) + DN::dc(I);
                    zero = false;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[lt(I, J)][r] =
                    zero ? 0.f : notor_p[(ti * 16 + slot * 4 + r) * KP + tj * 16 + sub];
        });
    });
#pragma unroll
    for (int i = 0; i < NL; ++i) yacc[i] = 0.f;

    const int first_slab = row_slab[row];
    if (first_slab >= 0) {
        const int64_t n = end - beg;
        // (reference-order plans: 256-entry chunks, all slabs of the row summed one after the
        // other into its first slab -- see als_plan.h)
        const int ch = chunk_rt > 0 ? chunk_rt : LK_ALS_CHUNK_BLK;
        const int ns = (int)((n + ch - 1) / ch);
        // many chunks: the groups were pre-summed into their heads (slab_group_reduce_kernel)
        const int step = chunk_rt > 0 ? ns : (ns > LK_ALS_SLAB_GROUP ? LK_ALS_SLAB_GROUP : 1);
        for (int s = 0; s < ns; s += step) {
            const float *slab =
                slabs + (size_t)(first_slab + s) * C::SLAB + (size_t)wave * C::SLAB_WAVE;
            sfor<0, C::T>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tt][r] += slab[(tt * 4 + r) * 64 + lane];
                // at most 8 tiles' worth of loads in flight (all T at once would need 4 T
                // temporaries next to the 4 T accumulators)
                if constexpr (tt % 8 == 7) __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int i = 0; i < NL; ++i) yacc[i] += slab[(C::T * 4 + i) * 64 + lane];
        }
    } else {
        constexpr int DR = NT == 16 ? LK_BLK_RING16 : LK_BLK_RING8;
        if (phantom)  // (wave-uniform) the phantom wave's role: donated tiles, see Don
            gram_accumulate<NT, EXPL, DR, true>(acc, yacc, indices, values, beg, end, other, lane,
                                                wr, wc);
        else
            gram_accumulate<NT, EXPL, DR, false>(acc, yacc, indices, values, beg, end, other, lane,
                                                 wr, wc);
        if constexpr (Don<NT>::ND > 0) {
            // hand the donated tiles to their owners through the second panel buffer (idle until
            // block step 1 publishes into it)
            using DN = Don<NT>;
            float *hand = lds + C::OFF_P1;
            if (phantom) {
                sfor<0, DN::ND>([&](auto pc) {
                    constexpr int pp = decltype(pc)::value;
                    *reinterpret_cast<f32x4 *>(&hand[pp * 256 + lane * 4]) = acc[lt(pp, pp)];
                });
            }
            __syncthreads();
            if (!phantom) {
                const int first = (wave == 0 ? 0 : (wave == 2 ? 1 : 2)) * DN::PER;  // wave-uniform
                sfor<0, DN::PER>([&](auto dc_) {
                    constexpr int d = decltype(dc_)::value;
                    acc[lt(d, NL - 1)] =
                        *reinterpret_cast<const f32x4 *>(&hand[(first + d) * 256 + lane * 4]);
                });
            }
        }
    }
    if (EXPL) {
        // explicit.rs:104-107: A[i][i] += reg * n on the real features (acc = -A)
        const float dg = reg * (float)(end - beg);
        if (wr == wc) {
            sfor<0, NL>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                const int f = sub * NT + C::pos(2 * I + wr);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (slot * 4 + r == sub && f < k) acc[lt(I, I)][r] -= dg;
            });
        }
    }
    // y: combine the four entry slots; the waves (wr, 0) publish blocks t = 2 I + wr
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        yacc[i] += __shfl_xor(yacc[i], 16, 64);
        yacc[i] += __shfl_xor(yacc[i], 32, 64);
    }
    if (wc == 0 && slot == 0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            // (y_ref: the right-hand side in the reference's summation order, als_rhs.hip --
            // one row of KP floats per TASK t < n_yref, natural feature order, primed (t, sub)
            // <-> feature sub * NT + pos(t))
            const int tb = 2 * i + wr;
            lds[C::OFF_Y + tb * 16 + sub] =
                (y_ref && t < n_yref) ? y_ref[(int64_t)t * KP + sub * NT + C::pos(tb)] : yacc[i];
        }
    }
    // (the first barrier of chol_step<0> orders these stores before thread 0 reads them)

    // -- phase 2: blocked Cholesky with the forward substitution riding along ------------------
    float minpiv = 3.0e38f;
    LK_BP_T(bp_gram);
    LK_BP_ADD(0, bp_begin, bp_gram);
#if LK_BLK_SOLVE_PRIO
    __builtin_amdgcn_s_setprio(LK_BLK_SOLVE_PRIO);  // see als_chol.hip, LK_ALS_SOLVE_PRIO
#endif
#ifndef LK_BLK_ROTATE
#define LK_BLK_ROTATE 1
#endif
    const int rot = LK_BLK_ROTATE ? (int)(t & 3) : 0;
    chol_all<NT>(acc, lds, tid, lane, wave, wr, wc, rot, minpiv LK_BP_PASS,
                 std::make_integer_sequence<int, NT>{});
    LK_BP_T(bp_chol);

    // -- phase 3: back substitution ----------------------------------------------------------
#ifdef LK_BLK_NO_BACK  // TIMING EXPERIMENT ONLY (wrong results): what the back substitution costs
    if (tid < KP) lds[C::OFF_X + tid] = lds[C::OFF_Z + tid];
    __syncthreads();
#else
    back_all<NT>(acc, lds, lane, wave, wr, wc, rot, std::make_integer_sequence<int, NT>{});
#endif
#if LK_BLK_SOLVE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    LK_BP_T(bp_back);
    LK_BP_ADD(5, bp_chol, bp_back);

    // -- output: un-prime through LDS so that the row is written coalesced --------------------
    if (tid < KP) {
        const int f = (tid & 15) * NT + C::pos(tid >> 4);
        lds[C::OFF_Y + f] = lds[C::OFF_X + tid];
    }
    __syncthreads();
    float dd = 0.f;
    bool bad = !(minpiv > 0.f);
    if (tid < KP && tid < k) {
        const float x = lds[C::OFF_Y + tid];
        const float old = xrow[tid];
        xrow[tid] = x;
        dd = x - old;
        bad = bad || !(fabsf(x) <= 3.0e38f);
    }
    const float d2 = wave_sum(dd * dd);
    if (lane == 0) lds[C::OFF_RED + wave] = d2;
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);
    __syncthreads();
    if (tid == 0) {
        row_delta[row] = ((lds[C::OFF_RED] + lds[C::OFF_RED + 1]) + lds[C::OFF_RED + 2]) +
                         lds[C::OFF_RED + 3];
        if constexpr (CTL) ctl_advance(ctl, 1);
    }
#ifdef LK_BLK_PHASES
    if (tid == 0 && lk_blk_phase_buf) {
        LK_BP_T(bp_end);
        ph[6] = (unsigned)(end - beg);
        ph[7] = (unsigned)(bp_end - bp_begin);
        for (int i = 0; i < 8; ++i) lk_blk_phase_buf[(size_t)row * 8 + i] = ph[i];
    }
#endif
}

#define LK_BLK_KERNEL(NAME, NTV, ATTR)                                                          \
    template <bool IS64, bool EXPL, bool CTL>                                                   \
    __global__ __launch_bounds__(256) ATTR void NAME(                                           \
        const typename IndPtr<IS64>::type *__restrict__ indptr,                                 \
        const int32_t *__restrict__ indices, const float *__restrict__ values,                  \
        const int32_t *__restrict__ order, int64_t n_rows, const int32_t *__restrict__ row_slab, \
        const float *__restrict__ other, float *__restrict__ this_,                             \
        const float *__restrict__ notor_p, const float *__restrict__ slabs,                     \
        float *__restrict__ row_delta, int *__restrict__ status, int k, float reg,              \
        TaskCtlDev ctl, const float *__restrict__ y_ref, int chunk_rt, int64_t n_yref)          \
    {                                                                                           \
        __shared__ __attribute__((aligned(16))) float lds[Cfg<NTV>::LDS_FLOATS];                \
        als_blk_solve_body<NTV, IS64, EXPL, CTL>(indptr, indices, values, order, n_rows,        \
                                                 row_slab, other, this_, notor_p, slabs,        \
                                                 row_delta, status, k, reg, ctl, lds,           \
                                                 (int64_t)blockIdx.x, y_ref, chunk_rt, n_yref); \
    }

LK_BLK_KERNEL(als_blk_solve_kernel16, 16, LK_ALS_BLK_ATTR16)
LK_BLK_KERNEL(als_blk_solve_kernel8, 8, LK_ALS_BLK_ATTR8)
#undef LK_BLK_KERNEL

// Dense solve of the rows [t_begin, t_end) of the plan order IF status[1] != 0, i.e. when the
// Woodbury kernels that normally take them had to stand down because OtOr^-1 does not exist
// (spd_inverse.hip set the flag ON THE DEVICE: no host round trip decides this).  A small fixed
// grid that strides over the rows: with the flag clear -- always, unless reg = 0 meets
// rank-deficient factors -- every workgroup returns after one scalar load.
template <int NT, bool IS64>
__global__ __launch_bounds__(256) void als_blk_fallback_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t t_begin,
    int64_t t_end, const int32_t *__restrict__ row_slab, const float *__restrict__ other,
    float *__restrict__ this_, const float *__restrict__ notor_p, const float *__restrict__ slabs,
    float *__restrict__ row_delta, int *__restrict__ status, int k)
{
    __shared__ __attribute__((aligned(16))) float lds[Cfg<NT>::LDS_FLOATS];
    if (status[1] == 0) return;
    for (int64_t t = t_begin + blockIdx.x; t < t_end; t += gridDim.x) {
        als_blk_solve_body<NT, IS64, false, false>(indptr, indices, values, order, t_end, row_slab,
                                                   other, this_, notor_p, slabs, row_delta,
                                                   status, k, 0.f, TaskCtlDev{}, lds, t);
        __syncthreads();
    }
}

// ---- Woodbury row solve for rows with 65 .. 128 entries at padded k = 256 (round 4) --------------
// Same identity as als_wb.hip / als_wb64_kernel:  S u' = sv o (S0 w),  S = I + diag(sv) S0 diag(sv),
// S0 = [q_i . z_j],  x = sum_j (w_j - sv_j u'_j) z_j  -- with S up to 128 x 128: the system the
// k = 128 instance of THIS file's blocked solver factors (4 workgroups per CU, 8 block steps)
// instead of a 256 x 256 normal matrix (2 per CU, 16 steps, 512 B of scratch per thread).
// S0 on the matrix cores without any transposition: the system index IS the entry slot (tile t,
// lane & 15 = entry 16 t + sub), the contraction runs over the 256 features, 16 per super-step:
// a lane loads ONE float4 of "its" gathered row per operand tile -- features 16 G + 4 slot .. + 3 --
// and MFMA step kk contracts feature 16 G + 4 slot + kk on BOTH operands (the enumeration trick
// of the trailing update), so four instructions cover the 16 features.  (A first version wrote the
// gathered rows transposed to a 256 KiB scratch per workgroup and streamed them back: bound by
// that traffic, no faster than the dense kernel.)
constexpr int WB128_N = 128;  // system size = Cfg<8>::KP

template <bool IS64>
__global__ __launch_bounds__(256) LK_ALS_BLK_ATTR8 void als_wb128_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_tasks,
    const float *__restrict__ other, const float *__restrict__ z, float *__restrict__ this_,
    float *__restrict__ row_delta, int *__restrict__ status)
{
    using C = Cfg<8>;
    constexpr int KF = 256, NL = 4;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ int s_col[WB128_N];
    __shared__ float s_w[WB128_N], s_sv[WB128_N], s_g[WB128_N], s_t[KF];
    if (status[1] != 0) return;  // Z unavailable (OtOr not positive definite): dense fallback
    const int64_t task = blockIdx.x;
    if (task >= n_tasks) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = lane_id();
    const int wr = wave & 1, wc = wave >> 1;
    const int sub = lane & 15, slot = lane >> 4;
    const bool phantom = wr > wc;
    const int row = order[task];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const int n = (int)(end - beg);  // 65 .. 128 (any 1 .. 128 is handled)
    float *xrow = this_ + (int64_t)row * KF;
    if (tid < WB128_N) {
        const int64_t e = beg + (tid < n ? tid : n - 1);
        const float v = tid < n ? values[e] : 0.f;
        s_col[tid] = indices[e];
        s_w[tid] = tid < n ? v + 1.0f : 0.f;  // `vals += 1.0` (implicit.rs:116)
        s_sv[tid] = __builtin_sqrtf(v);       // v < 0: NaN -> reported as not positive definite
    }
    __syncthreads();
    // ---- t = sum_j w_j z_j (thread = feature), then the right-hand side sv_i (q_i . t) -------------
    {
        float tf = 0.f;
        for (int j = 0; j < n; ++j) tf = fmaf(s_w[j], z[(int64_t)s_col[j] * KF + tid], tf);
        s_t[tid] = tf;
    }
    __syncthreads();
    for (int i = wave; i < WB128_N; i += 4) {
        float r = 0.f;
        if (i < n) {
            const f32x4 q4 =
                *reinterpret_cast<const f32x4 *>(other + (int64_t)s_col[i] * KF + lane * 4);
            r = q4.x * s_t[lane * 4] + q4.y * s_t[lane * 4 + 1] + q4.z * s_t[lane * 4 + 2] +
                q4.w * s_t[lane * 4 + 3];
            r = wave_sum(r);
        }
        if (lane == 0) lds[C::OFF_Y + i] = i < n ? s_sv[i] * r : 0.f;  // system index = entry slot
    }
    // ---- -S = -I - (sv q)(sv z)^T on the accumulators (upper tiles, 2 x 2 block-cyclic) ------------
    f32x4 acc[C::T];
    sfor<0, NL>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        sfor<0, J + 1>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            const bool diag = (I == J) && (wr == wc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[lt(I, J)][r] = (diag && (slot * 4 + r) == sub) ? -1.0f : 0.f;
        });
    });
    {
        // this lane's operand rows: A tiles ti = 2 I + wr, B tiles tj = 2 J + wc, entry 16 t + sub
        const float *arow[NL], *brow[NL];
        float asv[NL], bsv[NL];
#pragma unroll
        for (int I = 0; I < NL; ++I) {
            const int ea = 16 * (2 * I + wr) + sub, eb = 16 * (2 * I + wc) + sub;
            arow[I] = other + (int64_t)s_col[ea] * KF + 4 * slot;
            brow[I] = z + (int64_t)s_col[eb] * KF + 4 * slot;
            asv[I] = -s_sv[ea];  // (negated: the accumulators hold -S)
            bsv[I] = s_sv[eb];
        }
        f32x4 qa[2][NL], qb[2][NL];
        auto issue = [&](int buf, int G) {
#pragma unroll
            for (int I = 0; I < NL; ++I) {
                qa[buf][I] = *reinterpret_cast<const f32x4 *>(arow[I] + 16 * G);
                qb[buf][I] = *reinterpret_cast<const f32x4 *>(brow[I] + 16 * G);
            }
        };
        auto consume = [&](int buf) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float a[NL], b[NL];
#pragma unroll
                for (int I = 0; I < NL; ++I) {
                    a[I] = asv[I] * qa[buf][I][kk];
                    b[I] = bsv[I] * qb[buf][I][kk];
                }
                sfor<0, NL>([&](auto Jc) {
                    constexpr int J = decltype(Jc)::value;
                    sfor<0, J + 1>([&](auto Ic) {
                        constexpr int I = decltype(Ic)::value;
                        acc[lt(I, J)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I], b[J],
                                                                             acc[lt(I, J)], 0, 0, 0);
                    });
                });
            }
        };
        issue(0, 0);
        for (int G = 0; G < KF / 16; G += 2) {
            issue(1, G + 1);
            consume(0);
            issue(0, G + 2 < KF / 16 ? G + 2 : G + 1);
            consume(1);
        }
    }
    if (phantom) {  // wave (1, 0): its diagonal local tiles are lower tiles, never read
        sfor<0, NL>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            acc[lt(I, I)] = f32x4{0.f, 0.f, 0.f, 0.f};
        });
    }
    // ---- the 128 x 128 system through this file's blocked solver -----------------------------------
    float minpiv = 3.0e38f;
#ifdef LK_BLK_PHASES
    unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int rot = (int)(task & 3);
    chol_all<8>(acc, lds, tid, lane, wave, wr, wc, rot, minpiv LK_BP_PASS,
                std::make_integer_sequence<int, 8>{});
    back_all<8>(acc, lds, lane, wave, wr, wc, rot, std::make_integer_sequence<int, 8>{});
    // g_j = w_j - sv_j u'_j
    bool bad = !(minpiv > 0.f);
    if (tid < WB128_N) {
        const float g = s_w[tid] - s_sv[tid] * lds[C::OFF_X + tid];
        s_g[tid] = tid < n ? g : 0.f;
        bad = bad || !(fabsf(g) <= 3.0e38f);
    }
    __syncthreads();
    // ---- x = sum_j g_j z_j (thread = feature) ---------------------------------------------------------
    float x = 0.f;
    for (int j = 0; j < n; ++j) x = fmaf(s_g[j], z[(int64_t)s_col[j] * KF + tid], x);
    const float old = xrow[tid];
    xrow[tid] = x;
    const float d = x - old;
    const float d2 = wave_sum(d * d);
    if (lane == 0) lds[C::OFF_RED + wave] = d2;
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);
    __syncthreads();
    if (tid == 0)
        row_delta[row] = ((lds[C::OFF_RED] + lds[C::OFF_RED + 1]) + lds[C::OFF_RED + 2]) +
                         lds[C::OFF_RED + 3];
}

// -OtOr [k x k] -> primed [KP x KP] of this file's feature order, -1 on the pad diagonal
template <int NT>
__global__ void als_blk_prep_otor_kernel(const float *__restrict__ otor, int ld_otor, int k,
                                         float *__restrict__ notor_p)
{
    using C = Cfg<NT>;
    constexpr int KP = C::KP;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KP * KP) return;
    const int pr = idx / KP, pc = idx % KP;
    const int fr = (pr & 15) * NT + C::pos(pr >> 4), fc = (pc & 15) * NT + C::pos(pc >> 4);
    float v;
    if (fr < k && fc < k)
        v = otor ? -otor[fr * ld_otor + fc] : 0.f;  // explicit mode: no OtOr term
    else
        v = (pr == pc) ? -1.0f : 0.0f;
    notor_p[idx] = v;
}

// LK_ALS_WB4=0: rows with <= 4 entries take the wave-per-row kernel too (A/B timing, tests)
static bool als_wb8_enabled()
{
    const char *e = getenv("LK_ALS_WB8");
    return !(e && e[0] == '0');
}
static bool als_wb4_enabled()
{
    const char *e = getenv("LK_ALS_WB4");
    return !(e && e[0] == '0');
}

// LK_ALS_WB128=0: rows with 65 .. 128 entries stay on the dense kernel (A/B timing, tests)
static bool als_wb128_enabled()
{
    const char *e = getenv("LK_ALS_WB128");
    return !(e && e[0] == '0');
}

// LK_ALS_SIDE_STREAM=0: OtOr^-1 on the launch stream (A/B timing)
static bool side_stream_enabled()
{
    const char *e = getenv("LK_ALS_SIDE_STREAM");
    return !(e && e[0] == '0');
}

// LK_ALS_WB64=0: rows with 17 .. 64 entries stay on the dense kernel (A/B timing, tests)
static bool als_wb64_enabled()
{
    const char *e = getenv("LK_ALS_WB64");
    return !(e && e[0] == '0');
}

template <int NT, bool IS64, bool EXPL>
static int launch_blk(const lk_als_plan *p, const void *indptr, const int32_t *indices,
                      const float *values, int64_t n_rows, int64_t n_cols, int k, float *this_,
                      const float *other, const float *otor, int ld_otor, char *ws,
                      float *out_frob, hipStream_t st, float reg)
{
    using C = Cfg<NT>;
    using IT = typename IndPtr<IS64>::type;
    int *status = reinterpret_cast<int *>(ws + p->off_status);
    float *notor_p = reinterpret_cast<float *>(ws + p->off_otor);
    float *row_delta = reinterpret_cast<float *>(ws + p->off_delta);
    float *partial = reinterpret_cast<float *>(ws + p->off_partial);
    float *slabs = reinterpret_cast<float *>(ws + p->off_slabs);

    LK_REQUIRE(!p->ref_order || (p->d_yref && !p->ctl),
               "a reference-order ALS plan needs its rhs workspace (lk_als_plan_set_rhs_workspace) "
               "and no task-control block");
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 64, st));
    if (p->ctl) {
        LK_HIP_CHECK(hipMemsetAsync(row_delta, 0, (size_t)n_rows * sizeof(float), st));
        int rc = ctl_begin(p->ctl, n_rows, n_rows, st);
        if (rc != LK_OK) return rc;
    }
    hipLaunchKernelGGL(als_blk_prep_otor_kernel<NT>, dim3((C::KP * C::KP + 255) / 256), dim3(256),
                       0, st, otor, ld_otor, k, notor_p);
    // first task of the Woodbury kernels: rows <= 16 entries always; 17 .. 64 at padded k = 256
    // (17 .. 32 / 64 at k = 128 with LK_ALS_WB64_K128 = 32 / 64; LK_ALS_WB64=0: none)
    int64_t n_wb64_first = p->t_short;
    if (als_wb64_enabled()) {
        if (NT == 16) {
            n_wb64_first = p->t_mid;
        } else {
            const int lim = wb64_k128_limit();
            n_wb64_first = lim >= 64 ? p->t_mid : (lim >= 32 ? p->t_32 : p->t_short);
        }
    }
    const bool prefix = p->dense_limit >= 0;  // CG hybrid: the chunked rows only, no Woodbury
    const bool own_z = !EXPL && p->d_zbuf != nullptr && !p->ctl &&
                       (n_wb64_first < n_rows || p->z_for_others) && n_cols > 0 && !prefix;
    // OtOr^-1 to float64 accuracy (spd_inverse.hip; status[1] = its flag, tested by the Woodbury
    // kernels and by the fallback launch below): one workgroup for most of its time, so it goes
    // to the plan's side stream, under the chunk kernel, and the Z GEMM waits for it
    bool inv_on_side = false;
    if (own_z) {
        float *ginv = reinterpret_cast<float *>(ws + p->off_ginv);
        hipStream_t sv = st;
        if (side_stream_enabled() && p->n_chunks > 0) {
            if (!p->side) {
                p->side = lk::side_stream_acquire();
                LK_REQUIRE(p->side != nullptr, "als: no side stream");
                LK_HIP_CHECK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
                LK_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
            }
            LK_HIP_CHECK(hipEventRecord(p->ev_fork, st));  // (after the status memset above)
            LK_HIP_CHECK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
            sv = p->side;
            inv_on_side = true;
        }
        int rc = spd_inverse(otor, ld_otor, k, C::KP, ginv, status + 1, ws + p->off_invws, sv);
        if (rc != LK_OK) return rc;
        if (inv_on_side) LK_HIP_CHECK(hipEventRecord(p->ev_join, p->side));
    }
    const bool tm = p->timing && p->timing_n < lk_als_plan::TIMING_RING;
    if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][0], st));
    // rows whose right-hand side comes from the reference-order chain (als_rhs.hip): the long
    // rows (the first n_long tasks) of a hybrid plan, every dense row of a strict one.  The
    // chains run on the plan's second stream under the chunk and Woodbury kernels.
    float *yref = p->hybrid ? reinterpret_cast<float *>(ws + p->off_yref)
                            : (p->ctl ? nullptr : p->d_yref);
    const int ref_chunk = (p->ref_order || p->hybrid) ? p->chunk : 0;
    int64_t n_yhyb = 0;  // hybrid plans: the long rows' chains, forked here
    if (yref && p->hybrid) {
        n_yhyb = p->n_long;
        if (p->dense_limit >= 0 && p->dense_limit < n_yhyb) n_yhyb = p->dense_limit;
        if (n_yhyb > 0) {
            hipStream_t sr = st;
            int rc = plan_fork_rhs(p, st, &sr);
            if (rc != LK_OK) return rc;
            rc = launch_rhs_reference(p, indptr, IS64 ? 1 : 0, indices, values, p->d_order,
                                      n_yhyb, other, EXPL, yref, sr);
            if (rc != LK_OK) return rc;
        }
    }
    if (p->n_chunks > 0) {
        bool dma = false;
        if constexpr (NT == 16) dma = chunk_dma_enabled();
        if (dma) {
            if constexpr (NT == 16) {  // gathered rows staged through LDS (72 KiB, dynamic)
                static PerDeviceOnce attr_once;
                bool &attr_set = attr_once.flag();
                if (!attr_set) {
                    LK_HIP_CHECK(hipFuncSetAttribute(
                        reinterpret_cast<const void *>(&als_blk_chunk_dma_kernel<EXPL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHUNK_DMA_LDS_BYTES));
                    attr_set = true;
                }
                hipLaunchKernelGGL((als_blk_chunk_dma_kernel<EXPL>), dim3((unsigned)p->n_chunks),
                                   dim3(256), CHUNK_DMA_LDS_BYTES, st, indices, values,
                                   p->d_chunk_beg, p->d_chunk_len, other, slabs, p->d_chunk_slab,
                                   p->unit > p->chunk ? (int)p->chunk : 0);
            }
        } else {
            hipLaunchKernelGGL((als_blk_chunk_kernel<NT, EXPL>), dim3((unsigned)p->n_chunks),
                               dim3(256), 0, st, indices, values, p->d_chunk_beg, p->d_chunk_len,
                               other, slabs);
        }
    }
    {
        // hybrid plans: the ordered slab sums (pure HBM streaming) go to the chains' stream and
        // run under the Woodbury kernels; the dense launch joins that stream anyway
        // (LK_ALS_REDUCE_SIDE=0: launch stream)
        hipStream_t sg = st;
        const char *e = getenv("LK_ALS_REDUCE_SIDE");
        if (p->hybrid && n_yhyb > 0 && p->n_groups > 0 && p->side_rhs && !(e && e[0] == '0')) {
            const char *s = getenv("LK_ALS_SIDE_STREAM");
            if (!(s && s[0] == '0')) {
                int rc = plan_rhs_wait_main(p, st);
                if (rc != LK_OK) return rc;
                sg = p->side_rhs;
            }
        }
        int rc = launch_slab_group_reduce(p, slabs, (size_t)C::SLAB, sg);
        if (rc != LK_OK) return rc;
    }
    if (tm) LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][1], st));
    // short rows (<= 16 entries) of the implicit model: Woodbury kernel, when the caller
    // supplied Z = other * OtOr^-1 for this half-epoch (never with a task-control block: the
    // kernel does not poll it)
    const float *z = p->d_z;
    if (own_z) {
        // Z = other * OtOr^-1 for this half-epoch: one scoring GEMM (k-ordered f32 MFMA, topk.hip)
        float *ginv = reinterpret_cast<float *>(ws + p->off_ginv);
        if (inv_on_side) LK_HIP_CHECK(hipStreamWaitEvent(st, p->ev_join, 0));
        int rc = lk_score_dense(other, C::KP, n_cols, ginv, C::KP, C::KP, k, p->d_zbuf, C::KP, st);
        if (rc != LK_OK) return rc;
        z = p->d_zbuf;
    }
    if (!EXPL && p->d_zflag_src && !own_z)  // Z (and its validity) come from the leading slice
        LK_HIP_CHECK(hipMemcpyAsync(status + 1, p->d_zflag_src, sizeof(int),
                                    hipMemcpyDeviceToDevice, st));
    const bool shared_z = !EXPL && p->d_zflag_src != nullptr && !own_z;
    const bool use_wb = !EXPL && z != nullptr && !p->ctl && n_wb64_first < n_rows && !prefix;
    // (17 .. 64 entries: only at padded k = 256 -- at k = 128 the 64 x 64 system costs as much as
    // the dense solve of this file, measured on the ML-25M shape)
    // (65 .. 128 entries at padded k = 256: the same identity with a 128 x 128 system,
    // als_wb128_kernel; LK_ALS_WB128=0 keeps those rows on the dense kernel)
    const bool wb128 = NT == 16 && als_wb64_enabled() && als_wb128_enabled() && use_wb &&
                       p->t_128 < p->t_mid;
    const int64_t n_dense =
        prefix ? (p->dense_limit < n_rows ? p->dense_limit : n_rows)
               : (use_wb ? (wb128 ? p->t_128 : n_wb64_first) : n_rows);
    if (use_wb) {
        // <= 4 entries (and empty rows): four rows per wave; 5 .. 16: a wave per row
        // (5 .. 8 entries: two rows per wave, LK_ALS_WB8=0: a wave per row)
        const int64_t t4 = als_wb4_enabled() ? p->t_4 : n_rows;
        const int64_t t8 = als_wb8_enabled() ? p->t_8 : t4;
        int rc = als_wb_launch(p, indptr, IS64 ? 1 : 0, indices, values, p->t_short, t8, this_,
                               other, z, row_delta, status, st);
        if (rc != LK_OK) return rc;
        rc = als_wb4_launch(p, indptr, IS64 ? 1 : 0, indices, values, t8, t4, this_, other, z,
                            row_delta, status, st, 8);
        if (rc != LK_OK) return rc;
        rc = als_wb4_launch(p, indptr, IS64 ? 1 : 0, indices, values, t4, n_rows, this_, other,
                            z, row_delta, status, st, 4);
        if (rc != LK_OK) return rc;
        // rows with 17 .. 64 entries: the same identity with a 64 x 64 system
        rc = als_wb64_launch(p, indptr, IS64 ? 1 : 0, indices, values, n_wb64_first, p->t_short,
                             this_, other, z, row_delta, status, st);
        if (rc != LK_OK) return rc;
        if (wb128) {
            const int64_t nr = p->t_mid - p->t_128;
            hipLaunchKernelGGL((als_wb128_kernel<IS64>), dim3((unsigned)nr), dim3(256), 0, st,
                               static_cast<const IT *>(indptr), indices, values,
                               p->d_order + p->t_128, nr, other, z, this_, row_delta, status);
        }
        if (own_z || shared_z)  // no-op unless spd_inverse raised its flag
            hipLaunchKernelGGL((als_blk_fallback_kernel<NT, IS64>), dim3(1024), dim3(256), 0, st,
                               static_cast<const IT *>(indptr), indices, values, p->d_order,
                               n_dense, n_rows, p->d_row_slab, other, this_, notor_p, slabs,
                               row_delta, status, k);
    }
    if (n_dense > 0) {
        int64_t n_ytasks = 0;  // tasks [0, n_ytasks) take y from the chains
        if (p->hybrid) {
            if (n_yhyb > 0) {
                int rc = plan_join_rhs(p, st);
                if (rc != LK_OK) return rc;
            }
            n_ytasks = std::min<int64_t>(n_yhyb, n_dense);  // (long rows are always dense rows)
        } else if (yref) {  // strict plans: every dense row, in order on the launch stream
            int rc = launch_rhs_reference(p, indptr, IS64 ? 1 : 0, indices, values, p->d_order,
                                          n_dense, other, EXPL, yref, st);
            if (rc != LK_OK) return rc;
            n_ytasks = n_dense;
        }
        const dim3 grid((unsigned)n_dense), block(256);
        const IT *ip = static_cast<const IT *>(indptr);
#define LK_BLK_LAUNCH(KERN, CTLV)                                                                \
    hipLaunchKernelGGL((KERN<IS64, EXPL, CTLV>), grid, block, 0, st, ip, indices, values,        \
                       p->d_order, n_dense, p->d_row_slab, other, this_, notor_p, slabs,         \
                       row_delta, status, k, reg, (CTLV) ? p->ctl->dev() : TaskCtlDev{},        \
                       yref, ref_chunk, n_ytasks)
        if constexpr (NT == 16) {
            if (p->ctl)
                LK_BLK_LAUNCH(als_blk_solve_kernel16, true);
            else
                LK_BLK_LAUNCH(als_blk_solve_kernel16, false);
        } else {
            if (p->ctl)
                LK_BLK_LAUNCH(als_blk_solve_kernel8, true);
            else
                LK_BLK_LAUNCH(als_blk_solve_kernel8, false);
        }
#undef LK_BLK_LAUNCH
    }
    if (tm) {
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][2], st));
        p->timing_n++;
    }
    int rc = launch_delta_reduce(row_delta, n_rows, partial, out_frob, st);
    if (rc != LK_OK) return rc;
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace blk

size_t als_blk_slab_floats(int NT)
{
    return NT == 16 ? (size_t)blk::Cfg<16>::SLAB : (size_t)blk::Cfg<8>::SLAB;
}

// Exact half-epoch for KP = 128 / 256 (dispatch target of lk_als_implicit_half_epoch /
// lk_als_explicit_half_epoch); `otor` null = explicit model.
int als_blk_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                       const float *values, int64_t n_rows, int64_t n_cols, int k, float *this_,
                       const float *other, const float *otor, int ld_otor, char *ws,
                       float *out_frob, hipStream_t st, bool expl, float reg)
{
#define LK_BLK_ARGS p, indptr, indices, values, n_rows, n_cols, k, this_, other, otor, ld_otor, ws, out_frob, st, reg
#define LK_BLK_CASE(NTV)                                                                      \
    do {                                                                                      \
        if (expl)                                                                             \
            return is64 ? blk::launch_blk<NTV, true, true>(LK_BLK_ARGS)                       \
                        : blk::launch_blk<NTV, false, true>(LK_BLK_ARGS);                     \
        return is64 ? blk::launch_blk<NTV, true, false>(LK_BLK_ARGS)                          \
                    : blk::launch_blk<NTV, false, false>(LK_BLK_ARGS);                        \
    } while (0)
    if (p->KP == 256) LK_BLK_CASE(16);
    if (p->KP == 128) LK_BLK_CASE(8);
#undef LK_BLK_ARGS
#undef LK_BLK_CASE
    set_error("blocked Cholesky: unsupported padded embedding size %d", p->KP);
    return LK_E_INVALID;
}

}  // namespace lk

#ifdef LK_BLK_PHASES
extern "C" int lk_blk_phase_set(unsigned *d_buf)
{
    LK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(lk::blk::lk_blk_phase_buf), &d_buf, sizeof(d_buf)));
    return LK_OK;
}
#endif
