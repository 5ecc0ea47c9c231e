// amx_solver.hpp -- one-wavefront-per-voxel non-negative (elastic-net) least squares for gfx950.
//
// Replaces the per-voxel calls cyspams.interfaces.nnls / .lasso of the reference
// (amico/models.pyx:911, 926, 940, 1238, 1569) by ONE device routine:
//
//     min_x  1/2 || y - A diag(s) x ||^2 + lambda1 * sum(x) + lambda2/2 * ||x||^2 ,  x >= 0
//
// solved to the KKT point by an active-set method in A-space (never on the Gram matrix:
// the AMICO dictionaries are numerically rank deficient, cond(A) ~ 1e20, passive sets
// reach cond ~ 1e6..1e8, so normal equations lose the support decisions the reference's
// QR-based Lawson-Hanson solver gets right -- see DESIGN.md "Why not Gram space").
//
// Mapping onto a 64-lane wavefront
//   * dictionary slice A (nS x n_atoms) of the voxel's orientation: staged once per
//     workgroup in LDS, row-major with an ODD leading dimension (conflict-free both for
//     the row sweep of the gradient and for column gathers);
//   * "row space"  : lane l owns signal rows l, l+64, ...   (NR rows per lane)
//   * "atom space" : lane l owns atoms      l, l+64, ...   (NQ atoms per lane)
//   * "slot space" : lane s owns the s-th passive atom (its coefficient, one row of the
//     triangular factor R, one augmented ridge row of Q);
//   * thin QR of the passive columns kept in REGISTERS: Q[k][NR] (row space), built by
//     blocked classical Gram-Schmidt with re-orthogonalisation, down-dated by Givens
//     rotations when an atom leaves;
//   * gradient w = s * A'(y - A s x) - lambda1: one LDS sweep of A per outer iteration,
//     fp32 storage, fp64 accumulation; arg-max / sums by DPP wavefront reductions;
//   * ridge term: the sqrt(lambda2) identity rows of the augmented system only touch
//     passive atoms, so they live in slot space (Qa); the l1 term needs e = R^-T 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amx {

constexpr int kWave = 64;

// -DAMX_PHASES: wall-clock split of solve() by phase (s_memtime), accumulated per wavefront (diagnosis builds only)
#ifdef AMX_PHASES
#define AMX_PH(k) do { const long long t__ = (long long)__builtin_readcyclecounter(); ph[k] += t__ - pht; pht = t__; } while (0)
#else
#define AMX_PH(k) do { } while (0)
#endif
enum SolveStatus : int { kSolved = 0, kOverflow = 1, kIterCap = 2, kGuardSelect = 3, kGuardOuter = 4 };
// inputs of the dual-value screening in certify_seed (all of one voxel / orientation; Sf == nullptr: exact sweep instead)
struct SeedScreen {
    const float *Sf = nullptr;        // LDS: float32 compressed dictionary [KD][ld]
    int ld = 0;
    double kappa = 0.0;               // |a_j'r - s_j'(U'r)| <= kappa ||r|| for every atom of the orientation
    const double *ytil = nullptr;     // global: U'y of the voxel [KD]
    const double *Sg = nullptr;       // global: fp64 compressed dictionary of the orientation [n_atoms][KD]
    int *count = nullptr;             // statistics (AMX_STATS): atoms that needed the exact dot product
};
// support seed of a voxel (amx_seed.hpp): up to 8 atom ids, one per byte from the low end; bytes >= 0xf0 are empty
constexpr unsigned long long kSeedNone = ~0ull;

// ------------------------------------------------------------------ wavefront primitives
__device__ __forceinline__ double bcast(double v, int l)   // l must be wave-uniform
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ double dpp_move(double v, double ident)
{
    // lanes whose source is invalid or masked off keep `ident`
    int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWM, BANKM, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWM, BANKM, false);
    return __hiloint2double(hi, lo);
}
// DPP controls (GFX9 family): quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, row_shr:n = 0x110+n.
template <int CTRL>
__device__ __forceinline__ double dpp_zero(double v)       // source lanes that do not exist read as 0
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// The two row-swap steps of gfx950 (v_permlane16_swap / v_permlane32_swap exchange whole 16-lane rows between two
// registers; probed: tools/probes/permlane_probe.hip): a swap of a value with itself returns (rows 0,0,2,2) and
// (rows 1,1,3,3), resp. (lanes 0-31 twice) and (lanes 32-63 twice).
#define AMX_ROW_SWAP(BUILTIN, k, A, B)                                                                             \
    do {                                                                                                           \
        const auto lo_ = BUILTIN((unsigned)__double2loint(k), (unsigned)__double2loint(k), false, false);          \
        const auto hi_ = BUILTIN((unsigned)__double2hiint(k), (unsigned)__double2hiint(k), false, false);          \
        A = __hiloint2double((int)hi_[0], (int)lo_[0]); B = __hiloint2double((int)hi_[1], (int)lo_[1]);            \
    } while (0)
// Sum of the four 16-lane rows, lane by lane, left in every row (no LDS round trip as with ds_bpermute).
__device__ __forceinline__ double rows_allreduce(double k)
{
    double a, b;
    AMX_ROW_SWAP(__builtin_amdgcn_permlane16_swap, k, a, b); k = a + b;
    AMX_ROW_SWAP(__builtin_amdgcn_permlane32_swap, k, a, b); k = a + b;
    return k;
}
// Butterfly all-reductions: every lane ends with the result (xor 1, xor 2 by quad_perm; the mirrors act as xor 4 and
// xor 8 on values that are already equal within groups of 4 / 8 lanes; then the two row swaps).  All source lanes
// exist, so no identity has to be prepared: 6 levels of 2 moves + 1 op, no readlane.
__device__ __forceinline__ double wave_sum(double v)
{
    double t = v + dpp_zero<0xB1>(v);
    t += dpp_zero<0x4E>(t);
    t += dpp_zero<0x141>(t);
    t += dpp_zero<0x140>(t);
    return rows_allreduce(t);
}
__device__ __forceinline__ double wave_max(double v)
{
    double t = fmax(v, dpp_zero<0xB1>(v));
    t = fmax(t, dpp_zero<0x4E>(t));
    t = fmax(t, dpp_zero<0x141>(t));
    t = fmax(t, dpp_zero<0x140>(t));
    double a, b;
    AMX_ROW_SWAP(__builtin_amdgcn_permlane16_swap, t, a, b); t = fmax(a, b);
    AMX_ROW_SWAP(__builtin_amdgcn_permlane32_swap, t, a, b); t = fmax(a, b);
    return t;
}
__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }
// maximum over the four 16-lane rows, lane by lane, left in every row
__device__ __forceinline__ double rows_allmax(double k)
{
    double a, b;
    AMX_ROW_SWAP(__builtin_amdgcn_permlane16_swap, k, a, b); k = fmax(a, b);
    AMX_ROW_SWAP(__builtin_amdgcn_permlane32_swap, k, a, b); k = fmax(a, b);
    return k;
}

// Four wavefront sums at once (all lanes receive all four totals).  Instead of four 7-step
// reductions the values are "transposed" while they are summed: after the two quad_perm
// exchanges every lane owns ONE of the four partial sums (over its quad), which is then reduced
// across the 16 quads.  ~43 VALU instead of ~100.
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sum4(double (&p)[4], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
    double k0 = o1 ? p[2] : p[0], k1 = o1 ? p[3] : p[1];          // kept pair
    const double s0 = o1 ? p[0] : p[2], s1 = o1 ? p[1] : p[3];    // pair handed to lane ^ 1
    k0 += dpp_zero<0xB1>(s0);                                      // quad_perm:[1,0,3,2]
    k1 += dpp_zero<0xB1>(s1);
    double k = o2 ? k1 : k0;
    const double s = o2 ? k0 : k1;                                 // handed to lane ^ 2
    k += dpp_zero<0x4E>(s);                                        // quad_perm:[2,3,0,1]
    k += dpp_zero<0x114>(k);                                       // across the quads of a row
    k += dpp_zero<0x118>(k);
    k = rows_allreduce(k);                                         // across the four rows
    // lanes 12..15 hold the totals of value 2*(lane&1) + ((lane>>1)&1)
    p[0] = bcast(k, 12); p[2] = bcast(k, 13); p[1] = bcast(k, 14); p[3] = bcast(k, 15);
}

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __ballot(p); }
// 1/sqrt(a) to full double precision: hardware estimate + two Newton steps (a > 0, normal range)
__device__ __forceinline__ double inv_sqrt(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    y = y * (1.5 - 0.5 * a * y * y);
    y = y * (1.5 - 0.5 * a * y * y);
    return y;
}
// wave-uniform predicate as a scalar (SGPR) value: makes the branch on it a scalar branch
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// value of lane+1 (used to compact slot space after an atom left)
// (DPP wave_shl:1 -- a GFX9-family control -- instead of __shfl_down's ds_bpermute round trip; lane 63 reads 0)
__device__ __forceinline__ int from_next_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ double from_next_lane(double v)
{
    return __hiloint2double(from_next_lane(__double2hiint(v)), from_next_lane(__double2loint(v)));
}

// ------------------------------------------------------------------ the dictionary tile: in LDS, or -- gtile<T> -- where it lies
// A tile too large for a CU's LDS (an HCP-style protocol: 288 x 145 float32 = 167 KB) is read from HBM / L2 (amx_kernels.hpp: GT).
// The element TYPE says so: gtile<float> converts like a float, and the one place where it matters -- the row sweep u = A'v, a chain
// of L2 round trips if it loads two rows at a time as the LDS loop does -- dispatches on it: kSweepBatch rows' loads in flight together.
template <typename T> struct gtile {
    T v;
    __device__ __forceinline__ operator double() const { return (double)v; }
};
template <typename AT> struct is_global_tile { static constexpr bool value = false; };
template <typename T> struct is_global_tile<gtile<T>> { static constexpr bool value = true; };
constexpr int kSweepBatch = 8;

// u += (even rows), w2 += (odd rows) of A'v for the atoms lane + 64 q; ap = As + lane, v in the per-wave scratch rs
template <int NQ, typename AT>
__device__ __forceinline__ void tile_sweep(const AT *ap, int ldA, int nS, const double *rs, double (&u)[NQ], double (&w2)[NQ])
{
    int i = 0;
    if constexpr (is_global_tile<AT>::value) {
        for (; i + kSweepBatch <= nS; i += kSweepBatch) {
            AT av[kSweepBatch][NQ];
#pragma unroll
            for (int b = 0; b < kSweepBatch; b++) {
#pragma unroll
                for (int q = 0; q < NQ; q++) av[b][q] = ap[(i + b) * ldA + kWave * q];
            }
#pragma unroll
            for (int b = 0; b < kSweepBatch; b += 2) {
                const double r0 = rs[i + b], r1 = rs[i + b + 1];
#pragma unroll
                for (int q = 0; q < NQ; q++) { u[q] += (double)av[b][q] * r0; w2[q] += (double)av[b + 1][q] * r1; }
            }
        }
    }
    for (; i + 1 < nS; i += 2) {
        const double r0 = rs[i], r1 = rs[i + 1];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            u[q] += (double)ap[i * ldA + kWave * q] * r0;
            w2[q] += (double)ap[(i + 1) * ldA + kWave * q] * r1;
        }
    }
    if (i < nS) {
        const double r0 = rs[i];
#pragma unroll
        for (int q = 0; q < NQ; q++) u[q] += (double)ap[i * ldA + kWave * q] * r0;
    }
}

// The rows lane + 64 rr of column t: the loads under their guards into registers, the conversions OUTSIDE the guards -- all NR loads of
// the column in flight together; `cond ? (double)As[..] : 0` makes the compiler wait inside every guard (the raw columns of a 12-atom
// seed: 96 L2 round trips one after the other per left-over voxel of a 288-volume protocol, 24 LDS round trips at 99 volumes).
// Measured (profiles/r05b_tile_column_ab.txt): nothing at 288 volumes (those kernels wait elsewhere), 0.5 - 1 % of the 99-volume fit
// with the LDS tiles built the same way (AMX_TILE_COL_LDS=0: the guarded form).
#ifndef AMX_TILE_COL_LDS
#define AMX_TILE_COL_LDS 1
#endif
template <int NR, typename AT>
__device__ __forceinline__ void tile_column(const AT *As, int ldA, int nS, int t, int lane, const bool (&rowok)[NR], double (&col)[NR])
{
    if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
        AT raw[NR] = {};
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            if (i < nS && rowok[rr]) raw[rr] = As[i * ldA + t];
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            col[rr] = (i < nS && rowok[rr]) ? (double)raw[rr] : 0.0;
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            col[rr] = (i < nS && rowok[rr]) ? (double)As[i * ldA + t] : 0.0;
        }
    }
}

// ------------------------------------------------------------------ the solver
// NR   rows per lane   (nS      <= 64*NR)
// NQ   atoms per lane  (n_atoms <= 64*NQ)
// MAXP capacity of the passive set (slot space, <= 64)
// RIDGE lambda2 > 0 (augmented rows kept); AT storage type of A in LDS
template <int NR, int NQ, int MAXP, bool RIDGE, typename AT>
struct NNSolver {
    static_assert(MAXP <= kWave, "passive set lives in the lanes of one wavefront");
    // thin QR of the passive columns
    double Q[MAXP][NR];   // row space: Q[k][r] = q_k(row lane+64r)
    double *Ql;           // per-wave LDS (RIDGE): Ql[s*LDR + k] = entry of q_k in the ridge row of slot s's atom
    static constexpr int LDR = MAXP + 1;   // odd leading dimension: rows AND columns of R are conflict-free
    double *Rl;           // per-wave LDS: R[i][c] at Rl[i*LDR + c] (upper triangle used)
    double d, e, rinv;    // lane i: (Q'y)_i, (R^-T 1)_i, 1/R_ii
    double x, sc;         // lane s: coefficient and column scale of slot s
    double xprev;         // lane s: coefficient at the last update of the dual vector (Gram updates)
    int idx;              // lane s: atom of slot s
    int np;               // passive-set size (uniform)
    double r[NR];         // row space: residual y - A s x at exit
    int iters;
    int n_exact, n_gram;  // dual-vector evaluations: exact sweeps / Gram updates (statistics)
    int seed_why;         // diagnosis: why certify_seed refused (1 malformed, 2 pivot, 3 refinement, 4 x <= 0, 5 dual value > 0)
    int seeded;           // 1: the seed was certified (Q holds the raw passive columns, r the final residual), 0: not, -1: no seed
#ifdef AMX_PHASES
    long long ph[8], pht; // 0 sweep, 1 gram update, 2 selection, 3 column + Gram-Schmidt, 4 commit, 5 triangular solve, 6 step/removal, 7 other
#endif

    // fl: per-lane atom flags (bit q: allowed, bit 8+q: passive, bit 16+q: banned) of atom lane+64q
    __device__ __forceinline__ void remove_slot(int k, int lane, unsigned &fl)
    {
        const int a = bcast_i(idx, k);
        if (lane == (a & 63)) fl &= ~(0x100u << (a >> 6));
        // re-triangularise R without column k: rotate rows (j, j+1), j = k .. np-2
#pragma unroll
        for (int j = 0; j < MAXP - 1; j++) {
            if (j >= k && j < np - 1) {
                // rows j, j+1 of R: lane m owns column m (two contiguous LDS rows, no bank conflicts)
                const int lc = lane < MAXP ? lane : MAXP;      // column MAXP is padding
                const double ra = Rl[j * LDR + lc], rb = Rl[(j + 1) * LDR + lc];
                const double ga = bcast(ra, j + 1), gb = bcast(rb, j + 1);
                // one reciprocal square root (v_rsq_f64 + Newton) instead of sqrt and three divisions
                const double h2 = ga * ga + gb * gb;
                const double ri = (h2 > 0.0) ? inv_sqrt(h2) : 0.0;
                const double rr = h2 * ri;
                const double c = (h2 > 0.0) ? ga * ri : 1.0, s = gb * ri;
                if (lane > j && lane < np) {
                    Rl[j * LDR + lane] = c * ra + s * rb;
                    Rl[(j + 1) * LDR + lane] = c * rb - s * ra;
                }
#pragma unroll
                for (int rr_ = 0; rr_ < NR; rr_++) {
                    const double q0 = Q[j][rr_], q1 = Q[j + 1][rr_];
                    Q[j][rr_] = c * q0 + s * q1;
                    Q[j + 1][rr_] = c * q1 - s * q0;
                }
                if (RIDGE && lane < np) {
                    const double q0 = Ql[lane * LDR + j], q1 = Ql[lane * LDR + j + 1];
                    Ql[lane * LDR + j] = c * q0 + s * q1;
                    Ql[lane * LDR + j + 1] = c * q1 - s * q0;
                }
                {
                    const double d0 = bcast(d, j), d1 = bcast(d, j + 1);
                    const double e0 = bcast(e, j), e1 = bcast(e, j + 1);
                    if (lane == j) { d = c * d0 + s * d1; e = c * e0 + s * e1; rinv = ri; }
                    if (lane == j + 1) { d = c * d1 - s * d0; e = c * e1 - s * e0; }
                }
            }
        }
        // the rotations have moved the dropped direction into row np-1: clear it (rows beyond np stay zero)
#pragma unroll
        for (int m = 0; m < MAXP; m++) {       // (selects, not an indexed store: Q must stay in registers)
            const bool last = (m == np - 1);
#pragma unroll
            for (int rr_ = 0; rr_ < NR; rr_++) Q[m][rr_] = last ? 0.0 : Q[m][rr_];
        }
        // columns k+1.. move one to the left (rows stay where they are)
        for (int i = 0; i < np - 1; i++) {
            const double t = Rl[i * LDR + (lane < MAXP ? lane + 1 : MAXP)];
            if (lane >= k && lane < np - 1) Rl[i * LDR + lane] = t;
        }
        // slot-indexed data of lanes > k move one lane down
        {
            const double xn = from_next_lane(x), sn = from_next_lane(sc), pn = from_next_lane(xprev);
            const int in = from_next_lane(idx);
            if (lane >= k) { x = xn; sc = sn; idx = in; xprev = pn; }
            if (RIDGE) {
                // ridge rows follow their slots (row s <- row s+1 for s >= k); the vacated last
                // row and the dropped last column are cleared (rows/columns >= np stay zero)
                for (int m = 0; m < np; m++) {
                    const double qn = Ql[(lane < MAXP - 1 ? lane + 1 : MAXP - 1) * LDR + m];
                    if (lane >= k && lane < np - 1) Ql[lane * LDR + m] = qn;
                    if (lane == np - 1) Ql[lane * LDR + m] = 0.0;
                }
                if (lane < np) Ql[lane * LDR + np - 1] = 0.0;
            }
        }
        np = __builtin_amdgcn_readfirstlane(np - 1);
        if (lane >= np) { x = 0.0; xprev = 0.0; d = 0.0; e = 0.0; idx = -1; }
    }

    // yr     row space, 0 on rows >= nS and on rows excluded by rowok
    // rowok  row space, rows that belong to the problem
    // scl    atom space column scales, allowed[q] uniform bit masks of admissible atoms
    // rs     per-wave LDS scratch of NR*64 doubles; rl per-wave LDS for R and the ridge rows: 2*(MAXP+1)*LDR doubles
    // G      optional Gram matrix A'A of this orientation (global memory, row stride ldG >= 64*NQ,
    //        restricted to the rows in rowok): lets the dual vector follow coefficient changes by
    //        a few column loads instead of a full sweep of A; every decision near convergence
    //        (and the final KKT check) is still taken on the exactly recomputed dual vector.
    // Control flow is wave-uniform by construction; every branch condition goes through uni()
    // (v_readfirstlane) so that the compiler emits scalar branches and never masks EXEC around
    // the cross-lane operations.
    // r = y - Q (d - l1 e): residual of the current passive solution (the member `r` is only refreshed by the exact
    // sweeps of solve(); callers that need it afterwards -- the error maps -- recompute it here)
    __device__ __forceinline__ void residual(const double (&yr)[NR], double lam1)
    {
        if (seeded == 1) return;          // certify_seed() left the residual of the certified solution in r
#pragma unroll
        for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
        const double coef = d - lam1 * e;
#pragma unroll
        for (int k = 0; k < MAXP; k++) {
            if (k < np) {
                const double ck = bcast(coef, k);
#pragma unroll
                for (int rr = 0; rr < NR; rr++) r[rr] -= Q[k][rr] * ck;
            }
        }
    }


    // u = A'v (atom space) by one sweep over the LDS tile; v is handed over through the per-wave scratch rs
    __device__ __forceinline__ void sweep(const AT *As, int ldA, int nS, const double (&v)[NR], double *rs, int lane, double (&u)[NQ])
    {
#pragma unroll
        for (int rr = 0; rr < NR; rr++) rs[lane + kWave * rr] = v[rr];
        double w2[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) { u[q] = 0.0; w2[q] = 0.0; }
        tile_sweep<NQ, AT>(As + lane, ldA, nS, rs, u, w2);
#pragma unroll
        for (int q = 0; q < NQ; q++) u[q] += w2[q];
    }

    // lane s < np: sum over the rows of Q[s] * v (the raw passive columns while a seed is being certified)
    __device__ __forceinline__ double slot_dots(const double (&v)[NR], int lane)
    {
        double out = 0.0;
#pragma unroll
        for (int kb = 0; kb < MAXP; kb += 4) {
            if (kb < np) {
                double p[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    p[u] = 0.0;
                    if (kb + u < MAXP) {
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) p[u] += Q[kb + u][rr] * v[rr];
                    }
                }
                if (kb + 1 < np) wave_sum4(p, lane);
                else { p[0] = bcast(wave_sum(p[0]), 0); p[1] = 0.0; p[2] = 0.0; p[3] = 0.0; }
#pragma unroll
                for (int u = 0; u < 4; u++) out = (lane == kb + u) ? p[u] : out;
            }
        }
        return out;
    }

    // two triangular solves with the Cholesky factor kept as a square in Rl (lane = row): returns (L L')^-1 rhs in slot space
    __device__ __forceinline__ double chol_solve(double rhs, int lane)
    {
        const int li = (lane < MAXP ? lane : MAXP) * LDR;
        double f = (lane < np) ? rhs : 0.0;
        for (int k = 0; k < np; k++) {
            const double lk = Rl[li + k];
            const double wk = bcast(f * rinv, k);
            if (lane > k) f -= lk * wk;
        }
        double b = f * rinv;
        for (int k = np - 1; k >= 0; k--) {
            const double lk = Rl[k * LDR + (lane < k ? lane : k)];
            const double zk = bcast(b * rinv, k);
            if (lane < k) b -= lk * zk;
        }
        return (lane < np) ? b * rinv : 0.0;
    }

    // Certify a support seed in the full problem (unregularised NNLS only): least squares on the seeded columns by the
    // semi-normal equations -- Cholesky of the precomputed Gram block, refined on the TRUE residual until the passive dual
    // values are at rounding level, which is what the thin QR of the Lawson-Hanson path delivers too --, then one exact
    // sweep of the dual vector and the STRICT Kuhn-Tucker test of that path (x_P > 0, w_j <= 0 for every other admissible
    // atom).  true: np / idx / x / r hold the solution (a KKT point of a problem whose minimiser is unique whenever the
    // seeded columns are independent).  false: nothing is decided -- the caller starts Lawson-Hanson from the empty set.
    __device__ __forceinline__ bool certify_seed(const AT *As, int ldA, int nS, const double (&yr)[NR], const bool (&rowok)[NR],
                                                 unsigned fl, unsigned long long seed, double *rs, int lane,
                                                 const double *__restrict__ G, int ldG, const SeedScreen &scr)
    {
#ifdef AMX_PHASES
        for (int k = 0; k < 8; k++) ph[k] = 0;
        pht = (long long)__builtin_readcyclecounter();
#endif
        // decode: slot s = byte s
        const int my = (int)((seed >> (8 * (lane & 7))) & 0xffull);
        const unsigned long long present = ballot64(lane < MAXP && lane < 8 && my < 0xf0);
        const int n0 = __builtin_popcountll(present);
        seed_why = 0;
        if (present != ((1ull << n0) - 1ull)) { seed_why = 1; return false; }   // not a prefix: malformed
        np = n0;
        idx = (lane < np) ? my : -1;
        // every seeded atom must be admissible and distinct
        {
            bool bad = false;
            for (int s = 0; s < np; s++) {
                const int t = bcast_i(idx, s);
                const unsigned ft = (unsigned)bcast_i((int)fl, t & 63);
                bad = bad || !((ft >> (t >> 6)) & 1u) || (t >= 64 * NQ);
                if (lane > s && lane < np && idx == t) bad = true;
            }
            if (ballot64(bad) != 0ull) { np = 0; idx = -1; seed_why = 1; return false; }
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
        bool ok = true;
        AMX_PH(0);
        if (np > 0) {
            // the Gram block first (lane = row of the triangle): ALL its loads are in flight at once, and behind them the column
            // fetch and the first set of dot products (a loop that loaded and stored entry by entry waited np round trips)
            double gk[MAXP];
#pragma unroll
            for (int k = 0; k < MAXP; k++) {
                gk[k] = 0.0;
                if (k < np) {
                    const int tk = bcast_i(idx, k);
                    if (lane >= k && lane < np) gk[k] = G[(size_t)idx * ldG + tk];
                }
            }
            // raw columns (row space)
#pragma unroll
            for (int m = 0; m < MAXP; m++) {
                if (m < np) {
                    const int t = bcast_i(idx, m);
                    tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, Q[m]);
                }
            }
            const int li = (lane < MAXP ? lane : MAXP) * LDR;
            const double c0 = slot_dots(yr, lane);                        // A_P'y (independent of the factor)
#pragma unroll
            for (int k = 0; k < MAXP; k++) {
                if (k < np && lane >= k && lane < np) Rl[li + k] = gk[k];
            }
            AMX_PH(1);
            for (int k = 0; k < np; k++) {
                double t = Rl[li + k];
                const double hkk = bcast(t, k);
                for (int m = 0; m < k; m++) t -= Rl[li + m] * Rl[k * LDR + m];
                const double tk = bcast(t, k);
                if (!uni(tk > 1e-14 * hkk)) { ok = false; seed_why = 2; break; }
                const double iv = inv_sqrt(tk);
                if (lane >= k && lane < np) Rl[li + k] = t * iv;
                if (lane == k) rinv = iv;
            }
            AMX_PH(2);
            if (ok) {
                x = chol_solve(c0, lane);
#pragma unroll
                for (int m = 0; m < MAXP; m++) {
                    if (m < np) {
                        const double xs = bcast(x, m);
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) r[rr] -= Q[m][rr] * xs;
                    }
                }
                AMX_PH(3);
                // refinement on the true residual: g = A_P' r -> 0
                double gprev = __builtin_huge_val();
                for (int it = 0; it < 5; it++) {
                    const double g = slot_dots(r, lane);
                    const double gmax = wave_max((lane < np) ? fabs(g) : 0.0);
                    if (uni(gmax < 2e-15)) break;
                    if (uni(!(gmax < 0.25 * gprev))) { ok = uni(gmax < 1e-12); if (!ok) seed_why = 3; break; }      // stagnation: rounding level reached (or ill-conditioned)
                    gprev = gmax;
                    const double dx = chol_solve(g, lane);
                    x += dx;
#pragma unroll
                    for (int m = 0; m < MAXP; m++) {
                        if (m < np) {
                            const double ds = bcast(dx, m);
#pragma unroll
                            for (int rr = 0; rr < NR; rr++) r[rr] -= Q[m][rr] * ds;
                        }
                    }
                    if (it == 4) { ok = false; seed_why = 3; }
                }
                if (ok && ballot64(lane < np && !(x > 0.0)) != 0ull) { ok = false; seed_why = 4; }
                AMX_PH(4);
            }
        }
        if (ok) {
            unsigned pm = 0u;                                    // bit q: atom lane + 64 q is seeded
            for (int s = 0; s < np; s++) {
                const int t = bcast_i(idx, s);
                if (lane == (t & 63)) pm |= 1u << (t >> 6);
            }
            // ||r||^2 and ||y||^2 together; a residual at rounding level is a Kuhn-Tucker point whatever the signs of the
            // dual values (see the zero-residual exit of solve())
            double nrm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int rr = 0; rr < NR; rr++) { nrm[0] += r[rr] * r[rr]; nrm[1] += yr[rr] * yr[rr]; }
            wave_sum4(nrm, lane);
            const bool zero_res = np > 0 && nrm[0] <= 1e-28 * nrm[1];
            AMX_PH(5);
            if (zero_res) {
                n_exact++;
            } else if (scr.Sf == nullptr) {
                // exact dual vector of the candidate solution; strict test over the admissible atoms outside the seed
                double u[NQ];
                sweep(As, ldA, nS, r, rs, lane, u);
                n_exact++;
                bool viol = false;
#pragma unroll
                for (int q = 0; q < NQ; q++) viol = viol || (((fl & ~pm) >> q) & 1u && u[q] > 0.0);
                ok = ballot64(viol) == 0ull;
            } else {
                // SCREENED test.  a_j = U s_j + e_j with ||e_j|| tiny (the basis spans the dictionary to float32 rounding), so
                // a_j'r = s_j'(U'r) + e_j'r and |e_j'r| <= ||e_j|| ||r||: an atom whose compressed dual value is below
                // -kappa ||r|| cannot violate.  U'r = y~ - S_P x needs no reduction (lane d < KD owns component d); the
                // compressed dual values are KD float32 products per atom from an LDS table.  Only the atoms inside the
                // margin get their exact fp64 dual value (a column dot product), and the strict test of the sweep.
                constexpr int KDs = 12;
                // few admissible atoms outside the seed (stage 3: the LASSO support): all of them get the exact value at once
                int n_out = 0;
#pragma unroll
                for (int q = 0; q < NQ; q++) n_out += __builtin_popcountll(ballot64(((fl & ~pm) >> q) & 1u));
                const bool direct = n_out <= 12;
                float ut[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) ut[q] = 0.0f;
                const double rho2 = nrm[0];
                if (!direct) {
                    // (every passive atom's row at once: guarded by `m < np` each load sat in a branch of its own and was waited for
                    //  before the next one left -- np memory round trips in a row)
                    const int l12 = lane < KDs ? lane : 0;
                    double rt = scr.ytil[l12];
                    double sv[MAXP];
#pragma unroll
                    for (int m = 0; m < MAXP; m++) {
                        const int t = m < np ? bcast_i(idx, m) : 0;
                        sv[m] = scr.Sg[(size_t)t * KDs + l12];
                    }
                    rt = (lane < KDs) ? rt : 0.0;
#pragma unroll
                    for (int m = 0; m < MAXP; m++) {
                        if (m < np) {
                            const double xs = bcast(x, m);
                            if (lane < KDs) rt -= sv[m] * xs;
                        }
                    }
#pragma unroll
                    for (int dd = 0; dd < KDs; dd++) {
                        const float rd = (float)bcast(rt, dd);
#pragma unroll
                        for (int q = 0; q < NQ; q++) ut[q] += scr.Sf[dd * scr.ld + lane + kWave * q] * rd;
                    }
                }
                const float margin = direct ? __builtin_huge_valf() : (float)(1.0625 * scr.kappa * sqrt(rho2));
                AMX_PH(6);
                bool viol = false;
                // the atoms inside the margin, four exact dot products per batched reduction
                unsigned long long todo[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) todo[q] = ballot64((((fl & ~pm) >> q) & 1u) && !(ut[q] < -margin));
                for (int guard = 0; guard < kWave * NQ; guard++) {
                    int t4[4] = {-1, -1, -1, -1};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            if (t4[u] < 0 && todo[q] != 0ull) { t4[u] = kWave * q + __builtin_ctzll(todo[q]); todo[q] &= todo[q] - 1ull; }
                        }
                    }
                    if (t4[0] < 0) break;
                    double p[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        p[u] = 0.0;
                        const int t = t4[u] < 0 ? t4[0] : t4[u];
                        if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
                            double col[NR];
                            tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, col);
#pragma unroll
                            for (int rr = 0; rr < NR; rr++) p[u] += col[rr] * r[rr];
                        } else {
#pragma unroll
                            for (int rr = 0; rr < NR; rr++) {
                                const int i = lane + kWave * rr;
                                if (i < nS && rowok[rr]) p[u] += (double)As[i * ldA + t] * r[rr];
                            }
                        }
                    }
                    if (t4[1] >= 0) wave_sum4(p, lane);
                    else { p[0] = bcast(wave_sum(p[0]), 0); p[1] = 0.0; p[2] = 0.0; p[3] = 0.0; }
#pragma unroll
                    for (int u = 0; u < 4; u++) viol = viol || (t4[u] >= 0 && p[u] > 0.0);
#ifdef AMX_STATS
                    if (scr.count && lane == 0) atomicAdd(scr.count, (t4[0] >= 0) + (t4[1] >= 0) + (t4[2] >= 0) + (t4[3] >= 0));
#endif
                }
                n_exact++;
                ok = !viol;
                AMX_PH(7);
            }
            if (!ok) seed_why = 5;
        }
        if (!ok) {
            // leave the state as solve() initialised it
            np = 0; idx = -1; x = 0.0; rinv = 0.0;
#pragma unroll
            for (int m = 0; m < MAXP; m++) {
#pragma unroll
                for (int rr = 0; rr < NR; rr++) Q[m][rr] = 0.0;
            }
            return false;
        }
        sc = 1.0; xprev = x; d = 0.0; e = 0.0;
        if (lane >= np) { x = 0.0; xprev = 0.0; idx = -1; }
        return true;
    }

    __device__ __forceinline__ int solve(const AT *As, int ldA, int nS, int n_atoms,
                                         const double (&yr)[NR], const bool (&rowok)[NR],
                                         const double (&scl)[NQ],
                                         const unsigned long long (&allowed)[NQ],
                                         double lam1, double lam2, double *rs, double *rl, int lane,
                                         const double *__restrict__ G = nullptr, int ldG = 0,
                                         unsigned long long seed = kSeedNone, const SeedScreen &scr = SeedScreen())
    {
        Rl = rl;
        Ql = rl + (MAXP + 1) * LDR;
        // KKT tolerance on the (exactly recomputed) dual vector.  Lawson-Hanson -- the unregularised solver of the
        // reference -- continues while any admissible dual value is > 0 and leaves it to the z-test to refuse
        // candidates that are rounding noise; with nearly collinear atoms a dual value of 5e-13 still moves the maps
        // by 1e-3, so the NNLS problems use the same strict rule.  The l1 problems keep a small positive tolerance.
        const double tol = (lam1 > 0.0) ? 1e-12 : 0.0;
        // Lawson-Hanson's independence test of a candidate column, `unorm + |pivot| * 0.01 != unorm` with unorm = the
        // norm of its component inside span(Q) and pivot = the norm b of the component outside: accept iff
        // b * 0.01 exceeds half an ulp of unorm, i.e. b^2 > (1.1e-16 / 0.01)^2 unorm^2 (1.2e-28 .. 4.9e-28 depending
        // on where unorm sits in its binade; 2e-28 here).  A stricter test bans atoms the reference's solver accepts.
        const double dep2 = 2e-28;
        const double inf = __builtin_huge_val();
        const int itmax = 3 * n_atoms + 10;  // Lawson-Hanson's cap
        const double sqlam2 = RIDGE ? sqrt(lam2) : 0.0;
        // atom flags live per lane (VGPR bits), not as wave-uniform 64-bit masks: keeps SGPRs free
        unsigned fl = 0u;
#pragma unroll
        for (int q = 0; q < NQ; q++) fl |= (unsigned)((allowed[q] >> lane) & 1ull) << q;
        np = 0; d = 0.0; e = 0.0; rinv = 0.0; x = 0.0; xprev = 0.0; sc = 1.0; idx = -1; iters = 0;
        // invariant: rows of Q (and columns of the ridge block Ql) beyond the passive set are zero, so the Gram-Schmidt
        // blocks need no per-row conditions
#pragma unroll
        for (int m = 0; m < MAXP; m++) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) Q[m][rr] = 0.0;
        }
        if (RIDGE && lane < MAXP) {
            for (int m = 0; m < LDR; m++) Ql[lane * LDR + m] = 0.0;
        }
        int status = kSolved;
        int last_added = -1, second_looks = 0;
        bool cyc_banned = false, prefer = false;
        n_exact = 0; n_gram = 0;
        seeded = -1;
        if (!RIDGE && G != nullptr && seed != kSeedNone && lam1 == 0.0) {
            seeded = certify_seed(As, ldA, nS, yr, rowok, fl, seed, rs, lane, G, ldG, scr) ? 1 : 0;
            if (seeded == 1) return kSolved;
#ifndef AMX_NO_WARM_ORDER
            // A refused seed is still the best guess at the support (it is wrong in an atom or two, or its Gram block was too
            // ill-conditioned for the semi-normal equations): its atoms get to ENTER FIRST.  Lawson-Hanson may admit any atom
            // whose dual value is positive -- the arg-max is a heuristic, the descent proof only needs w_t > 0 -- so this is
            // the same finite algorithm with the same tests (independence, z-test, strict Kuhn-Tucker stop) on a path that
            // does not wander along the (kappa, v_ic) grid first: ~5 column additions instead of ~10 + 5 removals.
            // fl bit 24 + q: atom lane + 64 q is a seeded atom that has not entered or been refused yet.
            if (seed_why != 1) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) {
                    const int t = (int)((seed >> (8 * s8)) & 0xffull);
                    if (t < 0xf0 && t < n_atoms && t < kWave * NQ && lane == (t & 63)) fl |= 0x1000000u << (t >> 6);
                }
                prefer = true;
            }
#endif
        }
#ifdef AMX_PHASES
        if (seeded != 1) { for (int k = 0; k < 8; k++) ph[k] = 0; }
        pht = (long long)__builtin_readcyclecounter();
#endif
#ifndef AMX_GRAM_COLS
#define AMX_GRAM_COLS 4
#endif
        constexpr int kGramCols = AMX_GRAM_COLS;   // Gram columns in flight per trip of the dual-vector update
#ifndef AMX_GRAM_STEPS
#define AMX_GRAM_STEPS 48
#endif
        constexpr int kMaxGramSteps = AMX_GRAM_STEPS;    // bound the drift of the Gram-updated dual vector
        const double kExactBelow = 1e-7;     // decisions on smaller dual values use the exact sweep
        double u[NQ];                        // atom space: A' r (unscaled, without the l1 shift)
        bool have_u = false, force_exact = false;
        int gram_steps = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) u[q] = 0.0;

        // Zero-residual exit (unregularised NNLS): once ||r|| <= 1e-14 ||y|| every dual value a_j'r is rounding noise of
        // either sign, and Lawson-Hanson's strict `w > 0` rule would keep admitting atoms with coefficients ~1e-17 until
        // the iteration cap (a voxel whose signal IS an atom: golden fixture voxel 0).  The point is a Kuhn-Tucker point to
        // working precision: stop there.
        double ysq = 0.0;
        if (lam1 == 0.0 && !RIDGE) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) ysq += yr[rr] * yr[rr];
            ysq = wave_sum(ysq);
        }
        for (int outer = 0; status == kSolved; ++outer) {
            if (outer > 2 * itmax) { status = kGuardOuter; break; }   // never spin
            double w[NQ];
            const bool exact = (G == nullptr) || !have_u || force_exact || gram_steps >= kMaxGramSteps;
            if (exact) {
                // ---- residual of the current passive least-squares solution: r = y - Q (d - l1 e)
#pragma unroll
                for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
                {
                    const double coef = d - lam1 * e;
#pragma unroll
                    for (int k = 0; k < MAXP; k++) {
                        if (k < np) {
                            const double ck = bcast(coef, k);
#pragma unroll
                            for (int rr = 0; rr < NR; rr++) r[rr] -= Q[k][rr] * ck;
                        }
                    }
                }
                if (lam1 == 0.0 && !RIDGE && np > 0) {
                    double rsq = 0.0;
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) rsq += r[rr] * r[rr];
                    rsq = wave_sum(rsq);
                    if (uni(rsq <= 1e-28 * ysq)) break;
                }
                // ---- u = A' r (atom space): one sweep over the LDS tile
#pragma unroll
                for (int rr = 0; rr < NR; rr++) rs[lane + kWave * rr] = r[rr];
                double w2[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) { u[q] = 0.0; w2[q] = 0.0; }
                tile_sweep<NQ, AT>(As + lane, ldA, nS, rs, u, w2);
#pragma unroll
                for (int q = 0; q < NQ; q++) u[q] += w2[q];
                have_u = true; force_exact = false; gram_steps = 0; n_exact++;
                AMX_PH(0);
            } else {
                // ---- u -= G[:, P] (s (x - xprev)): the passive coefficients moved, nothing else did
                // (4 columns = 4*NQ loads in flight per trip: the loop is bound by L2/MALL latency)
                {
                    const double delta = sc * (x - xprev);
                    for (int s0 = 0; s0 < np; s0 += kGramCols) {
                        double gv[kGramCols][NQ], dls[kGramCols];
#pragma unroll
                        for (int t4 = 0; t4 < kGramCols; t4++) {
                            const int sl = (s0 + t4 < np) ? s0 + t4 : np - 1;
                            const double dv = bcast(delta, sl);
                            dls[t4] = (s0 + t4 < np) ? dv : 0.0;
                            const double *gc = G + (size_t)bcast_i(idx, sl) * ldG + lane;
#pragma unroll
                            for (int q = 0; q < NQ; q++) gv[t4][q] = gc[kWave * q];
                        }
#pragma unroll
                        for (int t4 = 0; t4 < kGramCols; t4++) {
#pragma unroll
                            for (int q = 0; q < NQ; q++) u[q] -= gv[t4][q] * dls[t4];
                        }
                    }
                }
                gram_steps++; n_gram++;
                AMX_PH(1);
            }
            xprev = x;
#pragma unroll
            for (int q = 0; q < NQ; q++) w[q] = scl[q] * u[q] - lam1;

            // ---- pick the most violating admissible atom; test it; maybe take the next one
            bool added = false, redo = false;
            for (int sel = 0; status == kSolved && !added; ++sel) {
                if (sel > kWave * NQ + 2) { status = kGuardSelect; break; }
                double best = -inf;
                int bj = -1;
                unsigned cm = fl & ~(fl >> 8) & ~(fl >> 16);     // bit q: allowed, not passive, not barred
                if (prefer) {
                    // seeded atoms with a positive dual value enter before anything else (see above)
                    const unsigned cp = cm & (fl >> 24);
                    bool anyp = false;
#pragma unroll
                    for (int q = 0; q < NQ; q++) anyp = anyp || (((cp >> q) & 1u) && w[q] > tol);
                    if (ballot64(anyp) != 0ull) cm = cp;
                    else prefer = false;                      // none qualifies now: the arg-max rule takes over for good
                }
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    if (((cm >> q) & 1u) && w[q] > best) { best = w[q]; bj = lane + kWave * q; }
                }
                const double wmax = wave_max(best);
                if (!exact) {
                    // every admissible dual value is negative by far more than the Gram updates can have drifted:
                    // the exact sweep would confirm the KKT point and change nothing
#ifndef AMX_ALWAYS_CONFIRM
                    if (uni(wmax < -kExactBelow)) break;
#endif
                    if (uni(!(wmax > kExactBelow))) { force_exact = true; redo = true; break; }
                }
                if (!uni(wmax > tol)) break;                      // KKT point reached
                const unsigned long long who = ballot64(best == wmax);
                if (uni(who == 0ull)) { status = kGuardSelect; break; }
                const int t = bcast_i(bj, __builtin_ctzll(who));
                if (uni(t < 0 || t >= n_atoms)) { status = kGuardSelect; break; }
                if (np >= MAXP) { status = kOverflow; break; }
                const int tq = t >> 6, tl = t & 63;
                double sct = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q == tq) sct = bcast(scl[q], tl);
                AMX_PH(2);
                // candidate column (row space) and its ridge rows (slot space)
                double v[NR];
                double vsq = 0.0;
                {
                    double col[NR];
                    tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, col);
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) {
                        const int i = lane + kWave * rr;
                        v[rr] = (i < nS && rowok[rr]) ? sct * col[rr] : 0.0;
                        vsq += v[rr] * v[rr];
                    }
                }
                double va = (RIDGE && lane == np) ? sqlam2 : 0.0;
                const int ls = (lane < MAXP ? lane : MAXP - 1) * LDR;   // this lane's ridge row in Ql
                const double vsq0 = vsq;          // |a_t|^2 of this lane's rows (reduced with the final batch below)
                double rho = 0.0;                 // lane k: R[k][new]
                // two Gram-Schmidt passes, 4 projections in flight at a time
#pragma unroll
#ifndef AMX_CGS_PASSES
#define AMX_CGS_PASSES 2
#endif
                for (int pass = 0; pass < AMX_CGS_PASSES; pass++) {
#pragma unroll
                    for (int kb = 0; kb < MAXP; kb += 4) {
                        if (kb < np) {
                            double p[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {      // rows of Q beyond np are zero: no per-row conditions
                                p[u] = 0.0;
                                if (kb + u < MAXP) {
#pragma unroll
                                    for (int rr = 0; rr < NR; rr++) p[u] += Q[kb + u][rr] * v[rr];
                                    if (RIDGE) p[u] += Ql[ls + kb + u] * va;
                                }
                            }
                            // (both paths end in readlanes: the projections stay scalar registers, no vector copies
                            // where the paths merge)
                            if (kb + 1 < np) wave_sum4(p, lane);        // >= 2 live projections: batched
                            else { p[0] = bcast(wave_sum(p[0]), 0); p[1] = 0.0; p[2] = 0.0; p[3] = 0.0; }
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                if (kb + u < MAXP) {
#pragma unroll
                                    for (int rr = 0; rr < NR; rr++) v[rr] -= p[u] * Q[kb + u][rr];
                                    if (RIDGE) va -= p[u] * Ql[ls + kb + u];
                                    if (lane == kb + u) rho += p[u];
                                }
                            }
                        }
                    }
                }
                // one batched reduction for everything the tests and the commit need: |a_t|^2, |v|^2, v'y, rho'e
                double fin[4];
                fin[0] = vsq0;
                fin[1] = va * va;
                fin[2] = 0.0;
                fin[3] = (lam1 != 0.0 && lane < np) ? rho * e : 0.0;        // only the l1 term needs e
#pragma unroll
                for (int rr = 0; rr < NR; rr++) { fin[1] += v[rr] * v[rr]; fin[2] += v[rr] * yr[rr]; }
                wave_sum4(fin, lane);
                const double n0 = fin[0] + lam2, b2 = fin[1];
                bool reject = !uni(b2 > dep2 * (n0 - b2));
                double beta = 0.0, binv = 0.0, dnew = 0.0, enew = 0.0;
                if (!reject) {
                    binv = inv_sqrt(b2);
                    beta = b2 * binv;
                    dnew = fin[2] * binv;
                    enew = (lam1 != 0.0) ? (1.0 - fin[3]) * binv : 0.0;
                    const double znew = (dnew - lam1 * enew) * binv;   // Lawson-Hanson "ztest"
                    reject = !uni(znew > 0.0);
                }
                if (reject) {
                    if (lane == tl) fl |= 0x10000u << tq;
                } else {
                    AMX_PH(3);
                    // ---- commit column np
                    const int kn = np;
#pragma unroll
                    for (int m = 0; m < MAXP; m++) {
                        if (m == kn) {
#pragma unroll
                            for (int rr = 0; rr < NR; rr++) Q[m][rr] = v[rr] * binv;
                        }
                    }
                    if (lane <= kn) Rl[lane * LDR + kn] = (lane == kn) ? beta : rho;     // column kn of R
                    if (RIDGE && lane <= kn) Ql[lane * LDR + kn] = va * binv;               // ridge rows of q_kn
                    if (lane == kn) { d = dnew; e = enew; rinv = binv; x = 0.0; sc = sct; idx = t; }
                    fl &= 0xff00ffffu; cyc_banned = false;                           // forget the rejected candidates
                    if (lane == tl) fl |= 0x100u << tq;
                    np = kn + 1;
                    last_added = t;
                    added = true;

                }
            }
            if (redo) continue;  // small dual values: decide on the exactly recomputed vector
            if (!added) {
                // An atom that left the passive set in the very step that brought it in is barred from re-entering until
                // another atom has been added (add/remove cycles on rounding noise).  Lawson-Hanson itself forgets such
                // history after every step, so before declaring a KKT point give the barred atoms another look -- on an
                // exact dual vector, a bounded number of times.
                if (status == kSolved && cyc_banned && second_looks < 3) {
                    fl &= 0xff00ffffu; cyc_banned = false; second_looks++; force_exact = true; last_added = -1;
                    continue;
                }
                break;   // KKT point (or a guard tripped)
            }

            AMX_PH(4);
            // ---- Lawson-Hanson inner loop: restore feasibility of the passive solution
            for (bool feasible = false; !feasible && status == kSolved;) {
                if (++iters > itmax) { status = kIterCap; break; }
                double rhs = d - lam1 * e;
                {
                    // R z = rhs.  Lane j's rhs is final once step j has
                    // run (only lanes < j are updated afterwards), so z = rhs * rinv is formed once, after the loop.
                    const int li = (lane < MAXP ? lane : MAXP - 1) * LDR;
                    for (int j = np - 1; j >= 0; j--) {
                        const double col = Rl[li + j];
                        const double zj = bcast(rhs * rinv, j);
                        if (lane < j) rhs -= col * zj;
                    }
                }
                const double z = (lane < np) ? rhs * rinv : 0.0;
                const bool act = lane < np;
                const bool neg = act && !(z > 0.0);
                // The heavy state (Q, R, d, ...) changes only inside the removal loop below, which makes zero trips for a
                // feasible solution: an if/else around it cost two full copies of that state per pass (phi copies).
                AMX_PH(5);
                const bool any = ballot64(neg) != 0ull;
                unsigned long long rem = 0ull;
                if (any) {
                    const double den = x - z;
                    const double ratio = neg ? ((den > 0.0) ? x / den : 0.0) : inf;
                    const double alpha = wave_min(ratio);
                    const unsigned long long hit = ballot64(neg && ratio == alpha);
                    const int kmin = hit ? __builtin_ctzll(hit) : -1;
                    x = act ? x + alpha * (z - x) : 0.0;
                    if (lane == kmin) x = 0.0;
                    rem = ballot64(act && !(x > 0.0));
                } else {
                    x = act ? z : 0.0;
                }
                for (int guard = 0; rem != 0ull && guard < kWave; ++guard) {
                    const int k = 63 - __builtin_clzll(rem);
                    rem &= ~(1ull << k);
                    const int a = bcast_i(idx, k);
                    if (a == last_added) { cyc_banned = true; if (lane == (a & 63)) fl |= 0x10000u << (a >> 6); }   // no add/remove cycling
                    if (G != nullptr) {       // the atom leaves with coefficient 0: fold its change into u now
                        const double dl = -bcast(sc * xprev, k);
                        const double *gc = G + (size_t)a * ldG + lane;
#pragma unroll
                        for (int q = 0; q < NQ; q++) u[q] -= gc[kWave * q] * dl;
                    }
                    if (lane == (a & 63)) fl &= ~(0x1000000u << (a >> 6));      // an atom that left is no longer preferred
                    remove_slot(k, lane, fl);
                }
                if (np == 0) x = 0.0;
                feasible = !any || np == 0;
                AMX_PH(6);
            }
        }
        return status;
    }
};

}  // namespace amx
