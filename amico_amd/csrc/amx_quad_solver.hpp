// amx_quad_solver.hpp -- FOUR voxels per wavefront: non-negative least squares on 16-lane DPP rows (gfx950).
//
// The unregularised solves of NODDI (models.pyx:911 stage 1, :940 stage 3) keep passive sets of <= 8 atoms, so a
// one-wavefront-per-voxel mapping (amx_solver.hpp) uses 8 of 64 lanes in slot space and spends most of its VALU issue
// slots moving data across 64 lanes (six-level reductions, v_readlane broadcasts).  Here a wavefront is split into its
// four 16-lane DPP rows; every row owns ONE voxel and the four voxels run the same Lawson-Hanson step in lock step:
//   * "row space"  : lane l of a row owns signal rows  l, l+16, ...      (NR per lane, nS      <= 16*NR)
//   * "atom space" : lane l owns atoms                 l, l+16, ...      (NQ per lane, n_atoms <= 16*NQ)
//   * "slot space" : lane s owns the s-th passive atom                   (MAXP <= 16)
//   * reductions end at `row_mirror` (4 DPP steps, no row exchange), broadcasts are ONE v_mov_b64_dpp row_newbcast,
//     and every vector instruction serves four voxels;
//   * per-row decisions (accept / reject / remove) are predicates, never branches: control flow stays wave-uniform
//     (a step is skipped only when no row needs it), so DPP never runs under a partial EXEC mask.
// Same algorithm and decision rules as NNSolver (thin QR of the passive columns in registers by blocked
// Gram-Schmidt with re-orthogonalisation, Givens down-dating, Lawson-Hanson's strict dual rule, z-test and
// independence test, Gram-column updates of the dual vector between exact sweeps); overflow of MAXP goes to the
// wavefront-per-voxel kernel.
#pragma once
#include "amx_solver.hpp"

namespace amx {

constexpr int kRow = 16;

// ------------------------------------------------------------------ 16-lane row primitives
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ double row_sum(double v)      // every lane of the row ends with the row's sum
{
    double t = v + dpp_zero<0xB1>(v);
    t += dpp_zero<0x4E>(t);
    t += dpp_zero<0x141>(t);
    t += dpp_zero<0x140>(t);
    return t;
}
__device__ __forceinline__ double row_max(double v)
{
    double t = fmax(v, dpp_zero<0xB1>(v));
    t = fmax(t, dpp_zero<0x4E>(t));
    t = fmax(t, dpp_zero<0x141>(t));
    t = fmax(t, dpp_zero<0x140>(t));
    return t;
}
__device__ __forceinline__ int row_min_i(int v)
{
    int t = min(v, dppi<0xB1>(v));
    t = min(t, dppi<0x4E>(t));
    t = min(t, dppi<0x141>(t));
    t = min(t, dppi<0x140>(t));
    return t;
}
// lane K of every row to all lanes of that row (DPP row_newbcast, gfx90a+: one v_mov_b64_dpp)
template <int K>
__device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, false); }
template <int K>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false); }
// k must fold to a constant (fully unrolled loops): the DPP control is an immediate
template <typename T>
__device__ __forceinline__ T row_bcast_c(T v, int k)
{
    switch (k) {
    case 0: return row_bcast<0>(v);   case 1: return row_bcast<1>(v);   case 2: return row_bcast<2>(v);   case 3: return row_bcast<3>(v);
    case 4: return row_bcast<4>(v);   case 5: return row_bcast<5>(v);   case 6: return row_bcast<6>(v);   case 7: return row_bcast<7>(v);
    case 8: return row_bcast<8>(v);   case 9: return row_bcast<9>(v);   case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
    case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v); case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
    }
}
// lane (row base + k), k a per-row (not compile-time) index: LDS crossbar, used on the rare removal path only
__device__ __forceinline__ int row_pick(int v, int k, int lane) { return __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, v); }
__device__ __forceinline__ double row_pick(double v, int k, int lane)
{
    return __hiloint2double(row_pick(__double2hiint(v), k, lane), row_pick(__double2loint(v), k, lane));
}
// value of the next lane of the row (lane 15 reads 0): DPP row_shl:1
__device__ __forceinline__ int row_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ double row_next(double v) { return __hiloint2double(row_next(__double2hiint(v)), row_next(__double2loint(v))); }

// four row sums at once (the transposing reduction of wave_sum4, ended inside the row): every lane gets all totals
__device__ __forceinline__ void row_sum4(double (&p)[4], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
    double k0 = o1 ? p[2] : p[0], k1 = o1 ? p[3] : p[1];
    const double s0 = o1 ? p[0] : p[2], s1 = o1 ? p[1] : p[3];
    k0 += dpp_zero<0xB1>(s0);
    k1 += dpp_zero<0xB1>(s1);
    double k = o2 ? k1 : k0;
    const double s = o2 ? k0 : k1;
    k += dpp_zero<0x4E>(s);
    k += dpp_zero<0x114>(k);           // row_shr:4
    k += dpp_zero<0x118>(k);           // row_shr:8 -> lanes 12..15 of the row hold the totals
    p[0] = row_bcast<12>(k); p[2] = row_bcast<13>(k); p[1] = row_bcast<14>(k); p[3] = row_bcast<15>(k);
}

// wave-uniform "any lane" as a scalar branch condition; per-row "any lane" / bit set as per-lane values
__device__ __forceinline__ bool wany(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ unsigned row_bits(bool p, int lane) { return (unsigned)(__ballot(p) >> (lane & 48)) & 0xffffu; }
__device__ __forceinline__ bool row_any(bool p, int lane) { return row_bits(p, lane) != 0u; }
// largest value of a row-uniform int over the four rows, as a scalar
__device__ __forceinline__ int rows_max(int v)
{
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// ------------------------------------------------------------------ the solver
template <int NR, int NQ, int MAXP>
struct QuadNNLS {
    static_assert(MAXP <= kRow, "the passive set lives in the lanes of one 16-lane row");
    static_assert(3 * NQ <= 32, "atom flags are one 32-bit word per lane");
    static constexpr int LDR = MAXP + 1;                    // odd leading dimension of R (MAXP even)
    static constexpr int kRlWords = (MAXP + 1) * LDR;       // doubles of per-row LDS for R
    static constexpr int kRsWords = kRow * NR;              // doubles of per-row LDS for the residual broadcast
    double Q[MAXP][NR];   // row space: Q[k][r] = q_k(row l + 16 r); rows >= np hold finite leftovers (masked)
    double d, rinv;       // lane i: (Q'y)_i, 1/R_ii
    double x, xprev;      // lane s: coefficient, coefficient at the last dual-vector update
    int idx;              // lane s: atom of slot s (-1 beyond np)
    int np;               // passive-set size (row-uniform)
    int status;           // row-uniform SolveStatus
    double u[NQ];         // atom space: A'r
    int n_exact, n_gram, iters;

    __device__ __forceinline__ void init_once()
    {
#pragma unroll
        for (int m = 0; m < MAXP; m++) {
#pragma unroll
            for (int rr = 0; rr < NR; rr++) Q[m][rr] = 0.0;
        }
    }

    // slot k (row-uniform, -1 = this row removes nothing) leaves the passive set of its row
    __device__ __forceinline__ void remove_slot(int k, double *Rl, int lane, unsigned &fl)
    {
        const int l = lane & 15;
        const bool rm = k >= 0;
        const int kc = rm ? k : 0;
        const int a = row_pick(idx, kc, lane);
        if (rm && l == (a & 15)) fl &= ~(1u << (NQ + (a >> 4)));
        const int lc = l < MAXP ? l : MAXP;                 // column MAXP is padding
#pragma unroll
        for (int j = 0; j < MAXP - 1; j++) {
            const bool active = rm && j >= k && j < np - 1;
            if (wany(active)) {
                const double ra = Rl[j * LDR + lc], rb = Rl[(j + 1) * LDR + lc];
                const double ga = row_bcast_c(ra, j + 1), gb = row_bcast_c(rb, j + 1);
                const double h2 = ga * ga + gb * gb;
                const bool pos = active && h2 > 0.0;
                const double ri = pos ? inv_sqrt(pos ? h2 : 1.0) : 0.0;
                const double c = pos ? ga * ri : 1.0, s = pos ? gb * ri : 0.0;     // identity for the rows that sit out
                if (active && l > j && l < np) {
                    Rl[j * LDR + l] = c * ra + s * rb;
                    Rl[(j + 1) * LDR + l] = c * rb - s * ra;
                }
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const double q0 = Q[j][rr], q1 = Q[j + 1][rr];
                    Q[j][rr] = c * q0 + s * q1;
                    Q[j + 1][rr] = c * q1 - s * q0;
                }
                const double d0 = row_bcast_c(d, j), d1 = row_bcast_c(d, j + 1);
                if (active && l == j) { d = c * d0 + s * d1; rinv = ri; }
                if (active && l == j + 1) d = c * d1 - s * d0;
            }
        }
        // columns k+1.. of R move one to the left (rows stay); row np-1 now holds the dropped direction (masked by np)
        const int npm = rows_max(rm ? np : 0);
#pragma unroll
        for (int i = 0; i < MAXP - 1; i++) {
            if (i < npm - 1) {
                const double t = Rl[i * LDR + (l < MAXP ? l + 1 : MAXP)];
                if (rm && l >= k && l < np - 1 && i < np - 1) Rl[i * LDR + l] = t;
            }
        }
        {
            const double xn = row_next(x), pn = row_next(xprev);
            const int in = row_next(idx);
            if (rm && l >= k) { x = xn; idx = in; xprev = pn; }
        }
        np -= rm ? 1 : 0;
        if (l >= np) { x = 0.0; xprev = 0.0; d = 0.0; idx = -1; }
    }

    // r = y - Q d of the current passive solution (row space)
    __device__ __forceinline__ void residual(const double (&yr)[NR], double (&r)[NR]) const
    {
        const int npm = rows_max(np);
#pragma unroll
        for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
#pragma unroll
        for (int k = 0; k < MAXP; k++) {
            if (k < npm) {
                const double ck = row_bcast_c(d, k);        // 0 beyond the row's own np
#pragma unroll
                for (int rr = 0; rr < NR; rr++) r[rr] -= Q[k][rr] * ck;
            }
        }
    }

    // As      dictionary tile in LDS (row-major, leading dimension ldA, fp32)
    // yr      row space, 0 on rows >= nS
    // allowed per-lane bit q: atom l + 16 q is admissible
    // live0   row-uniform: this row holds a voxel
    // rs, Rl  per-ROW LDS scratch (kRsWords / kRlWords doubles)
    // G       Gram matrix A'A of the tile's orientation (global, row stride ldG >= 16*NQ) or null
    __device__ __forceinline__ void solve(const float *As, int ldA, int nS, int n_atoms, const double (&yr)[NR],
                                          unsigned allowed, bool live0, double *rs, double *Rl, int lane,
                                          const double *__restrict__ G, int ldG)
    {
        const int l = lane & 15;
        const double inf = __builtin_huge_val();
        const double dep2 = 2e-28;                          // Lawson-Hanson's independence test, see NNSolver::solve
        const double kExactBelow = 1e-7;                    // decisions on smaller dual values use the exact sweep
        constexpr int kMaxGramSteps = AMX_GRAM_STEPS;
        constexpr unsigned kMaskQ = (1u << NQ) - 1u;
        const int itmax = 3 * n_atoms + 10;                 // Lawson-Hanson's cap
        unsigned fl = allowed & kMaskQ;                     // bits [0,NQ) allowed, [NQ,2NQ) passive, [2NQ,3NQ) barred
        np = 0; d = 0.0; rinv = 0.0; x = 0.0; xprev = 0.0; idx = -1; iters = 0; status = kSolved;
        n_exact = 0; n_gram = 0;
        int last_added = -1, second_looks = 0, gram_steps = 0;
        bool cyc_banned = false, force_exact = false;
        bool live = live0;
        bool have_u = false;
#pragma unroll
        for (int q = 0; q < NQ; q++) u[q] = 0.0;
        const int li = (l < MAXP ? l : MAXP) * LDR;         // this lane's row of R

        for (int outer = 0; wany(live); ++outer) {
            if (outer > 2 * itmax) { if (live) status = kGuardOuter; live = false; break; }   // never spin
            // ------------------------------------------------ dual vector u = A'(y - A x)
            const bool exact = (G == nullptr) || !have_u || wany(live && (force_exact || gram_steps >= kMaxGramSteps));
            if (exact) {
                double r[NR];
                residual(yr, r);
#pragma unroll
                for (int rr = 0; rr < NR; rr++) rs[l + kRow * rr] = r[rr];
#pragma unroll
                for (int q = 0; q < NQ; q++) u[q] = 0.0;
                const float *ap = As + l;
                for (int i = 0; i < nS; i++) {
                    const double ri = rs[i];
#pragma unroll
                    for (int q = 0; q < NQ; q++) u[q] += (double)ap[i * ldA + kRow * q] * ri;
                }
                have_u = true; force_exact = false; gram_steps = 0; n_exact++;
            } else {
                // u -= G[:, P] (x - xprev): only the passive coefficients moved
                const double delta = x - xprev;
                const int npm = rows_max(np);
#pragma unroll
                for (int s0 = 0; s0 < MAXP; s0 += 2) {
                    if (s0 < npm) {
                        double gv[2][NQ], dls[2];
#pragma unroll
                        for (int t2 = 0; t2 < 2; t2++) {
                            if (s0 + t2 < MAXP) {
                                dls[t2] = row_bcast_c(delta, s0 + t2);                 // 0 beyond the row's np
                                const int at = max(row_bcast_c(idx, s0 + t2), 0);
                                const double *gc = G + (size_t)at * ldG + l;
#pragma unroll
                                for (int q = 0; q < NQ; q++) gv[t2][q] = gc[kRow * q];
                            }
                        }
#pragma unroll
                        for (int t2 = 0; t2 < 2; t2++) {
                            if (s0 + t2 < MAXP) {
#pragma unroll
                                for (int q = 0; q < NQ; q++) u[q] -= gv[t2][q] * dls[t2];
                            }
                        }
                    }
                }
                gram_steps++; n_gram++;
            }
            xprev = x;

            // ------------------------------------------------ most violating admissible atom of every row; test it
            bool pending = live, added = false, redo = false;
            for (int sel = 0; wany(pending); ++sel) {
                if (sel > kRow * NQ + 2) { if (pending) { status = kGuardSelect; live = false; } pending = false; break; }
                const unsigned cm = fl & ~(fl >> NQ) & ~(fl >> (2 * NQ)) & kMaskQ;
                double best = -inf;
                int bj = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    if (((cm >> q) & 1u) && u[q] > best) { best = u[q]; bj = l + kRow * q; }
                }
                const double wmax = row_max(best);
                if (!exact) {
                    // every admissible dual value is negative by far more than the Gram updates can have drifted
                    const bool clear = pending && (wmax < -kExactBelow);
                    const bool near0 = pending && !clear && !(wmax > kExactBelow);
                    if (near0) { force_exact = true; redo = true; }
                    pending = pending && !clear && !near0;
                }
                pending = pending && (wmax > 0.0);          // Lawson-Hanson's strict rule: KKT point otherwise
                const int t = row_min_i((pending && best == wmax) ? bj : 0x7fffffff);
                if (pending && (t < 0 || t >= n_atoms)) { status = kGuardSelect; live = false; pending = false; }
                if (pending && np >= MAXP) { status = kOverflow; live = false; pending = false; }
                if (!wany(pending)) break;
                const int tc = pending ? t : 0;
                // candidate column (row space)
                double v[NR];
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const int i = l + kRow * rr;
                    v[rr] = (i < nS) ? (double)As[i * ldA + tc] : 0.0;
                }
                double rho = 0.0;                           // lane k: R[k][new]
                const int npm = rows_max(pending ? np : 0);
                // two Gram-Schmidt passes, 4 projections per batched row reduction
#pragma unroll
                for (int pass = 0; pass < 2; pass++) {
#pragma unroll
                    for (int kb = 0; kb < MAXP; kb += 4) {
                        if (kb < npm) {
                            double p[4];
#pragma unroll
                            for (int uu = 0; uu < 4; uu++) {
                                p[uu] = 0.0;
                                if (kb + uu < MAXP) {
#pragma unroll
                                    for (int rr = 0; rr < NR; rr++) p[uu] += Q[kb + uu][rr] * v[rr];
                                }
                            }
                            row_sum4(p, lane);
#pragma unroll
                            for (int uu = 0; uu < 4; uu++) {
                                if (kb + uu < MAXP) {
                                    const double pu = (kb + uu < np) ? p[uu] : 0.0;     // rows of Q beyond np are leftovers
#pragma unroll
                                    for (int rr = 0; rr < NR; rr++) v[rr] -= pu * Q[kb + uu][rr];
                                    if (l == kb + uu) rho += pu;
                                }
                            }
                        }
                    }
                }
                double p[4] = {0.0, 0.0, rho * rho, 0.0};
#pragma unroll
                for (int rr = 0; rr < NR; rr++) { p[0] += v[rr] * v[rr]; p[1] += v[rr] * yr[rr]; }
                row_sum4(p, lane);
                const double b2 = p[0], vy = p[1], un2 = p[2];   // |component outside span(Q)|^2, v'y, |component inside|^2
                bool reject = !(b2 > dep2 * un2) || !(b2 > 0.0);
                const double binv = inv_sqrt(reject ? 1.0 : b2);
                const double beta = b2 * binv, dnew = vy * binv;
                reject = reject || !(dnew * binv > 0.0);         // Lawson-Hanson "ztest"
                const bool acc = pending && !reject, rej = pending && reject;
                if (rej && l == (t & 15)) fl |= 1u << (2 * NQ + (t >> 4));
                // ---- commit column np of the accepting rows
#pragma unroll
                for (int m = 0; m < MAXP; m++) {
                    const bool here = acc && m == np;
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) Q[m][rr] = here ? v[rr] * binv : Q[m][rr];
                }
                if (acc && l <= np) Rl[li + np] = (l == np) ? beta : rho;
                if (acc && l == np) { d = dnew; rinv = binv; x = 0.0; idx = t; }
                if (acc) {
                    fl &= ~(kMaskQ << (2 * NQ)); cyc_banned = false;       // forget the rejected candidates
                    if (l == (t & 15)) fl |= 1u << (NQ + (t >> 4));
                    np += 1; last_added = t; added = true; pending = false;
                }
            }
            {
                // rows that found no atom to add: KKT point -- unless barred atoms deserve a second look (see NNSolver)
                const bool stop = live && !added && !redo;
                const bool look = stop && cyc_banned && second_looks < 3;
                if (look) { fl &= ~(kMaskQ << (2 * NQ)); cyc_banned = false; second_looks++; force_exact = true; last_added = -1; }
                if (stop && !look) live = false;
            }

            // ------------------------------------------------ Lawson-Hanson inner loop of the rows that added an atom
            bool need = added && live;
            while (wany(need)) {
                if (need) iters++;
                if (need && iters > itmax) { status = kIterCap; live = false; need = false; }
                const int npm = rows_max(np);
                double rhs = d;
#pragma unroll
                for (int j = MAXP - 1; j >= 0; j--) {
                    if (j < npm) {
                        const double col = Rl[li + j];
                        const double zj = row_bcast_c(rhs * rinv, j);      // 0 beyond the row's np (rhs = d = 0 there)
                        if (l < j) rhs -= col * zj;
                    }
                }
                const bool act = l < np;
                const double z = act ? rhs * rinv : 0.0;
                const bool neg = need && act && !(z > 0.0);
                const bool anyneg = row_any(neg, lane);
                unsigned rem = 0u;
                if (wany(neg)) {
                    const double den = x - z;
                    const double ratio = neg ? ((den > 0.0) ? x / den : 0.0) : inf;
                    const double alpha = -row_max(-ratio);
                    const int kmin = row_min_i((neg && ratio == alpha) ? l : 99);
                    double xn = act ? x + alpha * (z - x) : 0.0;
                    if (l == kmin) xn = 0.0;
                    if (need) x = anyneg ? xn : (act ? z : 0.0);
                    rem = row_bits(need && anyneg && act && !(x > 0.0), lane);
                } else {
                    if (need) x = act ? z : 0.0;
                }
                for (int guard = 0; wany(rem != 0u) && guard < kRow; ++guard) {
                    const int k = rem ? 31 - __builtin_clz(rem) : -1;
                    if (k >= 0) rem &= ~(1u << k);
                    const int kc = k >= 0 ? k : 0;
                    const int a = row_pick(idx, kc, lane);
                    if (k >= 0 && a == last_added) { cyc_banned = true; if (l == (a & 15)) fl |= 1u << (2 * NQ + (a >> 4)); }   // no add/remove cycling
                    if (G != nullptr) {                    // the atom leaves with coefficient 0: fold its change into u now
                        const double dl = (k >= 0) ? -row_pick(xprev, kc, lane) : 0.0;
                        const double *gc = G + (size_t)max(a, 0) * ldG + l;
#pragma unroll
                        for (int q = 0; q < NQ; q++) u[q] -= gc[kRow * q] * dl;
                    }
                    remove_slot(k, Rl, lane, fl);
                }
                if (np == 0) x = 0.0;
                need = need && anyneg && np > 0;
            }
        }
    }
};

}  // namespace amx
