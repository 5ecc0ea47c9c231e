// amx_gram_solver.hpp -- wavefront-per-voxel non-negative elastic net in GRAM space (lambda2 > 0).
//
// Same problem and same outer logic as NNSolver (amx_solver.hpp):
//     min_x 1/2||y - A diag(s) x||^2 + lambda1*sum(x) + lambda2/2*||x||^2 ,  x >= 0
// but the passive system is solved on H_PP = (S A'A S + lambda2 I)_PP by a Cholesky factor kept in the
// wavefront's LDS block instead of a thin QR of the passive columns in registers.  This is only
// legitimate because the ridge bounds cond(H_PP) (~1e5 for NODDI's LASSO stage): it is what SPAMS'
// LARS does as well.  It is NOT used for the unregularised NNLS stages ("Why not Gram
// space").  What it buys on gfx950: no Q[MAXP][NR] register block (-64..96 VGPRs => more wavefronts per
// SIMD) and no Gram-Schmidt reductions (the new row of the factor comes from MAXP gathered Gram entries
// and one forward substitution).  The dual vector is handled exactly as in NNSolver: Gram-column
// updates between exact sweeps of the LDS tile, and the final KKT decision on the exact vector.
#pragma once
#include "amx_solver.hpp"

namespace amx {

template <int NR, int NQ, int MAXP, typename AT>
struct GramSolver {
    static_assert(MAXP <= kWave, "passive set lives in the lanes of one wavefront");
    // per-wave LDS: lower triangles of H_PP and of its Cholesky factor, packed: (i, j <= i) at i(i+1)/2 + j
    static constexpr int kTri = (MAXP + 1) * (MAXP + 2) / 2;      // one spare row for lanes >= MAXP
    static constexpr int kLdsWords = 2 * kTri;
    double *Hl, *Ll;
    double x, xprev, sc;  // lane s: coefficient, coefficient at the last dual update, column scale
    double cs, linv;      // lane s: (s A'y - lambda1) of the slot's atom, 1 / L_ss
    int idx, np;
    double r[NR];         // row space: residual y - A s x at exit
    int iters, n_exact, n_gram;
    int seeded;           // 1: the seed was certified, 0: refused, -1: none given

    __device__ __forceinline__ static int tri(int i, int j) { return i * (i + 1) / 2 + j; }
    __device__ __forceinline__ static int row(int lane) { const int i = lane < MAXP ? lane : MAXP; return i * (i + 1) / 2; }

    // Cholesky of the np x np matrix in Hl from scratch (left-looking, lane = row)
    __device__ __forceinline__ void refactor(int lane)
    {
        const int ls = row(lane);
        for (int k = 0; k < np; k++) {
            double t = Hl[ls + k];
            for (int m = 0; m < k; m++) t -= Ll[ls + m] * Ll[tri(k, m)];
            const double iv = inv_sqrt(bcast(t, k));
            if (lane >= k && lane < np) Ll[ls + k] = t * iv;
            if (lane == k) linv = iv;
        }
    }

    // z = H_PP^-1 cs by the two triangular solves
    __device__ __forceinline__ double solve_passive(int lane)
    {
        const int ls = row(lane);
        // lane k's right-hand side is final once step k has run (later steps only touch lanes beyond k), so the
        // division by the diagonal happens once per sweep, after the loop, instead of a select in every step
        double f = (lane < np) ? cs : 0.0;
        for (int k = 0; k < np; k++) {
            const double lk = Ll[ls + k];
            const double wk = bcast(f * linv, k);
            if (lane > k) f -= lk * wk;
        }
        double b = f * linv;
        for (int k = np - 1; k >= 0; k--) {
            const double lk = Ll[tri(k, lane < k ? lane : k)];
            const double zk = bcast(b * linv, k);
            if (lane < k) b -= lk * zk;
        }
        return (lane < np) ? b * linv : 0.0;
    }

    __device__ __forceinline__ void remove_slot(int k, int lane, unsigned &fl)
    {
        const int a = bcast_i(idx, k);
        if (lane == (a & 63)) fl &= ~(0x100u << (a >> 6));
        // delete row and column k of H_PP in place: new (s, m <= s) <- old (s + [s>=k], m + [m>=k]);
        // ascending m, and every step reads before it writes, so no entry is consumed after its overwrite
        {
            const int s_new = lane < MAXP ? lane : MAXP;
            const int s_old = (s_new >= k && s_new < MAXP) ? s_new + 1 : s_new;
            for (int m = 0; m < np - 1; m++) {
                const int mo = (m >= k) ? m + 1 : m;
                const double v = Hl[tri(s_old, mo <= s_old ? mo : s_old)];
                if (lane < np - 1 && m <= s_new) Hl[tri(s_new, m)] = v;
            }
        }
        const double xn = from_next_lane(x), sn = from_next_lane(sc), pn = from_next_lane(xprev), cn = from_next_lane(cs);
        const int in = from_next_lane(idx);
        if (lane >= k) { x = xn; sc = sn; idx = in; xprev = pn; cs = cn; }
        np = __builtin_amdgcn_readfirstlane(np - 1);
        if (lane >= np) { x = 0.0; xprev = 0.0; cs = 0.0; idx = -1; }
        refactor(lane);
    }

    // append atom t (scale sct, s * A'y = uyt) as slot np: new row of H_PP and of its Cholesky factor; false if the atom is
    // numerically dependent on the passive ones (cannot happen with lambda2 > 0: d2 >= lambda2)
    __device__ __forceinline__ bool append(int t, double sct, double uyt, double lam1, double lam2, int lane, const double *__restrict__ G, int ldG)
    {
        const int ls = row(lane);
        const double h = (lane < np) ? sc * sct * G[(size_t)idx * ldG + t] : 0.0;
        const double htt = sct * sct * G[(size_t)t * ldG + t] + lam2;
        double hh = h;
        for (int k = 0; k < np; k++) {
            const double lk_ = Ll[ls + k];
            const double lk = bcast(hh * linv, k);
            if (lane > k) hh -= lk_ * lk;
        }
        const double lrow = (lane < np) ? hh * linv : 0.0;           // lane k's hh is final after step k
        const double d2 = htt - wave_sum(lrow * lrow);
        if (!uni(d2 > 1e-13 * htt)) return false;
        const int kn = np;
        const double iv = inv_sqrt(d2);
        if (lane < kn) { Hl[tri(kn, lane)] = h; Ll[tri(kn, lane)] = lrow; }
        if (lane == kn) {
            Hl[tri(kn, kn)] = htt; Ll[tri(kn, kn)] = d2 * iv; linv = iv;
            x = 0.0; xprev = 0.0; sc = sct; idx = t; cs = sct * uyt - lam1;
        }
        np = kn + 1;
        return true;
    }

    // Dense optima (strong ridge -- CylinderZeppelinBall: lambda2 = 4, 22 of 26 atoms): block principal pivoting from the FULL
    // set instead of one Lawson-Hanson addition per atom.  n_atoms <= 64, one atom per lane in atom space (NQ == 1), unit column
    // scales.  Every step rebuilds the factor by appending the atoms of P in ascending order (O(np^2) LDS steps in total), solves,
    // forms the dual vector from the Gram columns and exchanges ALL infeasible atoms (passive with a non-positive coefficient,
    // inactive with a positive dual value) while their number keeps falling, then `kBackup` more times, else only the one with
    // the largest index (Murty's rule: finite for a positive definite H).  Leaves np / idx / x like solve().
    // Cholesky factor of the FULL set -- the same for every voxel of an orientation: built once per workgroup by one wavefront
    // into Lf [kTri] / lf_inv [MAXP] (Hf [kTri]: scratch), read by the first step of solve_dense
    __device__ __forceinline__ void factor_full(int n_atoms, double lam2, double *Hf, double *Lf, double *lf_inv, int lane,
                                                const double *__restrict__ G, int ldG)
    {
        Hl = Hf; Ll = Lf;
        np = 0; x = 0.0; xprev = 0.0; sc = 1.0; cs = 0.0; linv = 0.0; idx = -1;
        for (int t = 0; t < n_atoms && t < MAXP; t++) append(t, 1.0, 0.0, 0.0, lam2, lane, G, ldG);
        if (lane < MAXP) lf_inv[lane] = (lane < np) ? linv : 0.0;
    }

    __device__ __forceinline__ int solve_dense(const AT *As, int ldA, int nS, int n_atoms, const double (&yr)[NR], double lam1,
                                               double lam2, double *rs, double *rl, int lane, const double *__restrict__ G, int ldG,
                                               const double *Lf = nullptr, const double *lf_inv = nullptr)
    {
        static_assert(NQ == 1, "one atom per lane");
        Hl = rl;
        Ll = rl + kTri;
        const double tol = 1e-12;
        constexpr int kBackup = 3;
        iters = 0; n_exact = 1; n_gram = 0;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) rs[lane + kWave * rr] = yr[rr];
        double uy = 0.0;                                             // lane j: a_j'y (lanes >= n_atoms: unused)
        if (lane < n_atoms)
            for (int i = 0; i < nS; i++) uy += (double)As[i * ldA + lane] * rs[i];
        const double cj = uy - lam1;
        unsigned long long P = n_atoms >= 64 ? ~0ull : ((1ull << n_atoms) - 1ull);
        int ninf = n_atoms + 1, backup = 0, status = kSolved;
        for (int it = 0;; ++it) {
            if (it > 4 * n_atoms + 16) { status = kIterCap; x = (x > 0.0) ? x : 0.0; break; }      // (the last iterate may be infeasible: the maps get a feasible one; ST_ITCAP reports it)
            if (it == 0 && Lf != nullptr) {
                // the full set: slot s = atom s, factor shared by the workgroup (read-only)
                Ll = const_cast<double *>(Lf);
                np = n_atoms;
                const bool act = lane < np;
                idx = act ? lane : -1; sc = 1.0; x = 0.0; xprev = 0.0;
                cs = act ? cj : 0.0;
                linv = act ? lf_inv[lane] : 0.0;
            } else {
                Hl = rl; Ll = rl + kTri;
                np = 0; x = 0.0; xprev = 0.0; sc = 1.0; cs = 0.0; linv = 0.0; idx = -1;
                for (unsigned long long rem = P; rem != 0ull; rem &= rem - 1ull) {
                    const int t = __builtin_ctzll(rem);
                    if (np >= MAXP) return kOverflow;
                    if (!append(t, 1.0, bcast(uy, t), lam1, lam2, lane, G, ldG)) P &= ~(1ull << t);
                }
            }
            const double z = solve_passive(lane);
            iters++;
            double g = cj;                                            // dual value of atom `lane`: c_j - sum_s G[j][atom_s] z_s
            for (int s = 0; s < np; s++) {
                const int as = bcast_i(idx, s);
                const double zs = bcast(z, s);
                g -= G[(size_t)as * ldG + lane] * zs;
            }
            n_gram++;
            const bool in_p = (P >> lane) & 1ull;
            const int rank = __builtin_popcountll(P & ((1ull << lane) - 1ull));       // slot of atom `lane`
            const double za = __shfl(z, rank);
            const bool v1 = in_p && !(za > 0.0);
            const bool v2 = !in_p && lane < n_atoms && g > tol;
            const unsigned long long bad = ballot64(v1 || v2);
            x = z;
            if (bad == 0ull) break;                                   // KKT point
            const int nbad = __builtin_popcountll(bad);
            bool block = false;
            if (nbad < ninf) { ninf = nbad; backup = kBackup; block = true; }
            else if (backup > 0) { backup--; block = true; }
            P ^= block ? bad : (1ull << (63 - __builtin_clzll(bad)));
        }
        return status;
    }


    // Certify a passive-set seed (bit mask over the atoms, amx_seed.hpp: k_lasso_seed) in the full problem: Cholesky solve on
    // H_PP = (S A'A S + lambda2 I)_PP from the Gram table -- exactly what the Lawson-Hanson path below does for its last
    // passive set --, then the exact dual vector by one sweep of the tile and the Kuhn-Tucker test of that path (x_P > 0,
    // g_j = s_j a_j'(y - A s x) - lambda1 <= tol for every other admissible atom).  With lambda2 > 0 the problem is strictly
    // convex: a point that passes IS the unique optimum.  false: nothing decided, the caller starts from the empty set.
    __device__ __forceinline__ bool certify_seed(const AT *As, int ldA, int nS, int n_atoms, const double (&yr)[NR], const bool (&rowok)[NR],
                                                 const double (&scl)[NQ], unsigned fl, const unsigned long long (&mask)[4],
                                                 double lam1, double lam2, double tol, double *rs, int lane,
                                                 const double *__restrict__ G, int ldG, const SeedScreen &scr)
    {
        const int n0 = __builtin_popcountll(mask[0]) + __builtin_popcountll(mask[1]) + __builtin_popcountll(mask[2]);
        if (mask[3] != 0ull || n0 > MAXP) return false;
        np = n0; idx = -1; sc = 1.0;
        {
            // slot s = the s-th set bit (ascending atom index), its column scale from atom space
            int s = 0;
#pragma unroll
            for (int w3 = 0; w3 < 3; w3++) {
                for (unsigned long long rem = mask[w3]; rem != 0ull; rem &= rem - 1ull) {
                    const int t = w3 * 64 + __builtin_ctzll(rem);
                    double sct = 1.0;
#pragma unroll
                    for (int q = 0; q < NQ; q++)
                        if (q == (t >> 6)) sct = bcast(scl[q], t & 63);
                    if (lane == s) { idx = t; sc = sct; }
                    s++;
                }
            }
        }
        {
            bool bad = false;
            for (int s = 0; s < np; s++) {
                const int t = bcast_i(idx, s);
                const unsigned ft = (unsigned)bcast_i((int)fl, t & 63);
                bad = bad || !((ft >> (t >> 6)) & 1u) || t >= n_atoms;
            }
            if (bad) { np = 0; idx = -1; return false; }
        }
        x = 0.0; xprev = 0.0; cs = 0.0; linv = 0.0;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
        if (np > 0) {
            for (int t = 0; t < np; t++) {
                const int it = bcast_i(idx, t);
                const double sct = bcast(sc, t);
                if (lane >= t && lane < np) Hl[tri(lane, t)] = sc * sct * G[(size_t)idx * ldG + it] + ((lane == t) ? lam2 : 0.0);
            }
            refactor(lane);
            // c_P = s a_p'y - lambda1: four columns per batched reduction
            for (int s0 = 0; s0 < np; s0 += 4) {
                double p[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = bcast_i(idx, (s0 + u < np) ? s0 + u : np - 1);
                    p[u] = 0.0;
                    if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
                        double col[NR];
                        tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, col);
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) p[u] += col[rr] * yr[rr];
                    } else {
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) {
                            const int i = lane + kWave * rr;
                            if (i < nS && rowok[rr]) p[u] += (double)As[i * ldA + t] * yr[rr];
                        }
                    }
                }
                wave_sum4(p, lane);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (lane == s0 + u) cs = sc * p[u] - lam1;
            }
            if (lane >= np) cs = 0.0;
            x = solve_passive(lane);
            if (ballot64(lane < np && !(x > 0.0)) != 0ull) { np = 0; idx = -1; x = 0.0; cs = 0.0; linv = 0.0; return false; }
            for (int s = 0; s < np; s++) {
                const int t = bcast_i(idx, s);
                const double cx = bcast(sc * x, s);
                if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
                    double col[NR];
                    tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, col);
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) r[rr] -= col[rr] * cx;
                } else {
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) {
                        const int i = lane + kWave * rr;
                        if (i < nS && rowok[rr]) r[rr] -= (double)As[i * ldA + t] * cx;
                    }
                }
            }
        }
        unsigned pm = 0u;
        for (int s = 0; s < np; s++) {
            const int t = bcast_i(idx, s);
            if (lane == (t & 63)) pm |= 1u << (t >> 6);
        }
        bool viol = false;
        if (scr.Sf == nullptr) {
            // exact dual vector
#pragma unroll
            for (int rr = 0; rr < NR; rr++) rs[lane + kWave * rr] = r[rr];
            double u[NQ], w2[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) { u[q] = 0.0; w2[q] = 0.0; }
            tile_sweep<NQ, AT>(As + lane, ldA, nS, rs, u, w2);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const double gq = scl[q] * (u[q] + w2[q]) - lam1;
                viol = viol || ((((fl & ~pm) >> q) & 1u) && gq > tol);
            }
        } else {
            // screened test (see NNSolver::certify_seed): s a_j'r = s2_j'(U2'r) + e_j'r, U2'r = y2~ - S2_P x; atoms whose
            // compressed dual value is below -kappa ||r|| cannot violate, the others get the exact dot product
            constexpr int KDs = 12;
            double rho2 = 0.0;
#pragma unroll
            for (int rr = 0; rr < NR; rr++) rho2 += r[rr] * r[rr];
            rho2 = wave_sum(rho2);
            double rt = (lane < KDs) ? scr.ytil[lane] : 0.0;
            for (int s = 0; s < np; s++) {
                const int t = bcast_i(idx, s);
                const double xs = bcast(x, s);
                if (lane < KDs) rt -= scr.Sg[(size_t)t * KDs + lane] * xs;
            }
            float ut[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) ut[q] = 0.0f;
#pragma unroll
            for (int dd = 0; dd < KDs; dd++) {
                const float rd = (float)bcast(rt, dd);
#pragma unroll
                for (int q = 0; q < NQ; q++) ut[q] += scr.Sf[dd * scr.ld + lane + kWave * q] * rd;
            }
            const float margin = (float)(1.0625 * scr.kappa * sqrt(rho2)), l1f = (float)lam1;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                unsigned long long todo = ballot64((((fl & ~pm) >> q) & 1u) && !(ut[q] - l1f < -margin));
                while (todo != 0ull) {
                    const int tl = __builtin_ctzll(todo);
                    const int t = kWave * q + tl;
                    todo &= todo - 1ull;
                    double p = 0.0;
                    if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
                        double col[NR];
                        tile_column<NR, AT>(As, ldA, nS, t, lane, rowok, col);
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) p += col[rr] * r[rr];
                    } else {
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) {
                            const int i = lane + kWave * rr;
                            if (i < nS && rowok[rr]) p += (double)As[i * ldA + t] * r[rr];
                        }
                    }
                    p = wave_sum(p);
                    const double gt = bcast(scl[q], tl) * p - lam1;
                    viol = viol || (gt > tol);
#ifdef AMX_STATS
                    if (scr.count && lane == 0) atomicAdd(scr.count, 1);
#endif
                }
            }
        }
        n_exact++;
        if (ballot64(viol) != 0ull) { np = 0; idx = -1; x = 0.0; cs = 0.0; linv = 0.0; return false; }
        xprev = x;
        if (lane >= np) { x = 0.0; xprev = 0.0; idx = -1; }
        return true;
    }

    // G: Gram matrix A'A of this orientation restricted to the rows in rowok (REQUIRED here)
    __device__ __forceinline__ int solve(const AT *As, int ldA, int nS, int n_atoms, const double (&yr)[NR],
                                         const bool (&rowok)[NR], const double (&scl)[NQ],
                                         const unsigned long long (&allowed)[NQ], double lam1, double lam2,
                                         double *rs, double *rl, int lane, const double *__restrict__ G, int ldG,
                                         const unsigned long long *seedmask = nullptr, const SeedScreen &scr = SeedScreen())
    {
        Hl = rl;
        Ll = rl + kTri;
        const double tol = 1e-12, inf = __builtin_huge_val();
        const int itmax = 3 * n_atoms + 10;
        constexpr int kMaxGramSteps = 12;
        const double kExactBelow = 1e-7;
        unsigned fl = 0u;
#pragma unroll
        for (int q = 0; q < NQ; q++) fl |= (unsigned)((allowed[q] >> lane) & 1ull) << q;
        np = 0; x = 0.0; xprev = 0.0; sc = 1.0; cs = 0.0; linv = 0.0; idx = -1; iters = 0; n_exact = 0; n_gram = 0;
        seeded = -1;
        if (seedmask != nullptr) {
            const unsigned long long m4[4] = {seedmask[0], seedmask[1], seedmask[2], seedmask[3]};
            seeded = certify_seed(As, ldA, nS, n_atoms, yr, rowok, scl, fl, m4, lam1, lam2, tol, rs, lane, G, ldG, scr) ? 1 : 0;
            if (seeded == 1) return kSolved;
            sc = 1.0;
        }
        int status = kSolved, last_added = -1, gram_steps = 0, second_looks = 0;
        bool cyc_banned = false;
        bool have_u = false, force_exact = false;
        double u[NQ], uy[NQ];                 // atom space: A'r and A'y (unscaled)
#pragma unroll
        for (int q = 0; q < NQ; q++) { u[q] = 0.0; uy[q] = 0.0; }
#ifndef AMX_NO_LASSO_WARM
        // WARM START from a refused seed (round 5).  The voxels that reach this solver with a seed that was not certified -- a support
        // wrong in an atom or two, or the passive set the seed solver held when it gave up at its trip cap (flag word set: then the
        // set is incomplete, not wrong) -- used to start from the EMPTY set: 20 - 30 additions, one factor row and one dual update
        // each, ~700 us of a wavefront for a thousand voxels per million that made up most of the left-over kernel's time.  Any set P
        // whose least-squares solution is positive is a state the active-set method may be in, so: the seed's atoms enter (one factor
        // row each), the atoms whose coefficient comes out non-positive leave (the most negative first) until all are positive, and
        // the method continues from there on an exactly computed dual vector.  The ridge makes the optimum unique: same support.
        if (seedmask != nullptr && np == 0) {
            const int n0 = __builtin_popcountll(seedmask[0]) + __builtin_popcountll(seedmask[1]) + __builtin_popcountll(seedmask[2]);
            if (n0 > 0 && n0 <= MAXP - 2) {
                // A'y first (the new slots' right-hand sides need it): the exact sweep at x = 0
#pragma unroll
                for (int rr = 0; rr < NR; rr++) { r[rr] = yr[rr]; rs[lane + kWave * rr] = r[rr]; }
                double w2[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) { u[q] = 0.0; w2[q] = 0.0; }
                tile_sweep<NQ, AT>(As + lane, ldA, nS, rs, u, w2);
#pragma unroll
                for (int q = 0; q < NQ; q++) { u[q] += w2[q]; uy[q] = u[q]; }
                have_u = true; n_exact++;
#pragma unroll
                for (int w3 = 0; w3 < 3; w3++) {
                    if (w3 >= NQ) break;
                    for (unsigned long long rem = seedmask[w3] & allowed[w3 < NQ ? w3 : 0]; rem != 0ull; rem &= rem - 1ull) {
                        const int tl = __builtin_ctzll(rem), t = w3 * 64 + tl;
                        if (t >= n_atoms) break;
                        double sct = 0.0, uyt = 0.0;
#pragma unroll
                        for (int q = 0; q < NQ; q++)
                            if (q == w3) { sct = bcast(scl[q], tl); uyt = bcast(uy[q], tl); }
                        if (append(t, sct, uyt, lam1, lam2, lane, G, ldG) && lane == tl) fl |= 0x100u << w3;
                    }
                }
                for (int guard = 0; guard <= MAXP && np > 0; guard++) {
                    const double z = solve_passive(lane);
                    const bool neg = lane < np && !(z > 0.0);
                    if (ballot64(neg) == 0ull) { x = (lane < np) ? z : 0.0; break; }
                    const double zmin = wave_min(neg ? z : inf);
                    const unsigned long long who = ballot64(neg && z == zmin);
                    remove_slot(who != 0ull ? __builtin_ctzll(who) : __builtin_ctzll(ballot64(neg)), lane, fl);
                }
                if (np == 0) x = 0.0;
                xprev = x;
                force_exact = true;              // the dual vector of this state comes from the true residual
            }
        }
#endif

        for (int outer = 0; status == kSolved; ++outer) {
            if (outer > 2 * itmax) { status = kGuardOuter; break; }
            const bool exact = !have_u || force_exact || gram_steps >= kMaxGramSteps;
            if (exact) {
                // ---- r = y - A (s x) from the tile columns of the passive atoms, then u = A'r
#pragma unroll
                for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
                for (int sl = 0; sl < np; sl++) {
                    const int a = bcast_i(idx, sl);
                    const double cx = bcast(sc * x, sl);
                    if constexpr (is_global_tile<AT>::value || AMX_TILE_COL_LDS != 0) {
                        double col[NR];
                        tile_column<NR, AT>(As, ldA, nS, a, lane, rowok, col);
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) r[rr] -= col[rr] * cx;
                    } else {
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) {
                            const int i = lane + kWave * rr;
                            if (i < nS && rowok[rr]) r[rr] -= (double)As[i * ldA + a] * cx;
                        }
                    }
                }
#pragma unroll
                for (int rr = 0; rr < NR; rr++) rs[lane + kWave * rr] = r[rr];
                double w2[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) { u[q] = 0.0; w2[q] = 0.0; }
                tile_sweep<NQ, AT>(As + lane, ldA, nS, rs, u, w2);
#pragma unroll
                for (int q = 0; q < NQ; q++) u[q] += w2[q];
                if (!have_u) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) uy[q] = u[q];          // x == 0: u is A'y
                }
                have_u = true; force_exact = false; gram_steps = 0; n_exact++;
            } else {
                const double delta = sc * (x - xprev);
                for (int s0 = 0; s0 < np; s0 += 4) {
                    double gv[4][NQ], dls[4];
#pragma unroll
                    for (int t4 = 0; t4 < 4; t4++) {
                        const int sl = (s0 + t4 < np) ? s0 + t4 : np - 1;
                        const double dv = bcast(delta, sl);
                        dls[t4] = (s0 + t4 < np) ? dv : 0.0;
                        const double *gc = G + (size_t)bcast_i(idx, sl) * ldG + lane;
#pragma unroll
                        for (int q = 0; q < NQ; q++) gv[t4][q] = gc[kWave * q];
                    }
#pragma unroll
                    for (int t4 = 0; t4 < 4; t4++) {
#pragma unroll
                        for (int q = 0; q < NQ; q++) u[q] -= gv[t4][q] * dls[t4];
                    }
                }
                gram_steps++; n_gram++;
            }
            xprev = x;
            double w[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) w[q] = scl[q] * u[q] - lam1;

            // ---- pick the most violating admissible atom
            bool added = false, redo = false;
            for (int sel = 0; status == kSolved && !added; ++sel) {
                if (sel > kWave * NQ + 2) { status = kGuardSelect; break; }
                double best = -inf;
                int bj = -1;
                const unsigned cm = fl & ~(fl >> 8) & ~(fl >> 16);     // bit q: allowed, not passive, not barred
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    if (((cm >> q) & 1u) && w[q] > best) { best = w[q]; bj = lane + kWave * q; }
                }
                const double wmax = wave_max(best);
                if (!exact) {
#ifndef AMX_ALWAYS_CONFIRM
                    if (uni(wmax < -kExactBelow)) break;           // clearly a KKT point: no exact confirmation needed
#endif
                    if (uni(!(wmax > kExactBelow))) { force_exact = true; redo = true; break; }
                }
                if (!uni(wmax > tol)) break;
                const unsigned long long who = ballot64(best == wmax);
                if (uni(who == 0ull)) { status = kGuardSelect; break; }
                const int t = bcast_i(bj, __builtin_ctzll(who));
                if (uni(t < 0 || t >= n_atoms)) { status = kGuardSelect; break; }
                if (np >= MAXP) { status = kOverflow; break; }
                const int tq = t >> 6, tl = t & 63;
                double sct = 0.0, uyt = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q == tq) { sct = bcast(scl[q], tl); uyt = bcast(uy[q], tl); }
                // ---- new row of H_PP and of its Cholesky factor
                const int ls = row(lane);
                const double h = (lane < np) ? sc * sct * G[(size_t)idx * ldG + t] : 0.0;
                const double htt = sct * sct * G[(size_t)t * ldG + t] + lam2;
                double hh = h;
                for (int k = 0; k < np; k++) {
                    const double lk_ = Ll[ls + k];
                    const double lk = bcast(hh * linv, k);
                    if (lane > k) hh -= lk_ * lk;
                }
                const double lrow = (lane < np) ? hh * linv : 0.0;     // lane k's hh is final after step k
                const double d2 = htt - wave_sum(lrow * lrow);
                if (!uni(d2 > 1e-13 * htt)) {                 // cannot happen with lambda2 > 0 (d2 >= lambda2)
                    if (lane == tl) fl |= 0x10000u << tq;
                } else {
                    const int kn = np;
                    const double iv = inv_sqrt(d2);
                    if (lane < kn) { Hl[tri(kn, lane)] = h; Ll[tri(kn, lane)] = lrow; }
                    if (lane == kn) {
                        Hl[tri(kn, kn)] = htt; Ll[tri(kn, kn)] = d2 * iv; linv = iv;
                        x = 0.0; xprev = 0.0; sc = sct; idx = t; cs = sct * uyt - lam1;
                    }
                    fl &= 0xffffu; cyc_banned = false;
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
                    fl &= 0xffffu; cyc_banned = false; second_looks++; force_exact = true; last_added = -1;
                    continue;
                }
                break;   // KKT point (or a guard tripped)
            }

            // ---- Lawson-Hanson inner loop
            for (bool feasible = false; !feasible && status == kSolved;) {
                if (++iters > itmax) { status = kIterCap; break; }
                const double z = solve_passive(lane);
                const bool act = lane < np;
                const bool neg = act && !(z > 0.0);
                // (state changes only inside the removal loop, which makes zero trips when feasible: see NNSolver)
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
                    if (a == last_added) { cyc_banned = true; if (lane == (a & 63)) fl |= 0x10000u << (a >> 6); }
                    {   // the atom leaves with coefficient 0: fold its change into u now
                        const double dl = -bcast(sc * xprev, k);
                        const double *gc = G + (size_t)a * ldG + lane;
#pragma unroll
                        for (int q = 0; q < NQ; q++) u[q] -= gc[kWave * q] * dl;
                    }
                    remove_slot(k, lane, fl);
                }
                if (np == 0) x = 0.0;
                feasible = !any || np == 0;
            }
        }
        return status;
    }
};

}  // namespace amx
