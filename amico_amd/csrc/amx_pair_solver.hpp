// amx_pair_solver.hpp -- TWO voxels per wavefront: non-negative least squares on 32-lane halves (gfx950).
//
// The unregularised solves of NODDI (models.pyx:911 stage 1, :940 stage 3) keep passive sets of <= 8 atoms, so a
// one-wavefront-per-voxel mapping (amx_solver.hpp) uses 8 of 64 lanes in slot space and spends most of its VALU issue
// slots moving data across 64 lanes (six-level reductions, v_readlane broadcasts).  Here every half of a wavefront
// (two 16-lane DPP rows) owns ONE voxel and the two voxels run the same Lawson-Hanson step in lock step:
//   * "row space"  : lane l (0..31) of a half owns signal rows  l, l+32, ...   (NR per lane, nS      <= 32*NR)
//   * "atom space" : lane l owns atoms                          l, l+32, ...   (NQ per lane, n_atoms <= 32*NQ)
//   * "slot space" : lane s (0..15) of EACH of the half's two rows owns the s-th passive atom -- slot data is kept in
//     both rows (same instructions, no extra cost), so a slot value reaches every lane of the half by ONE
//     v_mov_b64_dpp row_newbcast and reductions are 4 DPP steps + one v_permlane16_swap row exchange;
//   * per-voxel decisions (accept / reject / remove) are predicates, never branches: control flow stays wave-uniform
//     (a step is skipped only when neither voxel needs it), so DPP never runs under a partial EXEC mask.
// Same algorithm and decision rules as NNSolver (thin QR of the passive columns in registers by blocked
// Gram-Schmidt with re-orthogonalisation, Givens down-dating, Lawson-Hanson's strict dual rule, z-test and
// independence test, Gram-column updates of the dual vector between exact sweeps); overflow of MAXP goes to the
// wavefront-per-voxel kernel.  (Four voxels per wavefront -- one per DPP row -- need Q[8][7] per lane: with the rest
// of the state that is beyond 256 VGPRs, i.e. one wavefront per SIMD; measured by compilation, not built.)
#pragma once
#include "amx_solver.hpp"

namespace amx {

constexpr int kRow = 16;      // lanes of a DPP row = slots of a passive set

// ------------------------------------------------------------------ 16-lane row primitives
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// v_permlane16_swap of a value with itself returns (rows 0,0,2,2) and (rows 1,1,3,3): the partner row of every half
#define AMX_HALF_SWAP(k, A, B) AMX_ROW_SWAP(__builtin_amdgcn_permlane16_swap, k, A, B)
__device__ __forceinline__ double half_sum(double v)     // every lane of the half ends with the half's sum
{
    double t = v + dpp_zero<0xB1>(v);
    t += dpp_zero<0x4E>(t);
    t += dpp_zero<0x141>(t);
    t += dpp_zero<0x140>(t);
    double a, b;
    AMX_HALF_SWAP(t, a, b);
    return a + b;
}
__device__ __forceinline__ double half_max(double v)
{
    double t = fmax(v, dpp_zero<0xB1>(v));
    t = fmax(t, dpp_zero<0x4E>(t));
    t = fmax(t, dpp_zero<0x141>(t));
    t = fmax(t, dpp_zero<0x140>(t));
    double a, b;
    AMX_HALF_SWAP(t, a, b);
    return fmax(a, b);
}
__device__ __forceinline__ int half_min_i(int v)
{
    int t = min(v, dppi<0xB1>(v));
    t = min(t, dppi<0x4E>(t));
    t = min(t, dppi<0x141>(t));
    t = min(t, dppi<0x140>(t));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
    return min((int)r[0], (int)r[1]);
}
// slot-space reductions: the data is the same in both rows of a half, so the 4 DPP steps inside the row suffice
__device__ __forceinline__ double slot_max(double v)
{
    double t = fmax(v, dpp_zero<0xB1>(v));
    t = fmax(t, dpp_zero<0x4E>(t));
    t = fmax(t, dpp_zero<0x141>(t));
    return fmax(t, dpp_zero<0x140>(t));
}
__device__ __forceinline__ int slot_min_i(int v)
{
    int t = min(v, dppi<0xB1>(v));
    t = min(t, dppi<0x4E>(t));
    t = min(t, dppi<0x141>(t));
    return min(t, dppi<0x140>(t));
}
__device__ __forceinline__ double slot_sum(double v)
{
    double t = v + dpp_zero<0xB1>(v);
    t += dpp_zero<0x4E>(t);
    t += dpp_zero<0x141>(t);
    return t + dpp_zero<0x140>(t);
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
// slot k of the lane's own row, k a per-voxel (not compile-time) index: LDS crossbar, used on the rare removal path only
__device__ __forceinline__ int row_pick(int v, int k, int lane) { return __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, v); }
__device__ __forceinline__ double row_pick(double v, int k, int lane)
{
    return __hiloint2double(row_pick(__double2hiint(v), k, lane), row_pick(__double2loint(v), k, lane));
}
// value of the next lane of the row (lane 15 reads 0): DPP row_shl:1
__device__ __forceinline__ int row_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ double row_next(double v) { return __hiloint2double(row_next(__double2hiint(v)), row_next(__double2loint(v))); }

// four sums over a lane group at once (the transposing reduction of wave_sum4, ended inside the row or after one row
// exchange): every lane of the group gets all four totals
template <bool TWO_ROWS>
__device__ __forceinline__ void group_sum4(double (&p)[4], int lane)
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
    k += dpp_zero<0x118>(k);           // row_shr:8 -> lanes 12..15 of each row hold its totals
    if constexpr (TWO_ROWS) {
        double a, b;
        AMX_HALF_SWAP(k, a, b);
        k = a + b;
    }
    p[0] = row_bcast<12>(k); p[2] = row_bcast<13>(k); p[1] = row_bcast<14>(k); p[3] = row_bcast<15>(k);
}

// wave-uniform "any lane" as a scalar branch condition; per-row "any lane" / bit set as per-lane values
__device__ __forceinline__ bool wany(bool p) { return __ballot(p) != 0ull; }
// ------------------------------------------------------------------ a voxel's lane group: LPV = 32 (two DPP rows) or 16 (one)
template <int LPV>
struct Grp {
    static_assert(LPV == 16 || LPV == 32, "a voxel owns one or two 16-lane DPP rows");
    static __device__ __forceinline__ double sum(double v) { if constexpr (LPV == 32) return half_sum(v); else return slot_sum(v); }
    static __device__ __forceinline__ double vmax(double v) { if constexpr (LPV == 32) return half_max(v); else return slot_max(v); }
    static __device__ __forceinline__ int min_i(int v) { if constexpr (LPV == 32) return half_min_i(v); else return slot_min_i(v); }
    static __device__ __forceinline__ void sum4(double (&p)[4], int lane) { group_sum4<LPV == 32>(p, lane); }
    // bit set of a slot-space predicate (row 0 of the group) / "any lane of the group", as per-lane values
    static __device__ __forceinline__ unsigned slot_bits(bool p, int lane) { return (unsigned)(__ballot(p) >> (lane & (64 - LPV))) & 0xffffu; }
    static __device__ __forceinline__ bool any(bool p, int lane)
    {
        const unsigned long long m = __ballot(p) >> (lane & (64 - LPV));
        return (LPV == 32 ? (unsigned)m : ((unsigned)m & 0xffffu)) != 0u;
    }
    // largest value of a group-uniform int over the groups of the wavefront, as a scalar
    static __device__ __forceinline__ int gmax(int v)
    {
        int m = max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
        if constexpr (LPV == 16) m = max(m, max(__builtin_amdgcn_readlane(v, 16), __builtin_amdgcn_readlane(v, 48)));
        return m;
    }
    // slot-space data is written to LDS by one row of the group
    static __device__ __forceinline__ bool writer(int lane) { return LPV == 16 || (lane & 16) == 0; }
};

// ------------------------------------------------------------------ the solver
template <int LPV, int NR, int NQ, int MAXP>
struct PairNNLS {
    using G_ = Grp<LPV>;
    static_assert(MAXP <= kRow, "the passive set lives in the lanes of one 16-lane row (kept in both rows of the half)");
    static_assert(3 * NQ <= 32, "atom flags are one 32-bit word per lane");
    static constexpr int LDR = MAXP + 1;                    // odd leading dimension of R (MAXP even)
    static constexpr int kRlWords = (MAXP + 1) * LDR;       // doubles of per-row LDS for R
    static constexpr int kRsWords = LPV * NR;               // doubles of per-voxel LDS for the residual broadcast
    double Q[MAXP][NR];   // row space: Q[k][r] = q_k(row l + 32 r); rows >= np hold finite leftovers (masked)
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

    // slot k (half-uniform, -1 = this voxel removes nothing) leaves the passive set of its voxel
    __device__ __forceinline__ void remove_slot(int k, double *Rl, int lane, unsigned &fl)
    {
        const int sl = lane & 15, l = lane & (LPV - 1);
        const bool rm = k >= 0;
        const int kc = rm ? k : 0;
        const int a = row_pick(idx, kc, lane);
        if (rm && l == (a % LPV)) fl &= ~(1u << (NQ + (a / LPV)));
        const int lc = sl < MAXP ? sl : MAXP;               // column MAXP is padding
#pragma unroll
        for (int j = 0; j < MAXP - 1; j++) {
            const bool active = rm && j >= k && j < np - 1;
            if (wany(active)) {
                const double ra = Rl[j * LDR + lc], rb = Rl[(j + 1) * LDR + lc];
                const double ga = row_bcast_c(ra, j + 1), gb = row_bcast_c(rb, j + 1);
                const double h2 = ga * ga + gb * gb;
                const bool pos = active && h2 > 0.0;
                const double ri = pos ? inv_sqrt(pos ? h2 : 1.0) : 0.0;
                const double c = pos ? ga * ri : 1.0, s = pos ? gb * ri : 0.0;     // identity for the voxel that sits out
                if (active && sl > j && sl < np && G_::writer(lane)) {          // (one of the two rows writes)
                    Rl[j * LDR + sl] = c * ra + s * rb;
                    Rl[(j + 1) * LDR + sl] = c * rb - s * ra;
                }
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const double q0 = Q[j][rr], q1 = Q[j + 1][rr];
                    Q[j][rr] = c * q0 + s * q1;
                    Q[j + 1][rr] = c * q1 - s * q0;
                }
                const double d0 = row_bcast_c(d, j), d1 = row_bcast_c(d, j + 1);
                if (active && sl == j) { d = c * d0 + s * d1; rinv = ri; }
                if (active && sl == j + 1) d = c * d1 - s * d0;
            }
        }
        // columns k+1.. of R move one to the left (rows stay); row np-1 now holds the dropped direction (masked by np)
        const int npm = G_::gmax(rm ? np : 0);
#pragma unroll
        for (int i = 0; i < MAXP - 1; i++) {
            if (i < npm - 1) {
                const double t = Rl[i * LDR + (sl < MAXP ? sl + 1 : MAXP)];
                if (rm && sl >= k && sl < np - 1 && i < np - 1 && G_::writer(lane)) Rl[i * LDR + sl] = t;
            }
        }
        {
            const double xn = row_next(x), pn = row_next(xprev);
            const int in = row_next(idx);
            if (rm && sl >= k) { x = xn; idx = in; xprev = pn; }
        }
        np -= rm ? 1 : 0;
        if (sl >= np) { x = 0.0; xprev = 0.0; d = 0.0; idx = -1; }
    }

    // r = y - Q d of the current passive solution (row space)
    __device__ __forceinline__ void residual(const double (&yr)[NR], double (&r)[NR]) const
    {
        const int npm = G_::gmax(np);
#pragma unroll
        for (int rr = 0; rr < NR; rr++) r[rr] = yr[rr];
#pragma unroll
        for (int k = 0; k < MAXP; k++) {
            if (k < npm) {
                const double ck = row_bcast_c(d, k);        // 0 beyond the voxel's own np
#pragma unroll
                for (int rr = 0; rr < NR; rr++) r[rr] -= Q[k][rr] * ck;
            }
        }
    }

    // As      dictionary tile in LDS (row-major, leading dimension ldA, fp32)
    // yr      row space, 0 on rows >= nS
    // allowed per-lane bit q: atom l + 32 q is admissible
    // live0   half-uniform: this half holds a voxel
    // rs, Rl  per-VOXEL LDS scratch (kRsWords / kRlWords doubles)
    // G       Gram matrix A'A of the tile's orientation (global, row stride ldG >= 32*NQ) or null
    //
    // ONE flat loop: every trip advances each live voxel by one step of its own state machine
    //     kDual   -> refresh the dual vector, pick the most violating atom, orthogonalise it, z-test, commit  -> kSolve
    //     kSolve  -> triangular solve; feasible: accept (-> kDual); else step to the boundary                  -> kRemove
    //     kRemove -> one atom leaves (Givens down-date); more to go: stay; else                                -> kSolve
    // A section is skipped (scalar branch) only when NEITHER voxel is in that state.  Besides letting the two voxels
    // progress independently, the flat form keeps every update of the register-resident factor Q in straight-line,
    // predicated code of a single loop: nested loops around those updates made the register allocator keep several
    // copies of Q (measured: +130 VGPRs).
    __device__ __forceinline__ void solve(const float *As, int ldA, int nS, int n_atoms, const double (&yr)[NR],
                                          unsigned allowed, bool live0, double *rs, double *Rl, int lane,
                                          const double *__restrict__ G, int ldG)
    {
        enum : int { kDual = 0, kSolve = 1, kRemove = 2 };
        const int sl = lane & 15, l = lane & (LPV - 1);
        const bool row0 = G_::writer(lane);                 // LDS writes of slot-space data: one row of the group
        const double inf = __builtin_huge_val();
        const double dep2 = 2e-28;                          // Lawson-Hanson's independence test, see NNSolver::solve
        const double kExactBelow = 1e-7;                    // decisions on smaller dual values use the exact sweep
        constexpr int kMaxGramSteps = AMX_GRAM_STEPS;
        constexpr unsigned kMaskQ = (1u << NQ) - 1u;
        const int itmax = 3 * n_atoms + 10;                 // Lawson-Hanson's cap
        unsigned fl = allowed & kMaskQ;                     // bits [0,NQ) allowed, [NQ,2NQ) passive, [2NQ,3NQ) barred
        np = 0; d = 0.0; rinv = 0.0; x = 0.0; xprev = 0.0; idx = -1; iters = 0; status = kSolved;
        n_exact = 0; n_gram = 0;
        int last_added = -1, second_looks = 0, gram_steps = 0, st = kDual;
        unsigned rem = 0u;
        bool cyc_banned = false, force_exact = false, reselect = false, u_exact = false;
        bool live = live0;
        bool have_u = false;
#pragma unroll
        for (int q = 0; q < NQ; q++) u[q] = 0.0;
        const int li = (sl < MAXP ? sl : MAXP) * LDR;       // this lane's row of R

        for (int trip = 0; wany(live); ++trip) {
            if (trip > 8 * itmax) { if (live) status = kGuardOuter; live = false; break; }   // never spin
            // ------------------------------------------------ kDual, part 1: dual vector u = A'(y - A x)
            const bool wantdual = live && st == kDual && !reselect;
            if (wany(wantdual)) {
                const bool exact = (G == nullptr) || !have_u || wany(wantdual && (force_exact || gram_steps >= kMaxGramSteps));
                if (exact) {
                    double r[NR];
                    residual(yr, r);
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) rs[l + LPV * rr] = r[rr];
                    double un[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; q++) un[q] = 0.0;
                    const float *ap = As + l;
                    for (int i = 0; i < nS; i++) {
                        const double ri = rs[i];
#pragma unroll
                        for (int q = 0; q < NQ; q++) un[q] += (double)ap[i * ldA + LPV * q] * ri;
                    }
#pragma unroll
                    for (int q = 0; q < NQ; q++) u[q] = wantdual ? un[q] : u[q];       // a voxel in mid-update keeps its vector
                    have_u = true; n_exact++;
                    if (wantdual) { force_exact = false; gram_steps = 0; u_exact = true; }
                } else {
                    // u -= G[:, P] (x - xprev): only the passive coefficients moved
                    const double delta = wantdual ? x - xprev : 0.0;
                    const int npm = G_::gmax(wantdual ? np : 0);
                    constexpr int GC = (LPV == 32) ? 4 : 2;          // Gram columns in flight (register budget)
#pragma unroll
                    for (int s0 = 0; s0 < MAXP; s0 += GC) {
                        if (s0 < npm) {
                            double gv[GC][NQ], dls[GC];
#pragma unroll
                            for (int t4 = 0; t4 < GC; t4++) {
                                if (s0 + t4 < MAXP) {
                                    dls[t4] = row_bcast_c(delta, s0 + t4);                 // 0 beyond the voxel's np
                                    const int at = max(row_bcast_c(idx, s0 + t4), 0);
                                    const double *gc = G + (size_t)at * ldG + l;
#pragma unroll
                                    for (int q = 0; q < NQ; q++) gv[t4][q] = gc[LPV * q];
                                }
                            }
#pragma unroll
                            for (int t4 = 0; t4 < GC; t4++) {
                                if (s0 + t4 < MAXP) {
#pragma unroll
                                    for (int q = 0; q < NQ; q++) u[q] -= gv[t4][q] * dls[t4];
                                }
                            }
                        }
                    }
                    n_gram++;
                    if (wantdual) { gram_steps++; u_exact = false; }
                }
                if (wantdual) xprev = x;
            }
            reselect = false;

            // ------------------------------------------------ kDual, part 2: most violating admissible atom; test it
            bool pending = live && st == kDual;
            if (wany(pending)) {
                const bool cand = pending;
                const unsigned cm = fl & ~(fl >> NQ) & ~(fl >> (2 * NQ)) & kMaskQ;
                double best = -inf;
                int bj = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    if (((cm >> q) & 1u) && u[q] > best) { best = u[q]; bj = l + LPV * q; }
                }
                const double wmax = G_::vmax(best);
                // on a Gram-updated vector: a clearly negative maximum needs no confirmation, a small one is decided on
                // the exactly recomputed vector
                const bool clear = pending && !u_exact && (wmax < -kExactBelow);
                const bool near0 = pending && !u_exact && !clear && !(wmax > kExactBelow);
                if (near0) force_exact = true;
                pending = pending && !clear && !near0 && (wmax > 0.0);      // Lawson-Hanson's strict rule: KKT point otherwise
                const int t = G_::min_i((pending && best == wmax) ? bj : 0x7fffffff);
                if (pending && (t < 0 || t >= n_atoms)) { status = kGuardSelect; live = false; pending = false; }
                if (pending && np >= MAXP) { status = kOverflow; live = false; pending = false; }
                {
                    // no atom to add and no exact look pending: KKT point -- unless barred atoms deserve a second look
                    const bool stop = cand && live && !pending && !near0;
                    const bool look = stop && cyc_banned && second_looks < 3;
                    if (look) { fl &= ~(kMaskQ << (2 * NQ)); cyc_banned = false; second_looks++; force_exact = true; last_added = -1; }
                    if (stop && !look) live = false;
                }
                if (wany(pending)) {
                    const int tc = pending ? t : 0;
                    // candidate column (row space)
                    double v[NR];
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) {
                        const int i = l + LPV * rr;
                        v[rr] = (i < nS) ? (double)As[i * ldA + tc] : 0.0;
                    }
                    double rho = 0.0;                           // slot k: R[k][new]
                    const int npm = G_::gmax(pending ? np : 0);
                    // two Gram-Schmidt passes, 4 projections per batched reduction
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
                                G_::sum4(p, lane);
#pragma unroll
                                for (int uu = 0; uu < 4; uu++) {
                                    if (kb + uu < MAXP) {
                                        const double pu = (kb + uu < np) ? p[uu] : 0.0;     // rows of Q beyond np are leftovers
#pragma unroll
                                        for (int rr = 0; rr < NR; rr++) v[rr] -= pu * Q[kb + uu][rr];
                                        if (sl == kb + uu) rho += pu;
                                    }
                                }
                            }
                        }
                    }
                    // (rho lives in both rows of the half: only one of them may enter the sum)
                    double p[4] = {0.0, 0.0, row0 ? rho * rho : 0.0, 0.0};
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) { p[0] += v[rr] * v[rr]; p[1] += v[rr] * yr[rr]; }
                    G_::sum4(p, lane);
                    const double b2 = p[0], vy = p[1], un2 = p[2];   // |component outside span(Q)|^2, v'y, |component inside|^2
                    bool reject = !(b2 > dep2 * un2) || !(b2 > 0.0);
                    const double binv = inv_sqrt(reject ? 1.0 : b2);
                    const double beta = b2 * binv, dnew = vy * binv;
                    reject = reject || !(dnew * binv > 0.0);         // Lawson-Hanson "ztest"
                    const bool acc = pending && !reject, rej = pending && reject;
                    if (rej) { reselect = true; if (l == (t % LPV)) fl |= 1u << (2 * NQ + (t / LPV)); }   // next candidate, same dual vector
                    // ---- commit column np of the accepting voxels
#pragma unroll
                    for (int m = 0; m < MAXP; m++) {
                        const bool here = acc && m == np;
#pragma unroll
                        for (int rr = 0; rr < NR; rr++) Q[m][rr] = here ? v[rr] * binv : Q[m][rr];
                    }
                    if (acc && row0 && sl <= np) Rl[li + np] = (sl == np) ? beta : rho;
                    if (acc && sl == np) { d = dnew; rinv = binv; x = 0.0; idx = t; }
                    if (acc) {
                        fl &= ~(kMaskQ << (2 * NQ)); cyc_banned = false;       // forget the rejected candidates
                        if (l == (t % LPV)) fl |= 1u << (NQ + (t / LPV));
                        np += 1; last_added = t; st = kSolve;
                    }
                }
            }

            // ------------------------------------------------ kSolve: Lawson-Hanson inner step
            bool need = live && st == kSolve;
            if (wany(need)) {
                if (need) iters++;
                if (need && iters > itmax) { status = kIterCap; live = false; need = false; }
                const int npm = G_::gmax(np);
                double rhs = d;
#pragma unroll
                for (int j = MAXP - 1; j >= 0; j--) {
                    if (j < npm) {
                        const double col = Rl[li + j];
                        const double zj = row_bcast_c(rhs * rinv, j);      // 0 beyond the voxel's np (rhs = d = 0 there)
                        if (sl < j) rhs -= col * zj;
                    }
                }
                const bool act = sl < np;
                const double z = act ? rhs * rinv : 0.0;
                const bool neg = need && act && !(z > 0.0);
                const bool anyneg = G_::any(neg, lane);
                if (wany(neg)) {
                    const double den = x - z;
                    const double ratio = neg ? ((den > 0.0) ? x / den : 0.0) : inf;
                    const double alpha = -slot_max(-ratio);
                    const int kmin = slot_min_i((neg && ratio == alpha) ? sl : 99);
                    double xn = act ? x + alpha * (z - x) : 0.0;
                    if (sl == kmin) xn = 0.0;
                    if (need) x = anyneg ? xn : (act ? z : 0.0);
                    const unsigned rb = G_::slot_bits(need && anyneg && act && !(x > 0.0), lane);
                    if (need && anyneg) { rem = rb; st = kRemove; }
                } else {
                    if (need) x = act ? z : 0.0;
                }
                if (need && !anyneg) st = kDual;             // feasible: next atom
            }

            // ------------------------------------------------ kRemove: one atom per trip leaves its voxel's passive set
            const bool rmv = live && st == kRemove;
            if (wany(rmv)) {
                const int k = (rmv && rem) ? 31 - __builtin_clz(rem) : -1;
                if (k >= 0) rem &= ~(1u << k);
                const int kc = k >= 0 ? k : 0;
                const int a = row_pick(idx, kc, lane);
                if (k >= 0 && a == last_added) { cyc_banned = true; if (l == (a % LPV)) fl |= 1u << (2 * NQ + (a / LPV)); }   // no add/remove cycling
                if (G != nullptr) {                    // the atom leaves with coefficient 0: fold its change into u now
                    const double dl = (k >= 0) ? -row_pick(xprev, kc, lane) : 0.0;
                    const double *gc = G + (size_t)max(a, 0) * ldG + l;
#pragma unroll
                    for (int q = 0; q < NQ; q++) u[q] -= gc[LPV * q] * dl;
                }
                remove_slot(k, Rl, lane, fl);
                if (rmv && np == 0) x = 0.0;
                if (rmv && rem == 0u) st = (np > 0) ? kSolve : kDual;
            }
        }
    }
};

}  // namespace amx
