// amx_small.hip -- FreeWater / SANDI fit for small dictionaries (n_atoms <= 16): ONE VOXEL PER LANE.
//
// models.pyx:1231-1276 (FreeWater) and :1567-1619 (SANDI) solve, per voxel,
//     min_x 1/2||y - A x||^2 + lambda1*sum(x) + lambda2/2*||x||^2 ,  x >= 0      (cyspams lasso)
// with 11..15 atoms and lambda2 > 0.  A wavefront per voxel (amx_solver.hpp) leaves most lanes idle
// on such problems, so this unit maps one voxel to one LANE:
//   * the workgroup's voxels share one orientation (bucketing) => A (nS x n) and the regularised
//     Gram matrix H = A'A + lambda2*I (n x n, built by the workgroup itself in its prologue) sit in
//     LDS and are read with wave-uniform (broadcast) addresses;
//   * each lane forms c = A'y from its own signal row and runs a Lawson-Hanson active set on
//     (H, c) entirely in registers: the passive set is a bit mask, the passive system is solved by a
//     MASKED Cholesky factorisation (rows/columns outside the set replaced by identity), all loops
//     are fully unrolled over the compile-time dictionary size => no cross-lane traffic, and lane
//     divergence is plain SIMT predication.
// H is well conditioned thanks to the ridge (cond <= ~1e6 for AMICO's defaults), so Gram space is
// safe here (it is NOT for NODDI's unregularised NNLS stages).
#include "amx_launch.hpp"
using namespace amx;

namespace {

template <int N>
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// x / d for well-scaled operands (no denormal / overflow handling: v_rcp_f64, two Newton steps, one residual correction;
// ~9 instructions against ~35 of the IEEE sequence, within 1 ulp)
__device__ __forceinline__ double fast_div(double x, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-d, q, x), r, q);
}

// H / the dictionary are loop invariant: without these barriers the compiler keeps their entries in vector registers
// across the fully unrolled passes (hundreds of VGPRs) and spills everything else
#define AMX_RELOAD() asm volatile("" ::: "memory")

// Cholesky of H restricted to P and the two triangular solves: z = H_PP^-1 cc_P (0 elsewhere).
// The restriction costs ONE select per column: ivm_j = 1 / L_jj for j in P, 0 otherwise.  A zero ivm_j zeroes column j of
// the factor (so no row of P ever sees atom j) and z_j in both substitutions.  Row j itself is then computed from whatever
// the arithmetic gives (finite: sums of products of bounded entries; a negative pivot only feeds the discarded rsqrt) --
// nothing reads it, because every use of row j is multiplied by ivm_j or by z_j = 0.
template <int N>
__device__ __forceinline__ void lane_solve(const double *__restrict__ Hs, const double (&cc)[N], unsigned P, double (&z)[N])
{
    double L[N * (N + 1) / 2], ivm[N];
    AMX_RELOAD();
#pragma unroll
    for (int j = 0; j < N; j++) {
        double s = Hs[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= L[tri<N>(j, k)] * L[tri<N>(j, k)];
        const double iv = ((P >> j) & 1u) ? rsqrt(s) : 0.0;
        ivm[j] = iv;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double tt = Hs[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) tt -= L[tri<N>(i, k)] * L[tri<N>(j, k)];
            L[tri<N>(i, j)] = tt * iv;
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        double s = cc[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= L[tri<N>(j, k)] * z[k];
        z[j] = s * ivm[j];
    }
#pragma unroll
    for (int j = N - 1; j >= 0; j--) {
        double s = z[j];
#pragma unroll
        for (int i = j + 1; i < N; i++) s -= L[tri<N>(i, j)] * z[i];
        z[j] = s * ivm[j];
    }
}

// per-lane NNQP: min 1/2 x'Hx - cc'x, x >= 0 (cc = c - lambda1); Hs in LDS, row-major N x N.
// warm: start from all n_atoms atoms and drop the non-positive ones in blocks before the Lawson-Hanson loop takes over
// (unique optimum with the ridge; see lane_nnqp_rows).  returns 0, or 2 if an iteration cap tripped.
template <int N>
__device__ __forceinline__ int lane_nnqp(const double *__restrict__ Hs, const double (&cc)[N], double (&x)[N],
                                         int n_atoms = N, bool warm = false)
{
    const double tol = 1e-12, inf = __builtin_huge_val();
    double z[N];
    unsigned P = 0u;
    int status = 0;
#pragma unroll
    for (int j = 0; j < N; j++) x[j] = 0.0;
    if (warm) {
        P = (1u << n_atoms) - 1u;
        for (int round = 0; round < N && P != 0u; ++round) {
            lane_solve<N>(Hs, cc, P, z);
            unsigned negm = 0u;
#pragma unroll
            for (int j = 0; j < N; j++)
                if (((P >> j) & 1u) && !(z[j] > 0.0)) negm |= 1u << j;
            if (negm == 0u) {
#pragma unroll
                for (int j = 0; j < N; j++) x[j] = ((P >> j) & 1u) ? z[j] : 0.0;
                break;
            }
            P &= ~negm;
        }
    }
    for (int it = 0; status == 0; ++it) {
        if (it > 3 * N + 8) { status = 2; break; }
        // dual vector g = cc - H x, most violating atom outside the passive set
        AMX_RELOAD();
        double best = -inf;
        int t = -1;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double g = cc[j];
#pragma unroll
            for (int k = 0; k < N; k++) g -= Hs[j * N + k] * x[k];
            if (!((P >> j) & 1u) && g > best) { best = g; t = j; }
        }
        if (!(best > tol)) break;               // KKT point
        P |= 1u << t;
        for (int in = 0;; ++in) {
            if (in > N + 2) { status = 2; break; }
            lane_solve<N>(Hs, cc, P, z);
            bool feasible = true;
#pragma unroll
            for (int j = 0; j < N; j++)
                if (((P >> j) & 1u) && !(z[j] > 0.0)) feasible = false;
            if (feasible) {
#pragma unroll
                for (int j = 0; j < N; j++) x[j] = ((P >> j) & 1u) ? z[j] : 0.0;
                break;
            }
            double alpha = inf;
            int jm = -1;
#pragma unroll
            for (int j = 0; j < N; j++) {
                if (((P >> j) & 1u) && !(z[j] > 0.0)) {
                    const double den = x[j] - z[j];
                    const double r = (den > 0.0) ? x[j] / den : 0.0;
                    if (r < alpha) { alpha = r; jm = j; }
                }
            }
#pragma unroll
            for (int j = 0; j < N; j++) {
                if ((P >> j) & 1u) {
                    x[j] += alpha * (z[j] - x[j]);
                    if (j == jm || !(x[j] > 0.0)) { x[j] = 0.0; P &= ~(1u << j); }
                }
            }
            if (P == 0u) break;
        }
    }
    return status;
}

// workgroup prologue: tile -> LDS, H = A'A + lambda2*I (identity on the padding atoms)
template <int N, typename AT>
__device__ __forceinline__ void small_prologue(const AT *__restrict__ tile, int words, AT *As, double *Hs, int nS,
                                               int ldA, int n_atoms, double lam2)
{
    for (int k = threadIdx.x; k < words; k += blockDim.x) As[k] = tile[k];
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int j = e / N, k = e % N;
        double acc = (j == k) ? ((j < n_atoms) ? lam2 : 1.0) : 0.0;
        if (j < n_atoms && k < n_atoms)
            for (int i = 0; i < nS; i++) acc += (double)As[i * ldA + j] * (double)As[i * ldA + k];
        Hs[e] = acc;
    }
    __syncthreads();
}

// c = A'y for this lane's voxel (+ sum y^2); A read with wave-uniform LDS addresses
template <int N, typename AT>
__device__ __forceinline__ bool lane_aty(const AT *As, const double *__restrict__ yv, int nS, int ldA, int n_atoms,
                                         double (&c)[N], double &ysq)
{
    bool finite = true;
    ysq = 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) c[j] = 0.0;
    // the lane's signal row is strided in memory (one cache line per lane and load): keep sixteen loads in flight
    int i = 0;
    for (; i + 16 <= nS; i += 16) {
        double yb[16];
#pragma unroll
        for (int u = 0; u < 16; u++) yb[u] = yv[i + u];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const double yi = yb[u];
            finite = finite && (fabs(yi) <= 1.79769313486231570e308);
            ysq += yi * yi;
#pragma unroll
            for (int j = 0; j < N; j++)
                if (j < n_atoms) c[j] += (double)As[(i + u) * ldA + j] * yi;
        }
    }
    for (; i < nS; i++) {
        const double yi = yv[i];
        finite = finite && (fabs(yi) <= 1.79769313486231570e308);
        ysq += yi * yi;
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < n_atoms) c[j] += (double)As[i * ldA + j] * yi;
    }
    return finite;
}

template <int N, typename AT>
__device__ __forceinline__ double lane_rss(const AT *As, const double *__restrict__ yv, int nS, int ldA, int n_atoms,
                                           const double (&x)[N])
{
    double rss = 0.0;
    for (int i = 0; i < nS; i++) {
        double e = yv[i];
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < n_atoms) e -= (double)As[i * ldA + j] * x[j];
        rss += e * e;
    }
    return rss;
}

#define AMX_SMALL_LDS(AT)                                                                        \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];                      \
    const int words = a.c.nS * a.c.ldA;                                                          \
    AT *As = reinterpret_cast<AT *>(smem_s);                                                     \
    double *Hs = reinterpret_cast<double *>(smem_s + (((size_t)words * sizeof(AT) + 15) & ~(size_t)15));

template <int N>
__global__ void __launch_bounds__(256) k_freewater_lane(const FwArgs a)
{
    AMX_SMALL_LDS(float)
    const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.c.chunks[cid];
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_perp = a.n_perp;
    small_prologue<N, float>(reinterpret_cast<const float *>(a.c.tiles) + (size_t)ck.dir * a.c.tile_stride, words, As,
                             Hs, nS, ldA, n_atoms, a.c.lam2);
    for (int v = threadIdx.x; v < ck.count; v += blockDim.x) {
        const int vox = a.c.perm[ck.start + v];
        const double *yv = a.c.y + (size_t)vox * nS;
        double c[N], x[N], ysq;
        const bool ok = lane_aty<N, float>(As, yv, nS, ldA, n_atoms, c, ysq);
        double *e = a.est + (size_t)vox * a.n_maps;
        if (!ok) {
            const double nan = __builtin_nan("");
            for (int m = 0; m < a.n_maps; m++) e[m] = nan;
            if (a.rmse) a.rmse[vox] = nan;
            if (a.nrmse) a.nrmse[vox] = nan;
            if (a.ycorr) for (int i = 0; i < nS; i++) a.ycorr[(size_t)vox * nS + i] = nan;
            continue;
        }
#pragma unroll
        for (int j = 0; j < N; j++) c[j] -= a.c.lam1;
#ifdef AMX_FW_SKIP_SOLVE
#pragma unroll
        for (int j = 0; j < N; j++) x[j] = c[j] > 0.0 ? 1e-3 * c[j] : 0.0;
#else
        if (lane_nnqp<N>(Hs, c, x, n_atoms, amx_warm_start(a.c.lam2, a.c.flags)) != 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
#endif
        if (a.c.xdbg) {
#pragma unroll
            for (int j = 0; j < N; j++) if (j < n_atoms) a.c.xdbg[(size_t)vox * n_atoms + j] = x[j];
        }
        // models.pyx:1241-1256
        double x_sum = 0.0, x_perp = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) { x_sum += x[j]; if (j < n_perp) x_perp += x[j]; }
        x_sum += 1e-16;
        const double vv = x_perp / x_sum;
        e[0] = vv; e[1] = 1.0 - vv;
        if (a.is_mouse) {
            double xb = 0.0, xc = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) { if (j == n_perp) xb = x[j]; if (j == n_perp + 1) xc = x[j]; }
            e[2] = xb / x_sum; e[3] = xc / x_sum;
        }
        if (a.rmse || a.nrmse) {
            const double rss = lane_rss<N, float>(As, yv, nS, ldA, n_atoms, x);
            if (a.rmse) a.rmse[vox] = sqrt(rss / (double)nS);
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rss / ysq) : 0.0;
        }
        if (a.ycorr) {                                   // models.pyx:1264-1274
            for (int i = 0; i < nS; i++) {
                double fw = 0.0;
#pragma unroll
                for (int j = 0; j < N; j++)
                    if (j >= n_perp && j < n_atoms) fw += (double)As[i * ldA + j] * x[j];
                const double yc = yv[i] - fw;
                a.ycorr[(size_t)vox * nS + i] = yc < 0.0 ? 0.0 : yc;
            }
        }
    }
}

template <int N>
__global__ void __launch_bounds__(256) k_sandi_lane(const SandiArgs a)
{
    AMX_SMALL_LDS(double)
    const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.c.chunks[cid];
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_rs = a.n_rs, n_in = a.n_in;
    small_prologue<N, double>(reinterpret_cast<const double *>(a.c.tiles), words, As, Hs, nS, ldA, n_atoms, a.c.lam2);
    for (int v = threadIdx.x; v < ck.count; v += blockDim.x) {
        const int vox = a.c.perm[ck.start + v];
        const double *yv = a.c.y + (size_t)vox * nS;
        double c[N], x[N], ysq;
        const bool ok = lane_aty<N, double>(As, yv, nS, ldA, n_atoms, c, ysq);
        double *e = a.est + (size_t)vox * 6;
        if (!ok) {
            const double nan = __builtin_nan("");
            for (int m = 0; m < 6; m++) e[m] = nan;
            if (a.rmse) a.rmse[vox] = nan;
            if (a.nrmse) a.nrmse[vox] = nan;
            continue;
        }
#pragma unroll
        for (int j = 0; j < N; j++) c[j] -= a.c.lam1;
        if (lane_nnqp<N>(Hs, c, x, n_atoms, amx_warm_start(a.c.lam2, a.c.flags)) != 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
        // models.pyx:1570-1612
        double x_sum = 0.0, xsph = 0.0, xstk = 0.0, xiso = 0.0, Rsoma = 0.0, Din = 0.0, De = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (j < n_atoms) {
                x[j] *= a.norms[j];
                x_sum += x[j];
                if (j < n_rs) { xsph += x[j]; Rsoma += a.Rs[j] * x[j]; }
                else if (j < n_rs + n_in) { xstk += x[j]; Din += a.d_in[j - n_rs] * x[j]; }
                else { xiso += x[j]; De += a.d_isos[j - n_rs - n_in] * x[j]; }
            }
        }
        if (a.c.xdbg) {                                   // the rescaled x (models.pyx:1570-1571)
#pragma unroll
            for (int j = 0; j < N; j++) if (j < n_atoms) a.c.xdbg[(size_t)vox * n_atoms + j] = x[j];
        }
        x_sum += 1e-16;
        e[0] = fast_div(xsph, x_sum); e[1] = fast_div(xstk, x_sum); e[2] = fast_div(xiso, x_sum);
        e[3] = 1e6 * fast_div(Rsoma, xsph + 1e-16);
        e[4] = 1e3 * fast_div(Din, xstk + 1e-16);
        e[5] = 1e3 * fast_div(De, xiso + 1e-16);
        if (a.rmse || a.nrmse) {
            // quirk kept (models.pyx:1571 then 1615): errors use the RESCALED x with the NORMALISED A
            const double rss = lane_rss<N, double>(As, yv, nS, ldA, n_atoms, x);
            if (a.rmse) a.rmse[vox] = sqrt(rss / (double)nS);
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rss / ysq) : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-space solver of the SANDI problem (M = 6 values per voxel after the directional average, N = 15 atoms, ONE dictionary
// for all voxels):  min 1/2 ||y - A x||^2 + lambda1 sum(x) + lambda2/2 ||x||^2, x >= 0, lambda2 > 0.
// On a passive set P the solution is x_P = (c_P - A_P' w) / lambda2 with w = B^-1 A_P c_P, B = lambda2 I + A_P A_P' (Woodbury:
// a 6 x 6 Cholesky instead of a |P| x |P| one).  For j outside P the same expression is the dual value: A_P x_P = w exactly,
// so g_j = c_j - a_j'w -- the KKT test and the choice of the entering atom cost nothing extra.
// Tables of the dictionary (k_sandi_tables, once per (dictionary, lambda1, lambda2)), read with wave-uniform addresses (scalar loads):
//   T [N][kRowsTs]  packed lower triangles of a_j a_j'          (B is summed from them: no +- drift, half the arithmetic)
//   G [N][M], g0 [N]  z0 = G y + g0 = the unconstrained optimum on the FULL set (G = A' (lambda2 I + A A')^-1)
// Warm start: SANDI's optimum is dense (12 of 15 atoms), so the method starts from P0 = {z0 > 0} -- the full-set solve and
// the first block removal are one tabulated map -- and continues by block principal pivoting (below).  With lambda2 > 0
// the optimum is unique, so the path does not matter; the result satisfies the KKT conditions to 1e-12.
constexpr int kRowsTs = 22;                // stride of T: 21 entries of the 6 x 6 triangle, padded for 16-byte reads

template <int M, int N, typename TP>
__device__ __forceinline__ int lane_nnqp_rows(TP A, int ldA, TP T, TP G, TP g0, const double (&y)[M],
                                              double lam1, double lam2, double (&x)[N], int n_atoms, bool warm)
{
    static_assert(M * (M + 1) / 2 <= kRowsTs, "triangle of a_j a_j' fits its table row");
    constexpr int kTri = M * (M + 1) / 2;
    const double tol = 1e-12, il2 = 1.0 / lam2;
    // c = A'y - lambda1 is never stored (15 doubles = 30 registers the Cholesky would have to live with): c_j is recomputed for
    // the atoms that change sides (below), the dual values are g_j = a_j'(y - w) - lambda1.
    unsigned P = 0u;
    AMX_RELOAD();
#pragma unroll
    for (int j = 0; j < N; j++) {
        x[j] = 0.0;
        double sz = g0[j];
#pragma unroll
        for (int i = 0; i < M; i++) sz += G[j * M + i] * y[i];
        if (warm && j < n_atoms && sz > 0.0) P |= 1u << j;
    }
    constexpr int kBackup = 3;               // block exchanges allowed without progress (Kim & Park)
    int ninf = N + 1, backup = 0;
    // A_P A_P' and the right-hand side A_P c_P are CARRIED from trip to trip: only the atoms that changed sides are added or
    // subtracted (rank-one terms from the table, a_j c_j with c_j = a_j'y - lambda1 recomputed on the spot).  The branch per atom
    // is wave-uniform (ballot over the lanes still iterating): in the late trips of a lock-step wavefront few lanes are left and
    // they exchange one or two atoms each.  (+- accumulation: a handful of updates per voxel, errors of 1e-16 relative.)
    double Bp[kTri], rp[M];
#pragma unroll
    for (int t = 0; t < kTri; t++) Bp[t] = 0.0;
#pragma unroll
    for (int i = 0; i < M; i++) rp[i] = 0.0;
    unsigned flips = P;
    for (int it = 0;; ++it) {
        if (it > 4 * N + 16) return 2;
        double B[kTri], L[kTri], li[M], w[M];
        AMX_RELOAD();
#pragma unroll
        for (int j = 0; j < N; j++) {
            const bool fj = (flips >> j) & 1u;
            if (__ballot(fj) != 0ull) {
                const double dj = fj ? (((P >> j) & 1u) ? 1.0 : -1.0) : 0.0;
                double cj = -lam1;
#pragma unroll
                for (int i = 0; i < M; i++) cj += A[i * ldA + j] * y[i];
                cj *= dj;
#pragma unroll
                for (int t = 0; t < kTri; t++) Bp[t] += dj * T[j * kRowsTs + t];
#pragma unroll
                for (int i = 0; i < M; i++) rp[i] += cj * A[i * ldA + j];
            }
        }
#pragma unroll
        for (int t = 0; t < kTri; t++) B[t] = Bp[t];
#pragma unroll
        for (int i = 0; i < M; i++) { w[i] = rp[i]; B[tri<M>(i, i)] += lam2; }
#pragma unroll
        for (int j = 0; j < M; j++) {
            double d = B[tri<M>(j, j)];
#pragma unroll
            for (int k = 0; k < j; k++) d -= L[tri<M>(j, k)] * L[tri<M>(j, k)];
            const double iv = rsqrt(d);
            li[j] = iv;
#pragma unroll
            for (int i = j + 1; i < M; i++) {
                double tt = B[tri<M>(i, j)];
#pragma unroll
                for (int k = 0; k < j; k++) tt -= L[tri<M>(i, k)] * L[tri<M>(j, k)];
                L[tri<M>(i, j)] = tt * iv;
            }
        }
#pragma unroll
        for (int j = 0; j < M; j++) {
            double sacc = w[j];
#pragma unroll
            for (int k = 0; k < j; k++) sacc -= L[tri<M>(j, k)] * w[k];
            w[j] = sacc * li[j];
        }
#pragma unroll
        for (int j = M - 1; j >= 0; j--) {
            double sacc = w[j];
#pragma unroll
            for (int i = j + 1; i < M; i++) sacc -= L[tri<M>(i, j)] * w[i];
            w[j] = sacc * li[j];
        }
        AMX_RELOAD();
        // block principal pivoting: passive atoms with a non-positive coefficient leave, inactive atoms with a positive dual
        // value enter, all at once while the number of infeasibilities keeps falling (then kBackup more times); otherwise only
        // the infeasible atom with the largest index is exchanged (Murty's rule).  (Launched with the warm start only: smaller
        // lambda2 / AMX_COLD_START=1 go to k_sandi_lane's Lawson-Hanson loop, amx_launch_sandi_small.)
        unsigned v1 = 0u, v2 = 0u;
#pragma unroll
        for (int i = 0; i < M; i++) w[i] = y[i] - w[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            double g = -lam1;
#pragma unroll
            for (int i = 0; i < M; i++) g += A[i * ldA + j] * w[i];
            const bool pj = (P >> j) & 1u;
            x[j] = pj ? g * il2 : 0.0;
            if (pj && !(g > 0.0)) v1 |= 1u << j;
            if (!pj && j < n_atoms && g > tol) v2 |= 1u << j;
        }
        const unsigned bad = v1 | v2;
        if (bad == 0u) return 0;                                   // KKT point: x holds the solution
        const int nbad = __builtin_popcount(bad);
        bool block = false;
        if (nbad < ninf) { ninf = nbad; backup = warm ? kBackup : 0; block = warm; }
        else if (backup > 0) { backup--; block = true; }
        flips = block ? bad : (1u << (31 - __builtin_clz(bad)));
        P ^= flips;
    }
}

// T, G, g0 of one dictionary and one (lambda1, lambda2): one workgroup, thread 0 inverts the 6 x 6 (Gauss-Jordan on the SPD
// matrix, no pivoting needed).  out: T [N][kRowsTs] | G [N][M] | g0 [16] | A [M][16] (zero-padded rows)
template <int M, int N>
__global__ void __launch_bounds__(64) k_sandi_tables(const double *__restrict__ Ag, int ldA, int n_atoms, double lam1, double lam2,
                                                     double *__restrict__ out)
{
    __shared__ double A[M * 16], W[M * M], Bm[M * 2 * M];
    for (int e = threadIdx.x; e < M * 16; e += blockDim.x) A[e] = ((e % 16) < n_atoms) ? Ag[(e / 16) * ldA + (e % 16)] : 0.0;
    __syncthreads();
    double *T = out, *G = out + N * kRowsTs, *g0 = G + N * M;
    for (int e = threadIdx.x; e < M * 16; e += blockDim.x) g0[16 + e] = A[e];                // the dictionary, rows padded to 16
    for (int e = threadIdx.x; e < N * kRowsTs; e += blockDim.x) {
        const int j = e / kRowsTs, t = e % kRowsTs;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) i++;                 // t = i (i + 1) / 2 + k
        const int k = t - i * (i + 1) / 2;
        T[e] = (t < M * (M + 1) / 2) ? A[i * 16 + j] * A[k * 16 + j] : 0.0;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < M; i++)
            for (int k = 0; k < M; k++) {
                double acc = (i == k) ? lam2 : 0.0;
                for (int j = 0; j < N; j++) acc += A[i * 16 + j] * A[k * 16 + j];
                Bm[i * 2 * M + k] = acc; Bm[i * 2 * M + M + k] = (i == k) ? 1.0 : 0.0;
            }
        for (int c0 = 0; c0 < M; c0++) {
            const double pv = 1.0 / Bm[c0 * 2 * M + c0];
            for (int k = 0; k < 2 * M; k++) Bm[c0 * 2 * M + k] *= pv;
            for (int i = 0; i < M; i++)
                if (i != c0) {
                    const double f = Bm[i * 2 * M + c0];
                    for (int k = 0; k < 2 * M; k++) Bm[i * 2 * M + k] -= f * Bm[c0 * 2 * M + k];
                }
        }
        for (int i = 0; i < M; i++) for (int k = 0; k < M; k++) W[i * M + k] = Bm[i * 2 * M + M + k];
    }
    __syncthreads();
    // z0 = (c - A' W A c) / lambda2 with c = A'y - lambda1 1 and A A' = W^-1 - lambda2 I:
    //    = A' W y - (lambda1 / lambda2) (1 - A' W A 1)
    for (int e = threadIdx.x; e < N * M; e += blockDim.x) {
        const int j = e / M, i = e % M;
        double acc = 0.0;
        for (int k = 0; k < M; k++) acc += A[k * 16 + j] * W[k * M + i];
        G[e] = acc;
    }
    for (int j = threadIdx.x; j < 16; j += blockDim.x) {
        double acc = 0.0;
        if (j < n_atoms) {
            double awa = 0.0;
            for (int i = 0; i < M; i++) {
                double wi = 0.0;
                for (int k = 0; k < M; k++) { double a1 = 0.0; for (int jj = 0; jj < n_atoms; jj++) a1 += A[k * 16 + jj]; wi += W[i * M + k] * a1; }
                awa += A[i * 16 + j] * wi;
            }
            acc = -(lam1 / lam2) * (1.0 - awa);
        }
        g0[j] = acc;
    }
}
constexpr int kSandiTableWords = 15 * kRowsTs + 15 * 6 + 16 + 8 * 16;

#ifndef AMX_ROWS_OCC
#define AMX_ROWS_OCC 3
#endif

// SANDI, nS == M (<= 8) values per voxel: row-space solver; the dictionary and its tables come through the scalar cache
// (wave-uniform addresses), y from the voxel's row.
template <int M, int N>
__global__ void __launch_bounds__(256, AMX_ROWS_OCC) k_sandi_rows(const SandiArgs a)
{
    Chunk ck;
    const bool linear = a.n_lin > 0;           // SANDI has one dictionary: nothing to bucket, the voxels are taken in order
    if (linear) {
        ck.start = (int)blockIdx.x * 256; ck.count = a.n_lin - ck.start < 256 ? a.n_lin - ck.start : 256;
        if (ck.count <= 0) return;
    } else {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        ck = a.c.chunks[cid];
    }
    const int n_atoms = a.c.n_atoms, n_rs = a.n_rs, n_in = a.n_in;
    // the dictionary and its tables (4.6 KB, the same for every voxel) are read through the SCALAR cache: every address is
    // wave-uniform, so the loads are s_load_dwordx8/x16 into SGPRs and the fused multiply-adds take them as their scalar
    // operand -- no LDS instruction in the solver (from LDS the ~210 16-byte broadcast reads per trip cost as much of the
    // CU's time as the ~650 fp64 instructions they feed).
    using CD = const __attribute__((address_space(4))) double;
    constexpr int ldA = 16;
    CD *T = (CD *)a.tables, *G = T + N * kRowsTs, *g0 = G + N * M, *A = g0 + 16;
    const bool warm = amx_warm_start(a.c.lam2, a.c.flags);
    // the atoms' norms and model parameters (Rs | d_in | d_isos by atom class) once per workgroup: read in the maps section below
    // from their four arrays, every value was a load of its own under a wave-uniform guard, waited for before the next one left --
    // ~45 memory round trips one after the other per wavefront, as long as the solver itself
    __shared__ double s_par[2][16];
    if (threadIdx.x < 16) {
        const int j = threadIdx.x;
        s_par[0][j] = j < n_atoms ? a.norms[j] : 0.0;
        s_par[1][j] = j < n_rs ? a.Rs[j] : (j < n_rs + n_in ? a.d_in[j - n_rs] : (j < n_atoms ? a.d_isos[j - n_rs - n_in] : 0.0));
    }
    __syncthreads();
    for (int v = threadIdx.x; v < ck.count; v += blockDim.x) {
        const int vox = linear ? ck.start + v : a.c.perm[ck.start + v];
        const double *yv = a.c.y + (size_t)vox * M;
        double y[M], x[N], ysq = 0.0;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < M; i++) {
            y[i] = yv[i];
            ok = ok && (fabs(y[i]) <= 1.79769313486231570e308);
            ysq += y[i] * y[i];
        }
        double *e = a.est + (size_t)vox * 6;
        if (!ok) {
            const double nan = __builtin_nan("");
            for (int m = 0; m < 6; m++) e[m] = nan;
            if (a.rmse) a.rmse[vox] = nan;
            if (a.nrmse) a.nrmse[vox] = nan;
            continue;
        }
        if (lane_nnqp_rows<M, N>(A, ldA, T, G, g0, y, a.c.lam1, a.c.lam2, x, n_atoms, warm) != 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
        // models.pyx:1570-1612
        double x_sum = 0.0, xsph = 0.0, xstk = 0.0, xiso = 0.0, Rsoma = 0.0, Din = 0.0, De = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const double nj = s_par[0][j < 16 ? j : 0], pj = s_par[1][j < 16 ? j : 0];
            if (j < n_atoms) {
                x[j] *= nj;
                x_sum += x[j];
                if (j < n_rs) { xsph += x[j]; Rsoma += pj * x[j]; }
                else if (j < n_rs + n_in) { xstk += x[j]; Din += pj * x[j]; }
                else { xiso += x[j]; De += pj * x[j]; }
            }
        }
        if (a.c.xdbg) {                                   // the rescaled x (models.pyx:1570-1571)
#pragma unroll
            for (int j = 0; j < N; j++) if (j < n_atoms) a.c.xdbg[(size_t)vox * n_atoms + j] = x[j];
        }
        x_sum += 1e-16;
        e[0] = fast_div(xsph, x_sum); e[1] = fast_div(xstk, x_sum); e[2] = fast_div(xiso, x_sum);
        e[3] = 1e6 * fast_div(Rsoma, xsph + 1e-16);
        e[4] = 1e3 * fast_div(Din, xstk + 1e-16);
        e[5] = 1e3 * fast_div(De, xiso + 1e-16);
        if (a.rmse || a.nrmse) {
            // quirk kept (models.pyx:1571 then 1615): errors use the RESCALED x with the NORMALISED A
            double rss = 0.0;
#pragma unroll
            for (int i = 0; i < M; i++) {
                double ei = y[i];
#pragma unroll
                for (int j = 0; j < N; j++) ei -= A[i * ldA + j] * x[j];
                rss += ei * ei;
            }
            if (a.rmse) a.rmse[vox] = sqrt(rss / (double)M);
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rss / ysq) : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// FreeWater, lanes that never idle: k_freewater_refill.
//
// k_freewater_lane runs one active-set solve per lane and a wavefront waits for its slowest lane: voxels need 3..20
// steps (one Cholesky each), so a pass over 64 voxels costs the MAXIMUM step count (measured: ~23 steps' worth of
// instructions for a mean of ~8), and every lane walks its own 520-byte signal row with 8-byte loads (one cache line
// per lane and load; 64 rows = 33 KB per wavefront thrash the 32 KB L1).  Here
//   * phase 1 (per batch of 64 voxels of the wavefront): the signal rows are read with COALESCED 16-byte loads (8
//     lanes per row segment of 16 values), transposed through a small LDS tile (odd leading dimension: conflict-free
//     both ways) and contracted with the fp64 dictionary to c = A'y by the lane that owns the voxel; the c vectors
//     wait in an LDS buffer of the wavefront;
//   * phase 2: every lane runs ONE active-set step per trip of a flat loop (select an atom if its solution is
//     feasible, masked Cholesky + solves, accept or step back); a lane whose voxel reached its KKT point writes the
//     maps and takes the next c vector from the buffer at once -- the wavefront refills the buffer (phase 1) when it
//     runs dry.  The same per-voxel arithmetic as lane_nnqp, in the same order.
// Chunks of amx_refill_chunk() voxels of one orientation per workgroup (the buffer needs a pool to draw from).
#ifdef AMX_FW_PHASES
// diagnosis builds only (tools/fw_phases.py): s_memtime split of the refill kernel, summed over wavefronts
__device__ unsigned long long g_fw_ph[8], g_fw_pp[8];
#define FWPH_T() __builtin_readcyclecounter()
#define FWPH_ADD(k, t0) ph[k] += __builtin_readcyclecounter() - (t0)
#else
#define FWPH_T() 0ull
#define FWPH_ADD(k, t0) (void)(t0)
#endif
constexpr int kTileRows = 16;          // signal values per voxel and transposition pass
constexpr int kTileLd = 65;            // odd leading dimension (doubles) of the transposition tile [row][voxel]

// per orientation, once per (dictionary, lambda2): the fp64 dictionary A [nS][NP], H^-1 [N][NP] and H = A'A + lambda2 I
// [N][N] (NP = N rounded up to even).  One wavefront per orientation.
template <int N> constexpr int fw_prep_words(int nS) { return (nS * ((N + 1) & ~1) + N * ((N + 1) & ~1) + N * N + 1) & ~1; }   // even: 16-byte aligned tables

template <int N>
__global__ void __launch_bounds__(64) k_fw_orient_prep(const float *__restrict__ tiles, int tile_stride, int ldA, int nS, int n_atoms,
                                                       double lam2, double *__restrict__ prep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_o[];
    constexpr int NP = (N + 1) & ~1;
    double *Ad = reinterpret_cast<double *>(smem_o);
    double *Hs = Ad + (size_t)nS * NP;
    const float *tile = tiles + (size_t)blockIdx.x * tile_stride;
    double *out = prep + (size_t)blockIdx.x * fw_prep_words<N>(nS);
    double *Ag = out, *Ig = out + (size_t)nS * NP, *Hg = Ig + N * NP;
    for (int e = threadIdx.x; e < nS * NP; e += blockDim.x) {
        const int i = e / NP, j = e % NP;
        const double v = (j < n_atoms) ? (double)tile[i * ldA + j] : 0.0;
        Ad[e] = v; Ag[e] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int j = e / N, k = e % N;
        double acc = (j == k) ? ((j < n_atoms) ? lam2 : 1.0) : 0.0;
        if (j < n_atoms && k < n_atoms)
            for (int i = 0; i < nS; i++) acc += Ad[i * NP + j] * Ad[i * NP + k];
        Hs[e] = acc; Hg[e] = acc;
    }
    __syncthreads();
    if (threadIdx.x < N) {                       // row r of the (symmetric) inverse: H z = e_r
        const int r = threadIdx.x;
        double rhs[N], z[N];
#pragma unroll
        for (int j = 0; j < N; j++) rhs[j] = (j == r) ? 1.0 : 0.0;
        lane_solve<N>(Hs, rhs, (1u << N) - 1u, z);
#pragma unroll
        for (int j = 0; j < N; j++) Ig[r * NP + j] = z[j];
        if (NP > N) Ig[r * NP + N] = 0.0;
    }
}

// phase 1 as its own launch: c = A'y - lambda1 for every voxel, in bucket order, [ldC][NP] (voxel-major: a batch of 64
// voxels is one contiguous 6 KB block that the solver streams into its LDS with direct global->LDS loads).  A streaming kernel (520 B in, 92 B out per voxel) at four wavefronts per SIMD: the signal loads of one
// wavefront overlap the contractions of the others, which the solver's two wavefronts per SIMD could not do.
//
// It also takes the first step of the warm-started active-set method off the solver's hands: the unconstrained optimum on
// the full set is z0 = H^-1 c, and all the solver needs from it is which coefficients came out positive -- the passive set
// after the first block removal, p0 [ldC] (11 bits per voxel).  H^-1 is tabulated per orientation; the explicit inverse
// (cond(H) ~ 1e5 with the ridge) is good enough for a starting guess, the solver's answer does not depend on it.
#ifndef AMX_FW_PROJ_OCC
#define AMX_FW_PROJ_OCC 4
#endif
template <int N>
__global__ void __launch_bounds__(256, AMX_FW_PROJ_OCC) k_fw_project(const FwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.c.chunks[cid];
    const int nS = a.c.nS, n_atoms = a.c.n_atoms;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    constexpr int NP = (N + 1) & ~1;          // row length of Ad / Hi: even, so that a row is read with 16-byte LDS loads
    // LDS: Ad f64 [nS][NP] | Hi f64 [N][NP] | per wavefront: Tt [kTileRows][kTileLd], Vb int[64]
    double *Ad = reinterpret_cast<double *>(smem_p);
    double *Hi = Ad + (size_t)nS * NP;
    constexpr int kWaveWords = kTileRows * kTileLd + 32;
    double *Tt = Hi + N * NP + (size_t)wave * kWaveWords;
    int *Vb = reinterpret_cast<int *>(Tt + kTileRows * kTileLd);
    {
        const double *src = a.prep + (size_t)ck.dir * fw_prep_words<N>(nS);
        for (int e = threadIdx.x; e < (nS + N) * NP; e += blockDim.x) Ad[e] = src[e];                    // A then H^-1
        __syncthreads();
    }
    const int n_batches = (ck.count + 63) >> 6;
    const int seg = lane & 7, grp = lane >> 3;          // 8 lanes x 16 B = one 16-value segment of a row
#ifdef AMX_FW_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_all = FWPH_T();
#endif
    for (int b = wave; b < n_batches; b += nw) {
        unsigned long long t0 = FWPH_T();
        const int cnt = min(64, ck.count - (b << 6));
        const int pos = ck.start + (b << 6) + lane;
        Vb[lane] = (lane < cnt) ? a.c.perm[pos] : -1;
        int vrow[8];                                       // the eight voxels this lane loads segments of
#pragma unroll
        for (int it = 0; it < 8; it++) vrow[it] = Vb[it * 8 + grp];
        // the loads of the next 16 signal values are in flight while the current 16 are contracted
        double yn[8][2];
        auto issue = [&](int r0) {
            const int rows = min(kTileRows, nS - r0);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                double y0 = 0.0, y1 = 0.0;
                if (vrow[it] >= 0) {
                    const double *yp = a.c.y + (size_t)vrow[it] * nS + (r0 + 2 * seg);
                    if (2 * seg + 1 < rows) { y0 = yp[0]; y1 = yp[1]; }       // one 16-byte load
                    else if (2 * seg < rows) y0 = yp[0];
                }
                yn[it][0] = y0; yn[it][1] = y1;
            }
        };
        double cn[NP];
        bool finite = true;
#pragma unroll
        for (int j = 0; j < NP; j++) cn[j] = 0.0;
        issue(0);
        FWPH_ADD(0, t0);
        for (int r0 = 0; r0 < nS; r0 += kTileRows) {
            const int rows = min(kTileRows, nS - r0);
            t0 = FWPH_T();
#pragma unroll
            for (int it = 0; it < 8; it++) {
                Tt[(2 * seg) * kTileLd + it * 8 + grp] = yn[it][0];
                Tt[(2 * seg + 1) * kTileLd + it * 8 + grp] = yn[it][1];
            }
            FWPH_ADD(1, t0); t0 = FWPH_T();
            if (r0 + kTileRows < nS) issue(r0 + kTileRows);
            AMX_RELOAD();
            for (int r = 0; r < rows; r++) {
                const double yi = Tt[r * kTileLd + lane];
                finite = finite && (fabs(yi) <= 1.79769313486231570e308);
                const double *ar = Ad + (size_t)(r0 + r) * NP;
#pragma unroll
                for (int j = 0; j < NP; j++) cn[j] += ar[j] * yi;
            }
            AMX_RELOAD();
            FWPH_ADD(2, t0);
        }
        t0 = FWPH_T();
        if (lane < cnt) {
            unsigned p0 = 0u;
#pragma unroll
            for (int j = 0; j < N; j++) cn[j] -= a.c.lam1;
            double *crow = a.cproj + (size_t)pos * NP;            // voxel-major: the solver streams whole batches into LDS
#pragma unroll
            for (int j = 0; j < NP; j++) crow[j] = finite ? cn[j] : __builtin_nan("");
#pragma unroll
            for (int j = 0; j < N; j++) {
                double z0 = 0.0;
#pragma unroll
                for (int k = 0; k < N; k++) z0 += Hi[j * NP + k] * cn[k];
                if (j < n_atoms && z0 > 0.0) p0 |= 1u << j;
            }
            a.p0[pos] = p0;
        }
        FWPH_ADD(3, t0);
    }
#ifdef AMX_FW_PHASES
    ph[7] = FWPH_T() - t_all;
    ph[5] = (wave < n_batches) ? (n_batches - wave + nw - 1) / nw : 0;
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&g_fw_pp[k], ph[k]);
#endif
}

// The same projection on the matrix cores: C' [atoms x voxels] = A' [atoms x nS] * Y' [nS x voxels] is a tall-skinny fp64 GEMM
// (v_mfma_f64_16x16x4_f64; 16 atoms x 16 voxels x 4 signal values per instruction, exact fp64 accumulation).
//   * A' lives in registers (one f64 per lane and K-step, loaded once per workgroup: lane l holds A[4 ks + (l >> 4)][l & 15]);
//   * the signal rows stream global -> LDS with direct loads, two 8 KB tiles of 64 voxels x 16 values per wavefront
//     (the next tile is in flight while the current one is multiplied); 8 lanes fetch the 128 contiguous bytes of one voxel,
//     and the 16-byte pieces of a voxel are XOR-swizzled by the voxel number so that the operand reads (16 voxels x one
//     value per instruction) spread over the LDS banks;
//   * z0 = H^-1 c is a second small GEMM on the accumulator registers (the D layout of the first product IS the B operand
//     layout of the second: atom (l >> 4) + 4 r of voxel l & 15 sits in register r of lane l);
//   * the values of a voxel end up in four lanes (l & 15 equal): the passive-set bits are OR-ed across them.
// Rows beyond the last full tile of 16 (nS = 65: one row) go through ordinary loads, zero-padded to a K-step.
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kProjKS = 24;                // K-steps of dictionary registers: nS <= 96

// F32: the signals are float32 in HBM (amx_freewater_fit_device_f32; 260 instead of 520 bytes per voxel): a tile of 16 values
// of 64 voxels is 4 KB, four lanes fetch the 64 bytes of a voxel, pieces XOR-swizzled by (voxel >> 2) & 3, converted at the
// operand read.
template <int N, bool F32>
__global__ void __launch_bounds__(256, 2) k_fw_project_mfma(const FwArgs a)
{
    static_assert(N <= 16, "one 16-row MFMA tile of atoms");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
    const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.c.chunks[cid];
    const int nS = a.c.nS, n_atoms = a.c.n_atoms;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    constexpr int NP = (N + 1) & ~1;
    constexpr int kTile = 64 * 16;                       // doubles per tile
    double *T = reinterpret_cast<double *>(smem_m) + (size_t)wave * (2 * kTile + 32);
    int *Vb = reinterpret_cast<int *>(T + 2 * kTile);
    const int q = lane >> 4, v16 = lane & 15;
    const double *src = a.prep + (size_t)ck.dir * fw_prep_words<N>(nS);
    const int KS = (nS + 3) >> 2;
    double ad[kProjKS], hi[4];
#pragma unroll
    for (int ks = 0; ks < kProjKS; ks++) {
        const int row = 4 * ks + q;
        ad[ks] = (ks < KS && row < nS && v16 < NP) ? src[row * NP + v16] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const int k = 4 * ks + q;                        // H^-1 [i = l & 15][k], zero beyond N
        hi[ks] = (v16 < N && k < N) ? src[(size_t)nS * NP + v16 * NP + k] : 0.0;
    }
    const int n_pass = nS >> 4;                          // full tiles of 16 signal values
    const int n_batches = (ck.count + 63) >> 6;
    const int seg = F32 ? (lane & 3) : (lane & 7), grp = F32 ? (lane >> 2) : (lane >> 3);   // loader role: 8 (4) lanes x 16 B = the 128 (64) B of one voxel
    constexpr int NL = F32 ? 4 : 8;                      // load instructions per tile
    for (int b = wave; b < n_batches; b += nw) {
        const int cnt = min(64, ck.count - (b << 6));
        const int pos0 = ck.start + (b << 6);
        Vb[lane] = a.c.perm[pos0 + min(lane, cnt - 1)];   // (lanes past the end repeat the last voxel; their results are dropped)
        const char *yl[8];                                  // (fixed size: a dependent array type here makes the host pass drop the kernel stub)
#pragma unroll
        for (int it = 0; it < NL; it++) {
            const int vloc = F32 ? it * 16 + grp : it * 8 + grp;
            if (F32) yl[it] = reinterpret_cast<const char *>(a.c.y32 + (size_t)Vb[vloc] * nS + 4 * (seg ^ ((vloc >> 2) & 3)));
            else yl[it] = reinterpret_cast<const char *>(a.c.y + (size_t)Vb[vloc] * nS + 2 * (seg ^ ((vloc >> 1) & 7)));
        }
        auto issue = [&](int p) {
            char *dst = reinterpret_cast<char *>(T + (p & 1) * kTile);
#pragma unroll
            for (int it = 0; it < NL; it++)
                __builtin_amdgcn_global_load_lds(yl[it] + (F32 ? 64 : 128) * p, (__attribute__((address_space(3))) void *)(dst + it * 1024), 16, 0, 0);
        };
        v4d acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (n_pass > 0) issue(0);
#pragma unroll
        for (int p = 0; p < kProjKS / 4; p++) {              // (unrolled: the dictionary registers need compile-time indices)
            if (p < n_pass) {
                if (p + 1 < n_pass) {
                    issue(p + 1);
                    if (F32) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile p has landed, tile p + 1 (8 / 4 loads) may be in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const double *tile = T + (p & 1) * kTile;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = 4 * kk + q, s2 = k >> 1, half = k & 1;
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        const int vloc = mt * 16 + v16;
                        const double bv = F32 ? (double)reinterpret_cast<const float *>(tile)[vloc * 16 + ((kk ^ ((vloc >> 2) & 3)) << 2) + q]
                                              : tile[vloc * 16 + ((s2 ^ ((vloc >> 1) & 7)) << 1) + half];
                        acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[4 * p + kk], bv, acc[mt], 0, 0, 0);
                    }
                }
                asm volatile("" ::: "memory");
            } else if (p == n_pass) {
                // rows beyond the last full tile: ordinary loads, zero-padded to whole K-steps
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int row = 16 * p + 4 * kk + q;
                    if (16 * p + 4 * kk < nS) {
#pragma unroll
                        for (int mt = 0; mt < 4; mt++) {
                            const size_t yo = (size_t)Vb[mt * 16 + v16] * nS + row;
                            const double bv = (row < nS) ? (F32 ? (double)a.c.y32[yo] : a.c.y[yo]) : 0.0;
                            acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[4 * p + kk], bv, acc[mt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // c = A'y - lambda1; z0 = H^-1 c; passive set after the first block removal
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            v4d cf = acc[mt], z = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; r++) cf[r] = (q + 4 * r < n_atoms) ? cf[r] - a.c.lam1 : 0.0;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) z = __builtin_amdgcn_mfma_f64_16x16x4f64(hi[ks], cf[ks], z, 0, 0, 0);
            unsigned m = 0u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (q + 4 * r < n_atoms && z[r] > 0.0) m |= 1u << (q + 4 * r);
                if (!(fabs(cf[r]) <= 1.79769313486231570e308)) m |= 0x80000000u;       // non-finite signal
            }
            m |= __shfl_xor((int)m, 16);
            m |= __shfl_xor((int)m, 32);
            const int vloc = mt * 16 + v16;
            if (vloc < cnt) {
                const bool finite = !(m & 0x80000000u);
                double *crow = a.cproj + (size_t)(pos0 + vloc) * NP;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (q + 4 * r < NP) crow[q + 4 * r] = finite ? cf[r] : __builtin_nan("");
                if (q == 0) a.p0[pos0 + vloc] = m & 0x7fffffffu;
            }
        }
    }
}

// The solver is a set of PERSISTENT wavefronts: each draws sub-chunks of kSubChunk voxels (of one orientation) from a global
// ticket and keeps TWO H tables in its LDS block -- lanes still iterating on voxels of the previous sub-chunk keep theirs
// while the free lanes already take voxels of the next one, so lanes only idle at the very end of the launch (a
// workgroup-per-chunk version lost a third of its lane-trips to the tail of every chunk).
#ifndef AMX_FW_SUBCHUNK
#define AMX_FW_SUBCHUNK 256      // (measured: 128 -> 534 us, 256 -> 527 us, 512 -> 540 us for the solver on 2 M voxels)
#endif
constexpr int kSubChunk = AMX_FW_SUBCHUNK;

template <int N>
__global__ void __launch_bounds__(256, 2) k_freewater_refill(const FwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    const int n_atoms = a.c.n_atoms, n_perp = a.n_perp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NP = (N + 1) & ~1;
    // LDS per wavefront: Hs f64 [2][kSlot] | two voxel buffers, each Cb [64][NP], Vb int[64], Pb unsigned[64].
    // Everything arrives by direct global->LDS loads (no staging registers, and the wavefront keeps iterating while
    // they are in flight): 16 bytes per lane and instruction, destination = base + 16 * lane.
    constexpr int kHPieces = (N * N + 127) / 128;        // 1 KB loads per H table
    constexpr int kSlot = 128 * kHPieces + 2;            // banks shifted between the two slots
    constexpr int kBufWords = 64 * NP + 64;              // doubles
    constexpr int kWaveWords = 2 * kSlot + 2 + 2 * kBufWords;
    double *Hs = reinterpret_cast<double *>(smem_r) + (size_t)wave * kWaveWords;
    double *buf0 = Hs + 2 * kSlot + 2;
    const int n_chunks = *a.c.n_chunks;
    const int n_units = n_chunks * a.sub_per_chunk;
    const double tol = 1e-12, inf = __builtin_huge_val();
    // lane state
    const bool warm0 = amx_warm_start(a.c.lam2, a.c.flags);
    constexpr int kBackup = 3;               // block exchanges allowed without progress (Kim & Park)
    bool active = false;
    int vox = 0, its = 0, ninf = N + 1, backup = 0;
    unsigned P = 0u;
    const double *Hl = Hs;                   // this lane's H table (slot of the sub-chunk its voxel came from)
    double c[N], x[N];
#pragma unroll
    for (int j = 0; j < N; j++) { c[j] = 0.0; x[j] = 0.0; }
    // wave-uniform: the buffer being consumed, the buffer in flight, the current sub-chunk
    int cur = 0, buf_pos = 0, buf_cnt = 0, buf_slot = 0;
    int nxt_cnt = 0, nxt_slot = 0;
    int sub_pos = 0, sub_end = 0, cur_slot = 1;
    bool more = true;                        // the global queue may still have sub-chunks

#ifdef AMX_FW_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_all = FWPH_T();
#endif
    for (int trip = 0; trip < (1 << 24); ++trip) {
        unsigned long long t0 = FWPH_T();
        // ------------------------------------------------------------ prefetch: the next batch into the idle buffer
        if (nxt_cnt == 0 && (more || sub_pos < sub_end)) {
            if (sub_pos >= sub_end) {
                // next sub-chunk, its H into the other slot -- once no lane works with that slot any more
                const int ns = cur_slot ^ 1;
                const double *slot_ptr = Hs + ns * kSlot;
                const bool slot_busy = __ballot(active && Hl == slot_ptr) != 0ull || (buf_cnt > 0 && buf_slot == ns);
                if (!slot_busy) {
                    int u = 0;
                    if (lane == 0) u = atomicAdd(a.queue, 1);
                    u = __builtin_amdgcn_readfirstlane(u);
                    if (u >= n_units) {
                        more = false;
                    } else {
                        const Chunk ck = a.c.chunks[u / a.sub_per_chunk];
                        const int k0 = (u % a.sub_per_chunk) * kSubChunk;
                        if (k0 < ck.count) {
                            const double *hsrc = a.prep + (size_t)ck.dir * fw_prep_words<N>(a.c.nS) + (a.c.nS + N) * NP;
#pragma unroll
                            for (int i = 0; i < kHPieces; i++) {
                                const int hoff = min(i * 128 + lane * 2, (N * N - 1) & ~1);   // (the tail lanes re-read the last piece)
                                __builtin_amdgcn_global_load_lds(hsrc + hoff, (__attribute__((address_space(3))) void *)(Hs + ns * kSlot + i * 128), 16, 0, 0);
                            }
                            cur_slot = ns;
                            sub_pos = ck.start + k0;
                            sub_end = ck.start + min(ck.count, k0 + kSubChunk);
                        }
                    }
                }
            }
            if (sub_pos < sub_end) {
                const int cnt = min(64, sub_end - sub_pos);
                double *nb = buf0 + (cur ^ 1) * kBufWords;
                int *nVb = reinterpret_cast<int *>(nb + 64 * NP);
                unsigned *nPb = reinterpret_cast<unsigned *>(nVb + 64);
                const int pl = sub_pos + min(lane, cnt - 1);                              // lanes past the end repeat the last voxel
                __builtin_amdgcn_global_load_lds(a.c.perm + pl, (__attribute__((address_space(3))) void *)nVb, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(a.p0 + pl, (__attribute__((address_space(3))) void *)nPb, 4, 0, 0);
                const double *src = a.cproj + (size_t)sub_pos * NP;
                const int last = cnt * NP - 2;                                            // last 16-byte piece of the batch
#pragma unroll
                for (int i = 0; i < (64 * NP * 8) / 1024; i++) {
                    const int off = min(i * 128 + lane * 2, last);
                    __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(nb + i * 128), 16, 0, 0);
                }
                nxt_cnt = cnt; nxt_slot = cur_slot;
                sub_pos += cnt;
            }
        }
        const unsigned long long freem = __ballot(!active);
        // ------------------------------------------------------------ the buffer ran dry: switch to the one in flight
        if (freem != 0ull && buf_cnt == 0 && nxt_cnt > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cur ^= 1; buf_pos = 0; buf_cnt = nxt_cnt; buf_slot = nxt_slot; nxt_cnt = 0;
        }
        FWPH_ADD(0, t0); t0 = FWPH_T();
        // ------------------------------------------------------------ free lanes take the next voxels of the buffer
        if (freem != 0ull && buf_cnt > 0) {
            const double *Cb = buf0 + cur * kBufWords;
            const int *Vb = reinterpret_cast<const int *>(Cb + 64 * NP);
            const unsigned *Pb = reinterpret_cast<const unsigned *>(Vb + 64);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(freem >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)freem, 0u));
            const bool take = !active && rank < buf_cnt;
            const int e = take ? buf_pos + rank : 0;
            const int nv = Vb[e];
            const unsigned np0 = Pb[e];
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double cj = Cb[e * NP + j];
                c[j] = take ? cj : c[j];
            }
            // warm: the full-set solve and the first block removal happened in k_fw_project (p0: the atoms that stayed)
            if (take) { active = true; vox = nv; P = warm0 ? np0 : 0u; its = warm0 ? 1 : 0; ninf = N + 1; backup = 0; Hl = Hs + buf_slot * kSlot; }
            const int taken = min(__builtin_popcountll(freem), buf_cnt);
            buf_pos += taken; buf_cnt -= taken;
        }
        FWPH_ADD(1, t0); t0 = FWPH_T();
        if (__ballot(active) == 0ull) {
            if (!more && buf_cnt == 0 && nxt_cnt == 0 && sub_pos >= sub_end) break;
            continue;
        }
#ifdef AMX_FW_PHASES
        ph[5] += 1; ph[6] += __builtin_popcountll(__ballot(active));
#endif
        // ------------------------------------------------------------ phase 2: one pivoting step per lane
        // Block principal pivoting (Portugal, Judice, Vicente 1994; Kim & Park 2011) on the strictly convex problem: solve on
        // the passive set P, form the dual vector, then exchange ALL infeasible atoms at once -- passive atoms with a
        // non-positive coefficient leave, inactive atoms with a positive dual value enter.  The block exchange is allowed
        // while the number of infeasibilities keeps falling (and `kBackup` more times after that); otherwise only the
        // infeasible atom with the largest index is exchanged (Murty's rule, which terminates on its own).  The unique
        // KKT point is reached in 2.9 solves per voxel from P0 (3.2 for block removals followed by Lawson-Hanson steps; the
        // longest voxel needs 7 instead of 14), and no coefficient vector has to survive from one trip to the next.
        // (The kernel is only launched with the warm start: for lambda2 < 1e-5, where H may be nearly singular, and for
        // AMX_COLD_START=1 the fit goes to k_freewater_lane's Lawson-Hanson loop -- amx_fw_use_refill.)
        bool done = false;
        if (active && !(c[0] == c[0])) done = true;             // non-finite signal: NaN maps, never iterate
        const bool slv = active && !done;
        {
            lane_solve<N>(Hl, c, P, x);                           // x = H_PP^-1 c_P, 0 outside P
            FWPH_ADD(2, t0); t0 = FWPH_T();
            AMX_RELOAD();
            unsigned v1 = 0u, v2 = 0u;                            // primal / dual infeasible atoms
#pragma unroll
            for (int j = 0; j < N; j++) {
                double g = c[j];
#pragma unroll
                for (int k = 0; k < N; k++) g -= Hl[j * N + k] * x[k];
                const bool pj = (P >> j) & 1u;
                if (pj && !(x[j] > 0.0)) v1 |= 1u << j;
                if (!pj && j < n_atoms && g > tol) v2 |= 1u << j;
            }
            if (slv) {
                const unsigned bad = v1 | v2;
                const int nbad = __builtin_popcount(bad);
                if (bad == 0u || its > 4 * N + 16) {
                    done = true;                                  // KKT point (or the iteration cap)
                } else {
                    bool block = false;
                    if (nbad < ninf) { ninf = nbad; backup = warm0 ? kBackup : 0; block = warm0; }
                    else if (backup > 0) { backup--; block = true; }
                    const unsigned ex = block ? bad : (1u << (31 - __builtin_clz(bad)));
                    P ^= ex;                                      // infeasible passive atoms leave, infeasible inactive atoms enter
                    its++;
                }
            }
        }
        FWPH_ADD(3, t0); t0 = FWPH_T();
        // ------------------------------------------------------------ finished voxels: maps (models.pyx:1241-1256)
        if (__ballot(done) != 0ull) {
            if (done) {
                double *e = a.est + (size_t)vox * a.n_maps;
                if (!(c[0] == c[0])) {
                    const double nan = __builtin_nan("");
                    for (int m = 0; m < a.n_maps; m++) e[m] = nan;
                } else {
                    if (its > 4 * N + 16) {          // (the last block-pivoting iterate may be infeasible: the maps get a feasible one)
                        atomicAdd(&a.c.status[ST_ITCAP], 1);
#pragma unroll
                        for (int j = 0; j < N; j++) x[j] = (x[j] > 0.0) ? x[j] : 0.0;
                    }
                    if (a.c.xdbg) {
#pragma unroll
                        for (int j = 0; j < N; j++) if (j < n_atoms) a.c.xdbg[(size_t)vox * n_atoms + j] = x[j];
                    }
                    double x_sum = 0.0, x_perp = 0.0;
#pragma unroll
                    for (int j = 0; j < N; j++) { x_sum += x[j]; if (j < n_perp) x_perp += x[j]; }
                    x_sum += 1e-16;
                    const double vv = x_perp / x_sum;
                    e[0] = vv; e[1] = 1.0 - vv;
                    if (a.is_mouse) {
                        double xb = 0.0, xc = 0.0;
#pragma unroll
                        for (int j = 0; j < N; j++) { if (j == n_perp) xb = x[j]; if (j == n_perp + 1) xc = x[j]; }
                        e[2] = xb / x_sum; e[3] = xc / x_sum;
                    }
                }
                active = false;
            }
        }
        FWPH_ADD(4, t0);
    }
#ifdef AMX_FW_PHASES
    ph[7] = FWPH_T() - t_all;
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&g_fw_ph[k], ph[k]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// FreeWater in ONE kernel (round 5): a PRODUCER wavefront and seven CONSUMER wavefronts per CU.
//
// k_fw_project_mfma + k_freewater_refill move c = A'y - lambda1 (96 B) and the first passive set (4 B) of every voxel through HBM and
// run one after the other: 0.32 ms of streaming with idle vector units, then 0.40 ms of fp64 vector work with an idle memory system
// (per 2 M voxels).  Here wavefront 0 of a workgroup of eight does what the projection kernel did -- signal tiles global -> LDS by
// direct loads, kFuseAhead tiles in flight behind the one being multiplied, c and z0 = H^-1 c on the fp64 matrix cores -- and leaves
// each batch of 64 voxels (c, voxel numbers, first passive sets) in a slot of an LDS ring; wavefronts 1 .. 7 are the lanes-that-never-
// idle solver of k_freewater_refill, drawing their batches from the ring instead of from HBM (ticket = LDS atomic, slot state = one
// sequence word per slot: 2 t + 1 "batch t is here", 2 t + 2 "batch t has been taken").  c never leaves the chip; the loads of the
// next signals hide behind the pivoting of the current ones.  The per-voxel arithmetic is that of the two kernels, in the same order.
// Needs 64 <= nS <= 96 (whole tiles of 16 values dominate, the dictionary operand lives in 24 K-steps of registers): other protocols
// keep the two-kernel path.
// Workgroups of FOUR wavefronts, two per CU: one producer + three consumers each.  (One producer for seven consumers -- a workgroup of
// eight -- was measured first: 0.84 - 0.88 ms per 2 M voxels whatever the number of tiles in flight, against 0.735 for the kernel pair:
// the producer shares its SIMD with a consumer, both live on the fp64 pipe, and 7 k cycles of matrix instructions per batch at half a
// SIMD are ~6 us, while seven consumers want a batch every 3.7 us.)
#ifndef AMX_FUSE_RING
#define AMX_FUSE_RING 5
#endif
constexpr int kFuseRing = AMX_FUSE_RING;   // batches between producer and consumers
#ifndef AMX_FUSE_AHEAD
#define AMX_FUSE_AHEAD 2
#endif
constexpr int kFuseAhead = AMX_FUSE_AHEAD; // signal tiles in flight behind the one being multiplied (2 x 8 KB per workgroup, 32 KB per CU)
constexpr int kFuseTiles = kFuseAhead + 1;
constexpr int kFuseHs = 3;                 // H tables per consumer wavefront (lanes may still work on voxels of the last two orientations)
constexpr int kFuseSub = 1024;             // voxels per unit of the global queue (one orientation: the operand registers are loaded per unit,
                                           // and the pipeline of tiles fills once per unit)
constexpr int kFuseConsumers = 3;
static_assert(kFuseAhead * 8 <= 63 && kFuseAhead <= 7, "s_waitcnt vmcnt holds six bits");

template <int N> constexpr int fuse_hslot_words() { return 128 * ((N * N + 127) / 128) + 2; }
template <int N> constexpr int fuse_slot_words() { return 64 * ((N + 1) & ~1) + 64; }       // Cb [64][NP] | Vb int[64] | Pb unsigned[64]
template <int N> constexpr size_t fuse_lds_bytes(int n_tail)      // n_tail = nS % 16: signal rows beyond the last full tile
{
    return ((size_t)kFuseRing * fuse_slot_words<N>() + (size_t)kFuseTiles * 64 * 16 + (size_t)n_tail * 64 +
            (size_t)kFuseConsumers * kFuseHs * fuse_hslot_words<N>()) * sizeof(double) + (64 + (kFuseSub / 64) * 64) * sizeof(int);
}

template <int N, bool F32>
__global__ void __launch_bounds__(64 * (kFuseConsumers + 1), 2) k_freewater_fused(const FwArgs a)
{
    static_assert(N <= 16, "one 16-row MFMA tile of atoms");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];
    constexpr int NP = (N + 1) & ~1;
    constexpr int R = kFuseRing, TB = kFuseTiles, D = kFuseAhead;
    constexpr int kSlotW = fuse_slot_words<N>(), kHW = fuse_hslot_words<N>(), kHPieces = (N * N + 127) / 128;
    constexpr int kTile = 64 * 16;                              // doubles per signal tile (float32 signals use half of it)
    double *ring = reinterpret_cast<double *>(smem_u);
    double *tiles = ring + (size_t)R * kSlotW;
    unsigned *Ttail = reinterpret_cast<unsigned *>(tiles + (size_t)TB * kTile);     // [<= 30 dwords][64 voxels]: rows beyond the last full tile
    double *Hp = reinterpret_cast<double *>(Ttail) + (size_t)(a.c.nS & 15) * 64;            // (2 dwords x 64 voxels per row)
    int *ctl = reinterpret_cast<int *>(Hp + (size_t)kFuseConsumers * kFuseHs * kHW);
    int *seq = ctl, *sdir = ctl + R, *scnt = ctl + 2 * R, *head = ctl + 3 * R, *total = ctl + 3 * R + 1;
    int *Vu = ctl + 64;                                         // [kFuseSub / 64][64]: voxel numbers of the unit the producer works on
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nS = a.c.nS, n_atoms = a.c.n_atoms, n_perp = a.n_perp;
    if (threadIdx.x < R) { seq[threadIdx.x] = 2 * ((int)threadIdx.x - R) + 2; sdir[threadIdx.x] = -1; scnt[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { *head = 0; *total = -1; }
    __syncthreads();
    const int n_chunks = *a.c.n_chunks;
    const int n_units = n_chunks * a.sub_per_chunk;
    auto lds_load = [](const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_store = [](int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // Which wavefront produces: the two workgroups of a CU are dispatched half a grid apart (the first 256 fill the chip, the second 256
    // follow in the same order), and a workgroup's wavefront w runs on SIMD w & 3: wavefront 0 in the first half, wavefront 2 in the
    // second, so that the two producers of a CU -- matrix instructions on the fp64 pipe, like the consumers' vector work -- sit on
    // DIFFERENT SIMDs next to one consumer each (both on wavefront 0: a SIMD without any pivoting, 0.589 ms per 2 M voxels)
#ifndef AMX_FUSE_PW
#define AMX_FUSE_PW 2
#endif
    const int pw = ((int)blockIdx.x >= ((int)gridDim.x >> 1)) ? AMX_FUSE_PW : 0;
    if (wave == pw) {
        // ================================================================ producer
        __builtin_amdgcn_s_setprio(3);
        const int q = lane >> 4, v16 = lane & 15;
        const int n_pass = nS >> 4, n_tail = nS - 16 * n_pass;              // full tiles of 16 values, rows beyond them (< 16)
        const int seg = F32 ? (lane & 3) : (lane & 7), grp = F32 ? (lane >> 2) : (lane >> 3);
        constexpr int NL = F32 ? 4 : 8;                                     // direct loads per tile
        int batch_no = 0;                                                   // batches published so far
        for (;;) {
            int u = 0;
            if (lane == 0) u = atomicAdd(a.queue, 1);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= n_units) break;
            const Chunk ck = a.c.chunks[u / a.sub_per_chunk];
            const int k0 = (u % a.sub_per_chunk) * kFuseSub;
            if (k0 >= ck.count) continue;
            const int upos = ck.start + k0, ucnt = min(kFuseSub, ck.count - k0);
            const int nb = (ucnt + 63) >> 6;
            // the orientation's operands: A' in registers (lane l: A[4 ks + (l >> 4)][l & 15]), H^-1 likewise
            const double *src = a.prep + (size_t)ck.dir * fw_prep_words<N>(nS);
            const int KS = (nS + 3) >> 2;
            double ad[kProjKS], hi[4];
#pragma unroll
            for (int ks = 0; ks < kProjKS; ks++) {
                const int row = 4 * ks + q;
                ad[ks] = (ks < KS && row < nS && v16 < NP) ? src[row * NP + v16] : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int k = 4 * ks + q;
                hi[ks] = (v16 < N && k < N) ? src[(size_t)nS * NP + v16 * NP + k] : 0.0;
            }
            // ---- ring slots of the batches of this unit: batch b of the unit is batch (batch_no + b) of the workgroup
            auto slot_of = [&](int b) { return (batch_no + b) % R; };
            auto claim = [&](int b) {                                        // wait until the slot's previous batch has been taken
                const int t = batch_no + b, sl = t % R;
                for (int spin = 0; spin < (1 << 26); spin++) {
                    if (lds_load(&seq[sl]) == 2 * (t - R) + 2) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            };
            auto slot_vb = [&](int b) { return Vu + 64 * b; };          // (the unit's voxel numbers: loaded once, below)
            auto issue_perm = [&](int b) {                                   // the batch's voxel numbers, global -> LDS
                const int cnt = min(64, ucnt - 64 * b);
                const int pl = upos + 64 * b + min(lane, cnt - 1);           // (lanes past the end repeat the last voxel; their results are dropped)
                __builtin_amdgcn_global_load_lds(a.c.perm + pl, (__attribute__((address_space(3))) void *)slot_vb(b), 4, 0, 0);
            };
            const char *yl[8];                                               // this lane's piece of the issue batch's signal rows
            auto set_yl = [&](int b) {
                const int *Vb = slot_vb(b);
#pragma unroll
                for (int it = 0; it < NL; it++) {
                    const int vloc = F32 ? it * 16 + grp : it * 8 + grp;
                    if (F32) yl[it] = reinterpret_cast<const char *>(a.c.y32 + (size_t)Vb[vloc] * nS + 4 * (seg ^ ((vloc >> 2) & 3)));
                    else yl[it] = reinterpret_cast<const char *>(a.c.y + (size_t)Vb[vloc] * nS + 2 * (seg ^ ((vloc >> 1) & 7)));
                }
            };
            auto issue_tile = [&](int gi) {                                  // tile gi of the unit's stream: batch gi / n_pass, pass gi % n_pass
                const int b = gi / n_pass, p = gi - b * n_pass;
                char *dst = reinterpret_cast<char *>(tiles + (size_t)(gi % TB) * kTile);
#pragma unroll
                for (int it = 0; it < NL; it++)
                    __builtin_amdgcn_global_load_lds(yl[it] + (F32 ? 64 : 128) * p, (__attribute__((address_space(3))) void *)(dst + it * 1024), 16, 0, 0);
                if (p == n_pass - 1 && n_tail > 0) {
                    // rows beyond the last full tile: one dword column of the 64 voxels per load, Ttail[dword][voxel]
                    const int *Vb = slot_vb(b);
                    const int ndw = F32 ? n_tail : 2 * n_tail;
                    const unsigned *row0 = F32 ? reinterpret_cast<const unsigned *>(a.c.y32 + (size_t)Vb[lane] * nS + 16 * n_pass)
                                               : reinterpret_cast<const unsigned *>(a.c.y + (size_t)Vb[lane] * nS + 16 * n_pass);
                    for (int dw = 0; dw < ndw; dw++)
                        __builtin_amdgcn_global_load_lds(row0 + dw, (__attribute__((address_space(3))) void *)(Ttail + dw * 64), 4, 0, 0);
                }
            };
            // ---- prologue of the unit: every batch's voxel numbers, first tiles
            for (int b = 0; b < nb; b++) issue_perm(b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            set_yl(0);
            const int n_steps = nb * n_pass;
            for (int gi = 0; gi < D && gi < n_steps; gi++) {
                if (gi > 0 && gi % n_pass == 0) set_yl(gi / n_pass);        // (more tiles ahead than a batch has: the next batch's rows)
                issue_tile(gi);
            }
            v4d acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < n_steps; g++) {
                const int gi = g + D;
                asm volatile("" ::: "memory");
                if (gi < n_steps) {
                    const int bi = gi / n_pass, pi = gi - bi * n_pass;
                    if (pi == 0) set_yl(bi);                             // a new batch on the issue side
                    issue_tile(gi);
                }
                // tile g has landed when at most the loads of the tiles behind it are outstanding (their extra loads only make the wait longer)
                {
                    const int younger = n_steps - 1 - g < D ? n_steps - 1 - g : D;
#define AMX_FUSE_WAIT(K) case K: if (F32) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * K) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * K) : "memory"); break;
                    switch (younger) {
                        AMX_FUSE_WAIT(7) AMX_FUSE_WAIT(6) AMX_FUSE_WAIT(5) AMX_FUSE_WAIT(4) AMX_FUSE_WAIT(3) AMX_FUSE_WAIT(2) AMX_FUSE_WAIT(1)
                        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
#undef AMX_FUSE_WAIT
                }
                const int b = g / n_pass, p = g - b * n_pass;
                const double *tile = tiles + (size_t)(g % TB) * kTile;
                // the dictionary registers need compile-time indices: the pass number selects its four K-steps
#pragma unroll
                for (int pp = 0; pp < kProjKS / 4; pp++) {
                    if (pp == p) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = 4 * kk + q, s2 = k >> 1, half = k & 1;
#pragma unroll
                            for (int mt = 0; mt < 4; mt++) {
                                const int vloc = mt * 16 + v16;
                                const double bv = F32 ? (double)reinterpret_cast<const float *>(tile)[vloc * 16 + ((kk ^ ((vloc >> 2) & 3)) << 2) + q]
                                                      : tile[vloc * 16 + ((s2 ^ ((vloc >> 1) & 7)) << 1) + half];
                                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[4 * pp + kk], bv, acc[mt], 0, 0, 0);
                            }
                        }
                    }
                }
                asm volatile("" ::: "memory");
                if (p == n_pass - 1) {
                    // ---- the batch is complete: rows beyond the last full tile (the operand of a row >= nS is zero: a neighbour's value
                    // in the padding of the K-step must not reach the sum as NaN x 0)
                    if (n_tail > 0) {
#pragma unroll
                        for (int pp = 4; pp <= kProjKS / 4; pp++) {
                            if (pp == n_pass) {
#pragma unroll
                                for (int kk = 0; kk < 4; kk++) {
                                    if (pp < kProjKS / 4 && 4 * kk < n_tail) {
                                        const int rt = 4 * kk + q;               // row within the tail
#pragma unroll
                                        for (int mt = 0; mt < 4; mt++) {
                                            const int vloc = mt * 16 + v16;
                                            double bv = 0.0;
                                            if (rt < n_tail) {
                                                if (F32) bv = (double)__builtin_bit_cast(float, Ttail[rt * 64 + vloc]);
                                                else bv = __hiloint2double((int)Ttail[(2 * rt + 1) * 64 + vloc], (int)Ttail[(2 * rt) * 64 + vloc]);
                                            }
                                            acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[(4 * pp + kk) < kProjKS ? 4 * pp + kk : 0], bv, acc[mt], 0, 0, 0);
                                        }
                                    }
                                }
                            }
                        }
                    }
                    // c = A'y - lambda1; z0 = H^-1 c; passive set after the first block removal -> the batch's ring slot
                    claim(b);                                                // (the slot's previous batch has been taken: normally long ago)
                    const int sl = slot_of(b), cnt = min(64, ucnt - 64 * b);
                    double *Cb = ring + (size_t)sl * kSlotW;
                    int *Vs = reinterpret_cast<int *>(Cb + 64 * NP);
                    unsigned *Pb = reinterpret_cast<unsigned *>(Vs + 64);
                    Vs[lane] = Vu[64 * b + lane];
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        v4d cf = acc[mt], z = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; r++) cf[r] = (q + 4 * r < n_atoms) ? cf[r] - a.c.lam1 : 0.0;
#pragma unroll
                        for (int ks = 0; ks < 4; ks++) z = __builtin_amdgcn_mfma_f64_16x16x4f64(hi[ks], cf[ks], z, 0, 0, 0);
                        unsigned m = 0u;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (q + 4 * r < n_atoms && z[r] > 0.0) m |= 1u << (q + 4 * r);
                            if (!(fabs(cf[r]) <= 1.79769313486231570e308)) m |= 0x80000000u;       // non-finite signal
                        }
                        m |= __shfl_xor((int)m, 16);
                        m |= __shfl_xor((int)m, 32);
                        const int vloc = mt * 16 + v16;
                        const bool finite = !(m & 0x80000000u);
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (q + 4 * r < NP) Cb[vloc * NP + q + 4 * r] = finite ? cf[r] : __builtin_nan("");
                        if (q == 0) Pb[vloc] = m & 0x7fffffffu;
                        acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
                    }
                    if (lane == 0) { sdir[sl] = ck.dir; scnt[sl] = cnt; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) lds_store(&seq[sl], 2 * (batch_no + b) + 1);
                }
            }
            batch_no += nb;
        }
        if (lane == 0) lds_store(total, batch_no);
        return;
    }

    // ==================================================================== consumers: the solver of k_freewater_refill
    const int cw = wave < pw ? wave : wave - 1;
    double *Hs = Hp + (size_t)cw * kFuseHs * kHW;
    const double tol = 1e-12;
    const bool warm0 = amx_warm_start(a.c.lam2, a.c.flags);
    constexpr int kBackup = 3;
    bool active = false;
    int vox = 0, its = 0, ninf = N + 1, backup = 0;
    unsigned P = 0u;
    const double *Hl = Hs;
    double c[N], x[N];
#pragma unroll
    for (int j = 0; j < N; j++) { c[j] = 0.0; x[j] = 0.0; }
    // wave-uniform: the slot being consumed, the one claimed next, the orientations of the H tables
    int buf_slot = -1, buf_tick = -1, buf_pos = 0, buf_cnt = 0, buf_h = 0;
    int nxt_slot = -1, nxt_tick = -1, nxt_cnt = 0, nxt_h = 0;
    int tick = -1;
    bool fin = false;
    int hdir[kFuseHs];
#pragma unroll
    for (int h = 0; h < kFuseHs; h++) hdir[h] = -1;
    for (int trip = 0; trip < (1 << 26); ++trip) {
        // ------------------------------------------------------------ the next batch: ticket, then wait for the producer (without blocking)
        if (nxt_cnt == 0 && !fin) {
            if (tick < 0) {
                int t = 0;
                if (lane == 0) t = atomicAdd(head, 1);
                tick = __builtin_amdgcn_readfirstlane(t);
            }
            const int sl = tick % R;
            if (lds_load(&seq[sl]) == 2 * tick + 1) {
                const int d = __builtin_amdgcn_readfirstlane(sdir[sl]);
                // an H table for the batch's orientation: one that holds it already, else one no lane works with
                int hsel = -1;
#pragma unroll
                for (int h = 0; h < kFuseHs; h++) if (hsel < 0 && hdir[h] == d) hsel = h;
                bool need_load = false;
                if (hsel < 0) {
#pragma unroll
                    for (int h = 0; h < kFuseHs; h++) {
                        const bool busy = __ballot(active && Hl == Hs + h * kHW) != 0ull || (buf_cnt > 0 && buf_h == h);
                        if (hsel < 0 && !busy) hsel = h;
                    }
                    need_load = hsel >= 0;
                }
                if (hsel >= 0) {
                    if (need_load) {
                        const double *hsrc = a.prep + (size_t)d * fw_prep_words<N>(nS) + (size_t)(nS + N) * NP;
#pragma unroll
                        for (int i = 0; i < kHPieces; i++) {
                            const int hoff = min(i * 128 + lane * 2, (N * N - 1) & ~1);   // (the tail lanes re-read the last piece)
                            __builtin_amdgcn_global_load_lds(hsrc + hoff, (__attribute__((address_space(3))) void *)(Hs + hsel * kHW + i * 128), 16, 0, 0);
                        }
#pragma unroll
                        for (int h = 0; h < kFuseHs; h++) hdir[h] = (h == hsel) ? d : hdir[h];
                    }
                    nxt_slot = sl; nxt_tick = tick; nxt_cnt = __builtin_amdgcn_readfirstlane(scnt[sl]); nxt_h = hsel;
                    tick = -1;
                }
            } else {
                const int tot = lds_load(total);
                if (tot >= 0 && tick >= tot) { fin = true; tick = -1; }
            }
        }
        const unsigned long long freem = __ballot(!active);
        // ------------------------------------------------------------ the buffer ran dry: switch to the batch claimed next
        if (freem != 0ull && buf_cnt == 0 && nxt_cnt > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (its H table, if one was requested)
            buf_slot = nxt_slot; buf_tick = nxt_tick; buf_pos = 0; buf_cnt = nxt_cnt; buf_h = nxt_h; nxt_cnt = 0;
        }
        // ------------------------------------------------------------ free lanes take the next voxels of the batch
        if (freem != 0ull && buf_cnt > 0) {
            const double *Cb = ring + (size_t)buf_slot * kSlotW;
            const int *Vb = reinterpret_cast<const int *>(Cb + 64 * NP);
            const unsigned *Pb = reinterpret_cast<const unsigned *>(Vb + 64);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(freem >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)freem, 0u));
            const bool take = !active && rank < buf_cnt;
            const int e = take ? buf_pos + rank : 0;
            const int nv = Vb[e];
            const unsigned np0 = Pb[e];
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double cj = Cb[e * NP + j];
                c[j] = take ? cj : c[j];
            }
            if (take) { active = true; vox = nv; P = warm0 ? np0 : 0u; its = warm0 ? 1 : 0; ninf = N + 1; backup = 0; Hl = Hs + buf_h * kHW; }
            const int taken = min(__builtin_popcountll(freem), buf_cnt);
            buf_pos += taken; buf_cnt -= taken;
            if (buf_cnt == 0) {
                // every voxel of the batch is in a lane's registers: the slot goes back to the producer
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) lds_store(&seq[buf_slot], 2 * buf_tick + 2);
            }
        }
        if (__ballot(active) == 0ull) {
            if (fin && buf_cnt == 0 && nxt_cnt == 0) break;
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        // ------------------------------------------------------------ one pivoting step per lane (block principal pivoting: see k_freewater_refill)
        bool done = false;
        if (active && !(c[0] == c[0])) done = true;             // non-finite signal: NaN maps, never iterate
        const bool slv = active && !done;
        {
            lane_solve<N>(Hl, c, P, x);
            AMX_RELOAD();
            unsigned v1 = 0u, v2 = 0u;
#pragma unroll
            for (int j = 0; j < N; j++) {
                double g = c[j];
#pragma unroll
                for (int k = 0; k < N; k++) g -= Hl[j * N + k] * x[k];
                const bool pj = (P >> j) & 1u;
                if (pj && !(x[j] > 0.0)) v1 |= 1u << j;
                if (!pj && j < n_atoms && g > tol) v2 |= 1u << j;
            }
            if (slv) {
                const unsigned bad = v1 | v2;
                const int nbad = __builtin_popcount(bad);
                if (bad == 0u || its > 4 * N + 16) {
                    done = true;
                } else {
                    bool block = false;
                    if (nbad < ninf) { ninf = nbad; backup = warm0 ? kBackup : 0; block = warm0; }
                    else if (backup > 0) { backup--; block = true; }
                    const unsigned ex = block ? bad : (1u << (31 - __builtin_clz(bad)));
                    P ^= ex;
                    its++;
                }
            }
        }
        // ------------------------------------------------------------ finished voxels: maps (models.pyx:1241-1256)
        if (__ballot(done) != 0ull) {
            if (done) {
                double *e = a.est + (size_t)vox * a.n_maps;
                if (!(c[0] == c[0])) {
                    const double nan = __builtin_nan("");
                    for (int m = 0; m < a.n_maps; m++) e[m] = nan;
                } else {
                    if (its > 4 * N + 16) {
                        atomicAdd(&a.c.status[ST_ITCAP], 1);
#pragma unroll
                        for (int j = 0; j < N; j++) x[j] = (x[j] > 0.0) ? x[j] : 0.0;
                    }
                    if (a.c.xdbg) {
#pragma unroll
                        for (int j = 0; j < N; j++) if (j < n_atoms) a.c.xdbg[(size_t)vox * n_atoms + j] = x[j];
                    }
                    double x_sum = 0.0, x_perp = 0.0;
#pragma unroll
                    for (int j = 0; j < N; j++) { x_sum += x[j]; if (j < n_perp) x_perp += x[j]; }
                    x_sum += 1e-16;
                    const double vv = x_perp / x_sum;
                    e[0] = vv; e[1] = 1.0 - vv;
                    if (a.is_mouse) {
                        double xb = 0.0, xc = 0.0;
#pragma unroll
                        for (int j = 0; j < N; j++) { if (j == n_perp) xb = x[j]; if (j == n_perp + 1) xc = x[j]; }
                        e[2] = xb / x_sum; e[3] = xc / x_sum;
                    }
                }
                active = false;
            }
        }
    }
}

static size_t refill_lds_bytes(int N, int nw) { return (size_t)nw * (2 * (128 * ((N * N + 127) / 128) + 2) + 2 + 2 * (64 * ((N + 1) & ~1) + 64)) * sizeof(double); }
static size_t project_lds_bytes(int nS, int N, int nw)
{
    const int NP = (N + 1) & ~1;
    return ((size_t)(nS + N) * NP + (size_t)nw * (kTileRows * kTileLd + 32)) * sizeof(double);
}

template <typename Args, typename K>
int launch_lane(amx_ctx *ctx, Args &a, const Plan &pl, hipStream_t s, K kern, size_t elem, int N)
{
    const size_t lds = (((size_t)a.c.nS * a.c.ldA * elem + 15) & ~(size_t)15) + (size_t)N * N * sizeof(double);
    if (lds > 160 * 1024) { ctx->err = "dictionary tile does not fit the 160 KB LDS of a CU"; return AMX_E_BADARG; }
    int rc;
    if ((rc = set_lds(ctx, kern, lds))) return rc;
    rec(ctx, 2, s);
    hipLaunchKernelGGL(kern, dim3(((pl.max_chunks + 7) / 8) * 8), dim3(256), lds, s, a);
    amx_note(ctx, "lane-per-voxel solver (k_freewater_lane / k_sandi_lane)");
    AMX_TRACE(ctx, s, "lane-per-voxel solver");
    rec(ctx, 3, s);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

}  // namespace

typedef void (*FwKernel)(const FwArgs);
static int launch_refill(amx_ctx *ctx, FwArgs &a, const Plan &pl, hipStream_t s, FwKernel proj, FwKernel pmfma, FwKernel pmfma32, FwKernel kern, int N,
                         FwKernel fused, FwKernel fused32, size_t fused_lds)
{
    int rc;
    // one kernel (k_freewater_fused) when the protocol fits its pipeline: four or more full tiles of 16 values per voxel, dictionary
    // operand in registers (AMX_FW_NO_FUSE=1: the projection + solver pair, for A/B)
    if (a.c.nS >= 64 && a.c.nS <= 4 * kProjKS && !ctx->opt_fw_proj_valu && !ctx->opt_fw_no_fuse) {
        FwKernel k = a.c.y32 != nullptr ? fused32 : fused;
        if ((rc = set_lds(ctx, k, fused_lds))) return rc;
        a.queue = pl.n_chunks + 60;                            // (misc word 60: zeroed with the plan counters)
        a.sub_per_chunk = (amx_refill_chunk(ctx, (long long)pl.n) + kFuseSub - 1) / kFuseSub;
        rec(ctx, 2, s);
        hipLaunchKernelGGL(k, dim3(2 * ctx->n_cu), dim3(64 * (kFuseConsumers + 1)), fused_lds, s, a);
        amx_note(ctx, N == 11 ? "k_freewater_fused<11>" : "k_freewater_fused<12>");
        AMX_TRACE(ctx, s, "FreeWater: projection (producer wavefront) + lane-per-voxel solver (consumers), one kernel");
        rec(ctx, 3, s);
        HIPCHK(ctx, hipGetLastError());
        return AMX_OK;
    }
    // workspace of the projection: c [ldC][NP], p0 [ldC]
    a.ldC = (int)((pl.n + 63) & ~(size_t)63);
    const size_t cbytes = (size_t)((N + 1) & ~1) * a.ldC * sizeof(double);
    if ((rc = amx_ensure(ctx, ctx->cproj, cbytes + (size_t)a.ldC * sizeof(unsigned)))) return rc;
    a.cproj = (double *)ctx->cproj.p;
    a.p0 = (unsigned *)((char *)ctx->cproj.p + cbytes);
    const bool mfma = a.c.nS <= 4 * kProjKS && !ctx->opt_fw_proj_valu;
    const size_t lds_p = mfma ? (size_t)4 * (2 * 64 * 16 + 32) * sizeof(double) : project_lds_bytes(a.c.nS, N, 4), lds = refill_lds_bytes(N, 4);
    if ((rc = set_lds(ctx, proj, lds_p)) || (rc = set_lds(ctx, pmfma, lds_p)) || (rc = set_lds(ctx, pmfma32, lds_p)) || (rc = set_lds(ctx, kern, lds))) return rc;
    if (a.c.y32 != nullptr && !mfma) { ctx->err = "float32 signals need the matrix-core projection (amx_fw_native_f32)"; return AMX_E_BADARG; }
    const dim3 grid(((pl.max_chunks + 7) / 8) * 8);
    a.queue = pl.n_chunks + 60;                            // (misc word 60: zeroed with the plan counters)
    a.sub_per_chunk = (amx_refill_chunk(ctx, (long long)pl.n) + kSubChunk - 1) / kSubChunk;
    rec(ctx, 2, s);
    if (mfma && a.c.y32 != nullptr) hipLaunchKernelGGL(pmfma32, grid, dim3(256), lds_p, s, a);
    else if (mfma) hipLaunchKernelGGL(pmfma, grid, dim3(256), lds_p, s, a);
    else hipLaunchKernelGGL(proj, grid, dim3(256), lds_p, s, a);
    amx_note(ctx, mfma ? "k_fw_project_mfma" : "k_fw_project");
    AMX_TRACE(ctx, s, "A'y of every voxel");
    hipLaunchKernelGGL(kern, dim3(2 * ctx->n_cu), dim3(256), lds, s, a);
    amx_note(ctx, "k_freewater_refill");
    AMX_TRACE(ctx, s, "lane-per-voxel solver with refill");
    rec(ctx, 3, s);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

template <int N>
static int fw_prepare_n(amx_ctx *ctx, const amx_lut *lut, FwArgs &a, hipStream_t s)
{
    const size_t words = (size_t)fw_prep_words<N>(lut->nS);
    if (lut->fw_lam2 != a.c.lam2 || lut->fw_N != N || !lut->fw_prep) {
        // (a fit with the previous lambda2 may still be reading the old tables on another stream)
        if (lut->fw_prep) HIPCHK(ctx, hipDeviceSynchronize());
        if (!lut->fw_prep || lut->fw_N != N) {
            if (lut->fw_prep) HIPCHK(ctx, hipFree(lut->fw_prep));
            lut->fw_prep = nullptr;
            HIPCHK(ctx, hipMalloc((void **)&lut->fw_prep, words * lut->ndirs * sizeof(double) + 64));     // (+ slack: 16-byte reads of the last entries)
        }
        if (!lut->fw_ready) HIPCHK(ctx, hipEventCreateWithFlags(&lut->fw_ready, hipEventDisableTiming));
        const int NP = (N + 1) & ~1;
        const size_t lds = ((size_t)lut->nS * NP + N * N) * sizeof(double);
        int rc;
        if ((rc = set_lds(ctx, k_fw_orient_prep<N>, lds))) return rc;
        hipLaunchKernelGGL(k_fw_orient_prep<N>, dim3(lut->ndirs), dim3(64), lds, s, reinterpret_cast<const float *>(lut->tiles),
                           lut->tile_stride, lut->ldA, lut->nS, lut->n_atoms, a.c.lam2, lut->fw_prep);
        AMX_TRACE(ctx, s, "FreeWater per-orientation tables");
        HIPCHK(ctx, hipEventRecord(lut->fw_ready, s));
        lut->fw_lam2 = a.c.lam2; lut->fw_N = N;
    }
    HIPCHK(ctx, hipStreamWaitEvent(s, lut->fw_ready, 0));      // another stream may have launched the build
    a.prep = lut->fw_prep;
    return AMX_OK;
}

int amx_fw_prepare(amx_ctx *ctx, const amx_lut *lut, FwArgs &a, hipStream_t s)
{
    return lut->n_atoms <= 11 ? fw_prepare_n<11>(ctx, lut, a, s) : fw_prepare_n<12>(ctx, lut, a, s);
}

int amx_launch_fw_small(amx_ctx *ctx, FwArgs &a, const Plan &pl, hipStream_t s)
{
    // compile-time dictionary sizes: the reference's defaults (11 Human, 12 Mouse) exactly, 16 otherwise
    const int n = a.c.n_atoms;
    if (amx_fw_use_refill(ctx, n, a.c.nS, a.c.flags, a.c.lam2)) {
        if (n <= 11) return launch_refill(ctx, a, pl, s, k_fw_project<11>, k_fw_project_mfma<11, false>, k_fw_project_mfma<11, true>, k_freewater_refill<11>, 11,
                                          k_freewater_fused<11, false>, k_freewater_fused<11, true>, fuse_lds_bytes<11>(a.c.nS & 15));
        return launch_refill(ctx, a, pl, s, k_fw_project<12>, k_fw_project_mfma<12, false>, k_fw_project_mfma<12, true>, k_freewater_refill<12>, 12,
                             k_freewater_fused<12, false>, k_freewater_fused<12, true>, fuse_lds_bytes<12>(a.c.nS & 15));
    }
    if (n <= 11) return launch_lane(ctx, a, pl, s, k_freewater_lane<11>, sizeof(float), 11);
    if (n == 12) return launch_lane(ctx, a, pl, s, k_freewater_lane<12>, sizeof(float), 12);
    return launch_lane(ctx, a, pl, s, k_freewater_lane<16>, sizeof(float), 16);
}

// tables of the row-space solver, cached in the dictionary handle for one (lambda1, lambda2)
int amx_sandi_prepare(amx_ctx *ctx, const amx_lut *lut, SandiArgs &a, hipStream_t s)
{
    a.tables = nullptr;
    if (!(lut->nS == 6 && lut->n_atoms == 15 && amx_warm_start(a.c.lam2, a.c.flags))) return AMX_OK;   // other shapes / cold start: atom-space kernels
    if (lut->sandi_lam1 != a.c.lam1 || lut->sandi_lam2 != a.c.lam2 || !lut->sandi_prep) {
        if (lut->sandi_prep) HIPCHK(ctx, hipDeviceSynchronize());                              // (a fit with the old tables may still run)
        if (!lut->sandi_prep) HIPCHK(ctx, hipMalloc((void **)&lut->sandi_prep, kSandiTableWords * sizeof(double)));
        if (!lut->sandi_ready) HIPCHK(ctx, hipEventCreateWithFlags(&lut->sandi_ready, hipEventDisableTiming));
        hipLaunchKernelGGL((k_sandi_tables<6, 15>), dim3(1), dim3(64), 0, s, reinterpret_cast<const double *>(lut->tiles), lut->ldA,
                           lut->n_atoms, a.c.lam1, a.c.lam2, lut->sandi_prep);
        AMX_TRACE(ctx, s, "SANDI dictionary tables");
        HIPCHK(ctx, hipEventRecord(lut->sandi_ready, s));
        lut->sandi_lam1 = a.c.lam1; lut->sandi_lam2 = a.c.lam2;
    }
    HIPCHK(ctx, hipStreamWaitEvent(s, lut->sandi_ready, 0));
    a.tables = lut->sandi_prep;
    return AMX_OK;
}

int amx_launch_sandi_small(amx_ctx *ctx, SandiArgs &a, const Plan &pl, hipStream_t s)
{
    const int n = a.c.n_atoms;                // SANDI default: 5 + 5 + 5 = 15 atoms
    // the default protocol after the directional average (b0 + 5 shells = 6 values, 15 atoms): row-space solver
    // (a refill variant of this kernel -- lanes drawing the next voxel from a global counter -- was measured SLOWER,
    //  2.65 vs 2.29 ms per 1 M voxels: SANDI's optimum is dense, 12 of 15 atoms, so the lanes of a wavefront need
    //  nearly the same number of steps and there is no idle time to win back)
    if (a.c.nS == 6 && n == 15 && amx_warm_start(a.c.lam2, a.c.flags) && !ctx->opt_sandi_atom_space) {
        if (!a.tables) { ctx->err = "amx_launch_sandi_small: dictionary tables missing (amx_sandi_prepare)"; return AMX_E_BADARG; }
        rec(ctx, 2, s);
        hipLaunchKernelGGL((k_sandi_rows<6, 15>), dim3(a.n_lin > 0 ? (a.n_lin + 255) / 256 : ((pl.max_chunks + 7) / 8) * 8), dim3(256), 0, s, a);
        amx_note(ctx, "k_sandi_rows<6,15>");
        AMX_TRACE(ctx, s, "row-space SANDI solver");
        rec(ctx, 3, s);
        HIPCHK(ctx, hipGetLastError());
        return AMX_OK;
    }
    if (n <= 12) return launch_lane(ctx, a, pl, s, k_sandi_lane<12>, sizeof(double), 12);
    if (n <= 15) return launch_lane(ctx, a, pl, s, k_sandi_lane<15>, sizeof(double), 15);
    return launch_lane(ctx, a, pl, s, k_sandi_lane<16>, sizeof(double), 16);
}

#ifdef AMX_FW_PHASES
extern "C" int amx_debug_fw_phases(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fw_ph), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_fw_pp), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fw_ph), z, sizeof z) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fw_pp), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
