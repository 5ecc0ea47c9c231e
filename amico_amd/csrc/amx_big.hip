// amx_big.hip -- the slow exact solver behind the NODDI LASSO stage: passive sets of ANY size (round 6).
//
// The reference solves whatever it is given (models.pyx:926: cyspams lasso, LARS, no cap below min(m, n)); the fast kernels of this
// library hold at most 64 passive atoms per voxel (k_noddi<4, .., 64, 1, true>: the factor lives in one wavefront's LDS block, a slot per
// lane) and used to answer AMX_E_OVERFLOW beyond that -- NODDI with lambda1 = 0, a legal set_solver(), has a DENSE optimum on most of
// its 144 atoms and could not be fitted at all (missing 3).  k_noddi_lasso_big takes the voxels that overflow there -- or,
// when lambda1 = 0, every voxel, straight away -- one WORKGROUP per voxel:
//     min_x 1/2 ||y2 - A2 S x||^2 + lambda1 sum(x) + lambda2/2 ||x||^2,  x >= 0          (models.pyx:914-926; y2, A2, S as in noddi_voxel)
// in Gram space, H = S G_dwi S + lambda2 I from the orientation's Gram matrix, c = S A2'y2 - lambda1, by block principal pivoting from the
// FULL set (Judice & Pires; the rule of GramSolver::solve_dense): solve H_PP z = c_P by a dense Cholesky factorisation -- in LDS up to
// 176 candidate atoms, in a global scratch block per workgroup beyond --, dual values g = c - H z off P, exchange ALL infeasible atoms while
// their number keeps falling (then kBackup more times), else the one with the largest index (Murty: finite for a positive definite H).
// lambda2 > 0 makes the problem strictly convex: the point it stops at IS the optimum the reference's LARS walks to.  A dense 144-atom
// voxel costs ~6 factorisations of mostly 30 - 60 atoms, 0.7 ms of a workgroup: slow (0.4 M voxels/s -- four times the reference's CPU path on
// this box's 128 cores), exact, never an error.  Output = what noddi_voxel<4> leaves: the support
// bits of the voxel (and the coefficient vector for AMX_F_DEBUG_X).
#include "amx_launch.hpp"
using namespace amx;

namespace {

struct BigArgs {
    NoddiArgs a;
    const int *list;              // bucket positions (list_is_pos) or voxel numbers; null: every bucket position 0 .. n_all - 1
    const int *count;             // number of list entries (device), or null with n_all
    int n_all;
    double *Lg;                   // global factor blocks [gridDim.x][n_cand * n_cand] (n_cand > kBigLdsAtoms), or null
};

constexpr int kBigLdsAtoms = 176;         // largest candidate set whose packed factor fits the LDS next to the vectors (124 KB)
constexpr int kBigThreads = 256;

__device__ __forceinline__ int tri_at(int r, int s) { return r * (r + 1) / 2 + s; }      // (<= 256 atoms: 32 896 entries)

template <bool LDSL>
__global__ void __launch_bounds__(kBigThreads) k_noddi_lasso_big(const BigArgs b)
{
    const NoddiArgs &a = b.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n = a.n_wm;      // candidates: the wm atoms 0 .. n_wm - 1
    const int iso_atom = n_atoms - 1, dot_atom = a.is_exvivo ? n_atoms - 2 : -1;
    double *y2 = reinterpret_cast<double *>(smem_b);              // [nS]
    double *cv = y2 + ((nS + 1) & ~1);                             // [n] c_j = s_j a_j'y2 - lambda1
    double *zv = cv + ((n + 1) & ~1);                              // [n] solution on P (slot order), scratch of the solves
    double *sc = zv + ((n + 1) & ~1);                              // [n] column scales
    double *zat = sc + ((n + 1) & ~1);                             // [n] solution in atom order
    int *plist = reinterpret_cast<int *>(zat + ((n + 1) & ~1));    // [n] atoms of P, ascending
    int *inP = plist + ((n + 3) & ~3);                             // [n] 1 = passive
    int *bad = inP + ((n + 3) & ~3);                               // [n] 1 = infeasible this step
    int *shi = bad + ((n + 3) & ~3);                               // [8] scalars: 0 np, 1 n_bad, 2 largest bad atom, 3 pivot failure, 4 non-finite
    unsigned long long *wmask = reinterpret_cast<unsigned long long *>(shi + 8);     // [4]
    double *Ll = reinterpret_cast<double *>(wmask + 4);            // packed lower triangle (LDSL)
    double *L = LDSL ? Ll : b.Lg + (size_t)blockIdx.x * n * n;     // (global: packed as well)
    const int tid = threadIdx.x, nt = blockDim.x;
    const float *tiles = reinterpret_cast<const float *>(a.c.tiles);
    const int cnt = b.count ? *b.count : b.n_all;
    const double tol = 1e-12;
    constexpr int kBackup = 3, kStart = 32;

    for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
        const int e = b.list ? b.list[it] : it;
        const bool is_pos = b.list == nullptr || a.list_is_pos;
        const int vox = is_pos ? a.c.perm[e] : e;
        const int dir = a.c.lutidx[vox];
        const float *At = tiles + (size_t)dir * a.c.tile_stride;
        const double *G = a.gram_dwi + (size_t)dir * n_atoms * a.ldG;
        __syncthreads();
        if (tid < 8) shi[tid] = 0;
        if (tid < 4) wmask[tid] = 0ull;
        __syncthreads();
        // ---- y2 = max(0, y - x_iso iso (- x_dot)) on the stage-2 rows, 0 elsewhere (models.pyx:917-925; noddi_voxel)
        const double xiso = a.xiso[(size_t)vox * 2], xdot = a.xiso[(size_t)vox * 2 + 1];
        for (int i = tid; i < nS; i += nt) {
            const double yi = a.c.y32 ? (double)a.c.y32[(size_t)vox * nS + i] : a.c.y[(size_t)vox * nS + i];
            if (!(fabs(yi) <= 1.79769313486231570e308)) shi[4] = 1;
            double t = 0.0;
            if (a.rowdwi[i]) {
                t = yi - xiso * (double)At[(size_t)i * ldA + iso_atom];
                if (a.is_exvivo) t -= xdot * 1.0;
                if (t < 0.0) t = 0.0;
            }
            y2[i] = t;
        }
        __syncthreads();
        if (shi[4] || !(fabs(xiso) <= 1.79769313486231570e308)) {       // non-finite signal: no support (stage 3 writes the NaN maps)
            if (tid < 4) a.supp[(size_t)vox * 4 + tid] = 0ull;
            continue;
        }
        // ---- c, scales; the first passive set = the kStart atoms that fit the signal best on their own (c_j / sqrt(H_jj), c_j > 0).  Block
        // pivoting converges from any start (Murty's rule is finite for a positive definite H); from the FULL set its first factorisation is
        // n^3 / 6 = 500 000 updates for 144 atoms, seven eighths of a voxel's time when the optimum holds 37 -- from 32 atoms it is 5 000,
        // and a dense optimum is reached by the block additions of the next steps (50 000 voxels, lambda1 = 0: 419 -> 129 ms with the 16 x 16 mapping of the trailing update, 7.4 -> 6.2 steps per voxel)
        for (int j = tid; j < n; j += nt) {
            double acc = 0.0;
            for (int i = 0; i < nS; i++) acc += (double)At[(size_t)i * ldA + j] * y2[i];
            const double s = a.colscale[j];
            sc[j] = s; cv[j] = s * acc - a.c.lam1;
            zat[j] = cv[j] > 0.0 ? cv[j] / sqrt(s * s * G[(size_t)j * a.ldG + j] + a.c.lam2) : -1.0;       // (score; zat is free until the first solve)
        }
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            const double sj = zat[j];
            int rank = 0;
            for (int k = 0; k < n; k++) rank += (zat[k] > sj || (zat[k] == sj && k < j)) ? 1 : 0;
            inP[j] = (sj > 0.0 && rank < kStart) ? 1 : 0;
        }
        __syncthreads();
        int ninf = n + 1, backup = 0, status = kSolved, steps_done = 0;
        for (int step = 0;; ++step) {
            if (step > 4 * n + 16) { status = kIterCap; break; }
            steps_done = step + 1;
            // ---- P in ascending order
            for (int j = tid; j < n; j += nt) {
                if (inP[j]) {
                    int r = 0;
                    for (int k = 0; k < j; k++) r += inP[k];
                    plist[r] = j;
                }
            }
            if (tid == 0) { int c2 = 0; for (int k = 0; k < n; k++) c2 += inP[k]; shi[0] = c2; shi[1] = 0; shi[2] = -1; }
            __syncthreads();
            const int np = shi[0];
            // ---- H_PP (packed lower triangle) and the right-hand side
            for (int r = tid; r < np; r += nt) {
                const int pr = plist[r];
                const double sr = sc[pr];
                const double *Gr = G + (size_t)pr * a.ldG;
                for (int s = 0; s <= r; s++) L[tri_at(r, s)] = sr * sc[plist[s]] * Gr[plist[s]] + (s == r ? a.c.lam2 : 0.0);
                zv[r] = cv[pr];
            }
            __syncthreads();
            // ---- Cholesky in place, right-looking (lambda2 > 0: every pivot >= lambda2)
            for (int k = 0; k < np; k++) {
                const double dk = L[tri_at(k, k)];
                if (!(dk > 0.0)) { if (tid == 0) shi[3] = 1; break; }
                const double d = sqrt(dk), di = 1.0 / d;
                __syncthreads();
                for (int r = k + tid; r < np; r += nt) L[tri_at(r, k)] = (r == k) ? d : L[tri_at(r, k)] * di;
                __syncthreads();
                // (16 x 16 threads over the trailing triangle: a thread per row left the short rows' threads idle)
                for (int r = k + 1 + (tid >> 4); r < np; r += 16) {
                    const double lrk = L[tri_at(r, k)];
                    for (int s = k + 1 + (tid & 15); s <= r; s += 16) L[tri_at(r, s)] -= lrk * L[tri_at(s, k)];
                }
                __syncthreads();
            }
            __syncthreads();
            if (shi[3]) { status = kGuardOuter; break; }
            // ---- L w = c_P, L'z = w (column oriented: one step per pivot)
            for (int k = 0; k < np; k++) {
                const double wk = zv[k] / L[tri_at(k, k)];
                __syncthreads();
                if (tid == 0) zv[k] = wk;
                for (int r = k + 1 + tid; r < np; r += nt) zv[r] -= L[tri_at(r, k)] * wk;
                __syncthreads();
            }
            for (int k = np - 1; k >= 0; k--) {
                const double zk = zv[k] / L[tri_at(k, k)];
                __syncthreads();
                if (tid == 0) zv[k] = zk;
                for (int r = tid; r < k; r += nt) zv[r] -= L[tri_at(k, r)] * zk;
                __syncthreads();
            }
            for (int j = tid; j < n; j += nt) zat[j] = 0.0;
            __syncthreads();
            for (int r = tid; r < np; r += nt) zat[plist[r]] = zv[r];
            __syncthreads();
            // ---- infeasible atoms: passive with z <= 0, inactive with a positive dual value g_j = c_j - sum_s H_js z_s
            for (int j = tid; j < n; j += nt) {
                bool v;
                if (inP[j]) v = !(zat[j] > 0.0);
                else {
                    const double *Gj = G + (size_t)j * a.ldG;
                    double g = 0.0;
                    for (int r = 0; r < np; r++) g += sc[plist[r]] * Gj[plist[r]] * zv[r];
                    v = (cv[j] - sc[j] * g) > tol;
                }
                bad[j] = v ? 1 : 0;
                if (v) { atomicAdd(&shi[1], 1); atomicMax(&shi[2], j); }
            }
            __syncthreads();
            const int nbad = shi[1], top = shi[2];
            if (nbad == 0) break;                                   // Kuhn-Tucker point of a strictly convex problem: the optimum
            bool block = false;
            if (nbad < ninf) { ninf = nbad; backup = kBackup; block = true; }
            else if (backup > 0) { backup--; block = true; }
            __syncthreads();
            for (int j = tid; j < n; j += nt)
                if (block ? bad[j] != 0 : j == top) inP[j] ^= 1;
            __syncthreads();
        }
        if (status == kIterCap && tid == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
        if (status > kIterCap && tid == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + status; }
        // ---- the stage's output: support bits (x > 0), coefficients for AMX_F_DEBUG_X
        for (int j = tid; j < n; j += nt)
            if (inP[j] && zat[j] > 0.0) atomicOr(&wmask[j >> 6], 1ull << (j & 63));
        __syncthreads();
        if (tid < 4) a.supp[(size_t)vox * 4 + tid] = wmask[tid];
        if (a.c.xdbg) {
            double *dst = a.c.xdbg + ((size_t)vox * 3 + 1) * n_atoms;
            for (int j = tid; j < n_atoms; j += nt) dst[j] = (j < n && inP[j] && zat[j] > 0.0) ? zat[j] : 0.0;
            __syncthreads();
            if (tid == 0) { dst[iso_atom] = xiso; if (dot_atom >= 0) dst[dot_atom] = xdot; }
        }
        if (tid == 0) { atomicAdd(&a.c.status[ST_GRAM + 1], 1); atomicAdd(&a.c.status[ST_ITERS + 1], steps_done); }      // (AMX_DEBUG=1 prints them: voxels, pivoting steps)
    }
}

size_t big_lds(int nS, int n, bool ldsl)
{
    size_t w = (size_t)((nS + 1) & ~1) + 4 * (size_t)((n + 1) & ~1);                                  // doubles
    size_t bytes = w * sizeof(double) + 3 * (size_t)((n + 3) & ~3) * sizeof(int) + 8 * sizeof(int) + 4 * sizeof(unsigned long long);
    bytes = (bytes + 15) & ~(size_t)15;
    if (ldsl) bytes += ((size_t)n * (n + 1) / 2) * sizeof(double);
    return bytes;
}

}  // namespace

// list == nullptr: every voxel of the call (n_all of them, bucket order); else the bucket positions / voxel numbers of *count entries
int amx_launch_noddi_big(amx_ctx *ctx, const NoddiArgs &a, const Plan &pl, hipStream_t s, const int *list, const int *count, int n_all)
{
    BigArgs b;
    memset(&b, 0, sizeof b);
    b.a = a; b.list = list; b.count = count; b.n_all = n_all;
    const int n = a.n_wm;
    const bool ldsl = n <= kBigLdsAtoms;
    const int grid = list ? 256 : (n_all < 2 * ctx->n_cu ? (n_all > 0 ? n_all : 1) : 2 * ctx->n_cu);
    if (!ldsl) {
        int rc;
        if ((rc = amx_ensure(ctx, ctx->big, (size_t)grid * n * n * sizeof(double)))) return rc;
        b.Lg = (double *)ctx->big.p;
    }
    const size_t lds = big_lds(a.c.nS, n, ldsl);
    if (lds > kLdsPerCU) { ctx->err = "k_noddi_lasso_big: protocol too long for the LDS vectors"; return AMX_E_BADARG; }
    int rc;
    if (ldsl) {
        if ((rc = set_lds(ctx, k_noddi_lasso_big<true>, lds))) return rc;
        hipLaunchKernelGGL(k_noddi_lasso_big<true>, dim3(grid), dim3(kBigThreads), lds, s, b);
    } else {
        if ((rc = set_lds(ctx, k_noddi_lasso_big<false>, lds))) return rc;
        hipLaunchKernelGGL(k_noddi_lasso_big<false>, dim3(grid), dim3(kBigThreads), lds, s, b);
    }
    amx_note(ctx, list ? "k_noddi_lasso_big (supports beyond 64 atoms)" : "k_noddi_lasso_big (all voxels: lambda1 = 0)");
    AMX_TRACE(ctx, s, "LASSO, any support size (block principal pivoting, workgroup per voxel)");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}
