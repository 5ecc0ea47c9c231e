// amx_czb.hip -- CylinderZeppelinBall solver kernel (models.pyx:526-652)
#include "amx_launch.hpp"
using namespace amx;

template <int NR>
static int go(amx_ctx *ctx, CzbArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 1, MP = 32, MB = 64;      // 26 atoms by default: the main kernel's passive set holds them all
    constexpr int NW = 8;
    return launch_pair<NW>(ctx, a, pl, s, k_czb<NR, NQ, MP, NW, false>, k_czb<NR, NQ, MB, 1, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, true) + ((size_t)a.c.n_atoms * a.ldG + (size_t)(MP + 1) * (MP + 2) + MP + 1) * sizeof(double) + 16; },
                           fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, true), 0, 2);
}

// lambda2 below what the Gram form can take: thin QR in A-space (the other models route small ridges there too)
template <int NR>
static int go_qr(amx_ctx *ctx, CzbArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 1, MP = 32, MB = 64;
    constexpr int NW = 4;
    return launch_pair<NW>(ctx, a, pl, s, k_czb_qr<NR, NQ, MP, NW, false>, k_czb_qr<NR, NQ, MB, 1, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP); }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB), 0, 2);
}

int amx_launch_czb(amx_ctx *ctx, CzbArgs &a, const Plan &pl, hipStream_t s)
{
    // (8 rows per lane: protocols of up to 512 volumes -- the <= 64-atom tile still fits the LDS: 512 x 65 float32 = 133 KB)
    if (a.c.lam2 < 1e-6) return a.c.nS <= 128 ? go_qr<2>(ctx, a, pl, s) : (a.c.nS <= 256 ? go_qr<4>(ctx, a, pl, s) : go_qr<8>(ctx, a, pl, s));
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : (a.c.nS <= 256 ? go<4>(ctx, a, pl, s) : go<8>(ctx, a, pl, s));
}

// ================================================================== the default problem, fast: complementary form, one voxel per lane
// CylinderZeppelinBall's lasso has lambda1 = 0, lambda2 = 4 (models.pyx:439): H = A'A + lambda2 I is well conditioned and the optimum
// is DENSE -- 22 of the 26 atoms are passive -- so the small set is the one CLAMPED to zero.  With M = H^-1 (26 x 26 per orientation,
// shared by all its voxels) and z0 = M (A'y - lambda1) (the unconstrained optimum: a GEMM over the voxels), the optimum with the atoms
// Z clamped is
//     nu = -M_ZZ^-1 z0_Z,    x = z0 + M[:, Z] nu  (x_Z = 0),    gradient on Z = nu    (Kuhn-Tucker: x_P > 0, nu >= 0)
// i.e. a |Z| x |Z| Cholesky (|Z| ~ 5) per pivoting step instead of a |P| x |P| one, which fits a LANE's registers: block principal
// pivoting from Z0 = {z0 <= 0} reaches the unique optimum in 1.8 solves per voxel (tools/lab/czb_schur_lab.py: max |x - oracle| 8e-15).
// When Z is the larger half (noisy voxels: 8 % of the bench's) the same lane solves the textbook form on the passive set instead,
// H_PP x_P = c_P, gradient on Z = H[Z, P] x_P - c_Z, with c = A'y - lambda1 from the same GEMM: min(|Z|, |P|) <= 13 of 26 atoms.
//   k_czb_tables   once per (dictionary, lambda2): M, H, M 1 and the operands B = M A', A' per orientation
//   k_czb_project  z0 = B y - lambda1 M 1 and c = A'y - lambda1 for every voxel on the fp64 matrix cores (the only pass over the signals)
//   k_czb_lane     the pivoting, one voxel per lane, M and H in LDS; a voxel with both sets above kCzbZ (dictionaries of more than
//                  26 atoms only) goes to the wavefront-per-voxel kernel through the overflow list
constexpr int kCzbN = 32;          // atoms, padded
constexpr int kCzbZ = 13;          // atoms a lane's factor can hold: the clamped set Z, or -- when Z is the larger half -- the passive set P
constexpr int kCzbLd = 33;         // row stride of M in LDS (odd: per-lane row gathers spread over the banks)

struct CzbFastArgs {
    const double *y; const float *y32;
    const int *perm; const Chunk *schunks; const int *n_schunks;
    const double *tables;          // per orientation: M [32][33] | H [32][33] | m1 [32] | B [32][nSp] | A' [32][nSp]
    int table_stride, nSp, nS, n_atoms, n_rs, n_perp;
    const double *Rs;
    double lam1;
    double *Zb;                    // [n_blocks][64][64]: z0 (rows 0 .. 31) and c (rows 32 .. 63) of 64 voxels, atom-major
    double *est, *xdbg;
    int *status, *ovf_list, *ovf_count;
};

__global__ void __launch_bounds__(64) k_czb_tables(const float *__restrict__ tiles, int tile_stride, int nS, int ldA, int n_atoms,
                                                   const double *__restrict__ gram, int ldG, double lam2, double *__restrict__ out, int table_stride, int nSp)
{
    __shared__ double H[kCzbN][kCzbLd], L[kCzbN][kCzbLd], Mi[kCzbN][kCzbLd];
    const int lane = threadIdx.x, dir = blockIdx.x;
    const double *G = gram + (size_t)dir * n_atoms * ldG;
    double *T = out + (size_t)dir * table_stride;
    for (int e = lane; e < kCzbN * kCzbN; e += 64) {
        const int i = e / kCzbN, j = e - i * kCzbN;
        const bool in = i < n_atoms && j < n_atoms;
        H[i][j] = in ? G[(size_t)i * ldG + j] + (i == j ? lam2 : 0.0) : (i == j ? 1.0 : 0.0);
        T[kCzbN * kCzbLd + i * kCzbLd + j] = in ? H[i][j] : 0.0;        // the table's copy of H: zero outside the dictionary
        L[i][j] = 0.0;
    }
    if (lane < kCzbN) T[kCzbN * kCzbLd + lane * kCzbLd + kCzbN] = 0.0;
    __syncthreads();
    // Cholesky H = L L', right-looking on the lower triangle (lane i owns row i)
    for (int k = 0; k < kCzbN; k++) {
        const double dk = sqrt(H[k][k]);
        __syncthreads();
        if (lane >= k && lane < kCzbN) L[lane][k] = H[lane][k] / dk;
        __syncthreads();
        if (lane > k && lane < kCzbN) { for (int j = k + 1; j <= lane; j++) H[lane][j] -= L[lane][k] * L[j][k]; }
        __syncthreads();
    }
    // column c of the inverse by lane c: L w = e_c, L' m = w
    if (lane < kCzbN) {
        double w[kCzbN];
        for (int i = 0; i < kCzbN; i++) {
            double f = (i == lane) ? 1.0 : 0.0;
            for (int m = 0; m < i; m++) f -= L[i][m] * w[m];
            w[i] = f / L[i][i];
        }
        for (int i = kCzbN - 1; i >= 0; i--) {
            double f = w[i];
            for (int m = i + 1; m < kCzbN; m++) f -= L[m][i] * w[m];
            w[i] = f / L[i][i];
        }
        for (int i = 0; i < kCzbN; i++) Mi[i][lane] = w[i];
    }
    __syncthreads();
    for (int e = lane; e < kCzbN * kCzbLd; e += 64) {
        const int i = e / kCzbLd, j = e - i * kCzbLd;
        T[e] = (j < kCzbN && i < n_atoms && j < n_atoms) ? 0.5 * (Mi[i][j] + Mi[j][i]) : 0.0;
    }
    if (lane < kCzbN) {
        double s1 = 0.0;
        for (int j = 0; j < n_atoms; j++) s1 += (lane < n_atoms) ? Mi[lane][j] : 0.0;
        T[2 * kCzbN * kCzbLd + lane] = s1;
    }
    const float *tile = tiles + (size_t)dir * tile_stride;
    double *B = T + 2 * kCzbN * kCzbLd + kCzbN;
    double *At = B + (size_t)kCzbN * nSp;
    for (int i = lane; i < nSp; i += 64) {
        for (int r = 0; r < kCzbN; r++) {
            double s = 0.0;
            if (i < nS && r < n_atoms) for (int k = 0; k < n_atoms; k++) s += Mi[r][k] * (double)tile[i * ldA + k];
            B[(size_t)r * nSp + i] = s;
            At[(size_t)r * nSp + i] = (i < nS && r < n_atoms) ? (double)tile[i * ldA + r] : 0.0;
        }
    }
}

template <int KS>
__global__ void __launch_bounds__(512) k_czb_project(const CzbFastArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    double *A64 = reinterpret_cast<double *>(smem_p);                    // [4][KS][64]: B | A' in MFMA operand order
    double *m1 = A64 + 4 * KS * 64;                                       // [32] M 1
    const int cid = xcd_chunk((int)blockIdx.x, *a.n_schunks);
    if (cid < 0) return;
    const Chunk ck = a.schunks[cid];
    if (ck.count == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int q = lane >> 4, c16 = lane & 15, nS = a.nS;
    const double *T = a.tables + (size_t)ck.dir * a.table_stride;
    const double *B = T + 2 * kCzbN * kCzbLd + kCzbN;                     // rows 0 .. 31: M A', rows 32 .. 63: A'
    {
        // (four elements of a thread in flight: one guarded load per trip of a plain loop is a memory round trip per element)
        constexpr int UB = 4, NE = 4 * KS * 64;
        for (int e0 = threadIdx.x; e0 < NE; e0 += UB * (int)blockDim.x) {
            double v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int e = e0 + u * (int)blockDim.x, ec = e < NE ? e : 0;
                const int l = ec & 63, ks = (ec >> 6) % KS, mt = (ec >> 6) / KS;
                const int r = 16 * mt + (l & 15), i = 4 * ks + (l >> 4);
                v[u] = B[(size_t)r * a.nSp + (i < a.nSp ? i : 0)];
                v[u] = (i < a.nSp) ? v[u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UB; u++) { const int e = e0 + u * (int)blockDim.x; if (e < NE) A64[e] = v[u]; }
        }
    }
    if (threadIdx.x < kCzbN) m1[threadIdx.x] = T[2 * kCzbN * kCzbLd + threadIdx.x];
    __syncthreads();
    const int n_groups = (ck.count + 15) >> 4;
    double bn[KS];
    auto issue = [&](int g) {
        const int k = 16 * g + c16;
        const int vox = a.perm[ck.start + (k < ck.count ? k : ck.count - 1)];
        if (a.y32 != nullptr) {
            const float *yv = a.y32 + (size_t)vox * nS + q;
            float bf32[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bf32[ks] = (4 * ks + q < nS) ? yv[4 * ks] : 0.0f;      // (conversion outside the guard: loads in flight)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bn[ks] = (double)bf32[ks];
        } else {
            const double *yv = a.y + (size_t)vox * nS + q;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bn[ks] = (4 * ks + q < nS) ? yv[4 * ks] : 0.0;
        }
    };
    typedef double v4d __attribute__((ext_vector_type(4)));
    if (wave < n_groups) issue(wave);
    for (int g = wave; g < n_groups; g += nw) {
        const bool live = 16 * g + c16 < ck.count;
        double b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[ks] = live ? bn[ks] : 0.0;
        if (g + nw < n_groups) issue(g + nw);
        double *out = a.Zb + (size_t)(ck.pad + (g >> 2)) * 2 * kCzbN * 64 + 16 * (g & 3) + c16;
#pragma unroll 2
        for (int mt = 0; mt < 4; mt++) {
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A64[(mt * KS + ks) * 64 + lane], b[ks], acc, 0, 0, 0);
            if (live) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int row = 16 * mt + 4 * rr + q;
                    out[(size_t)row * 64] = acc[rr] - a.lam1 * (row < kCzbN ? m1[row] : ((row - kCzbN) < a.n_atoms ? 1.0 : 0.0));
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256, 1) k_czb_lane(const CzbFastArgs a)
{
    constexpr int N = kCzbN, ZM = kCzbZ, LD = kCzbLd, NT = ZM * (ZM + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
    double *Ml = reinterpret_cast<double *>(smem_l);                      // [32][33] M = H^-1
    double *Hl = Ml + N * LD;                                             // [32][33] H
    double *zt = Hl + N * LD + 2 + (threadIdx.x >> 6) * (2 * N * 64);     // this wavefront's block: z0 [32][64] | x [32][64]; a lane touches its own column only
    double *xt = zt + N * 64;
    const int cid = xcd_chunk((int)blockIdx.x, *a.n_schunks);
    if (cid < 0) return;
    const Chunk ck = a.schunks[cid];
    if (ck.count == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const double *T = a.tables + (size_t)ck.dir * a.table_stride;
    {
        constexpr int UB = 4, NE = 2 * N * LD;
        for (int e0 = threadIdx.x; e0 < NE; e0 += UB * (int)blockDim.x) {
            double v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) { const int e = e0 + u * (int)blockDim.x; v[u] = T[e < NE ? e : 0]; }
#pragma unroll
            for (int u = 0; u < UB; u++) { const int e = e0 + u * (int)blockDim.x; if (e < NE) Ml[e] = v[u]; }
        }
    }
    __syncthreads();
    const int n_atoms = a.n_atoms;
    const unsigned valid_atoms = n_atoms >= 32 ? ~0u : ((1u << n_atoms) - 1u);
    constexpr int kBackup = 3;
    const int n_blocks = (ck.count + 63) >> 6;
    for (int bl = wave; bl < n_blocks; bl += nw) {
        const int k = 64 * bl + lane;
        const bool valid = k < ck.count;
        const int vox = a.perm[ck.start + (valid ? k : ck.count - 1)];
        const double *src = a.Zb + (size_t)(ck.pad + bl) * 2 * N * 64 + lane;
#pragma unroll
        for (int j = 0; j < N; j++) zt[j * 64 + lane] = src[(size_t)j * 64];
        unsigned Z = 0u;
        bool finite = true;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const double z = zt[j * 64 + lane];
            finite = finite && (fabs(z) <= 1.79769313486231570e308);
            if (!(z > 0.0)) Z |= 1u << j;
        }
        Z &= valid_atoms;
        // the dual tests compare gradients (units of c = A'y - lambda1) with zero: the tolerance follows the voxel's own scale, so that
        // un-normalised signals (doNormalizeSignal = False: y ~ 1e3) do not flip a near-degenerate atom back and forth on rounding noise
        // c = A'y - lambda1 (rows 32 .. 63 of the block) waits in the lane's column of the x block: x is written when the lane is done
        // with the voxel, c is read only until then.  (Read from the hand-over block in HBM / L2 inside the pivoting loop, every
        // one of the 13 + 32 entries of a trip was a guarded load waited for in turn -- 45 memory round trips one after the other in
        // every trip in which some lane of the wavefront factors its passive set: most of the kernel.)
        double cmax = 1.0;
        {
            const double *cz0 = src + (size_t)N * 64;
            double cv[N];
#pragma unroll
            for (int j = 0; j < N; j++) cv[j] = cz0[(size_t)j * 64];
#pragma unroll
            for (int j = 0; j < N; j++) { xt[j * 64 + lane] = cv[j]; const double cj = fabs(cv[j]); cmax = (((valid_atoms >> j) & 1u) && cj > cmax) ? cj : cmax; }
        }
        const double tol = 1e-12 * cmax;
        bool active = valid && finite, overflow = false;
        int ninf = N + 1, backup = 0, its = 0;
        for (int guard = 0; guard < 6 * N + 32; guard++) {
            if (__ballot(active) == 0ull) break;
            // the smaller of the two sets gets the factor: zform -- the clamped atoms on M; else the passive atoms on H
            const int nz = __builtin_popcount(Z);
            const bool zform = nz <= ZM;
            const unsigned S = zform ? Z : (valid_atoms & ~Z);
            const int ns_ = __builtin_popcount(S);
            if (active && ns_ > ZM) { overflow = true; active = false; }
            const int ns = active ? ns_ : 0;
            int si[ZM];
            {
                unsigned rem = active ? S : 0u;
#pragma unroll
                for (int s = 0; s < ZM; s++) { si[s] = rem ? __builtin_ctz(rem) : 0; rem &= rem - 1u; }
            }
            const double *Src = zform ? Ml : Hl;
            const double *cz = xt + lane;                                // c = A'y - lambda1, the lane's column (see above)
            double Tt[NT], dinv[ZM], sol[ZM];
#pragma unroll
            for (int s = 0; s < ZM; s++) {
#pragma unroll
                for (int t = 0; t <= s; t++) Tt[s * (s + 1) / 2 + t] = (s < ns) ? Src[si[s] * LD + si[t]] : (s == t ? 1.0 : 0.0);
                sol[s] = (s < ns) ? (zform ? -zt[si[s] * 64 + lane] : cz[(size_t)si[s] * 64]) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < ZM; j++) {
                double dj = Tt[j * (j + 1) / 2 + j];
#pragma unroll
                for (int m = 0; m < j; m++) dj -= Tt[j * (j + 1) / 2 + m] * Tt[j * (j + 1) / 2 + m];
                dinv[j] = (dj > 0.0) ? amx::inv_sqrt(dj) : 0.0;
                Tt[j * (j + 1) / 2 + j] = dj * dinv[j];
#pragma unroll
                for (int i = j + 1; i < ZM; i++) {
                    double v = Tt[i * (i + 1) / 2 + j];
#pragma unroll
                    for (int m = 0; m < j; m++) v -= Tt[i * (i + 1) / 2 + m] * Tt[j * (j + 1) / 2 + m];
                    Tt[i * (i + 1) / 2 + j] = v * dinv[j];
                }
            }
#pragma unroll
            for (int j = 0; j < ZM; j++) {
                double f = sol[j];
#pragma unroll
                for (int m = 0; m < j; m++) f -= Tt[j * (j + 1) / 2 + m] * sol[m];
                sol[j] = f * dinv[j];
            }
#pragma unroll
            for (int j = ZM - 1; j >= 0; j--) {
                double f = sol[j];
#pragma unroll
                for (int m = j + 1; m < ZM; m++) f -= Tt[m * (m + 1) / 2 + j] * sol[m];
                sol[j] = f * dinv[j];
            }
            // acc = base + Src[S, :]' sol over all atoms (rows of the symmetric matrices: consecutive LDS words per lane):
            //   zform: base = z0, acc = x on the passive atoms, the gradient on Z is sol itself
            //   else : base = -c, acc = the gradient H x - c on the clamped atoms, x on P is sol itself
            double acc[N];
#pragma unroll
            for (int j = 0; j < N; j++) acc[j] = zform ? zt[j * 64 + lane] : -cz[(size_t)j * 64];
#pragma unroll
            for (int s = 0; s < ZM; s++) {
                if (__ballot(s < ns) == 0ull) break;
                const double *row = Src + si[s] * LD;
                const double w = (s < ns) ? sol[s] : 0.0;
#pragma unroll
                for (int j = 0; j < N; j++) acc[j] += row[j] * w;
            }
            unsigned v1 = 0u, v2 = 0u;                                    // primal / dual infeasible atoms
#pragma unroll
            for (int j = 0; j < N; j++) {
                const bool inz = (Z >> j) & 1u, ok = (valid_atoms >> j) & 1u;
                if (zform && !inz && ok && !(acc[j] > 0.0)) v1 |= 1u << j;
                if (!zform && inz && ok && acc[j] < -tol) v2 |= 1u << j;
            }
#pragma unroll
            for (int s = 0; s < ZM; s++) {
                if (s < ns && zform && sol[s] < -tol) v2 |= 1u << si[s];
                if (s < ns && !zform && !(sol[s] > 0.0)) v1 |= 1u << si[s];
            }
            if (active) {
                const unsigned bad = v1 | v2;
                const int nbad = __builtin_popcount(bad);
                if (bad == 0u || its > 4 * N + 16) {
                    active = false;
                    if (bad != 0u) overflow = true;      // no optimum within the cap: the wavefront-per-voxel solver takes the voxel (overflow list), nothing approximate is emitted
                    // the coefficients, in the lane's column of the LDS block (a clamped or infeasible iterate never reaches the maps negative)
#pragma unroll
                    for (int j = 0; j < N; j++) xt[j * 64 + lane] = (zform && !((Z >> j) & 1u) && acc[j] > 0.0) ? acc[j] : 0.0;
                    if (!zform) {
#pragma unroll
                        for (int s = 0; s < ZM; s++) if (s < ns && sol[s] > 0.0) xt[si[s] * 64 + lane] = sol[s];
                    }
                } else {
                    bool block = false;
                    if (nbad < ninf) { ninf = nbad; backup = kBackup; block = true; }
                    else if (backup > 0) { backup--; block = true; }
                    const unsigned ex = block ? bad : (1u << (31 - __builtin_clz(bad)));
                    Z ^= ex;
                    its++;
                }
            }
        }
        if (valid && overflow) { const int kq = atomicAdd(a.ovf_count, 1); a.ovf_list[kq] = vox; }
        if (valid && !overflow) {
            double *e = a.est + (size_t)vox * 3;
            if (!finite) {
                const double nan = __builtin_nan("");
                e[0] = nan; e[1] = nan; e[2] = nan;
                if (a.xdbg) for (int j = 0; j < n_atoms; j++) a.xdbg[(size_t)vox * n_atoms + j] = nan;
            } else {
                // models.pyx:616-633
                double f1 = 0.0, f2 = 0.0, am = 0.0;
#pragma unroll
                for (int j = 0; j < N; j++) {
                    const double xj = xt[j * 64 + lane];
                    if (j < a.n_rs) { f1 += xj; am += a.Rs[j] * xj; }
                    else if (j < a.n_rs + a.n_perp) f2 += xj;
                    if (a.xdbg && j < n_atoms) a.xdbg[(size_t)vox * n_atoms + j] = xj;
                }
                f2 += 1e-16;
                const double v = f1 / (f1 + f2 + 1e-16);
                f1 += 1e-16;
                am = 1e6 * 2.0 * am / f1;
                e[0] = v; e[1] = am; e[2] = (4.0 * v) / (3.14159265358979323846 * (am * am) + 1e-16);
            }
        }
    }
}

size_t amx_czb_table_stride(int nS) { const int nSp = nS <= 100 ? 100 : 160; return (size_t)2 * kCzbN * kCzbLd + kCzbN + (size_t)2 * kCzbN * nSp; }

// tables of the fast path, cached in the dictionary handle for one lambda2
int amx_czb_prepare(amx_ctx *ctx, const amx_lut *lut, double lam2, hipStream_t s)
{
    const int nSp = lut->nS <= 100 ? 100 : 160;
    const size_t stride = amx_czb_table_stride(lut->nS);
    if (lut->czb_lam2 != lam2 || !lut->czb_prep) {
        if (lut->czb_prep) HIPCHK(ctx, hipDeviceSynchronize());            // (a fit with the old tables may still run)
        if (!lut->czb_prep) HIPCHK(ctx, hipMalloc((void **)&lut->czb_prep, (size_t)lut->ndirs * stride * sizeof(double) + 64));
        if (!lut->czb_ready) HIPCHK(ctx, hipEventCreateWithFlags(&lut->czb_ready, hipEventDisableTiming));
        hipLaunchKernelGGL(k_czb_tables, dim3(lut->ndirs), dim3(64), 0, s, (const float *)lut->tiles, lut->tile_stride, lut->nS, lut->ldA, lut->n_atoms,
                           (const double *)lut->gram, lut->ldG, lam2, lut->czb_prep, (int)stride, nSp);
        AMX_TRACE(ctx, s, "CylinderZeppelinBall tables (M = H^-1, M A')");
        HIPCHK(ctx, hipEventRecord(lut->czb_ready, s));
        lut->czb_lam2 = lam2;
    }
    HIPCHK(ctx, hipStreamWaitEvent(s, lut->czb_ready, 0));
    return AMX_OK;
}

int amx_launch_czb_fast(amx_ctx *ctx, const amx_lut *lut, CzbArgs &a, const Plan &pl, hipStream_t s)
{
    CzbFastArgs f;
    memset(&f, 0, sizeof f);
    f.y = a.c.y; f.y32 = a.c.y32; f.perm = pl.perm; f.schunks = pl.schunks; f.n_schunks = pl.n_chunks + 1;
    f.tables = lut->czb_prep; f.table_stride = (int)amx_czb_table_stride(lut->nS); f.nSp = lut->nS <= 100 ? 100 : 160;
    f.nS = lut->nS; f.n_atoms = lut->n_atoms; f.n_rs = lut->n_rs; f.n_perp = lut->n_perp; f.Rs = lut->Rs; f.lam1 = a.c.lam1;
    f.Zb = (double *)ctx->cgemm.p; f.est = a.est; f.xdbg = a.c.xdbg; f.status = a.c.status;
    f.ovf_count = pl.ovf_count; f.ovf_list = pl.ovf_list;
    const dim3 grid(((pl.max_schunks + 7) / 8) * 8);
    int rc;
    rec(ctx, 2, s);
    if (lut->nS <= 100) {
        const size_t lds = ((size_t)4 * 25 * 64 + kCzbN) * sizeof(double);
        if ((rc = set_lds(ctx, k_czb_project<25>, lds))) return rc;
        hipLaunchKernelGGL(k_czb_project<25>, grid, dim3(512), lds, s, f);
    } else {
        const size_t lds = ((size_t)4 * 40 * 64 + kCzbN) * sizeof(double);
        if ((rc = set_lds(ctx, k_czb_project<40>, lds))) return rc;
        hipLaunchKernelGGL(k_czb_project<40>, grid, dim3(512), lds, s, f);
    }
    amx_note(ctx, lut->nS <= 100 ? "k_czb_project<25>" : "k_czb_project<40>");
    AMX_TRACE(ctx, s, "z0 = M A'y on the matrix cores");
    const size_t lds2 = ((size_t)2 * kCzbN * kCzbLd + 2 + (size_t)4 * 2 * kCzbN * 64) * sizeof(double);
    if ((rc = set_lds(ctx, k_czb_lane, lds2))) return rc;
    hipLaunchKernelGGL(k_czb_lane, grid, dim3(256), lds2, s, f);
    amx_note(ctx, "k_czb_lane");
    AMX_TRACE(ctx, s, "complementary-form pivoting, one voxel per lane");
    // voxels with more clamped atoms than a lane holds: the wavefront-per-voxel kernel, one wavefront per workgroup
    {
        CzbArgs b = a;
        b.c.ovf_count = pl.ovf_count + 8; b.c.ovf_list = pl.ovf_list + 3 * pl.n;
        b.c.list = pl.ovf_list; b.c.list_count = pl.ovf_count;
        const size_t lds_list = fit_lds_bytes<float>(a.c.nS, a.c.ldA, 2, 1, 1, 64, true);
        if (a.c.nS <= 128) {
            if ((rc = set_lds(ctx, (k_czb<2, 1, 64, 1, true>), lds_list))) return rc;
            hipLaunchKernelGGL((k_czb<2, 1, 64, 1, true>), dim3(kListGrid), dim3(64), lds_list, s, b);
        } else {
            const size_t l4 = fit_lds_bytes<float>(a.c.nS, a.c.ldA, 4, 1, 1, 64, true);
            if ((rc = set_lds(ctx, (k_czb<4, 1, 64, 1, true>), l4))) return rc;
            hipLaunchKernelGGL((k_czb<4, 1, 64, 1, true>), dim3(kListGrid), dim3(64), l4, s, b);
        }
    }
    rec(ctx, 3, s);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}
