// amx_seed.hpp -- support SEEDS for the NNLS stages of the NODDI fit (models.pyx:911, 940).
//
// Lawson-Hanson reaches the 4.3 atoms of a NODDI stage-1 optimum by ~10 column additions and ~5.6 removals -- the path
// walks along the (kappa, v_ic) grid of nearly collinear atoms.  Two thirds of the solver's work is path, not destination
// (DESIGN.md section 5).  But WHICH atoms end up in the support is decided by a dozen directions of signal space: the
// dictionary of one orientation has singular values 1, .16, .11, .016, .009, .003, 6e-4, 1e-4, 2e-5, 2e-6, 7e-7, 4e-8 and
// then the float32 rounding floor of its entries (4e-9); the NNLS problem projected onto a rank-12 basis has the SAME
// support as the full problem in 97 % of the voxels (tools/lab/*.py).  So:
//
//   k_build_basis   once per dictionary and orientation: U [nS][KD] = an orthonormal basis of the dominant column space
//                   (pivoted Gram-Schmidt with re-orthogonalisation), S = U'A [n_atoms][KD]
//   k_noddi_project y~ = U'y of every voxel (bucket order), 12 doubles per voxel
//   k_nnls_seed     ONE VOXEL PER LANE: Lawson-Hanson on min ||S x - y~||, x >= 0 -- 12 rows instead of 99, so a voxel's
//                   whole state (y~, the passive set, the 8x8 Gram block of the passive columns) lives in the lane's
//                   registers and nothing crosses lanes; the dual vector is one sweep over S with wave-uniform (scalar)
//                   operands.  Output: the passive set, 8 bytes per voxel.
//   NNSolver::solve (amx_solver.hpp) then only CERTIFIES the seed in the full 99-row problem: least squares on the seeded
//                   columns (semi-normal equations with refinement on the true residual), one exact sweep of the dual
//                   vector, strict KKT test.  A seed that fails ANY test falls through to the unchanged Lawson-Hanson
//                   solver from the empty set, so the result is always the full problem's own KKT point: the compressed
//                   problem only proposes, it never decides.
#pragma once
#include "amx_kernels.hpp"

namespace amx {

constexpr int kSeedMax = 8;      // passive-set capacity of the seed solver (= MAXP of the NNLS stage kernels)
constexpr int kSeed3ListRow = 36;      // bytes per lane of the stage-3 candidate lists in LDS (k_nnls_seed<3>)
#ifndef SEED3_SCAN_MAX
#define SEED3_SCAN_MAX 32      // (diagnosis: a smaller value truncates the stage-3 candidate scan)
#endif
// one column of S for THIS lane from the LDS copy (row stride kSeedLd doubles = 112 bytes: 16-byte aligned rows): 16-byte reads --
// the per-lane gathers are bound by the NUMBER of LDS instructions (a float32 copy with half the bytes changed nothing, reads of
// twice the width took 9 % off the stage-3 seed solver)
// A chunk's tables global -> LDS with UB elements of a thread in flight together.  (As `dst[e] = src[e]` in a plain loop the compiler
// keeps ONE load per thread in flight -- it may not hoist a load over the guard of its iteration --, so staging a chunk was a chain
// of 7 + 8 memory round trips: ~13 us per workgroup and chunk, all of it exposed in small calls and in the table GEMM, whose one
// workgroup per CU multiplies nothing meanwhile.  Here the loads are unconditional at clamped indices and the guards sit on the stores.)
#ifndef AMX_STAGE_UBN
#define AMX_STAGE_UBN 2
#endif
#ifndef AMX_STAGE_UB
#define AMX_STAGE_UB 4
#endif
template <int KD, int LD, int UB = AMX_STAGE_UB>
__device__ __forceinline__ void stage_rows(double *Sl, const double *__restrict__ Sg, int n, int src_ld)
{
    const int N = n * KD;
    for (int e0 = threadIdx.x; e0 < N; e0 += UB * (int)blockDim.x) {
        double v[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = e0 + u * (int)blockDim.x, ec = e < N ? e : 0, j = ec / KD, d = ec - j * KD;
            v[u] = Sg[(size_t)j * src_ld + d];
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = e0 + u * (int)blockDim.x, j = e / KD, d = e - j * KD;
            if (e < N) Sl[j * LD + d] = v[u];
        }
    }
}
// the same rows in MFMA operand order: Aop[(mt KS + ks) 64 + l] = S[16 mt + (l & 15)][4 ks + (l >> 4)], 0 beyond n rows;
// NORM: every row scaled to unit length (k_nnls_seed stage 1)
template <int KS, int MT, bool NORM = false, int UB = AMX_STAGE_UB>
__device__ __forceinline__ void stage_operand(double *Aop, const double *__restrict__ Sg, int n, int src_ld)
{
    constexpr int KD = 4 * KS, N = MT * KS * 64;
    for (int e0 = threadIdx.x; e0 < N; e0 += UB * (int)blockDim.x) {
        double v[UB], n2[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = e0 + u * (int)blockDim.x, ec = e < N ? e : 0;
            const int l = ec & 63, ks = (ec >> 6) % KS, mt = (ec >> 6) / KS;
            const int atom = 16 * mt + (l & 15), d = 4 * ks + (l >> 4);
            const double *row = Sg + (size_t)(atom < n ? atom : 0) * src_ld;
            v[u] = row[d];
            n2[u] = 1.0;
            if (NORM) {
                double t2 = 0.0;
#pragma unroll
                for (int dd = 0; dd < KD; dd++) { const double t = row[dd]; t2 += t * t; }
                n2[u] = t2;
            }
            v[u] = atom < n ? v[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int e = e0 + u * (int)blockDim.x;
            double w = v[u];
            if (NORM) w = n2[u] > 0.0 ? w * inv_sqrt(n2[u]) : 0.0;
            if (e < N) Aop[e] = w;
        }
    }
}

template <int KD>
__device__ __forceinline__ void seed_col(const double *col, double (&cv)[KD])
{
    const double2 *c2 = reinterpret_cast<const double2 *>(col);
#pragma unroll
    for (int d = 0; d < KD; d += 2) { const double2 p = c2[d >> 1]; cv[d] = p.x; cv[d + 1] = p.y; }
}
constexpr int kSeedLd = 14;      // LDS row stride of S (odd: per-lane column gathers spread over the banks)
constexpr unsigned long long kNoSeed = kSeedNone;
// round 6 (AMX_FORK bit 1): a voxel the LASSO certificates left over is finished on a SIDE stream (k_noddi<4> -> k_noddi<3>, a wavefront per
// voxel) while the stage-3 lane kernels run: k_nnls_seed<3> marks it so, k_nnls_gcert<3> neither certifies it nor lists it
constexpr unsigned long long kSeedForked = 0xfffffffffffffffdull;

// ------------------------------------------------------------------ basis of one orientation
// One workgroup per orientation.  tile: float [nS][ldA]; rowsel (optional): rows of the sub-problem (others zero);
// colscale (optional): column scales.  Out: U f64 [nS][KD] (zero rows where rowsel == 0), S f64 [n_cols][KD].
__global__ void __launch_bounds__(256) k_build_basis(const float *__restrict__ tiles, int tile_stride, int nS, int ldA, int n_cols,
                                                     const unsigned char *__restrict__ rowsel, const double *__restrict__ colscale,
                                                     double *__restrict__ Ub, double *__restrict__ Sb, int KD,
                                                     float *__restrict__ Sf = nullptr, double *__restrict__ kap = nullptr,
                                                     double *__restrict__ kap0 = nullptr, double *__restrict__ Rg = nullptr, int dir0 = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // deflated columns [nS][ldA]: in LDS when they fit next to Q (protocols of up to ~128 volumes), else in a global scratch block of
    // this workgroup (one-off per dictionary upload: the L2 serves it)
    double *R = Rg != nullptr ? Rg + (size_t)blockIdx.x * nS * ldA : reinterpret_cast<double *>(smem_b);
    double *Q = Rg != nullptr ? reinterpret_cast<double *>(smem_b) : R + (size_t)nS * ldA;                          // [KD][nS]
    double *red = Q + (size_t)KD * nS;                         // [256]
    const int dir = (int)blockIdx.x + dir0;
    __shared__ int s_j;
    __shared__ double s_n, s_coef[kSeedKD];
    const int tid = threadIdx.x, nt = blockDim.x;
    const float *g = tiles + (size_t)dir * tile_stride;
    for (int e = tid; e < nS * ldA; e += nt) {
        const int i = e / ldA, j = e - i * ldA;
        double v = (j < n_cols && (rowsel == nullptr || rowsel[i])) ? (double)g[e] : 0.0;
        if (colscale != nullptr && j < n_cols) v *= colscale[j];
        R[e] = v;
    }
    __syncthreads();
    for (int d = 0; d < KD; d++) {
        double nr = -1.0;
        if (tid < n_cols) { nr = 0.0; for (int i = 0; i < nS; i++) { const double v = R[i * ldA + tid]; nr += v * v; } }
        red[tid] = nr;
        __syncthreads();
        if (tid == 0) {
            int bj = 0; double bn = red[0];
            for (int j = 1; j < n_cols && j < nt; j++) if (red[j] > bn) { bn = red[j]; bj = j; }
            s_j = bj; s_n = bn;
        }
        __syncthreads();
        const double nn = s_n;
        const bool live = nn > 1e-280;
        for (int i = tid; i < nS; i += nt) Q[d * nS + i] = live ? R[i * ldA + s_j] / sqrt(nn) : 0.0;
        __syncthreads();
        // re-orthogonalise against the previous directions, twice, and renormalise
        for (int pass = 0; pass < 2 && live; pass++) {
            if (tid < d) { double c = 0.0; for (int i = 0; i < nS; i++) c += Q[tid * nS + i] * Q[d * nS + i]; s_coef[tid] = c; }
            __syncthreads();
            for (int i = tid; i < nS; i += nt) { double v = Q[d * nS + i]; for (int e = 0; e < d; e++) v -= s_coef[e] * Q[e * nS + i]; Q[d * nS + i] = v; }
            __syncthreads();
            if (tid == 0) { double c = 0.0; for (int i = 0; i < nS; i++) c += Q[d * nS + i] * Q[d * nS + i]; s_n = c; }
            __syncthreads();
            const double n2 = s_n;
            for (int i = tid; i < nS; i += nt) Q[d * nS + i] = n2 > 0.0 ? Q[d * nS + i] / sqrt(n2) : 0.0;
            __syncthreads();
        }
        if (tid < n_cols) {
            double c = 0.0;
            for (int i = 0; i < nS; i++) c += Q[d * nS + i] * R[i * ldA + tid];
            for (int i = 0; i < nS; i++) R[i * ldA + tid] -= c * Q[d * nS + i];
        }
        __syncthreads();
    }
    // what the basis leaves out: the deflated columns ARE e_j = (I - U U') a_j.  kappa bounds, per unit of ||r||, how far the
    // compressed dual value s_j'(U'r) (evaluated from the float32 copy Sf of S) can be from a_j'r: max ||e_j|| plus the
    // rounding of the float32 table and products, 2e-6 max ||a_j||  (NNSolver::certify_seed screens with it)
    if (kap != nullptr) {
        double en = 0.0, an = 0.0;
        if (tid < n_cols) {
            for (int i = 0; i < nS; i++) {
                const double v = R[i * ldA + tid]; en += v * v;
                double o = (rowsel == nullptr || rowsel[i]) ? (double)g[i * ldA + tid] : 0.0;
                if (colscale != nullptr) o *= colscale[tid];
                an += o * o;
            }
        }
        red[tid] = en;
        __syncthreads();
        if (tid == 0) { double m = 0.0; for (int j = 0; j < n_cols && j < nt; j++) m = red[j] > m ? red[j] : m; s_n = m; }
        __syncthreads();
        const double emax = sqrt(s_n);
        __syncthreads();
        red[tid] = an;
        __syncthreads();
        if (tid == 0) { double m = 0.0; for (int j = 0; j < n_cols && j < nt; j++) m = red[j] > m ? red[j] : m; kap[dir] = emax + 2e-6 * sqrt(m); if (kap0) kap0[dir] = emax; }
        __syncthreads();
    }
    double *U = Ub + (size_t)dir * nS * KD;
    for (int e = tid; e < nS * KD; e += nt) { const int i = e / KD, d = e - i * KD; U[e] = Q[d * nS + i]; }
    // S = U'A from the ORIGINAL (scaled, row-selected) columns
    double *S = Sb + (size_t)dir * n_cols * KD;
    for (int e = tid; e < n_cols * KD; e += nt) {
        const int j = e / KD, d = e - j * KD;
        double c = 0.0;
        for (int i = 0; i < nS; i++) {
            double v = (rowsel == nullptr || rowsel[i]) ? (double)g[i * ldA + j] : 0.0;
            c += Q[d * nS + i] * v;
        }
        if (colscale != nullptr) c *= colscale[j];
        S[e] = c;
        if (Sf != nullptr) Sf[(size_t)dir * KD * kScreenLd + d * kScreenLd + j] = (float)c;      // [d][atom], zero padded
    }
}

// ------------------------------------------------------------------ y~ = U'y (wavefront per voxel, lane = signal rows)
struct SeedArgs {
    const double *y;              // [n_vox][nS]
    const float *y32;             // float32 signals instead (or null)
    const int *perm;
    const Chunk *chunks;          // 256-voxel chunks of the solver kernels (projection)
    const int *n_chunks;
    const Chunk *schunks;         // larger chunks of the seed kernel
    const int *n_schunks;
    const double *Ub, *Sb;        // [ndirs][nS][KD], [ndirs][n_atoms][KD]
    double *ytil;                 // [n_vox][KD], bucket order (k_noddi_project), or
    unsigned long long *seeds;    // [n_vox], bucket order: up to 8 atom ids, one per byte, 0xff = empty; kNoSeed = none
    // stage 3: the candidate byte lists the LASSO certificates left in the seeds2 array (k_lasso_gcert) for the voxels they settled
    // (cdone[pos] == 1): up to 31 atom ids in ascending order, their number in the top byte of word 3; null = none
    const unsigned long long *cand8; const unsigned char *cdone;
    const unsigned long long *supp;   // stage 3: [n_vox][4] stage-2 support bit set (voxel order), null for stage 1
    int nS, n_atoms, iso_atom, dot_atom;
    int *gcount;                  // [max_schunks + 1]: voxels of each chunk handed out so far (zeroed before the launch); last: helpers
    int n_gcount;
    int *stats;                   // optional counters (AMX_STATS): [0] trips, [1] lane-trips in use, [2] voxels, [3] no-seed voxels
    int trip_cap;                 // a voxel still on its way after this many trips is given up (no seed: left-over list)
    int fork_skip;                // stage 3: 1 = the voxels the LASSO certificates did not settle (cdone != 1) are finished elsewhere (kSeedForked)
};

template <int NR>
__global__ void __launch_bounds__(1024) k_noddi_project(const SeedArgs a)
{
    constexpr int KD = kSeedKD;
    const int cid = xcd_chunk((int)blockIdx.x, *a.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.chunks[cid];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int nS = a.nS;
    // this lane's rows of U stay in registers for the whole chunk
    double ur[NR][KD];
    const double *U = a.Ub + (size_t)ck.dir * nS * KD;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int i = lane + kWave * rr;
#pragma unroll
        for (int d = 0; d < KD; d++) ur[rr][d] = (i < nS) ? U[i * KD + d] : 0.0;
    }
    for (int k = wave; k < ck.count; k += nw) {
        const int pos = ck.start + k;
        const size_t yo = (size_t)a.perm[pos] * nS;
        double yr[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) { const int i = lane + kWave * rr; yr[rr] = (i < nS) ? (a.y32 ? (double)a.y32[yo + i] : a.y[yo + i]) : 0.0; }
        double out = 0.0;
#pragma unroll
        for (int b = 0; b < KD; b += 4) {
            double p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                p[u] = 0.0;
#pragma unroll
                for (int rr = 0; rr < NR; rr++) p[u] += ur[rr][b + u] * yr[rr];
            }
            wave_sum4(p, lane);
#pragma unroll
            for (int u = 0; u < 4; u++) out = (lane == b + u) ? p[u] : out;
        }
        if (lane < KD) a.ytil[(size_t)pos * KD + lane] = out;
    }
}

// ------------------------------------------------------------------ lane-per-voxel Lawson-Hanson in the compressed space
__device__ __forceinline__ double seed_div(double x, double d)      // x / d, d > 0 normal: reciprocal + two Newton steps
{
    double r = __builtin_amdgcn_rcp(d);
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    const double q = x * r;
    return q + r * (x - q * d);
}

template <int N> __device__ __forceinline__ constexpr int stri(int i, int j) { return i * (i + 1) / 2 + j; }


// Dual vector of the compressed problem for the 64 voxels of a wavefront on the fp64 matrix cores:
// W [atoms x voxels] = S' [atoms x KD] * R [KD x voxels], v_mfma_f64_16x16x4_f64 (exact fp64 products and sums).
//   * A operand (S'): prepared once per chunk in LDS in operand order, Aop[(mt * KS + ks) * 64 + lane] = S[atom 16 mt +
//     (lane & 15)][4 ks + (lane >> 4)] (zero beyond n_atoms);
//   * B operand (R): every lane publishes the residual of ITS voxel in the wavefront's LDS block Rb [64][4 KS + 1] and
//     reads back the element the operand layout asks of it -- r[4 ks + (lane >> 4)] of voxel 16 nt + (lane & 15);
//   * D: register r of lane l holds the dual value of atom 16 mt + 4 r + (l >> 4) for voxel 16 nt + (l & 15).  The arg-max
//     rides in the value itself: the low 8 mantissa bits are replaced by the candidate's code (mt * 4 + r, later the row
//     l >> 4), so one v_max_f64 per value keeps both; the four rows that share a voxel are combined by the two row
//     swaps, and the lane that owns voxel v finds its result in row v >> 4.  (2^-44 relative: far below what the choice
//     of the entering atom, or the 1e-10 stopping test, can see.)
__device__ __forceinline__ double seed_max(double a, double b)     // plain v_max_f64 (no canonicalisation of the operands)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ double seed_min(double a, double b)
{
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

typedef double seed_v4d __attribute__((ext_vector_type(4)));
template <int KS, int MT, bool PIPE = true>
__device__ __forceinline__ void seed_scan_mfma(const double *Aop, double *Rb, const double (&r)[4 * KS], int lane, double &best, int &bj)
{
    constexpr int KDP = 4 * KS + 1;
    const int q = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int d = 0; d < 4 * KS; d++) Rb[lane * KDP + d] = r[d];
    double b[4][KS];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[nt][ks] = Rb[(16 * nt + c16) * KDP + 4 * ks + q];
    }
    const double ninf = -__builtin_huge_val();
    double bv[4] = {ninf, ninf, ninf, ninf};
    // software pipeline: the 12 products of tile mt + 1 are issued BEFORE the tag / max work on tile mt's results, so the matrix
    // pipe runs while the vector pipe reduces (one wavefront per SIMD: nothing else would overlap the two)
    auto products = [&](int mt, seed_v4d (&acc)[4]) {
        double av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) av[ks] = Aop[(mt * KS + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[nt] = (seed_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], b[nt][ks], acc[nt], 0, 0, 0);
        }
    };
    seed_v4d cur[4], nxt[PIPE ? 4 : 1];
    if (PIPE) products(0, cur);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        if (PIPE) { if (mt + 1 < MT) products(mt + 1, reinterpret_cast<seed_v4d (&)[4]>(nxt)); }
        else products(mt, cur);                       // (two wavefronts per SIMD: the other wavefront fills the matrix pipe's shadow)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const double v = cur[nt][rr];
                const unsigned lo = ((unsigned)__double2loint(v) & 0xffffff00u) | (unsigned)(mt * 4 + rr);
                bv[nt] = seed_max(bv[nt], __hiloint2double(__double2hiint(v), (int)lo));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PIPE) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) cur[nt] = nxt[PIPE ? nt : 0];
        }
    }
    double mine = ninf;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const double t = __hiloint2double(__double2hiint(bv[nt]), (int)((unsigned)__double2loint(bv[nt]) | (unsigned)(q << 6)));
        const double m = rows_allmax(t);
        mine = (q == nt) ? m : mine;
    }
    const unsigned code = (unsigned)__double2loint(mine) & 0xffu;
    best = mine;
    bj = 16 * (int)((code >> 2) & 15u) + 4 * (int)(code & 3u) + (int)(code >> 6);
}

// Per-lane state of one voxel: the passive atoms idx[], their coefficients x[], c = S_P' y~, and ONE packed triangle T that
// holds the Cholesky factor of H_PP = S_P' S_P (slots >= np: zero rows, zero reciprocal pivots, so every loop runs over
// all MS slots without predicates).  Removing a slot turns T back into H_PP in place (L L'), deletes the row and column
// in place and factors again in place: no second triangle is ever live.
template <int MS>
struct SeedLane {
    static constexpr int NT = MS * (MS + 1) / 2;
    double T[NT], dinv[MS], c[MS], x[MS];
    int idx[MS], np;

    __device__ __forceinline__ void clear()
    {
        np = 0;
#pragma unroll
        for (int s = 0; s < MS; s++) { idx[s] = 0; x[s] = 0.0; c[s] = 0.0; dinv[s] = 0.0; }
#pragma unroll
        for (int e = 0; e < NT; e++) T[e] = 0.0;
    }
    // z = (L L')^-1 c
    __device__ __forceinline__ void solve(double (&z)[MS]) const
    {
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double f = c[j];
#pragma unroll
            for (int m = 0; m < j; m++) f -= T[stri<MS>(j, m)] * z[m];
            z[j] = f * dinv[j];
        }
#pragma unroll
        for (int j = MS - 1; j >= 0; j--) {
            double f = z[j];
#pragma unroll
            for (int m = j + 1; m < MS; m++) f -= T[stri<MS>(m, j)] * z[m];
            z[j] = f * dinv[j];
        }
    }
    // z = (L L')^-1 b
    __device__ __forceinline__ void solve_rhs(const double (&b)[MS], double (&z)[MS]) const
    {
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double f = b[j];
#pragma unroll
            for (int m = 0; m < j; m++) f -= T[stri<MS>(j, m)] * z[m];
            z[j] = f * dinv[j];
        }
#pragma unroll
        for (int j = MS - 1; j >= 0; j--) {
            double f = z[j];
#pragma unroll
            for (int m = j + 1; m < MS; m++) f -= T[stri<MS>(m, j)] * z[m];
            z[j] = f * dinv[j];
        }
    }
    // in-place Cholesky of the matrix held in T (slots >= np are zero); false: a pivot of a live slot is not positive
    __device__ __forceinline__ bool factor()
    {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double dj = T[stri<MS>(j, j)];
            const double hjj = dj;
#pragma unroll
            for (int m = 0; m < j; m++) dj -= T[stri<MS>(j, m)] * T[stri<MS>(j, m)];
            const bool good = dj > 1e-15 * hjj;
            ok = ok && (j >= np || good);
            dinv[j] = (j < np && good) ? inv_sqrt(dj) : 0.0;
            T[stri<MS>(j, j)] = dj * dinv[j];
#pragma unroll
            for (int i = j + 1; i < MS; i++) {
                double v = T[stri<MS>(i, j)];
#pragma unroll
                for (int m = 0; m < j; m++) v -= T[stri<MS>(i, m)] * T[stri<MS>(j, m)];
                T[stri<MS>(i, j)] = v * dinv[j];
            }
        }
        return ok;
    }
    // slot k leaves (per-lane k < np; k = MS: nothing happens, bit for bit -- the call is branch-free for the whole wavefront,
    // a divergent branch around it made the compiler keep a second copy of the triangle).  Rows below k lose their
    // entry of column k -- the vector l -- and move up one slot, columns right of k move left; the trailing block then takes
    // the rank-one update L22 L22' + l l' (the textbook Cholesky update: one rotation per column, identity where l is
    // zero, i.e. left of k and beyond the passive set).  Everything happens in place in the one triangle.
    __device__ __forceinline__ void remove(int k)
    {
        double l[MS];
#pragma unroll
        for (int i = 0; i < MS - 1; i++) {                   // l[i] = old T[i + 1][k] for i >= k, else 0
            double v = 0.0;
#pragma unroll
            for (int kk = 0; kk <= i; kk++) v = (k == kk) ? T[stri<MS>(i + 1, kk)] : v;
            l[i] = v;
        }
        l[MS - 1] = 0.0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MS - 1; i++) {                   // ascending: every entry is read before it is overwritten
#pragma unroll
            for (int j = 0; j <= i; j++) {
                const double v00 = T[stri<MS>(i, j)], v10 = T[stri<MS>(i + 1, j)], v11 = T[stri<MS>(i + 1, j + 1)];
                T[stri<MS>(i, j)] = (i < k) ? v00 : ((j < k) ? v10 : v11);
            }
            __builtin_amdgcn_sched_barrier(0);               // row by row: a reordered schedule needs a second copy of the triangle
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MS; m++) T[stri<MS>(MS - 1, m)] = (k < MS) ? 0.0 : T[stri<MS>(MS - 1, m)];
#pragma unroll
        for (int s = 0; s < MS - 1; s++) {
            const bool mv = s >= k;
            idx[s] = mv ? idx[s + 1] : idx[s]; x[s] = mv ? x[s + 1] : x[s]; c[s] = mv ? c[s + 1] : c[s]; dinv[s] = mv ? dinv[s + 1] : dinv[s];
        }
        if (k < MS) { idx[MS - 1] = 0; x[MS - 1] = 0.0; c[MS - 1] = 0.0; dinv[MS - 1] = 0.0; np -= 1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < MS - 1; j++) {
            const double a = T[stri<MS>(j, j)], b = l[j];
            const bool rot = b != 0.0;
            const double n2 = a * a + b * b;
            const double ri = rot ? inv_sqrt(n2) : dinv[j];
            const double ss = rot ? b * dinv[j] : 0.0;                   // sin / cos of the rotation
            const double cc = rot ? n2 * ri * dinv[j] : 1.0;             // 1 / cos
            const double ci = rot ? a * ri : 1.0;                        // cos
            T[stri<MS>(j, j)] = rot ? n2 * ri : a;
            dinv[j] = ri;
#pragma unroll
            for (int i = j + 1; i < MS - 1; i++) {
                const double t = (T[stri<MS>(i, j)] + ss * l[i]) * ci;
                l[i] = cc * l[i] - ss * t;
                T[stri<MS>(i, j)] = t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // step from x towards z: returns the slot that reaches zero first (-1: z is feasible and becomes x).  The smallest ratio
    // x_s / (x_s - z_s) over the infeasible slots is found by cross-multiplication (denominators are positive): one
    // division per call instead of one per slot
    __device__ __forceinline__ int step(const double (&z)[MS])
    {
        double nb = 0.0, db = 1.0;
        int kmin = -1;
#pragma unroll
        for (int s = 0; s < MS; s++) {
            const bool neg = s < np && !(z[s] > 0.0);
            const double den = x[s] - z[s];
            const double num = (den > 0.0) ? x[s] : 0.0, dd = (den > 0.0) ? den : 1.0;
            const bool better = neg && (kmin < 0 || num * db < nb * dd);
            nb = better ? num : nb; db = better ? dd : db; kmin = better ? s : kmin;
        }
        const double alpha = seed_div(nb, db);                 // (0 / 1 when nothing is infeasible: unused)
#pragma unroll
        for (int s = 0; s < MS; s++) x[s] = (kmin < 0) ? z[s] : ((s < np) ? x[s] + alpha * (z[s] - x[s]) : 0.0);
        return kmin;
    }
};

// How the lanes of the seed solvers get their voxels (round 4).  A chunk's voxels are handed out from a GLOBAL counter of the chunk in
// private blocks per wavefront (64 voxels, 16 near the chunk's end): inside a block a take is register arithmetic, and -- the point --
// several workgroups can work on ONE chunk.  One workgroup per orientation chunk fills the chip exactly once (~500 chunks on 512
// slots), so a launch used to last as long as its LARGEST chunk (1.15 x the mean at 1 M voxels, more on real data); now a workgroup
// whose chunk is exhausted joins the largest chunks that still have voxels (the chunk list is in order of decreasing size), staging
// that orientation's tables again.  Every decision of a lane still depends on its own voxel only: results are bit-identical.
struct SeedFeed {
    int lo, hi;                   // this wavefront's block [lo, hi) of the chunk's voxels (wave-uniform)
    bool more;                    // the chunk's counter may still have voxels beyond it
    int left;                     // voxels left in the chunk at the last fetch
    __device__ __forceinline__ void reset() { lo = 0; hi = 0; more = true; left = 1 << 30; }
    __device__ __forceinline__ bool pending() const { return more || lo < hi; }
    // the lanes of `needm` want a voxel: this lane's voxel number within the chunk, or -1 (served next trip, or none left)
    __device__ __forceinline__ int take(unsigned long long needm, int *gcount, int count, int lane)
    {
        int mine = -1;
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(needm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)needm, 0u));
        int served = 0;                         // lanes of needm (in rank order) that have their voxel
        const int want_n = __builtin_popcountll(needm);
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            if (lo == hi && more) {
                // blocks of 64 voxels; near the chunk's end smaller ones, so that the wavefronts (and workgroups) working on it end together
                const int want = left > 1024 ? 64 : (left > 256 ? 32 : 16);
                int old = 0;
                if (lane == 0) old = atomicAdd(gcount, want);
                old = __builtin_amdgcn_readfirstlane(old);
                lo = old < count ? old : count;
                hi = old + want < count ? old + want : count;
                left = count - hi;
                if (hi >= count) more = false;
            }
            const int have_n = hi - lo, todo = want_n - served;
            const int give = todo < have_n ? todo : have_n;
            if (((needm >> lane) & 1ull) && rank >= served && rank < served + give) mine = lo + (rank - served);
            lo += give; served += give;
            if (served >= want_n || !more) break;
        }
        return mine;
    }
};

// the chunk a workgroup works on next: its own first, then -- once that is exhausted -- the largest chunks that still have voxels
// (steal[0] counts the helpers; the p-th dispatched workgroup's chunk is the p-th largest: k_order_schunks)
__device__ __forceinline__ int seed_next_chunk(int round, int own, const Chunk *schunks, int n_schunks, const int *gcount, int *steal, int *lds_slot, int min_left)
{
    if (round == 0) return own;
#ifdef AMX_SEED_NO_STEAL
    return -1;
#endif
    if (threadIdx.x == 0) {
        int pick = -1;
        for (int tries = 0; tries < 12 && pick < 0; tries++) {
            const int h = atomicAdd(steal, 1);
            const int per = (n_schunks + 7) >> 3;
            if (h >= 8 * per) break;
            const int cand = xcd_chunk(h, n_schunks);
            if (cand < 0) continue;
            const int left = schunks[cand].count - __hip_atomic_load(&gcount[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left >= min_left) pick = cand;
        }
        *lds_slot = pick;
    }
    __syncthreads();
    const int pick = *lds_slot;
    __syncthreads();
    return pick;
}

// The kernels that walk a chunk block by block (the certificates: 64 voxels, the A'y GEMM: groups of 16) share their chunks the
// same way: a wavefront draws blocks from the chunk's global counter -- the ticket of the NEXT block is requested when the current
// one is handed out, so the atomic's latency hides behind a block's work -- and a workgroup whose chunk is exhausted joins the
// largest open ones.
// (chunks come largest first, k_order_schunks: when even the first is too small to share, nobody needs help -- small calls skip
//  the search.  Not in seed_next_chunk itself: k_nnls_seed<3, 6> ran 0.81 -> 1.15 ms per 1 M voxels with this test compiled in.)
__device__ __forceinline__ int block_next_chunk(int round, int own, const Chunk *schunks, int n_schunks, const int *gcount, int *steal, int *lds_slot, int min_left)
{
    if (round > 0 && schunks[xcd_chunk(0, n_schunks)].count < min_left) return -1;
    return seed_next_chunk(round, own, schunks, n_schunks, gcount, steal, lds_slot, min_left);
}

template <int UNIT>
struct BlockFeed {
    int raw;                      // lane 0: voxel offset of this wavefront's next block (the atomic's return value, read when needed)
    __device__ __forceinline__ void start(int *gcount, int lane) { raw = 0; if (lane == 0) raw = atomicAdd(gcount, UNIT); }
    __device__ __forceinline__ int next(int *gcount, int count, int lane)        // block index within the chunk, or -1
    {
        const int cur = __builtin_amdgcn_readfirstlane(raw);
        if (cur >= count) return -1;
        raw = 0;
        if (lane == 0) raw = atomicAdd(gcount, UNIT);
        return cur / UNIT;
    }
};

// STAGE 1: all atoms are candidates; STAGE 3: the atoms of the stage-2 support plus iso (dot)
// (MS = 8: the compiler needs ~540 registers for this kernel; at two wavefronts per SIMD it spilled 290 of them and the kernel moved
//  27 GB + 9 GB of scratch per 1 M voxels (rocprofv3 FETCH_SIZE / WRITE_SIZE) -- at one wavefront per SIMD, with the accumulation
//  registers as overflow, none: 5.1 -> 3.4 ms)
#ifndef AMX_SEED1_OCC
#define AMX_SEED1_OCC 1
#endif
// (stage 3: PINNED at two wavefronts per SIMD.  Left to itself -- bound 1 -- the compiler landed on either side of 256 registers
//  with every unrelated edit of this file, and the kernel ran 0.81 or 1.16 ms per 1 M voxels accordingly.  Without the next-voxel
//  prefetch: 204 VGPRs, 0.79 ms; with it 256 + 4 spilled, 0.82 ms.)
#ifndef AMX_SEED3_OCC
#define AMX_SEED3_OCC 2
#endif
#ifndef AMX_SEED3_PREF
#define AMX_SEED3_PREF 0
#endif
// OCC2 (stage 1, calls of >= ~0.5 M voxels): two wavefronts per SIMD -- no next voxel reserved in registers, no software pipeline
// in the scan (244 VGPRs, no scratch): a trip is a third longer, but two wavefronts hide each other's dependent chains and the
// kernel is throughput bound at that size (1 M voxels: 2.80 -> 2.14 ms).  Small calls are bound by the longest single voxel's path:
// there the one-wavefront build with the shorter trip wins (50 000 voxels: 0.54 against 0.73 ms).
template <int STAGE, int MS, bool OCC2 = false>
__global__ void __launch_bounds__(256, (MS > 6 ? (OCC2 ? 2 : AMX_SEED1_OCC) : AMX_SEED3_OCC)) k_nnls_seed(const SeedArgs a)
{
    constexpr int KD = kSeedKD, LD = kSeedLd;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    double *Sl = reinterpret_cast<double *>(smem_s);             // [n_atoms][LD]
    const int n_atoms = a.n_atoms;
    unsigned *ticket = reinterpret_cast<unsigned *>(Sl + (size_t)n_atoms * LD);
    // STAGE 1: the compressed dictionary once more in MFMA operand order, and a block per wavefront for the residuals
    constexpr int KS = KD / 4, MT = 10, KDP = KD + 1;
    static_assert(KD % 4 == 0, "whole K-steps");
    double *Aop = reinterpret_cast<double *>(ticket + 4);
    double *Rb = Aop + (STAGE == 1 ? MT * KS * 64 : 0) + (threadIdx.x >> 6) * (64 * KDP);
    const int n_sch = *a.n_schunks;
    const int own = xcd_chunk((int)blockIdx.x, n_sch);
    if (own < 0) return;
    const int lane = threadIdx.x & 63;
#ifdef AMX_STATS
    int st_trips = 0, st_used = 0;
    long long ph[6] = {0, 0, 0, 0, 0, 0}, pt = (long long)__builtin_readcyclecounter();
#define SEED_PH(k) do { const long long t__ = (long long)__builtin_readcyclecounter(); ph[k] += t__ - pt; pt = t__; } while (0)
#else
#define SEED_PH(k) do { } while (0)
#endif
    // the workgroup's own chunk first, then the largest chunks that still have voxels (SeedFeed)
    for (int round = 0; round < 256; round++) {
    const int cid = seed_next_chunk(round, own, a.schunks, n_sch, a.gcount, a.gcount + a.n_gcount, reinterpret_cast<int *>(ticket), 32 * (int)(blockDim.x >> 6));
    if (cid < 0) break;
    const Chunk ck = a.schunks[cid];
    const double *__restrict__ Sg = a.Sb + (size_t)ck.dir * n_atoms * KD;
    stage_rows<KD, LD>(Sl, Sg, n_atoms, KD);
    if (STAGE == 1) {
        // The scan's operand holds the atoms NORMALISED (s_j / ||s_j||): the entering atom is then the one whose direction fits the
        // residual best, not the one with the largest dual value -- Lawson-Hanson may admit any atom with a positive dual value, and
        // this rule needs 10.4 instead of 11.0 trips per voxel (95th percentile 15 instead of 17; tools/lab/two_add_lab.py rule 4).
        // The sign of a dual value, and with it the stopping test, is unchanged; the append below works on the atoms themselves.
#ifndef AMX_SEED_SCAN_NORM
#define AMX_SEED_SCAN_NORM 1
#endif
        stage_operand<KS, MT, AMX_SEED_SCAN_NORM != 0, AMX_STAGE_UBN>(Aop, Sg, n_atoms, KD);
    }
    __syncthreads();
#ifndef AMX_SEED_ISO_FIRST
#define AMX_SEED_ISO_FIRST 3          // bit 0: stage 1 (two-wavefront build), bit 1: stage 3
#endif
    constexpr bool iso_first = (STAGE == 1 && OCC2 && (AMX_SEED_ISO_FIRST & 1)) || (STAGE == 3 && (AMX_SEED_ISO_FIRST & 2));
    const double tol = 1e-10, inf = __builtin_huge_val();
    const int trip_cap = a.trip_cap;

    bool active = false;
    int pos = 0, trips = 0, last_added = -1, ban0 = -1, ban1 = -1;
    SeedLane<MS> V;
    V.clear();
    unsigned long long allow[STAGE == 3 ? 4 : 1];
    unsigned long long cand[STAGE == 3 ? 4 : 1];      // stage 3: the voxel's admissible atoms as a byte list (<= 32), made once per voxel
    int ncand = 0;
    SeedFeed feed;
    feed.reset();
    // Stage 1 (one wavefront per SIMD, registers to spare): the voxel's y~ stays in registers, and a lane RESERVES its next voxel
    // while it works on the current one -- the 12 loads of the next y~ are in flight for a whole solve instead of being waited
    // for in every trip in which some lane of the wavefront refills (measured: 15 % of the kernel).
    // (stage 3 could do the same, AMX_SEED3_PREF=1, with the admissible-atom mask of the next voxel next to its y~: no gain at two
    //  wavefronts per SIMD, see AMX_SEED3_OCC)
    constexpr bool PREF = (MS > 6 && !OCC2 && AMX_SEED1_OCC == 1) || (STAGE == 3 && AMX_SEED3_PREF);
    double yv[PREF ? KD : 1], ynext[PREF ? KD : 1];
    unsigned long long nallow[(PREF && STAGE == 3) ? 4 : 1];
    int next_pos = -1;
    bool have_next = false;
    for (int guard = 0; guard < (1 << 20); ++guard) {
        // ------------------------------------------------------------ free lanes take the next voxels of the chunk
        if (PREF) {
            if (!active && have_next) {
                pos = next_pos; have_next = false;
                bool finite = true;
#pragma unroll
                for (int d = 0; d < KD; d++) { yv[PREF ? d : 0] = ynext[PREF ? d : 0]; finite = finite && (fabs(yv[PREF ? d : 0]) <= 1.79769313486231570e308); }
                trips = 0; last_added = -1; ban0 = -1; ban1 = -1;
                V.clear();
                if (STAGE == 3) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) allow[STAGE == 3 ? q4 : 0] = nallow[(PREF && STAGE == 3) ? q4 : 0];
                    ncand = -1;                                 // (byte list below)
                }
                if (finite) active = true;
                else a.seeds[pos] = kNoSeed;
            }
            const unsigned long long needm = __ballot(!have_next);
            if (needm != 0ull && feed.pending()) {
                const int k = feed.take(needm, a.gcount + cid, ck.count, lane);
                if (k >= 0) {
                    next_pos = ck.start + k; have_next = true;
                    const double *yp = a.ytil + (size_t)next_pos * KD;
#pragma unroll
                    for (int d = 0; d < KD; d++) ynext[PREF ? d : 0] = yp[d];
                    if (STAGE == 3) {
                        const int vox = a.perm[next_pos];
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) nallow[(PREF && STAGE == 3) ? q4 : 0] = a.supp[(size_t)vox * 4 + q4];
                        nallow[(PREF && STAGE == 3) ? (a.iso_atom >> 6) : 0] |= 1ull << (a.iso_atom & 63);
                        if (a.dot_atom >= 0) nallow[(PREF && STAGE == 3) ? (a.dot_atom >> 6) : 0] |= 1ull << (a.dot_atom & 63);
                    }
                }
            }
            if (__ballot(active) == 0ull) {
                if (!feed.pending() && __ballot(have_next) == 0ull) break;
                continue;
            }
        } else {
        const unsigned long long freem = __ballot(!active);
        if (freem != 0ull && feed.pending()) {
            const int k = feed.take(freem, a.gcount + cid, ck.count, lane);
            if (k >= 0) {
                pos = ck.start + k;
                // (no look at the voxel's y~ here: `finite = finite && |y~_d| <= max` over a pointer compiles to KD loads that each
                //  wait for the one before -- a chain of 12 memory round trips in nearly every trip of the wavefront, the "taking
                //  voxels" fifth of this kernel.  A non-finite y~ needs no test of its own: NaN dual values never beat -inf (v_max_f64
                //  drops them, an index in the mantissa of an infinity is a NaN as well), so the voxel is done in its first trip or at
                //  the trip cap at the latest, and the certificate refuses it on ||y||^2 whatever seed it got: the wavefront-per-voxel
                //  kernel writes its NaN maps)
                bool finite = true;                             // (stage 3, forked fit: false for the voxels finished on the side stream)
                trips = 0; last_added = -1; ban0 = -1; ban1 = -1;
                V.clear();
                ncand = -1;                                     // (byte list below)
                if (STAGE == 3) {
                    // the candidate list as the LASSO certificate left it (a settled voxel: 99.4 % of them), else from the support bits
                    bool ready = false;
                    if (a.cand8 != nullptr) {
                        const unsigned long long *cl = a.cand8 + (size_t)pos * 4;
                        const unsigned long long c0 = cl[0], c1 = cl[1], c2 = cl[2], c3 = cl[3];
                        const int flag = (int)a.cdone[pos];
                        ready = flag == 1;
                        if (ready) { cand[0] = c0; cand[STAGE == 3 ? 1 : 0] = c1; cand[STAGE == 3 ? 2 : 0] = c2; cand[STAGE == 3 ? 3 : 0] = c3 & 0x00ffffffffffffffull; ncand = (int)(c3 >> 56); }
                    }
                    if (a.fork_skip && !ready) { finite = false; ncand = 0; }
                    else if (__ballot(!ready) != 0ull) {
                        if (!ready) {
                            const int vox = a.perm[pos];
#pragma unroll
                            for (int q = 0; q < 4; q++) allow[q] = a.supp[(size_t)vox * 4 + q];
                            allow[a.iso_atom >> 6] |= 1ull << (a.iso_atom & 63);
                            if (a.dot_atom >= 0) allow[a.dot_atom >> 6] |= 1ull << (a.dot_atom & 63);
                        }
                    }
                }
                if (finite) active = true;
                else a.seeds[pos] = (STAGE == 3 && a.fork_skip) ? kSeedForked : kNoSeed;
            }
        }
        if (__ballot(active) == 0ull) {
            if (!feed.pending()) break;
            continue;
        }
        }
        if (STAGE == 3) {
            // new voxels: set bits of the candidate mask -> byte list, once per voxel instead of once per trip.  Some lane is fresh in
            // nearly every trip, so the WHOLE wavefront walks this loop every trip: the bytes go to the lane's 36-byte row of an LDS
            // block (one ds_write_b8 per atom; an odd number of words per row: no bank conflicts) instead of being shifted into four
            // 64-bit registers under selects (~60 instructions per atom -- the conversion was a quarter of this kernel)
            if (__ballot(ncand < 0) != 0ull) {
                const bool fresh = ncand < 0;
                unsigned char *row = reinterpret_cast<unsigned char *>(ticket + 4) + (size_t)threadIdx.x * kSeed3ListRow;
                if (fresh) {
#pragma unroll
                    for (int w8 = 0; w8 < 8; w8++) reinterpret_cast<unsigned *>(row)[w8] = 0u;
                }
                int nc = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned long long rem = fresh ? allow[q] : 0ull;
                    for (int it = 0; it < 64; it++) {
                        if (__ballot(rem != 0ull) == 0ull) break;
                        if (rem != 0ull) {
                            const int j = q * 64 + __builtin_ctzll(rem);
                            rem &= rem - 1ull;
                            if (nc < 32) row[nc] = (unsigned char)j;
                            nc++;
                        }
                    }
                }
                if (fresh) {
#pragma unroll
                    for (int w4 = 0; w4 < 4; w4++) {
                        const unsigned lo = reinterpret_cast<const unsigned *>(row)[2 * w4], hi = reinterpret_cast<const unsigned *>(row)[2 * w4 + 1];
                        cand[STAGE == 3 ? w4 : 0] = ((unsigned long long)hi << 32) | (unsigned long long)lo;
                    }
                    ncand = nc;
                    if (nc > 32 && active) { a.seeds[pos] = kNoSeed; active = false; }      // (never seen: LASSO supports end at 23 atoms)
                }
            }
        }
#ifdef AMX_STATS
        st_trips++; st_used += __builtin_popcountll(__ballot(active));
#endif
        trips++;
        SEED_PH(0);
        // ------------------------------------------------------------ least squares on the passive set; step back and drop
        // one atom if it is infeasible, then solve again at once (a second removal waits for the next trip)
        bool scan = active;
        {
            double z[MS];
            V.solve(z);
            int kmin = V.step(z);
            if (active && kmin < 0) { ban0 = -1; ban1 = -1; }     // the last addition stood: forget the refused candidates
            if (__ballot(active && kmin >= 0) != 0ull) {
                bool ok = true;
                if (active && kmin >= 0) {
                    int gone = 0;
#pragma unroll
                    for (int s = 0; s < MS; s++) gone = (s == kmin) ? V.idx[s] : gone;
                    if (gone == last_added) { ban1 = ban0; ban0 = gone; }
                }
#ifndef SEED_NO_REMOVE
                __builtin_amdgcn_sched_barrier(0);
                V.remove((active && kmin >= 0) ? kmin : MS);
                __builtin_amdgcn_sched_barrier(0);
#endif
                V.solve(z);
                if (active && kmin >= 0) {
                    if (!ok) scan = false;                     // (numerically dependent set: the next trip's step sorts it out)
                    else if (V.step(z) >= 0) scan = false;     // still infeasible: the drop happens next trip (x already stepped)
                }
            }
        }
        SEED_PH(1);
        // ------------------------------------------------------------ dual vector in the compressed space, entering atom
        bool done = false;
        bool noseed = false;
        if (__ballot(scan) != 0ull) {
            // (the voxel's projected signal is read again in every trip -- 96 contiguous bytes, cache resident -- instead of
            //  occupying 24 registers for the whole solve)
            const double *yp = a.ytil + (size_t)pos * KD;
            double r[KD];
#pragma unroll
            for (int d = 0; d < KD; d++) r[d] = PREF ? yv[PREF ? d : 0] : yp[d];
#pragma unroll
            for (int s = 0; s < MS; s++) {         // (slots >= np: x = 0, idx = 0 -- no predicate needed)
                const double *col = Sl + V.idx[s] * LD;
                double cv[KD];
                seed_col<KD>(col, cv);
#pragma unroll
                for (int d = 0; d < KD; d++) r[d] -= V.x[s] * cv[d];
            }
            SEED_PH(2);
            // the explicit guard for non-finite signals (round 6): r = y~ - S x with finite x, so r[0] is finite iff y~[0] is, and
            // y~[0] = u_0'y sums EVERY sample of the voxel (NaN x 0 and Inf x 0 are NaN): one compare on a value that is in a register anyway
            // -- the lane mask lives in scalar registers -- instead of the chain of 12 loads that the take used to wait for
            const bool rfin = fabs(r[0]) <= 1.79769313486231570e308;
            double best = -inf;
            int bj = -1;
            if (STAGE == 1) {
                if (n_atoms <= 16 * MT) {
#ifndef SEED_NO_MFMA
                    seed_scan_mfma<KS, MT, (!OCC2 && AMX_SEED1_OCC == 1)>(Aop, Rb, r, lane, best, bj);
#else
                    best = r[0] + r[5]; bj = (int)(r[1] * 100.0) & 127;
#endif
                } else {
                    for (int j = 0; j < n_atoms; j++) {
                        const double *sp = Sg + (size_t)j * KD;              // wave-uniform address: scalar loads
                        double w = 0.0;
#pragma unroll
                        for (int d = 0; d < KD; d++) w += sp[d] * r[d];
                        const bool ok = w > best;
                        best = ok ? w : best; bj = ok ? j : bj;
                    }
                }
                // a refused candidate may not come back before another atom has entered: look again without it (rare)
                if (__ballot(scan && (bj == ban0 || bj == ban1)) != 0ull) {
#ifdef AMX_STATS
                    ph[5] += 1024;                  // (diagnosis: the "store" slot counts the trips that take this path)
#endif
                    best = -inf; bj = -1;
                    for (int j = 0; j < n_atoms; j++) {
                        const double *sp = Sg + (size_t)j * KD;
                        double w = 0.0;
#pragma unroll
                        for (int d = 0; d < KD; d++) w += sp[d] * r[d];
                        const bool ok = (j != ban0) && (j != ban1) && (w > best);
                        best = ok ? w : best; bj = ok ? j : bj;
                    }
                }
            } else {
                // candidates = the lane's byte list (<= 32 atoms): per-lane gathers of their columns, two per step (two
                // independent chains: the loop is a chain of LDS latencies and dependent FMAs otherwise)
#pragma unroll
                for (int it = 0; it < SEED3_SCAN_MAX; it += 2) {
                    if (__ballot(scan && it < ncand) == 0ull) break;
                    const int j0 = (int)((cand[STAGE == 3 ? (it >> 3) : 0] >> (8 * (it & 7))) & 0xffull);
                    const int j1 = (int)((cand[STAGE == 3 ? ((it + 1) >> 3) : 0] >> (8 * ((it + 1) & 7))) & 0xffull);
                    const double2 *c0 = reinterpret_cast<const double2 *>(Sl + j0 * LD), *c1 = reinterpret_cast<const double2 *>(Sl + j1 * LD);
                    double u0[KD], u1[KD];
#pragma unroll
                    for (int d = 0; d < KD; d += 2) {           // 16-byte LDS reads (row stride 14 doubles: 16-byte aligned rows)
                        const double2 p0 = c0[d >> 1], p1 = c1[d >> 1];
                        u0[d] = p0.x; u0[d + 1] = p0.y; u1[d] = p1.x; u1[d + 1] = p1.y;
                    }
                    double w0a = 0.0, w0b = 0.0, w1a = 0.0, w1b = 0.0;
#pragma unroll
                    for (int d = 0; d < KD; d += 2) { w0a += u0[d] * r[d]; w0b += u0[d + 1] * r[d + 1]; w1a += u1[d] * r[d]; w1b += u1[d + 1] * r[d + 1]; }
                    const double w0 = w0a + w0b, w1 = w1a + w1b;
                    // (a voxel's first atom is chosen among the others: the isotropic atom enters with it, see the append below)
                    const bool ok0 = scan && (it < ncand) && (j0 != ban0) && (j0 != ban1) && (w0 > best) && !(iso_first && V.np == 0 && j0 == a.iso_atom);
                    best = ok0 ? w0 : best; bj = ok0 ? j0 : bj;
                    const bool ok1 = scan && (it + 1 < ncand) && (j1 != ban0) && (j1 != ban1) && (w1 > best) && !(iso_first && V.np == 0 && j1 == a.iso_atom);
                    best = ok1 ? w1 : best; bj = ok1 ? j1 : bj;
                }
                if (iso_first && __ballot(scan && V.np == 0 && !(best > tol)) != 0ull) {
                    // nothing but the isotropic atom may want in: it is a candidate after all
                    const double *ci = Sl + a.iso_atom * LD;
                    double si[KD], wi = 0.0;
                    seed_col<KD>(ci, si);
#pragma unroll
                    for (int d = 0; d < KD; d++) wi += si[d] * r[d];
                    const bool oki = scan && V.np == 0 && !(best > tol) && wi > best;
                    best = oki ? wi : best; bj = oki ? a.iso_atom : bj;
                }
            }
            SEED_PH(3);
            if (scan) {
                bool inp = false;
#pragma unroll
                for (int s = 0; s < MS; s++) inp = inp || (s < V.np && V.idx[s] == bj);
                if (!rfin) {
                    done = true; noseed = true; V.np = 0;                // non-finite signal: no seed, the wavefront-per-voxel kernel writes the NaN maps
                } else if (!(best > tol) || bj < 0 || inp) {
                    done = true;                                         // KKT point of the compressed problem
                } else if (V.np >= MS || trips > trip_cap) {
                    done = true; noseed = true;                          // no usable seed
                } else {
                    if (iso_first) {      // (stage 1: not in the one-wavefront build of small calls -- 0.279 -> 0.297 ms at 50 000 voxels with it)
                        // The first atom of a voxel does not enter alone: the isotropic atom, which nearly every optimum holds, takes
                        // slot 0 in the same trip (unsolved, x = 0 -- if its dual value is positive, as Lawson-Hanson asks of any entering
                        // atom; should its coefficient come out negative, the next trip's step drops it again).  Two trips fewer per voxel
                        // (10.4 -> 8.5 on the bench mix, 6.0 -> 4.5 on the hard mix: tools/lab/two_add_lab.py rule 8).
                        const double *ci = Sl + a.iso_atom * LD;
                        double si[KD], ciso = 0.0, hii = 0.0;
                        seed_col<KD>(ci, si);
#pragma unroll
                        for (int d = 0; d < KD; d++) { ciso += si[d] * (PREF ? yv[PREF ? d : 0] : yp[d]); hii += si[d] * si[d]; }
                        const bool pre = V.np == 0 && bj != a.iso_atom && ciso > tol && hii > 0.0;
                        const double di0 = pre ? inv_sqrt(hii) : 0.0;
                        V.T[stri<MS>(0, 0)] = pre ? hii * di0 : V.T[stri<MS>(0, 0)];
                        V.dinv[0] = pre ? di0 : V.dinv[0];
                        V.idx[0] = pre ? a.iso_atom : V.idx[0];
                        V.c[0] = pre ? ciso : V.c[0];
                        V.x[0] = pre ? 0.0 : V.x[0];
                        V.np += pre ? 1 : 0;
                    }
                    // append atom bj as slot np: new row of the factor by one forward substitution
                    const double *ct = Sl + bj * LD;
                    double st[KD], cn = 0.0, htt = 0.0;
                    seed_col<KD>(ct, st);
#pragma unroll
                    for (int d = 0; d < KD; d++) { cn += st[d] * (PREF ? yv[PREF ? d : 0] : yp[d]); htt += st[d] * st[d]; }
                    double h[MS];
#pragma unroll
                    for (int s = 0; s < MS; s++) {
                        const double *col = Sl + V.idx[s] * LD;
                        double cv[KD];
                        seed_col<KD>(col, cv);
                        double v = 0.0;
#pragma unroll
                        for (int d = 0; d < KD; d++) v += cv[d] * st[d];
                        h[s] = (s < V.np) ? v : 0.0;
                    }
                    double dd = htt;
#pragma unroll
                    for (int j = 0; j < MS; j++) {
                        double f = h[j];
#pragma unroll
                        for (int m = 0; m < j; m++) f -= V.T[stri<MS>(j, m)] * h[m];
                        h[j] = f * V.dinv[j];                                     // (slots >= np: dinv = 0)
                        dd -= h[j] * h[j];
                    }
                    const bool good = dd > 1e-15 * htt;
                    const double di = good ? inv_sqrt(dd) : 0.0;
                    if (!good) {
                        ban1 = ban0; ban0 = bj;                          // numerically inside the span of the passive atoms
                    } else {
#pragma unroll
                        for (int i = 0; i < MS; i++) {
                            const bool here = (V.np == i);
#pragma unroll
                            for (int j = 0; j < i; j++) V.T[stri<MS>(i, j)] = here ? h[j] : V.T[stri<MS>(i, j)];
                            V.T[stri<MS>(i, i)] = here ? dd * di : V.T[stri<MS>(i, i)];
                            V.dinv[i] = here ? di : V.dinv[i];
                            V.idx[i] = here ? bj : V.idx[i]; V.c[i] = here ? cn : V.c[i]; V.x[i] = here ? 0.0 : V.x[i];
                        }
                        last_added = bj;
                        V.np += 1;
                    }
                }
            }
        }
        SEED_PH(4);
        if (active && !done && trips > 2 * trip_cap) { done = true; noseed = true; }
        if (done) {
            unsigned lo = 0xffffffffu, hi = 0xffffffffu;                 // one byte per slot, 0xff = empty
            // (a voxel given up at the trip cap or at MS atoms hands on the support it holds: not a solution of the compressed problem,
            //  so the certificate will most likely refuse it -- but the left-over kernel lets a refused seed's atoms enter first, and
            //  these are the voxels with the longest paths from the empty set)
            if (!noseed || (STAGE == 1 && V.np > 0)) {      // (stage 3: measured, no gain -- its left-over kernel went 0.249 -> 0.265 ms per 1 M voxels)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s < V.np) lo = (lo & ~(0xffu << (8 * s))) | ((unsigned)(V.idx[s] & 0xff) << (8 * s));
                    if (s + 4 < V.np && s + 4 < MS) hi = (hi & ~(0xffu << (8 * s))) | ((unsigned)(V.idx[s + 4 < MS ? s + 4 : 0] & 0xff) << (8 * s));
                }
                if (V.np == 0) lo = 0xfffffffeu;                          // certified-empty support: not the "no seed" pattern
            }
            a.seeds[pos] = ((unsigned long long)hi << 32) | (unsigned long long)lo;
            active = false;
        }
#ifdef AMX_STATS
        pt = (long long)__builtin_readcyclecounter();          // (slot 5 counts the trips that repeat the scan, see above)
#endif
    }
    __syncthreads();                          // every wavefront is through with this chunk's tables
    }
#ifdef AMX_STATS
    if (a.stats && lane == 0) {
        atomicAdd(&a.stats[0], st_trips); atomicAdd(&a.stats[1], st_used);
        if (STAGE == 1) for (int k = 0; k < 6; k++) atomicAdd(&a.stats[8 + k], (int)(ph[k] >> 10));
        if (STAGE == 3) for (int k = 0; k < 6; k++) atomicAdd(&a.stats[50 + k], (int)(ph[k] >> 10));
    }
#endif
}

// ================================================================== C = [A | U | U2 | 1_b0]' Y on the fp64 matrix cores
// The one genuinely GEMM-shaped piece of the NODDI fit: c_j = a_j'y for every atom and voxel, the 12 projections
// u_d'y, ||y||^2 -- everything the Gram-space certificates of the NNLS stages (k_nnls_gcert) and the seed solvers need from the
// signal -- AND what the LASSO stage needs of it (round 4): the stage-2 problem of models.pyx:914-926 works on
// y2 = max(0, y_dwi - x_iso iso_dwi), which is NOT linear in y; but for every voxel in which no row clips it is affine, and the
// b0 rows of every atom are exactly 1.0 (lut.pyx:298, 305: resample_kernel leaves them at the ones it starts from), so
//     a_j,dwi' y2 = (c_j - sum_b0 y) - x_iso G_dwi[j, iso],   U2' y2 = U2'y - x_iso U2'iso,
//     ||y2||^2 = (||y||^2 - sum_b0 y^2) - 2 x_iso (c_iso - sum_b0 y) + x_iso^2 G_dwi[iso, iso]
// follow from ONE pass over y: the table gains the rows U2'y, sum_b0 y, sum_b0 y^2 and t_min = min_dwi y_i / iso_i (the voxel clips
// iff x_iso > t_min).  k_s2_prep sorts the voxels after stage 1; only the clipped ones (0.07 % of the bench's voxels, 3 % at
// SNR 10, 16 % of the hard mix) are multiplied again, exactly, by the LASSO instantiation of this kernel over a compact list.
// v_mfma_f64_16x16x4_f64; one workgroup (8 wavefronts) per chunk of the second plan (one orientation), a wavefront takes 16 voxels
// at a time:
//   * A' operand: the orientation's dictionary once per workgroup in LDS in operand order, float32 for the tiles that hold atoms
//     only (the dictionary IS float32: exact), fp64 for the tiles with the bases and the b0 indicator;
//   * B operand: loaded from HBM straight in operand order (lane (q, c16): samples q, 4 + q, ... of voxel c16), one group
//     ahead; every lane keeps its KS operand values in registers for all atom tiles;
//   * output in blocks of 64 voxels, row-major: Cb[block][rows][64] -- a D tile stores four 128-byte row pieces.
// Rows: atoms 0 .. n_atoms - 1, then (aux0 = n_atoms) the kAux* rows below, padded to a multiple of 16.
// (kAux*, gemm_rows: amx_kernels.hpp)
struct GemmArgs {
    // LASSO variant (k_noddi_gemm<true>): the signal is y2 = max(0, y - x_iso iso (- x_dot)) on the stage-2 rows, 0 elsewhere
    // (models.pyx:917-925), the basis is U2, and row j of the output is scaled by colscale[j] (column-normalised atoms); it
    // runs over the compact lists of k_s2_prep (clist / ccount) and writes a compact table + the exact y2~ of those voxels
    const double *xiso;           // [n_vox][2]
    const unsigned char *rowdwi;  // [nS]
    const double *colscale;       // [n_atoms]
    int iso_atom, is_exvivo, n_wm;
    const double *y;              // [n_vox][nS]
    const float *y32;             // float32 signals instead (or null)
    const int *perm;
    const Chunk *schunks;         // pad = first block of the chunk
    const int *n_schunks;
    const float *tiles;           // [ndirs][nS][ldA]
    int tile_stride, ldA, nS, n_atoms;
    int rows, aux0;               // rows of a table block (gemm_rows), first auxiliary row (= n_atoms)
    const double *Ub;             // [ndirs][nS][12]: U (NNLS) / U2 (LASSO)
    const double *U2b;            // NNLS: [ndirs][nS][12] U2 as well (rows kAuxU2), or null
    double *Cb;                   // [n_blocks][rows][64]
    double *ytil;                 // [n][12] bucket order (copy of rows aux0 + kAuxU ..), or null
    const int *clist, *ccount;    // LASSO: bucket positions of the chunk's clipped voxels (compact from the chunk's start), their number
    int *gcount; int n_gcount;    // NNLS: chunk counters of this launch (BlockFeed), helpers' counter at [n_gcount]
    // K-window of this launch: the samples k0 <= row < k1 of every voxel.  A protocol of more than 160 volumes (an HCP-style one: 288)
    // takes several passes of <= 160 rows each, every pass with its rows' operands in LDS: the first writes the table, the others add
    // to it (accumulate = 1); the voxel-major copy of y~ is written by the last (last = 1), which holds the totals
    int k0, k1, accumulate, last;
};

__device__ __forceinline__ double rows_allmin(double k)
{
    double a, b;
    AMX_ROW_SWAP(__builtin_amdgcn_permlane16_swap, k, a, b); k = fmin(a, b);
    AMX_ROW_SWAP(__builtin_amdgcn_permlane32_swap, k, a, b); k = fmin(a, b);
    return k;
}

// MTF > 0: the number of float32 tiles is known at compile time (9 for the 144 + 1 atoms of the default dictionary) and their loop
// is unrolled -- the LDS reads of the next tiles overlap the products of the current ones; MTF = 0: any dictionary
// WIN: this launch covers a window of the samples only (GemmArgs::k0, k1, accumulate, last) -- a template switch, so that the
// one-pass builds keep their constants (the default dictionary's build sits at 246 registers: four run-time values more spilled 58)
template <bool LASSO, int KS, int MTF = 0, bool WIN = false>
__global__ void __launch_bounds__(512) k_noddi_gemm(const GemmArgs a)
{
    constexpr int NB = KS > 25 ? 1 : 3;            // atom tiles in flight together (registers: KS operand values each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    const int nS = a.nS, n_atoms = a.n_atoms, ldA = a.ldA, aux0 = a.aux0;
    const int MT = a.rows >> 4;
    const int n_cols = LASSO ? a.n_wm : n_atoms;
    const int MTf = MTF > 0 ? MTF : (n_cols >> 4);                        // tiles of atoms only: float32 operands
    float *A32 = reinterpret_cast<float *>(smem_g);                       // [MTf][KS][64]
    double *A64 = reinterpret_cast<double *>(A32 + (size_t)MTf * KS * 64);   // [MT - MTf][KS][64]
    double *IsoT = A64 + (size_t)(MT - MTf) * KS * 64;                    // [4 KS] LASSO: iso atom by signal row; NNLS: 1 / iso on the stage-2 rows
    double *ScT = IsoT + 4 * KS;                                          // [rows] column scale by output row, 0 beyond n_wm (LASSO)
    const int n_sch = *a.n_schunks;
    const int own = xcd_chunk((int)blockIdx.x, n_sch);
    if (own < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int k0 = WIN ? a.k0 : 0, k1 = WIN ? a.k1 : nS;    // this launch's window of samples (GemmArgs)
    unsigned long long maskE = 0ull, maskD = 0ull;      // bit ks: sample k0 + 4 ks + q exists / is a stage-2 row
    {
        // (unconditional loads at clamped rows: as `if (row < k1) ... rowdwi[row]` the KS byte loads each sat in a branch of their own,
        //  25 memory round trips one after the other before a workgroup's first product)
        unsigned char dw[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) { const int row = k0 + 4 * ks + q; dw[ks] = a.rowdwi[row < k1 ? row : k0]; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int row = k0 + 4 * ks + q;
            maskE |= (row < k1) ? (1ull << ks) : 0ull;
            maskD |= (row < k1 && dw[ks] != 0) ? (1ull << ks) : 0ull;
        }
    }
    const unsigned long long rowmask = LASSO ? maskD : maskE;
    // every voxel's table: the workgroup's own chunk, then the largest chunks still open, groups of 16 voxels per wavefront (BlockFeed);
    // the exact stage-2 pass over the (few) clipped voxels: its own chunk's list, groups of 16 dealt round robin
    for (int round = 0; round < (LASSO ? 1 : 256); round++) {
    const int cid = LASSO ? own : block_next_chunk(round, own, a.schunks, n_sch, a.gcount, a.gcount + a.n_gcount, reinterpret_cast<int *>(ScT + a.rows), 32 * nw);
    if (cid < 0) break;
    const Chunk ck = a.schunks[cid];
    const int count = LASSO ? a.ccount[cid] : ck.count;
    if (count == 0) { if (LASSO) return; else continue; }
    const float *tile = a.tiles + (size_t)ck.dir * a.tile_stride;
    const double *U = a.Ub + (size_t)ck.dir * nS * kSeedKD;
    const double *U2 = (!LASSO && a.U2b) ? a.U2b + (size_t)ck.dir * nS * kSeedKD : nullptr;
    // The operands of a chunk, SG elements of a thread in flight together (as `A32[e] = tile[..]` in a plain loop every element waited
    // for the one before: ~28 + 7 memory round trips per chunk during which the CU's one workgroup multiplies nothing)
    constexpr int SG = 8;
    for (int e0 = threadIdx.x; e0 < MTf * KS * 64; e0 += SG * blockDim.x) {
        float v[SG];
#pragma unroll
        for (int u = 0; u < SG; u++) {
            const int e = e0 + u * (int)blockDim.x;
            const int l = e & 63, ks = (e >> 6) % KS, mt = (e >> 6) / KS;
            const int atom = 16 * mt + (l & 15), row = k0 + 4 * ks + (l >> 4);
            const bool in = e < MTf * KS * 64 && row < k1;
            v[u] = tile[in ? row * ldA + atom : 0];
            v[u] = in ? v[u] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < SG; u++) { const int e = e0 + u * (int)blockDim.x; if (e < MTf * KS * 64) A32[e] = v[u]; }
    }
    for (int e0 = threadIdx.x; e0 < (MT - MTf) * KS * 64; e0 += SG * blockDim.x) {
        // one load per element whatever its kind: 0 atom (float32), 1 basis (fp64), 2 b0 indicator (byte), 3 zero
        float vf[SG]; double vd[SG]; unsigned char vb[SG]; int kind[SG];
#pragma unroll
        for (int u = 0; u < SG; u++) {
            const int e = e0 + u * (int)blockDim.x;
            const int l = e & 63, ks = (e >> 6) % KS, mt = MTf + (e >> 6) / KS;
            const int r = 16 * mt + (l & 15), row = k0 + 4 * ks + (l >> 4);
            int kd = 3;
            const double *pd = U;
            int od = 0, of = 0;
            if (e < (MT - MTf) * KS * 64 && row < k1) {
                if (r < n_cols) { kd = 0; of = row * ldA + r; }
                else if (r >= aux0 + kAuxU && r < aux0 + kAuxU + kSeedKD) { kd = 1; od = row * kSeedKD + (r - aux0 - kAuxU); }
                else if (U2 != nullptr && r >= aux0 + kAuxU2 && r < aux0 + kAuxU2 + kSeedKD) { kd = 1; pd = U2; od = row * kSeedKD + (r - aux0 - kAuxU2); }
                else if (!LASSO && r == aux0 + kAuxB0) kd = 2;
            }
            kind[u] = kd;
            vf[u] = tile[of];
            vd[u] = pd[od];
            vb[u] = a.rowdwi[kd == 2 ? row : k0];
        }
#pragma unroll
        for (int u = 0; u < SG; u++) {
            const int e = e0 + u * (int)blockDim.x;
            const double v = kind[u] == 0 ? (double)vf[u] : (kind[u] == 1 ? vd[u] : (kind[u] == 2 ? (vb[u] ? 0.0 : 1.0) : 0.0));
            if (e < (MT - MTf) * KS * 64) A64[e] = v;
        }
    }
    for (int e = threadIdx.x; e < 4 * KS; e += blockDim.x) {
        const int row = k0 + e;
        const double iso = row < k1 ? (double)tile[row * ldA + a.iso_atom] : 0.0;
        IsoT[e] = LASSO ? iso : ((row < k1 && a.rowdwi[row] && iso > 0.0) ? 1.0 / iso : 0.0);
    }
    if (LASSO) for (int e = threadIdx.x; e < a.rows; e += blockDim.x) ScT[e] = e < a.n_wm ? a.colscale[e] : 0.0;
    __syncthreads();
    const int n_groups = (count + 15) >> 4;
    // the signals of a group of 16 voxels straight in operand order: lane (q, c16) reads samples q, 4 + q, 8 + q, ... of voxel c16
    // (four lanes share every 32-byte sector; all bytes of a row are used by the KS loads).  The loads of group g + nw are
    // issued before the products of group g, so they are long done when their turn comes; no LDS staging.
    double bn[KS], xi_n = 0.0, xd_n = 0.0;
    int pos_n = 0;
    auto issue = [&](int g) {
        const int k = 16 * g + c16;
        const int kk = k < count ? k : count - 1;
        pos_n = LASSO ? a.clist[ck.start + kk] : ck.start + kk;
        const int vox = a.perm[pos_n];
        if (a.y32 != nullptr) {
            // (the loads under their guards, the conversions outside: converted inside, every load was waited for before the next left)
            const float *yv = a.y32 + (size_t)vox * nS + k0 + q;
            float bf32[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bf32[ks] = (k0 + 4 * ks + q < k1) ? yv[4 * ks] : 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bn[ks] = (double)bf32[ks];
        } else {
            const double *yv = a.y + (size_t)vox * nS + k0 + q;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bn[ks] = (k0 + 4 * ks + q < k1) ? yv[4 * ks] : 0.0;
        }
        if (LASSO) { xi_n = a.xiso[(size_t)vox * 2]; xd_n = a.is_exvivo ? a.xiso[(size_t)vox * 2 + 1] : 0.0; }
    };
    const bool accumulate = WIN && a.accumulate != 0, last = !WIN || a.last != 0;
    BlockFeed<16> bf;
    auto advance = [&](int g) -> int {                  // the group after g for this wavefront, -1: none
        if (LASSO) { const int n = g < 0 ? wave : g + nw; return n < n_groups ? n : -1; }
        return bf.next(a.gcount + cid, count, lane);
    };
    if (!LASSO) bf.start(a.gcount + cid, lane);
    int g = advance(-1);
    if (g >= 0) issue(g);
    while (g >= 0) {
        const bool live = 16 * g + c16 < count;
        const int pos = pos_n;
        double b[KS], yy = 0.0, yb = 0.0, tmin = __builtin_huge_val();
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            double t = bn[ks];
            if (LASSO) {
                t = t - xi_n * IsoT[4 * ks + q] - xd_n;
                t = t < 0.0 ? 0.0 : t;                                   // (NaN stays NaN: comparisons with NaN are false)
            }
            b[ks] = (live && ((rowmask >> ks) & 1ull)) ? t : 0.0;
            yy += b[ks] * b[ks];
            if (!LASSO) {
                const bool inD = (maskD >> ks) & 1ull;
                yb += inD ? 0.0 : b[ks] * b[ks];
                const double r = b[ks] * IsoT[4 * ks + q];               // y_i / iso_i on the stage-2 rows
                tmin = (inD && r < tmin) ? r : tmin;
            }
        }
        yy = rows_allreduce(yy);
        if (!LASSO) { yb = rows_allreduce(yb); tmin = rows_allmin(tmin); }
        const int g_nxt = advance(g);
        if (g_nxt >= 0) issue(g_nxt);
        const int blk = ck.pad + (g >> 2), col = 16 * (g & 3) + c16;
        double *out = a.Cb + (size_t)blk * a.rows * 64 + col;
#pragma unroll (MTF > 0 && !LASSO ? 3 : 1)
        for (int mt = 0; mt < MTf; mt += NB) {
            // the operands of NB atom tiles first (all their LDS reads in flight together), then the products back to back
            float af[NB][KS];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                if (mt + u < MTf) {
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) af[u][ks] = A32[((mt + u) * KS + ks) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            seed_v4d acc[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) acc[u] = (seed_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int u = 0; u < NB; u++)
                    if (mt + u < MTf) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)af[u][ks], b[ks], acc[u], 0, 0, 0);
            }
            if (live) {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    if (mt + u < MTf) {
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            const int row = 16 * (mt + u) + 4 * rr + q;
                            double v = LASSO ? ScT[row] * acc[u][rr] : acc[u][rr];
                            if (accumulate) v += out[(size_t)row * 64];            // (the earlier windows' share; uniform branch)
                            out[(size_t)row * 64] = v;
                        }
                    }
                }
            }
        }
        for (int mt = MTf; mt < MT; mt++) {
            seed_v4d acc = (seed_v4d){0.0, 0.0, 0.0, 0.0};
            const double *Am = A64 + (size_t)(mt - MTf) * KS * 64;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Am[ks * 64 + lane], b[ks], acc, 0, 0, 0);
            if (live) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int row = 16 * mt + 4 * rr + q;
                    double v = (LASSO && row < a.n_wm) ? ScT[row] * acc[rr] : acc[rr];
                    v = (row == aux0 + kAuxYY) ? yy : v;
                    if (!LASSO) { v = (row == aux0 + kAuxYB) ? yb : v; v = (row == aux0 + kAuxTmin) ? tmin : v; }
                    if (accumulate) {
                        const double old = out[(size_t)row * 64];
                        v = (!LASSO && row == aux0 + kAuxTmin) ? (old < v ? old : v) : v + old;       // sums add up over the windows, t_min is a minimum
                    }
                    out[(size_t)row * 64] = v;
                    // the projections once more in voxel-major order for the kernels that take one voxel at a time
                    if (a.ytil != nullptr && last && row >= aux0 + kAuxU && row < aux0 + kAuxU + kSeedKD) a.ytil[(size_t)pos * kSeedKD + (row - aux0 - kAuxU)] = v;
                }
            }
        }
        g = g_nxt;
    }
    __syncthreads();                          // every wavefront is through with this chunk's operands
    }
}

// ================================================================== after stage 1: which voxels' stage-2 signal clips?
// One voxel per lane over the blocks of the table: y2~ = U2'y - x_iso U2'iso for every voxel (the LASSO seed solver's input),
// and the voxels with x_iso > t_min -- at least one row of y - x_iso iso is negative, models.pyx:924-925 clips it -- compacted per
// chunk for the exact pass of k_noddi_gemm<true>; cslot[pos] = place in that compact list, -1 for the (unclipped) majority.
struct S2PrepArgs {
    const int *perm;
    const Chunk *schunks;
    const int *n_schunks;
    const double *Cb;             // table of k_noddi_gemm<false>
    int rows, aux0;
    const double *xiso;           // [n_vox][2]
    const double *u2iso;          // [ndirs][12] U2'iso (amx_build_basis)
    double *ytil2;                // [n][12] bucket order
    int *clist, *ccount, *cslot;
    int *status;
    int force_all;                // every voxel takes the exact pass (dictionary whose b0 rows are not all ones; AMX_S2_EXACT=1)
};

__global__ void __launch_bounds__(256) k_s2_prep(const S2PrepArgs a)
{
    const int cid = xcd_chunk((int)blockIdx.x, *a.n_schunks);
    if (cid < 0) return;
    const Chunk ck = a.schunks[cid];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const double *__restrict__ u2 = a.u2iso + (size_t)ck.dir * kSeedKD;
    const int n_blocks = (ck.count + 63) >> 6;
    int nclip = 0;
    for (int bl = wave; bl < n_blocks; bl += nw) {
        const int k = 64 * bl + lane;
        const bool valid = k < ck.count;
        const int pos = ck.start + (valid ? k : ck.count - 1);
        const double *Crow = a.Cb + (size_t)(ck.pad + bl) * a.rows * 64 + lane;
        const double xi = a.xiso[(size_t)a.perm[pos] * 2];
        const double tmin = Crow[(size_t)(a.aux0 + kAuxTmin) * 64];
        const bool clipped = valid && (a.force_all != 0 || !(xi <= tmin));       // (NaN: the exact pass carries it on)
        double yt[kSeedKD];
#pragma unroll
        for (int d = 0; d < kSeedKD; d++) yt[d] = Crow[(size_t)(a.aux0 + kAuxU2 + d) * 64] - xi * u2[d];
        if (valid) {
            double2 *dst = reinterpret_cast<double2 *>(a.ytil2 + (size_t)pos * kSeedKD);
#pragma unroll
            for (int d = 0; d < kSeedKD; d += 2) dst[d >> 1] = make_double2(yt[d], yt[d + 1]);
        }
        const unsigned long long cm = __ballot(clipped);
        int slot = -1;
        if (cm != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&a.ccount[cid], __builtin_popcountll(cm));
            base = __builtin_amdgcn_readfirstlane(base);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
            if (clipped) { slot = base + rank; a.clist[ck.start + slot] = pos; }
            nclip += __builtin_popcountll(cm);
        }
        if (valid) a.cslot[pos] = slot;
    }
    if (lane == 0 && nclip > 0) atomicAdd(&a.status[ST_CLIP], nclip);
}

// U2'iso of every orientation (once per dictionary upload): what x_iso takes out of the projected stage-2 signal
__global__ void k_u2iso(const float *__restrict__ tiles, int tile_stride, int nS, int ldA, int iso_atom, const double *__restrict__ U2b, double *__restrict__ out)
{
    const int d = threadIdx.x;
    if (d >= kSeedKD) return;
    const float *tile = tiles + (size_t)blockIdx.x * tile_stride;
    const double *U2 = U2b + (size_t)blockIdx.x * nS * kSeedKD;
    double s = 0.0;
    for (int i = 0; i < nS; i++) s += U2[i * kSeedKD + d] * (double)tile[i * ldA + iso_atom];      // (rows of U2 outside the stage-2 rows are zero)
    out[(size_t)blockIdx.x * kSeedKD + d] = s;
}

// Which atoms of which voxel have a compressed dual value above the voxel's threshold?  One fp64 MFMA product for the 64 voxels
// of the wavefront (operands as in seed_scan_mfma); every lane publishes its residual r~, its threshold and the mask of the
// atoms it wants examined in the wavefront's LDS block Rb [64][16], the lanes that hold the D tiles test "candidate and
// value > threshold" and the flag words travel back to the owner through the two row swaps.  ex[3]: flagged atoms of the own voxel.
template <int KS, int MT>
__device__ __forceinline__ void seed_flags_mfma(const double *Aop, double *Rb, int lane, const double (&rt)[4 * KS], bool good, double thr_own,
                                                const unsigned long long (&cand)[3], unsigned long long (&ex)[3])
{
    constexpr int RBW = 16;
    const int q = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int d = 0; d < 4 * KS; d++) Rb[lane * RBW + d] = good ? rt[d] : 0.0;
    Rb[lane * RBW + 12] = thr_own;                                     // flag when the compressed dual value exceeds this
    unsigned long long *Mb = reinterpret_cast<unsigned long long *>(Rb + lane * RBW + 13);
#pragma unroll
    for (int w3 = 0; w3 < 3; w3++) Mb[w3] = good ? cand[w3] : 0ull;
    double b[4][KS], thr[4];
    unsigned cq[4][6];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const int src = 16 * nt + c16;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[nt][ks] = Rb[src * RBW + 4 * ks + q];
        thr[nt] = Rb[src * RBW + 12];
        const unsigned long long *Ms = reinterpret_cast<const unsigned long long *>(Rb + src * RBW + 13);
#pragma unroll
        for (int w3 = 0; w3 < 3; w3++) { const unsigned long long m = Ms[w3] >> q; cq[nt][2 * w3] = (unsigned)m; cq[nt][2 * w3 + 1] = (unsigned)(m >> 32); }
    }
    unsigned fq[4][6];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
#pragma unroll
        for (int w6 = 0; w6 < 6; w6++) fq[nt][w6] = 0u;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        double av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) av[ks] = Aop[(mt * KS + ks) * 64 + lane];
        seed_v4d acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[nt] = (seed_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], b[nt][ks], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int bit = 16 * mt + 4 * rr;            // position in the row-shifted masks
                const bool hit = ((cq[nt][bit >> 5] >> (bit & 31)) & 1u) && (acc[nt][rr] > thr[nt]);
                fq[nt][bit >> 5] |= hit ? (1u << (bit & 31)) : 0u;
            }
        }
    }
    // the four rows that share a voxel: OR of their flag words (shifted back by the row), owner = row of the voxel
#pragma unroll
    for (int w3 = 0; w3 < 3; w3++) {
        unsigned long long mine = 0ull;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const unsigned long long f = (((unsigned long long)fq[nt][2 * w3 + 1] << 32) | (unsigned long long)fq[nt][2 * w3]) << q;
            unsigned lo = (unsigned)f, hi = (unsigned)(f >> 32);
            {
                const auto s1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); lo = s1[0] | s1[1];
                const auto s2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); lo = s2[0] | s2[1];
                const auto s3 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false); hi = s3[0] | s3[1];
                const auto s4 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); hi = s4[0] | s4[1];
            }
            const unsigned long long m = ((unsigned long long)hi << 32) | (unsigned long long)lo;
            mine = (q == nt) ? m : mine;
        }
        ex[w3] = mine;
    }
}

// ================================================================== Gram-space certificates of the NNLS seeds, one voxel per lane
// With c = A'y (k_noddi_gemm) and the orientation's Gram matrix G = A'A the Kuhn-Tucker conditions of a seeded support P
// need neither the signal nor the dictionary:  x_P = G_PP^-1 c_P (Cholesky in the lane's registers),  dual value of atom t:
// u_t = c_t - G_tP x_P,  ||r||^2 = ||y||^2 - x_P'c_P.  Normal equations square the condition number, so a lane only
// certifies what they can certify: the pivot ratio min L_ii / max L_ii (a lower bound of 1 / cond(A_P)) must exceed
// kGcertPivot{1,3} (tools/lab/gram_certify_lab.py: at 3e-3 x agrees with the QR solution to 2.5e-8 relative at worst, 2e-12
// in the median, for 94 % of the voxels; at 1e-3: 5.5e-8, 98 %), every coefficient must be positive and every examined dual value below -1e-10
// (the Gram-form value is exact to ~1e-11).  Which atoms are examined: stage 3 -- all admissible ones (the LASSO support,
// a handful); stage 1 -- those whose compressed dual value s_t'(y~ - S_P x) is within kappa ||r|| of zero or above, found
// for the 64 voxels of the wavefront by one fp64 MFMA product (see seed_scan_mfma).  Everything else -- refused seeds,
// ill-conditioned supports, ambiguous signs, non-finite signals -- is left to the wavefront-per-voxel kernel (k_noddi with
// the done[] flags: it skips the certified voxels), whose certificate works on the true residual.
//
// RESCUE pass (second launch, over the first pass's left-over lists): five in six of the voxels the first pass leaves are refused
// for conditioning alone (1 M voxels of the bench mix: 40 093 of 47 620 at stage 1, 23 299 of 24 788 at stage 3).  Their support
// is right; only x_P = G_PP^-1 c_P is not good enough.  The lane corrects it with the signal and the atoms themselves --
// x <- x + G_PP^-1 A_P'(y - A_P x), the corrected semi-normal equations, factor already in registers, atoms from the orientation's
// float32 tile in LDS, one pass over the voxel's nS samples per correction -- until the correction is below kRescueStep |x|
// (tools/lab/csne_lab.py: two or three corrections bring x to within 7e-10 of the 80-bit solution, where the QR solution of the
// wavefront-per-voxel kernel itself is at 3e-9), and then runs the same tests.  What does not settle in kRescuePasses corrections,
// has a pivot ratio below kRescuePivot, a coefficient too close to zero to call, or fails a dual test stays on the list.
constexpr double kGcertPivot1 = 1e-3, kGcertPivot3 = 3e-3;   // stage 1 only hands x_iso to the LASSO stage; stage 3's x becomes the maps
constexpr double kRescuePivot = 1e-6, kRescueStep = 1e-9, kRescueTiny = 1e-7;
constexpr int kRescuePasses = 4;
struct GcertArgs {
    const int *perm;
    const Chunk *schunks;
    const int *n_schunks;
    const unsigned long long *seeds;   // [n], bucket order
    const double *Cb;                  // [n_blocks][rows][64]
    int rows, aux0;                    // table layout (gemm_rows)
    const double *gram;                // [ndirs][n_atoms][ldG]
    int ldG, n_atoms, n_wm, nS, iso_atom, dot_atom, n_maps;
    const double *Sb;                  // [ndirs][n_atoms][12]
    const double *kappa0;              // [ndirs] max ||(I - U U') a_j||
    const unsigned long long *supp;    // stage 3: [n_vox][4]
    const float *icvf, *kappa;         // stage 3: maps
    unsigned char *done;               // [n], bucket order: 1 = certified here
    int *gcount; int n_gcount;         // chunk counters of this launch (BlockFeed; zeroed before the fit), helpers' counter at [n_gcount]
    int *rlist, *rcount;               // refused voxels: positions, compact from the chunk's own start; count per chunk
    const int *rlist_in, *rcount_in;   // RESCUE pass: the left-over lists of the first pass
    const double *y; const float *y32; // RESCUE pass: the signals [n_vox][nS] (one of the two)
    const float *tiles; size_t tile_stride; int ldA, tile_in_lds;   // RESCUE pass: the orientations' atoms [nS][ldA]
    double *xiso;                      // stage 1 out: [n_vox][2]
    double *est, *rmse, *nrmse, *mod;  // stage 3 out
    double *xdbg;                      // AMX_F_DEBUG_X: [n_vox][3][n_atoms]
    int *stats;
    int fork_skip;                     // stage 3: 1 = voxels marked kSeedForked are neither certified nor listed (finished on the side stream)
};

// REPAIR (round 5; long protocols, where the compressed-space seed is wrong in 4.5 % of the voxels instead of 0.9 %): a lane whose seed
// fails for ONE reason the lane itself can mend gets a second look in the same block -- a coefficient that came out non-positive: those
// atoms leave; a positive dual value outside the support: the most violating atom enters -- and the whole certificate (gather, factor,
// solve, screening, exact dual values) runs once more on the mended support.  tools/lab/repair_lab.py: at 288 volumes one such step
// settles 2.8 of the 4.5 % (the left-over kernel reads its tile from L2 there: ~300 us per voxel); at 99 volumes 0.08 of 0.87 %, and
// every third block would pay a second pass for it: off there.
#ifndef AMX_REPAIR_ROUNDS
#define AMX_REPAIR_ROUNDS 1
#endif
template <int STAGE, bool RESCUE = false, bool REPAIR = false>
__global__ void __launch_bounds__(256, RESCUE ? 1 : 2) k_nnls_gcert(const GcertArgs a)
{
    constexpr int KD = kSeedKD, KS = KD / 4, MT = 10, MS = 8, LD = kSeedLd, RBW = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    double *Sl = reinterpret_cast<double *>(smem_c);             // [n_atoms][LD]
    const int n_atoms = a.n_atoms;
    double *Aop = Sl + (size_t)n_atoms * LD + 2;                  // [MT][KS][64]
    double *Rb = Aop + MT * KS * 64 + (threadIdx.x >> 6) * (64 * RBW);   // per wavefront [64][16]: r~ | margin | masks
    float *Atl = reinterpret_cast<float *>(Aop + MT * KS * 64 + ((int)blockDim.x >> 6) * (64 * RBW));   // RESCUE: the orientation's atoms [nS][ldA]
    const int n_sch = *a.n_schunks;
    const int own = xcd_chunk((int)blockIdx.x, n_sch);
    if (own < 0) return;
    const int lane = threadIdx.x & 63, nw = (int)blockDim.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
#ifdef AMX_STATS
    long long gph[5] = {0, 0, 0, 0, 0}, gpt = (long long)__builtin_readcyclecounter();
#define GC_PH(k) do { const long long t__ = (long long)__builtin_readcyclecounter(); gph[k] += t__ - gpt; gpt = t__; } while (0)
#else
#define GC_PH(k) do { } while (0)
#endif
    for (int round = 0; round < (RESCUE ? 1 : 256); round++) {
    const int cid = RESCUE ? own : block_next_chunk(round, own, a.schunks, n_sch, a.gcount, a.gcount + a.n_gcount, reinterpret_cast<int *>(Sl + (size_t)n_atoms * LD), 64 * nw);
    if (cid < 0) break;
    const Chunk ck = a.schunks[cid];
    const int n_items = RESCUE ? a.rcount_in[cid] : ck.count;
    if (RESCUE && n_items == 0) return;
    const double *__restrict__ Sg = a.Sb + (size_t)ck.dir * n_atoms * KD;
    const double *__restrict__ Gd = a.gram + (size_t)ck.dir * n_atoms * a.ldG;
    const float *__restrict__ At = nullptr;    // RESCUE: the atoms in LDS when the tile fits beside the rest, else the L2-resident original
    if (RESCUE) {
        At = a.tiles + (size_t)ck.dir * a.tile_stride;
        if (a.tile_in_lds) {
            // (eight elements of a thread in flight: the plain loop kept one -- 56 memory round trips for a 57 KB tile, most of this pass's time)
            constexpr int UB = 8;
            const int NE = a.nS * a.ldA;
            for (int e0 = threadIdx.x; e0 < NE; e0 += UB * (int)blockDim.x) {
                float v[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) { const int e = e0 + u * (int)blockDim.x; v[u] = At[e < NE ? e : 0]; }
#pragma unroll
                for (int u = 0; u < UB; u++) { const int e = e0 + u * (int)blockDim.x; if (e < NE) Atl[e] = v[u]; }
            }
        }
    }
    stage_rows<KD, LD>(Sl, Sg, n_atoms, KD);
    stage_operand<KS, MT>(Aop, Sg, n_atoms, KD);
    __syncthreads();
    const double kap = a.kappa0[ck.dir];
    BlockFeed<64> bf;
    if (!RESCUE) bf.start(a.gcount + cid, lane);
    for (int bls = (int)(threadIdx.x >> 6); ; bls += nw) {
        int bl = bls;
        if (RESCUE) { if (64 * bl >= n_items) break; }
        else { bl = bf.next(a.gcount + cid, ck.count, lane); if (bl < 0) break; }
        const int k = 64 * bl + lane;
        const bool valid0 = k < n_items;
        const int kk = valid0 ? k : n_items - 1;
        const int pos = RESCUE ? a.rlist_in[ck.start + kk] : ck.start + kk;
        const int rel = pos - ck.start;                                   // (first pass: rel = 64 bl + lane)
        const double *Crow = a.Cb + (size_t)(ck.pad + (rel >> 6)) * a.rows * 64 + (rel & 63);
        const unsigned long long seed = a.seeds[pos];
        const bool valid = valid0 && !(STAGE == 3 && a.fork_skip && seed == kSeedForked);
        const int vox = a.perm[pos];
        SeedLane<MS> V;
        V.clear();
        bool okv = valid && seed != kNoSeed;
        // ---- decode (bytes from the low end; >= 0xf0: empty)
        {
            int n0 = 0; bool tail = false;
#pragma unroll
            for (int s = 0; s < MS; s++) {
                const int b = (int)((seed >> (8 * s)) & 0xffull);
                const bool on = b < 0xf0;
                if (on && tail) okv = false;                      // not a prefix
                if (!on) tail = true;
                if (on && b >= n_atoms) okv = false;
                V.idx[s] = (on && b < n_atoms) ? b : 0;
                n0 += on ? 1 : 0;
            }
            V.np = okv ? n0 : 0;
        }
        static_assert(!(RESCUE && REPAIR), "the rescue pass takes the first pass's lists as they are");
        bool cert_any = false, easy_any = false, live = true;
#pragma unroll 1
        for (int rep = 0; rep < (REPAIR ? AMX_REPAIR_ROUNDS + 1 : 1); rep++) {
        unsigned long long cand[3] = {~0ull, ~0ull, ~0ull};          // admissible atoms outside the seed
        if (STAGE == 3) {
#pragma unroll
            for (int w3 = 0; w3 < 3; w3++) cand[w3] = a.supp[(size_t)vox * 4 + w3];
            cand[a.iso_atom >> 6] |= 1ull << (a.iso_atom & 63);
            if (a.dot_atom >= 0) cand[a.dot_atom >> 6] |= 1ull << (a.dot_atom & 63);
        } else {
#pragma unroll
            for (int w3 = 0; w3 < 3; w3++) {
                const int cnt = n_atoms - 64 * w3;
                cand[w3] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
            }
        }
#pragma unroll
        for (int s = 0; s < MS; s++) {
            if (s < V.np) {
                const int t = V.idx[s];
                unsigned long long wsel = 0ull;
#pragma unroll
                for (int w3 = 0; w3 < 3; w3++) wsel = ((t >> 6) == w3) ? cand[w3] : wsel;
                if (!((wsel >> (t & 63)) & 1ull)) okv = false;      // seeded atom not admissible (or twice in the seed)
#pragma unroll
                for (int w3 = 0; w3 < 3; w3++) cand[w3] = ((t >> 6) == w3) ? (cand[w3] & ~(1ull << (t & 63))) : cand[w3];
            }
        }
        if (!okv) V.np = 0;
        GC_PH(0);
        // ---- Gram block, c_P, ||y||^2
#pragma unroll
        for (int s = 0; s < MS; s++) {
#pragma unroll
            for (int t = 0; t <= s; t++) V.T[stri<MS>(s, t)] = (s < V.np) ? Gd[(size_t)V.idx[s] * a.ldG + V.idx[t]] : 0.0;
            V.c[s] = (s < V.np) ? Crow[(size_t)V.idx[s] * 64] : 0.0;
        }
        const double yy = Crow[(size_t)(a.aux0 + kAuxYY) * 64];
        double pmax = 0.0, pmin = __builtin_huge_val();
        bool piv = V.factor();
#pragma unroll
        for (int s = 0; s < MS; s++) {
            const double di = V.T[stri<MS>(s, s)];
            if (s < V.np) { pmax = di > pmax ? di : pmax; pmin = di < pmin ? di : pmin; }
        }
        const bool ill = V.np > 0 && !(pmin > (STAGE == 1 ? kGcertPivot1 : kGcertPivot3) * pmax);
        if (!RESCUE && ill) piv = false;
        double z[MS];
        V.solve(z);
        bool feas = true;
        double rho2 = yy;
#pragma unroll
        for (int s = 0; s < MS; s++) { V.x[s] = z[s]; if (s < V.np && !(z[s] > 0.0)) feas = false; rho2 -= z[s] * V.c[s]; }
        rho2 = rho2 > 0.0 ? rho2 : 0.0;
        if constexpr (RESCUE) {
            // only the supports refused for conditioning come back (the others failed a test that would fail again)
            bool need = valid && okv && piv && ill && (pmin > kRescuePivot * pmax) && (yy <= 1.79769313486231570e308);
            bool conv = false;
            const double *yv = a.y ? a.y + (size_t)vox * a.nS : nullptr;
            const float *yv32 = a.y32 ? a.y32 + (size_t)vox * a.nS : nullptr;
            for (int pass = 0; pass < kRescuePasses; pass++) {
                if (__ballot(need && !conv) == 0ull) break;
                double rho[MS], rr = 0.0;
#pragma unroll
                for (int s = 0; s < MS; s++) rho[s] = 0.0;
                // rows in batches of RB: a batch's loads (the lane's own samples -- 8 bytes at a stride of one voxel -- and the
                // atoms' entries) are issued together, the next batch's samples before this batch's arithmetic
                constexpr int RB = 9;
                auto rows = [&](const auto *Ap) {
                    double yb[RB], yn[RB];
                    auto fetch = [&](int i0, double (&dst)[RB]) {
                        // (the signal's type is chosen once per batch, not per element: `yv ? yv[i] : yv32[i]` made every load wait for the one before)
                        if (yv != nullptr) {
#pragma unroll
                            for (int u = 0; u < RB; u++) { const int i = i0 + u < a.nS ? i0 + u : a.nS - 1; dst[u] = yv[i]; }
                        } else {
                            float t32[RB];
#pragma unroll
                            for (int u = 0; u < RB; u++) { const int i = i0 + u < a.nS ? i0 + u : a.nS - 1; t32[u] = yv32[i]; }
#pragma unroll
                            for (int u = 0; u < RB; u++) dst[u] = (double)t32[u];
                        }
                    };
                    fetch(0, yn);
                    for (int i0 = 0; i0 < a.nS; i0 += RB) {
#pragma unroll
                        for (int u = 0; u < RB; u++) yb[u] = yn[u];
                        float av[RB][MS];
#pragma unroll
                        for (int u = 0; u < RB; u++) {
                            const int i = i0 + u < a.nS ? i0 + u : a.nS - 1;
#pragma unroll
                            for (int s = 0; s < MS; s++) av[u][s] = Ap[(size_t)i * a.ldA + V.idx[s]];     // (slots >= np: idx = 0, x = 0)
                        }
                        if (i0 + RB < a.nS) fetch(i0 + RB, yn);
#pragma unroll
                        for (int u = 0; u < RB; u++) {
                            double ri = yb[u];
#pragma unroll
                            for (int s = 0; s < MS; s++) ri -= (double)av[u][s] * V.x[s];
                            ri = i0 + u < a.nS ? ri : 0.0;
#pragma unroll
                            for (int s = 0; s < MS; s++) rho[s] += (double)av[u][s] * ri;
                            rr += ri * ri;
                        }
                    }
                };
                if (a.tile_in_lds) rows(Atl); else rows(At);
                double dz[MS];
                V.solve_rhs(rho, dz);
                double xm = 0.0, dm = 0.0;
                const bool upd = need && !conv;
#pragma unroll
                for (int s = 0; s < MS; s++) {
                    if (upd && s < V.np) V.x[s] += dz[s];
                    const double ax = fabs(V.x[s]), ad = fabs(dz[s]);
                    if (s < V.np) { xm = ax > xm ? ax : xm; dm = ad > dm ? ad : dm; }
                }
                if (upd) { rho2 = rr; if (dm <= kRescueStep * xm) conv = true; }      // (||r||^2 of the x before this -- negligible -- correction)
            }
            piv = need && conv;
            feas = true;
            double xm = 0.0;
#pragma unroll
            for (int s = 0; s < MS; s++) xm = (s < V.np && V.x[s] > xm) ? V.x[s] : xm;
#pragma unroll
            for (int s = 0; s < MS; s++) if (s < V.np && !(V.x[s] > kRescueTiny * xm)) feas = false;
        }
        bool good = okv && piv && feas && (yy <= 1.79769313486231570e308);
        GC_PH(1);
        // ---- which atoms need their exact dual value
        // (stage 3 as well: its <= 25 admissible atoms used to be examined one by one -- 9 scattered loads each, the texture path of
        //  the CU was the bottleneck; the MFMA screening clears all but ~1 of them)
        unsigned long long ex[3] = {cand[0], cand[1], cand[2]};
        {
            double rt[KD];
#pragma unroll
            for (int d = 0; d < KD; d++) rt[d] = Crow[(size_t)(a.aux0 + kAuxU + d) * 64];
#pragma unroll
            for (int s = 0; s < MS; s++) {
                const double *col = Sl + V.idx[s] * LD;
                double cv[KD];
                seed_col<KD>(col, cv);
#pragma unroll
                for (int d = 0; d < KD; d++) rt[d] -= V.x[s] * cv[d];
            }
            seed_flags_mfma<KS, MT>(Aop, Rb, lane, rt, good, good ? -1.0625 * kap * sqrt(rho2) - 1e-12 : __builtin_huge_val(), cand, ex);
        }
        GC_PH(2);
        // ---- exact (Gram-form) dual values of the flagged atoms: u_t = c_t - G_tP x
        bool viol = false;
        int n_ex = 0;
        double umax = 0.0;                      // REPAIR: the largest positive exact dual value and its atom
        int tmax = -1;
        {
            // (UN flagged atoms per step: their 1 + np loads are independent and in flight together -- one atom per step made this loop,
            //  a chain of L2 round trips as long as the wavefront's LONGEST list, 72 % of the kernel)
            constexpr int UN = 2;
            unsigned long long rem[3] = {good ? ex[0] : 0ull, good ? ex[1] : 0ull, good ? ex[2] : 0ull};
            for (int it = 0; it < 192; it += UN) {
                if (__ballot((rem[0] | rem[1] | rem[2]) != 0ull) == 0ull) break;
                int tt[UN];
                bool on[UN];
#pragma unroll
                for (int u4 = 0; u4 < UN; u4++) {
                    int wq = -1;
#pragma unroll
                    for (int qq = 2; qq >= 0; qq--) wq = (rem[qq] != 0ull) ? qq : wq;
                    unsigned long long word = 0ull;
#pragma unroll
                    for (int qq = 0; qq < 3; qq++) word = (wq == qq) ? rem[qq] : word;
                    const int t = (wq >= 0) ? wq * 64 + __builtin_ctzll(word) : 0;
#pragma unroll
                    for (int qq = 0; qq < 3; qq++) rem[qq] = (wq == qq) ? (rem[qq] & (rem[qq] - 1ull)) : rem[qq];
                    on[u4] = wq >= 0 && t < n_atoms;
                    tt[u4] = on[u4] ? t : 0;
                }
                // Loads under the lanes' own masks, used outside the guards.  A scattered 8-byte load costs the CU's address unit about a
                // cycle per ACTIVE lane, and that unit -- not HBM, not the vector ALUs -- is what these certificates run against
                // (~190 load instructions x 64 lanes per block of 64 voxels with every lane loading whether it had a flagged atom
                // or not: 0.3 ms per 1 M voxels at one lane per cycle and CU).  A guard that also holds the USE of its load makes
                // the compiler wait inside the guard (a round trip per load): hence load inside, arithmetic outside.
                double uu[UN], gg[UN][MS];
#pragma unroll
                for (int u4 = 0; u4 < UN; u4++) {
                    uu[u4] = 0.0;
#pragma unroll
                    for (int s = 0; s < MS; s++) gg[u4][s] = 0.0;
                    if (on[u4]) {
                        uu[u4] = Crow[(size_t)tt[u4] * 64];
                        const double *gt = Gd + (size_t)tt[u4] * a.ldG;
#pragma unroll
                        for (int s = 0; s < MS; s++) if (s < V.np) gg[u4][s] = gt[V.idx[s]];
                    }
                }
#pragma unroll
                for (int u4 = 0; u4 < UN; u4++) {
                    double u = uu[u4];
#pragma unroll
                    for (int s = 0; s < MS; s++) u -= gg[u4][s] * V.x[s];
                    if (on[u4] && !(u < -1e-10)) viol = true;             // positive, or too close to zero for the Gram form to call
                    if (REPAIR && on[u4] && u > umax) { umax = u; tmax = tt[u4]; }
                    n_ex += on[u4] ? 1 : 0;
                }
            }
        }
        GC_PH(3);
        const bool cert = live && good && !viol;
        // done = 2: refused for the conditioning of its Gram block alone -- the support is most likely right, and the left-over kernel's
        // certificate on the true residual settles it in ~15 us; 0: wrong or no seed, ~140 us of Lawson-Hanson.  The left-over kernel
        // starts the long ones first (k_noddi)
        const bool easy = !RESCUE && live && okv && ill && (pmin > kRescuePivot * pmax);
        cert_any = cert_any || cert;
        if (rep == 0) easy_any = easy;
#ifdef AMX_STATS
        if (a.stats && !RESCUE && rep == 0) {
            const int nc = __builtin_popcountll(__ballot(cert)), nv = __builtin_popcountll(__ballot(valid));
            const int npv = __builtin_popcountll(__ballot(valid && okv && !piv)), nfe = __builtin_popcountll(__ballot(valid && okv && piv && !feas)), nvi = __builtin_popcountll(__ballot(valid && good && viol));
            if (lane == 0) { atomicAdd(&a.stats[0], nv); atomicAdd(&a.stats[1], nc); atomicAdd(&a.stats[2], npv); atomicAdd(&a.stats[3], nfe); atomicAdd(&a.stats[4], nvi); }
            int ne = n_ex;
            for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o);
            if (lane == 0) atomicAdd(&a.stats[5], ne);
        }
#endif
        if (cert) {
            double xi = 0.0, xd = 0.0;
#pragma unroll
            for (int s = 0; s < MS; s++) { if (s < V.np && V.idx[s] == a.iso_atom) xi = V.x[s]; if (s < V.np && V.idx[s] == a.dot_atom) xd = V.x[s]; }
            if (STAGE == 1) {
                a.xiso[(size_t)vox * 2] = xi; a.xiso[(size_t)vox * 2 + 1] = xd;
            } else {
                // models.pyx:945-967
                double sum_atoms = 1e-16;
#pragma unroll
                for (int s = 0; s < MS; s++) sum_atoms += (s < V.np) ? V.x[s] : 0.0;
                double sum_wm = 0.0;
#pragma unroll
                for (int s = 0; s < MS; s++) sum_wm += (s < V.np && V.idx[s] < a.n_wm) ? V.x[s] / sum_atoms : 0.0;
                sum_wm += 1e-16;
                double f1 = 0.0, f2 = 0.0, k1 = 0.0;
                // (the atoms' parameters first, all 2 MS loads in flight: inside the guard below each load was waited for in turn)
                float icv[MS], kpv[MS];
#pragma unroll
                for (int s = 0; s < MS; s++) { const int j = V.idx[s] < a.n_wm ? V.idx[s] : 0; icv[s] = a.icvf[j]; kpv[s] = a.kappa[j]; }
#pragma unroll
                for (int s = 0; s < MS; s++) {
                    if (s < V.np && V.idx[s] < a.n_wm) {
                        const float ic = icv[s];
                        const double t = V.x[s] / sum_atoms / sum_wm;
                        f1 += (double)ic * t;
                        f2 += (double)((float)(1.0 - (double)ic)) * t;
                        k1 += (double)kpv[s] * t;
                    }
                }
                const double ndi = f1 / (f1 + f2 + 1e-16);
                const double odi = odi_from_kappa(k1);
                const double fwf = xi / sum_atoms;
                double *e = a.est + (size_t)vox * a.n_maps;
                e[0] = ndi; e[1] = odi; e[2] = fwf;
                if (a.dot_atom >= 0) e[3] = xd / sum_atoms;
                if (a.rmse) a.rmse[vox] = sqrt(rho2 / (double)a.nS);
                if (a.nrmse) a.nrmse[vox] = (yy > 1e-16) ? sqrt(rho2 / yy) : 0.0;
                if (a.mod) { const double tf = 1.0 - fwf; a.mod[(size_t)vox * 2] = ndi * tf; a.mod[(size_t)vox * 2 + 1] = odi * tf; }
            }
            if (a.xdbg) {
                double *dst = a.xdbg + ((size_t)vox * 3 + (STAGE == 1 ? 0 : 2)) * n_atoms;
                for (int j = 0; j < n_atoms; j++) dst[j] = 0.0;
#pragma unroll
                for (int s = 0; s < MS; s++) if (s < V.np) dst[V.idx[s]] = V.x[s];
            }
        }
        if (REPAIR && rep < AMX_REPAIR_ROUNDS) {
            // who gets a second look: a well-conditioned block with a non-positive coefficient (those atoms leave), or with every
            // coefficient positive and a positive dual value outside (that atom enters)
            const bool base = live && valid && okv && !cert && piv && !ill && (yy <= 1.79769313486231570e308) && V.np > 0;
            const bool drop = base && !feas;
            const bool add = base && feas && viol && tmax >= 0 && V.np < MS;
            if (__ballot(drop || add) == 0ull) break;
            if (drop) {
                int nidx[MS], nn = 0;
#pragma unroll
                for (int d = 0; d < MS; d++) nidx[d] = 0;
#pragma unroll
                for (int s2 = 0; s2 < MS; s2++) {
                    const bool keep = s2 < V.np && V.x[s2] > 0.0;
#pragma unroll
                    for (int d = 0; d < MS; d++) nidx[d] = (keep && nn == d) ? V.idx[s2] : nidx[d];
                    nn += keep ? 1 : 0;
                }
#pragma unroll
                for (int d = 0; d < MS; d++) V.idx[d] = nidx[d];
                V.np = nn;
            }
            if (add) {
#pragma unroll
                for (int d = 0; d < MS; d++) V.idx[d] = (d == V.np) ? tmax : V.idx[d];
                V.np += 1;
            }
            live = drop || add;
            if (!live) V.np = 0;
            okv = live;
        }
        GC_PH(4);
        }                                         // rep
        if (valid) a.done[pos] = cert_any ? 1 : (easy_any ? 2 : 0);
        {
            // the voxels left to the wavefront-per-voxel kernel, compacted per chunk (that kernel then shares out real work only)
            const unsigned long long rm = __ballot(valid && !cert_any);
            if (rm != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&a.rcount[cid], __builtin_popcountll(rm));
                base = __builtin_amdgcn_readfirstlane(base);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(rm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rm, 0u));
                if (valid && !cert_any) a.rlist[ck.start + base + rank] = pos;
            }
        }
    }
    __syncthreads();                          // every wavefront is through with this chunk's tables
    }
#ifdef AMX_STATS
    if (RESCUE) return;
    if (a.stats && lane == 0) for (int q5 = 0; q5 < 5; q5++) atomicAdd(&a.stats[(STAGE == 1 ? 36 : 35) + q5], (int)(gph[q5] >> 10));
#endif
}

// ================================================================== Gram-space certificate of the LASSO seeds, one voxel per lane
// Same idea as k_nnls_gcert for  min 1/2 ||y2 - A2 x||^2 + lambda1 sum(x) + lambda2/2 ||x||^2, x >= 0:  with c2 = A2'y2
// (k_noddi_gemm<true>) and the stage-2 Gram matrix,  H_PP x_P = c2_P - lambda1,  H = S G2 S + lambda2 I  (S = column scales),
// dual value of atom t: g_t = c2_t - (S G2 S)_tP x_P - lambda1.  The ridge bounds cond(H), so no pivot guard is needed
// -- this is the arithmetic GramSolver::certify_seed performs too.  A lane holds the factor of up to 12 passive atoms
// (87 % of the voxels; the rest, and everything refused, goes to k_noddi<4> through the left-over lists).
// The passive system of one voxel in as few registers as it takes (round 5): ONE packed triangle -- the factor's off-diagonal entries
// with the RECIPROCAL pivots on its diagonal --, ONE vector that is the right-hand side going in and the solution coming out, the
// atoms.  SeedLane (T, dinv, c, x, idx + the caller's z and column scales) asked 18 atoms for ~520 registers of a lane's 512:
// k_lasso_gcert<18, wide> spilled 129 of them to scratch, the 11-atom pass 33.  x'c of the solution falls out of the forward
// substitution (x'c = x'Hx = ||L'x||^2 = ||L^-1 c||^2), so c need not survive the solve.
template <int MS>
struct LeanLane {
    static constexpr int NT = MS * (MS + 1) / 2;
    double T[NT], c[MS];
    int idx[MS], np;
    // in-place Cholesky of the matrix in T (slots >= np: zero rows); false: a pivot of a live slot is not positive
    __device__ __forceinline__ bool factor()
    {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double dj = T[stri<MS>(j, j)];
            const double hjj = dj;
#pragma unroll
            for (int m = 0; m < j; m++) dj -= T[stri<MS>(j, m)] * T[stri<MS>(j, m)];
            const bool good = dj > 1e-15 * hjj;
            ok = ok && (j >= np || good);
            const double di = (j < np && good) ? inv_sqrt(dj) : 0.0;
            T[stri<MS>(j, j)] = di;
#pragma unroll
            for (int i = j + 1; i < MS; i++) {
                double v = T[stri<MS>(i, j)];
#pragma unroll
                for (int m = 0; m < j; m++) v -= T[stri<MS>(i, m)] * T[stri<MS>(j, m)];
                T[stri<MS>(i, j)] = v * di;
            }
        }
        return ok;
    }
    // c <- (L L')^-1 c; returns c_in' x = ||L^-1 c_in||^2
    __device__ __forceinline__ double solve()
    {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double f = c[j];
#pragma unroll
            for (int m = 0; m < j; m++) f -= T[stri<MS>(j, m)] * c[m];
            c[j] = f * T[stri<MS>(j, j)];
            q += c[j] * c[j];
        }
#pragma unroll
        for (int j = MS - 1; j >= 0; j--) {
            double f = c[j];
#pragma unroll
            for (int m = j + 1; m < MS; m++) f -= T[stri<MS>(m, j)] * c[m];
            c[j] = f * T[stri<MS>(j, j)];
        }
        return q;
    }
};

#ifndef AMX_GCERT2_MAX
#define AMX_GCERT2_MAX 11      // (12: 37 spilled registers in the first pass, LASSO group 3.44 ms; 11: none, 3.37 ms; 10: 3.48 ms)
#endif
constexpr int kGcert2Max = AMX_GCERT2_MAX;
#ifndef AMX_GCERT2_WIDE
#define AMX_GCERT2_WIDE 18
#endif
constexpr int kGcert2Wide = AMX_GCERT2_WIDE;   // second pass (k_lasso_gcert<.., true>)
#ifndef AMX_GCERT2_WIDE3
#define AMX_GCERT2_WIDE3 24
#endif
constexpr int kGcert2Wide3 = AMX_GCERT2_WIDE3; // third pass over what the second left: a RUN-TIME choice (amx_launch_noddi_gcert2: shapes whose left-over kernel reads its tile from L2).  Round 3, 99 volumes, 1 M voxels, fit ms:
                                               // 12 / 16: 10.46; 12 / 18: 10.33 (121 spilled registers, but the left-over kernel sees 1.0 % instead of 2.3 %
                                               // of the voxels); 12 / 19: 10.35; 12 / 20: 10.42; 12 / 16 / 18: 10.34; 12 / 16 / 20: 10.42
struct Gcert2Args {
    const int *perm;
    const Chunk *schunks;
    const int *n_schunks;
    const unsigned long long *seeds2;  // [n][4], bucket order
    unsigned long long *cand8;         // = seeds2: a settled voxel's entry is overwritten with stage 3's candidate byte list (see below)
    const double *Cb;                  // [n_blocks][rows][64] of k_noddi_gemm<false>: c2, y2~, ||y2||^2 of the unclipped voxels derive from it
    const double *Cb2;                 // compact table of k_noddi_gemm<true>: the clipped voxels, exactly
    const int *cslot;                  // [n] bucket order: place in the chunk's compact list, -1 = not clipped
    const double *u2iso;               // [ndirs][12] U2'iso
    int rows, aux0;
    const double *gram;                // [ndirs][n_atoms][ldG] stage-2 rows
    const double *colscale;            // [n_atoms]
    int ldG, n_atoms, n_wm, iso_atom, dot_atom;
    const double *Sb;                  // [ndirs][n_wm][12]
    const double *kappa0;              // [ndirs]
    double lam1, lam2;
    unsigned long long *supp;          // out: [n_vox][4]
    const double *xiso;                // [n_vox][2] (AMX_F_DEBUG_X only)
    unsigned char *done;
    int *gcount; int n_gcount;         // first pass: chunk counters (BlockFeed), helpers' counter at [n_gcount]
    int *rlist, *rcount;
    const int *rlist_in, *rcount_in;   // WIDE pass: the left-over lists of the first pass
    double *xdbg;
    int *stats;
    int min_items;                     // third pass: a chunk with fewer CANDIDATES (below) is handed on as it is (round 6: the pass decides per chunk)
    int *qcount;                       // second pass, out: per chunk, the voxels it leaves whose complete seed holds MS + 1 .. q_hi atoms (what a third pass could settle)
    const int *qcount_in;              // third pass, in: that count
    int q_hi;
};

// WIDE = false: every voxel of the chunk, supports of up to 11 atoms, two wavefronts per SIMD.  WIDE = true: second pass over the
// left-over lists of the first for the supports of 12 .. 18 atoms (another 22 % of the voxels at the default lambdas), one
// wavefront per SIMD -- the triangle lives in the whole register file (18: 121 spilled registers); what it cannot settle goes on to k_noddi<4>.
#ifndef AMX_GCERT2_OCC
#define AMX_GCERT2_OCC 2
#endif
template <int MS, bool WIDE, int LOW = kGcert2Max>
__global__ void __launch_bounds__(256, WIDE ? 1 : AMX_GCERT2_OCC) k_lasso_gcert(const Gcert2Args a)
{
    constexpr int KD = kSeedKD, KS = KD / 4, MT = 9, LD = kSeedLd, RBW = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c2[];
    double *Sl = reinterpret_cast<double *>(smem_c2);             // [n_wm][LD]
    const int n_wm = a.n_wm;
    double *scl = Sl + (size_t)n_wm * LD + 2;                      // [n_wm]
    double *giso = scl + ((n_wm + 1) & ~1);                        // [n_wm] G_dwi[j][iso]: what x_iso takes out of a_j,dwi'y
    double *Aop = giso + ((n_wm + 1) & ~1);                        // [MT][KS][64]
    double *Rb = Aop + MT * KS * 64 + (threadIdx.x >> 6) * (64 * RBW);
    const int n_sch = *a.n_schunks;
    const int own = xcd_chunk((int)blockIdx.x, n_sch);
    if (own < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
#ifdef AMX_STATS
    long long gph[5] = {0, 0, 0, 0, 0}, gpt = (long long)__builtin_readcyclecounter();
#endif
    // first pass: the workgroup's own chunk, then the largest chunks still open (BlockFeed); the wide pass walks its own chunk's list only
    for (int round = 0; round < (WIDE ? 1 : 256); round++) {
    const int cid = WIDE ? own : block_next_chunk(round, own, a.schunks, n_sch, a.gcount, a.gcount + a.n_gcount, reinterpret_cast<int *>(Sl + (size_t)n_wm * LD), 64 * nw);
    if (cid < 0) break;
    const Chunk ck = a.schunks[cid];
    const int n_items = WIDE ? a.rcount_in[cid] : ck.count;
    if (WIDE && n_items == 0) return;
    if (WIDE && a.qcount_in != nullptr && a.qcount_in[cid] < a.min_items) {
        // not worth this chunk's tables and a 256-register wavefront's time: the list goes on to the next consumer unchanged.  The third pass
        // pays where a chunk still holds a block's worth of supports it can hold, 19 .. 24 atoms (counted by the second pass) -- a 105-volume
        // protocol: 80 per chunk of a 1 M-voxel call, 8.58 -> 8.0 ms -- and costs 0.3 - 0.4 ms where it does not: a dozen per chunk on the
        // default protocol, and at 150 volumes, where most of what is left holds MORE than 24 atoms (profiles/r06_protocols_ab.txt)
        for (int e = threadIdx.x; e < n_items; e += blockDim.x) a.rlist[ck.start + e] = a.rlist_in[ck.start + e];
        if (threadIdx.x == 0) atomicAdd(&a.rcount[cid], n_items);
        return;
    }
    const double *__restrict__ Sg = a.Sb + (size_t)ck.dir * n_wm * KD;
    const double *__restrict__ Gd = a.gram + (size_t)ck.dir * a.n_atoms * a.ldG;
    stage_rows<KD, LD>(Sl, Sg, n_wm, KD);
    for (int e = threadIdx.x; e < n_wm; e += blockDim.x) { scl[e] = a.colscale[e]; giso[e] = Gd[(size_t)e * a.ldG + a.iso_atom]; }
    stage_operand<KS, MT>(Aop, Sg, n_wm, KD);
    __syncthreads();
    const double kap = a.kappa0[ck.dir], lam1 = a.lam1, lam2 = a.lam2;
    const double gii = Gd[(size_t)a.iso_atom * a.ldG + a.iso_atom];      // ||iso_dwi||^2
    const double *__restrict__ u2 = a.u2iso + (size_t)ck.dir * kSeedKD;
    const int n_blocks = (n_items + 63) >> 6;
    BlockFeed<64> bf;
    if (!WIDE) bf.start(a.gcount + cid, lane);
    for (int bls = wave; ; bls += nw) {
        int bl = bls;
        if (WIDE) { if (bl >= n_blocks) break; }
        else { bl = bf.next(a.gcount + cid, ck.count, lane); if (bl < 0) break; }
        const int k = 64 * bl + lane;
        const bool valid = k < n_items;
        const int pos = WIDE ? a.rlist_in[ck.start + (valid ? k : n_items - 1)] : ck.start + (valid ? k : n_items - 1);
        // The voxel's column: in the table of stage 1 (place in the chunk) -- then c2_j = s_j (c_j - sum_b0 y - x_iso G_dwi[j, iso]) etc.
        // (see k_noddi_gemm) -- or, clipped, in the compact table of the exact pass, whose rows ARE c2, y2~, ||y2||^2.  One pointer
        // per lane, one formula: value = mul_j (load - sub - xq G_dwi[j, iso]) with sub = xq = 0, mul = 1 for the clipped voxels.
        const int cs = a.cslot[pos];
        const bool clip = cs >= 0;
        const int kk = clip ? cs : pos - ck.start;
        const double *Crow = (clip ? a.Cb2 : a.Cb) + (size_t)(ck.pad + (kk >> 6)) * a.rows * 64 + (kk & 63);
        const int vox = a.perm[pos];
        const double xi = a.xiso[(size_t)vox * 2];
        const double xq = clip ? 0.0 : xi;
        const double sub = clip ? 0.0 : Crow[(size_t)(a.aux0 + kAuxB0) * 64];
        const unsigned long long *sd = a.seeds2 + (size_t)pos * 4;
        const unsigned long long flag = sd[3];
        int cnt;
        bool okv;
        LeanLane<MS> V;
        unsigned long long wl[4] = {0ull, 0ull, 0ull, 0ull};    // the support's atoms as a byte list (stage 3's candidates: written out below)
        {
            // slots = set bits in ascending order
            const unsigned long long P[3] = {sd[0], sd[1], sd[2]};
            cnt = __builtin_popcountll(P[0]) + __builtin_popcountll(P[1]) + __builtin_popcountll(P[2]);
            okv = valid && flag == 0ull && cnt <= MS && (!WIDE || cnt > LOW);
            unsigned long long rem[3] = {okv ? P[0] : 0ull, okv ? P[1] : 0ull, okv ? P[2] : 0ull};
            int n0 = 0;
#pragma unroll
            for (int s = 0; s < MS; s++) {
                int wq = -1;
#pragma unroll
                for (int qq = 2; qq >= 0; qq--) wq = (rem[qq] != 0ull) ? qq : wq;
                unsigned long long word = 0ull;
#pragma unroll
                for (int qq = 0; qq < 3; qq++) word = (wq == qq) ? rem[qq] : word;
                const int t = (wq >= 0) ? wq * 64 + __builtin_ctzll(word) : 0;
#pragma unroll
                for (int qq = 0; qq < 3; qq++) rem[qq] = (wq == qq) ? (rem[qq] & (rem[qq] - 1ull)) : rem[qq];
                if (wq >= 0 && t >= n_wm) okv = false;
                V.idx[s] = (wq >= 0 && t < n_wm) ? t : 0;
                wl[s >> 3] |= (unsigned long long)((wq >= 0 && t < n_wm) ? t : 0) << (8 * (s & 7));
                n0 += (wq >= 0) ? 1 : 0;
            }
            V.np = okv ? n0 : 0;
        }
        GC_PH(0);
        // (the column scales are read from LDS where they are needed -- scl[idx] -- instead of living in MS registers through the
        //  factorisation: the kernel's register peak is T + c + z + idx there)
        {
            // Every entry lands in the register it will live in, under the lane's own mask (s < np), and is USED outside that guard.
            // As `(s < np) ? sc sc Gd[..] + .. : 0` each load sat in a branch of its own with a wait behind it: 77 (wide pass: 189)
            // memory round trips one after the other per block of 64 voxels -- the whole kernel (SQ_WAIT_ANY 76 - 81 % of its cycles).
            const unsigned ldg = (unsigned)a.ldG;
            unsigned rowo[MS];
#pragma unroll
            for (int s = 0; s < MS; s++) rowo[s] = (unsigned)V.idx[s] * ldg;
#pragma unroll
            for (int s = 0; s < MS; s++) {
#pragma unroll
                for (int t = 0; t <= s; t++) V.T[stri<MS>(s, t)] = 0.0;
                V.c[s] = 0.0;
                if (s < V.np) {                     // (loads under the lane's mask, used outside it: see k_nnls_gcert's exact dual values)
#pragma unroll
                    for (int t = 0; t <= s; t++) V.T[stri<MS>(s, t)] = Gd[rowo[s] + (unsigned)V.idx[t]];
                    V.c[s] = Crow[(size_t)V.idx[s] * 64];
                }
                if (WIDE) __builtin_amdgcn_sched_barrier(0);      // row by row: 171 addresses computed ahead of their loads were 342 registers of their own
            }
#pragma unroll
            for (int s = 0; s < MS; s++) {
                const double ss = scl[V.idx[s]];
#pragma unroll
                for (int t = 0; t <= s; t++)
                    V.T[stri<MS>(s, t)] = (s < V.np) ? ss * scl[V.idx[t]] * V.T[stri<MS>(s, t)] + ((s == t) ? lam2 : 0.0) : 0.0;
                V.c[s] = (s < V.np) ? (clip ? 1.0 : ss) * (V.c[s] - sub - xq * giso[V.idx[s]]) - lam1 : 0.0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        double yy = Crow[(size_t)(a.aux0 + kAuxYY) * 64];
        if (!clip) {
            // ||y2||^2 = (||y||^2 - sum_b0 y^2) - 2 x_iso (c_iso - sum_b0 y) + x_iso^2 ||iso_dwi||^2
            const double ciso = Crow[(size_t)a.iso_atom * 64], yb = Crow[(size_t)(a.aux0 + kAuxYB) * 64];
            yy = (yy - yb) - 2.0 * xi * (ciso - sub) + xi * xi * gii;
            yy = yy > 0.0 ? yy : (yy <= 0.0 ? 0.0 : yy);                 // (rounding below zero; NaN stays NaN)
        }
        const bool piv = V.factor();
        // x = H_PP^-1 (c2_P - lambda1) in place; ||r||^2 = ||y2||^2 - [x'(c2 - lambda1) + 2 lambda1 sum x + lambda2 x'x]
        // (x'(c2 - lambda1) comes with the forward substitution: LeanLane::solve)
        double rho2 = yy - V.solve();
        bool feas = true;
#pragma unroll
        for (int s = 0; s < MS; s++) {
            const double zs = V.c[s];
            if (s < V.np && !(zs > 0.0)) feas = false;
            rho2 -= 2.0 * lam1 * zs + lam2 * zs * zs;
        }
        rho2 = rho2 > 0.0 ? rho2 : 0.0;
        const bool good = okv && piv && feas && (yy <= 1.79769313486231570e308);
        GC_PH(1);
        unsigned long long ex[3] = {0ull, 0ull, 0ull};
        // (the support bits are read AGAIN here -- 24 bytes the L2 still holds -- instead of living in six registers through the factorisation,
        //  and the candidates are made from them only now: twelve registers less at the kernel's peak -- first pass 42 -> 30 spilled registers.
        //  Round 6 also tried the lane's atom table as packed bytes, 18 registers less on paper: the compiler spilled MORE, 145 -> 172)
        const unsigned long long Pq[3] = {sd[0], sd[1], sd[2]};
        unsigned long long cand[3];
#pragma unroll
        for (int w3 = 0; w3 < 3; w3++) {
            const int c = n_wm - 64 * w3;
            const unsigned long long all = c >= 64 ? ~0ull : (c > 0 ? ((1ull << c) - 1ull) : 0ull);
            cand[w3] = all & ~Pq[w3];
        }
        {
            double rt[KD];
            const double *Cu = Crow + (size_t)(a.aux0 + (clip ? kAuxU : kAuxU2)) * 64;
#pragma unroll
            for (int d = 0; d < KD; d++) rt[d] = Cu[(size_t)d * 64] - xq * u2[d];
#pragma unroll
            for (int s = 0; s < MS; s++) {
                const double *col = Sl + V.idx[s] * LD;
                double cv[KD];
                seed_col<KD>(col, cv);
#pragma unroll
                for (int d = 0; d < KD; d++) rt[d] -= V.c[s] * cv[d];
            }
            // compressed dual value s2_t'r~ - lambda1 > -kappa ||r||  <=>  s2_t'r~ > lambda1 - kappa ||r||
            seed_flags_mfma<KS, MT>(Aop, Rb, lane, rt, good, good ? lam1 - 1.0625 * kap * sqrt(rho2) - 1e-12 : __builtin_huge_val(), cand, ex);
        }
        GC_PH(2);
        bool viol = false;
        int n_ex = 0;
        {
            double xs[MS];                                 // the coefficients with their columns' scales (the factor is dead by now)
#pragma unroll
            for (int s = 0; s < MS; s++) xs[s] = scl[V.idx[s]] * V.c[s];
            unsigned long long rem[3] = {good ? ex[0] : 0ull, good ? ex[1] : 0ull, good ? ex[2] : 0ull};
            for (int it = 0; it < 192; it++) {
                int wq = -1;
#pragma unroll
                for (int qq = 2; qq >= 0; qq--) wq = (rem[qq] != 0ull) ? qq : wq;
                if (__ballot(wq >= 0) == 0ull) break;
                unsigned long long word = 0ull;
#pragma unroll
                for (int qq = 0; qq < 3; qq++) word = (wq == qq) ? rem[qq] : word;
                const int t = (wq >= 0) ? wq * 64 + __builtin_ctzll(word) : 0;
#pragma unroll
                for (int qq = 0; qq < 3; qq++) rem[qq] = (wq == qq) ? (rem[qq] & (rem[qq] - 1ull)) : rem[qq];
                const bool on = wq >= 0 && t < n_wm;
                const int tc = on ? t : 0;
                const double st = on ? scl[tc] : 0.0;
                double ct = 0.0, gv[MS];                               // (loads under the lane's mask, all in flight together, used outside it)
#pragma unroll
                for (int s = 0; s < MS; s++) gv[s] = 0.0;
                if (on) {
                    ct = Crow[(size_t)tc * 64];
                    const double *gt = Gd + (size_t)tc * a.ldG;
#pragma unroll
                    for (int s = 0; s < MS; s++) if (s < V.np) gv[s] = gt[V.idx[s]];
                }
                double g = on ? (clip ? 1.0 : st) * (ct - sub - xq * giso[tc]) - lam1 : -1.0;
#pragma unroll
                for (int s = 0; s < MS; s++) { const double xv = (s < V.np && on) ? xs[s] : 0.0; g -= st * gv[s] * xv; }    // (the expression the guarded loop had: same contraction, same bits)
                if (on && !(g < -1e-10)) viol = true;
                n_ex += on ? 1 : 0;
            }
        }
        GC_PH(3);
        const bool cert = good && !viol;
        if (valid) a.done[pos] = cert ? 1 : 0;
        if (WIDE && a.qcount != nullptr) {
            const unsigned long long qm = __ballot(valid && !cert && flag == 0ull && cnt > MS && cnt <= a.q_hi);
            if (qm != 0ull && lane == 0) atomicAdd(&a.qcount[cid], __builtin_popcountll(qm));
        }
        {
            const unsigned long long rm = __ballot(valid && !cert);
            if (rm != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&a.rcount[cid], __builtin_popcountll(rm));
                base = __builtin_amdgcn_readfirstlane(base);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(rm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rm, 0u));
                if (valid && !cert) a.rlist[ck.start + base + rank] = pos;
            }
        }
#ifdef AMX_STATS
        if (a.stats) {
            const int nc = __builtin_popcountll(__ballot(cert)), nv = __builtin_popcountll(__ballot(valid));
            const int nbig = __builtin_popcountll(__ballot(valid && flag == 0ull && cnt > MS)), nfe = __builtin_popcountll(__ballot(valid && okv && piv && !feas)), nvi = __builtin_popcountll(__ballot(valid && good && viol));
            if (lane == 0) { atomicAdd(&a.stats[0], nv); atomicAdd(&a.stats[1], nc); atomicAdd(&a.stats[2], nbig); atomicAdd(&a.stats[3], nfe); atomicAdd(&a.stats[4], nvi); }
            int ne = n_ex;
            for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o);
            if (lane == 0) atomicAdd(&a.stats[5], ne);
        }
#endif
        if (cert) {
            unsigned long long *sp = a.supp + (size_t)vox * 4;
            sp[0] = Pq[0]; sp[1] = Pq[1]; sp[2] = Pq[2]; sp[3] = 0ull;
            if (a.cand8 != nullptr) {
                // What the stage-3 seed solver makes of these bits when it takes the voxel -- the admissible atoms (support, dot, iso) as a
                // byte list in ascending order -- is in this lane's registers already: it goes where the voxel's seed was (nobody reads
                // the seed of a settled voxel again).  The seed solver then takes a voxel with ONE 32-byte load by position instead
                // of position -> voxel -> bits and a bit-by-bit conversion that every lane of the wavefront walked in every trip (a
                // third of that kernel).
                static_assert(MS + 2 <= 31, "the list is 31 bytes and its length");
                const int n_extra = a.dot_atom >= 0 ? 2 : 1;
                const int e0 = a.dot_atom >= 0 ? a.dot_atom : a.iso_atom, e1 = a.iso_atom;
                unsigned long long w[4] = {wl[0], wl[1], wl[2], wl[3]};        // (the support's atoms were packed while the bits were decoded)
#pragma unroll
                for (int s = 1; s < MS + 2; s++) {
                    const int b = (s == V.np ? e0 : ((s == V.np + 1 && n_extra == 2) ? e1 : 0));
                    w[s >> 3] |= (unsigned long long)(b & 0xff) << (8 * (s & 7));
                }
                if (V.np == 0) w[0] |= (unsigned long long)(e0 & 0xff);
                w[3] = (w[3] & 0x00ffffffffffffffull) | ((unsigned long long)(V.np + n_extra) << 56);
                unsigned long long *cl = a.cand8 + (size_t)pos * 4;
                cl[0] = w[0]; cl[1] = w[1]; cl[2] = w[2]; cl[3] = w[3];
            }
            if (a.xdbg) {
                double *dst = a.xdbg + ((size_t)vox * 3 + 1) * a.n_atoms;
                for (int j = 0; j < a.n_atoms; j++) dst[j] = 0.0;
#pragma unroll
                for (int s = 0; s < MS; s++) if (s < V.np) dst[V.idx[s]] = V.c[s];
                dst[a.iso_atom] = xi;
                if (a.dot_atom >= 0) dst[a.dot_atom] = a.xiso[(size_t)vox * 2 + 1];
            }
        }
        GC_PH(4);
    }
    __syncthreads();                          // every wavefront is through with this chunk's tables
    }
#ifdef AMX_STATS
    if (!WIDE && a.stats && lane == 0) for (int q5 = 0; q5 < 5; q5++) atomicAdd(&a.stats[34 + q5], (int)(gph[q5] >> 10));
#endif
}

// ================================================================== LASSO stage (models.pyx:914-926): seeds in Woodbury form
// The stage-2 problem  min 1/2 ||y2 - A2 x||^2 + lambda1 sum(x) + lambda2/2 ||x||^2, x >= 0  (column-normalised atoms, DWI rows,
// clipped signal) has a ridge, so on a passive set P
//     x_P = (c_P - S_P' w) / lambda2,   (lambda2 I_k + S_P S_P') w = S_P c_P,   c_j = s_j' y~ - lambda1,
// in the rank-k compressed space (k = 8: the support of the compressed problem equals the full problem's in tools/lab/s2_lab.py
// on every voxel tried): the passive system is k x k WHATEVER the size of P.  With t_j = s_j' (y~ - w) - lambda1:
// passive atoms have x_j = t_j / lambda2, all others have the dual value t_j -- one product S'(y~ - w) serves both.
// A lane therefore carries no per-atom state at all: the passive set as a bit mask, the Cholesky factor of
// M = lambda2 I + S_P S_P' (8 x 8) and g = S_P c_P.  One trip = solve, scan (fp64 MFMA for all 64 voxels of the wavefront),
// then ONE rank-one change of the factor: the most negative passive atom leaves (down-date), or, if none is negative, the
// atom with the largest dual value enters (update).  Greedy, no step back -- it is a proposal, certified afterwards by
// GramSolver::certify_seed in the full problem.
constexpr int kSeed2KD = 8;       // components the LASSO seed solver works with (the first 8 of the 12 stored: the pivoted basis is nested)
constexpr int kSeed2Ld = kSeedKD; // stride of U2 / S2 / y2~ rows
struct Seed2Args {
    const double *y;              // [n_vox][nS]
    const float *y32;             // float32 signals instead (or null)
    const int *perm;
    const Chunk *chunks;
    const int *n_chunks;
    const Chunk *schunks;
    const int *n_schunks;
    const float *tiles;           // dictionary tiles (the iso column of the orientation)
    int tile_stride, ldA;
    const unsigned char *rowdwi;
    const double *xiso;           // [n_vox][2] stage-1 x_iso, x_dot
    const double *Ub, *Sb;        // [ndirs][nS][8], [ndirs][n_wm][8]
    double *ytil;                 // [n_vox][8], bucket order
    unsigned long long *seeds;    // [n_vox][4], bucket order: passive-set bits; word 3 = all ones: no seed
    int nS, n_wm, iso_atom, is_exvivo;
    double lam1, lam2;
    int *gcount;                  // [max_schunks + 1]: see SeedArgs
    int n_gcount;
    int *stats;
    double *trace;                // SEED2_TRACE: per-trip records of the voxel at bucket position 0
    int trip_cap;                 // see SeedArgs
    int max_atoms;                // a voxel whose passive set reaches this many atoms and wants more is given up (its set goes on as an incomplete seed)
};

template <int NR>
__global__ void __launch_bounds__(1024) k_noddi_project2(const Seed2Args a)
{
    constexpr int KD = kSeed2Ld;
    const int cid = xcd_chunk((int)blockIdx.x, *a.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.chunks[cid];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int nS = a.nS;
    double ur[NR][KD], isov[NR];
    bool rowok[NR];
    const double *U = a.Ub + (size_t)ck.dir * nS * KD;
    const float *tile = a.tiles + (size_t)ck.dir * a.tile_stride;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int i = lane + kWave * rr;
        rowok[rr] = (i < nS) && a.rowdwi[i];
        isov[rr] = (i < nS) ? (double)tile[i * a.ldA + a.iso_atom] : 0.0;
#pragma unroll
        for (int d = 0; d < KD; d++) ur[rr][d] = (i < nS) ? U[i * KD + d] : 0.0;
    }
    for (int k = wave; k < ck.count; k += nw) {
        const int pos = ck.start + k;
        const int vox = a.perm[pos];
        const size_t yo = (size_t)vox * nS;
        const double xi = a.xiso[(size_t)vox * 2], xd = a.xiso[(size_t)vox * 2 + 1];
        double yr[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            double t = 0.0;
            if (rowok[rr]) {                      // models.pyx:917-925
                t = (a.y32 ? (double)a.y32[yo + i] : a.y[yo + i]) - xi * isov[rr];
                if (a.is_exvivo) t -= xd * 1.0;
                if (t < 0.0) t = 0.0;
            }
            yr[rr] = t;
        }
        double out = 0.0;
#pragma unroll
        for (int b = 0; b < KD; b += 4) {
            double p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                p[u] = 0.0;
#pragma unroll
                for (int rr = 0; rr < NR; rr++) p[u] += ur[rr][b + u] * yr[rr];
            }
            wave_sum4(p, lane);
#pragma unroll
            for (int u = 0; u < 4; u++) out = (lane == b + u) ? p[u] : out;
        }
        if (lane < KD) a.ytil[(size_t)pos * KD + lane] = out;
    }
}

#ifndef AMX_SEED2_OCC
#define AMX_SEED2_OCC 1
#endif
template <bool OCC2 = false>
__global__ void __launch_bounds__(256, OCC2 ? 2 : AMX_SEED2_OCC) k_lasso_seed(const Seed2Args a)
{
    constexpr int KD = kSeed2KD, KS = KD / 4, MT = 9, KDP = KD + 1, NT = KD * (KD + 1) / 2, LD = KD + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
    double *Sl = reinterpret_cast<double *>(smem_l);             // [n_wm][LD] for the per-lane gathers
    const int n_wm = a.n_wm;
    unsigned *ticket = reinterpret_cast<unsigned *>(Sl + (size_t)n_wm * LD);
    double *Aop = reinterpret_cast<double *>(ticket + 4);         // [MT][KS][64] in MFMA operand order
    double *Rb = Aop + MT * KS * 64 + (threadIdx.x >> 6) * (64 * KDP + 64 * 3);
    unsigned long long *Pb = reinterpret_cast<unsigned long long *>(Rb + 64 * KDP);      // [64][3] passive-set bits of the lanes' voxels
    const int n_sch = *a.n_schunks;
    const int own = xcd_chunk((int)blockIdx.x, n_sch);
    if (own < 0) return;
    const int lane = threadIdx.x & 63, q = lane >> 4, c16 = lane & 15;
    constexpr int SLD = kSeed2Ld;
#ifdef AMX_STATS
    int st_trips = 0, st_used = 0;
#endif
    // the workgroup's own chunk first, then the largest chunks that still have voxels (SeedFeed)
    for (int round = 0; round < 256; round++) {
    const int cid = seed_next_chunk(round, own, a.schunks, n_sch, a.gcount, a.gcount + a.n_gcount, reinterpret_cast<int *>(ticket), 32 * (int)(blockDim.x >> 6));
    if (cid < 0) break;
    const Chunk ck = a.schunks[cid];
    const double *__restrict__ Sg = a.Sb + (size_t)ck.dir * n_wm * SLD;
    stage_rows<KD, LD>(Sl, Sg, n_wm, SLD);
    stage_operand<KS, MT>(Aop, Sg, n_wm, SLD);
    __syncthreads();
    const double lam1 = a.lam1, lam2 = a.lam2, tol = 1e-9, inf = __builtin_huge_val();
    const double sl2 = sqrt(lam2), isl2 = 1.0 / sl2;
    const int trip_cap = a.trip_cap;

    bool active = false;
    int pos = 0, trips = 0;
    double T[NT], dinv[KD], g[KD];
    unsigned long long P[3] = {0ull, 0ull, 0ull};
#pragma unroll
    for (int e = 0; e < NT; e++) T[e] = 0.0;
#pragma unroll
    for (int d = 0; d < KD; d++) { dinv[d] = 0.0; g[d] = 0.0; }
    SeedFeed feed;
    feed.reset();
    // (one wavefront per SIMD: y~ of the voxel in registers, the next voxel reserved -- and its y~ loading -- one solve ahead, as in
    //  k_nnls_seed<1>)
    constexpr bool PREF2 = !OCC2 && AMX_SEED2_OCC == 1;       // (see k_nnls_seed: 1 M voxels 1.67 -> 1.28 ms at two wavefronts per SIMD)
    double yv[KD], ynext[PREF2 ? KD : 1];
    int next_pos = -1;
    bool have_next = false;
    for (int guard = 0; guard < (1 << 20); ++guard) {
        if (PREF2) {
        if (!active && have_next) {
            pos = next_pos; have_next = false;
            bool finite = true;
#pragma unroll
            for (int d = 0; d < KD; d++) { yv[d] = ynext[PREF2 ? d : 0]; finite = finite && (fabs(yv[d]) <= 1.79769313486231570e308); }
            trips = 0;
            P[0] = 0ull; P[1] = 0ull; P[2] = 0ull;
#pragma unroll
            for (int i = 0; i < KD; i++) {                 // M = lambda2 I
#pragma unroll
                for (int j = 0; j <= i; j++) T[stri<KD>(i, j)] = (i == j) ? sl2 : 0.0;
                dinv[i] = isl2; g[i] = 0.0;
            }
            if (finite) active = true;
            else { a.seeds[(size_t)pos * 4 + 3] = ~0ull; }
        }
        {
            const unsigned long long needm = __ballot(!have_next);
            if (needm != 0ull && feed.pending()) {
                const int k = feed.take(needm, a.gcount + cid, ck.count, lane);
                if (k >= 0) {
                    next_pos = ck.start + k; have_next = true;
                    const double *yp = a.ytil + (size_t)next_pos * SLD;
#pragma unroll
                    for (int d = 0; d < KD; d++) ynext[PREF2 ? d : 0] = yp[d];
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (!feed.pending() && __ballot(have_next) == 0ull) break;
            continue;
        }
        } else {
            // two wavefronts per SIMD: no voxel reserved ahead (its 8 values would cost 16 registers for a whole solve) -- the
            // other wavefront of the SIMD covers the load
            const unsigned long long freem = __ballot(!active);
            if (freem != 0ull && feed.pending()) {
                const int k = feed.take(freem, a.gcount + cid, ck.count, lane);
                if (k >= 0) {
                    pos = ck.start + k;
                    const double *yp = a.ytil + (size_t)pos * SLD;
                    bool finite = true;
#pragma unroll
                    for (int d = 0; d < KD; d++) { yv[d] = yp[d]; finite = finite && (fabs(yv[d]) <= 1.79769313486231570e308); }
                    trips = 0;
                    P[0] = 0ull; P[1] = 0ull; P[2] = 0ull;
#pragma unroll
                    for (int i = 0; i < KD; i++) {                 // M = lambda2 I
#pragma unroll
                        for (int j = 0; j <= i; j++) T[stri<KD>(i, j)] = (i == j) ? sl2 : 0.0;
                        dinv[i] = isl2; g[i] = 0.0;
                    }
                    if (finite) active = true;
                    else { a.seeds[(size_t)pos * 4 + 3] = ~0ull; }
                }
            }
            if (__ballot(active) == 0ull) {
                if (!feed.pending()) break;
                continue;
            }
        }
#ifdef AMX_STATS
        st_trips++; st_used += __builtin_popcountll(__ballot(active));
#endif
        trips++;
        // ------------------------------------------------------------ w = M^-1 g, r = y~ - w
        double r[KD];
        {
            double w[KD];
#pragma unroll
            for (int j = 0; j < KD; j++) {
                double f = g[j];
#pragma unroll
                for (int m = 0; m < j; m++) f -= T[stri<KD>(j, m)] * w[m];
                w[j] = f * dinv[j];
            }
#pragma unroll
            for (int j = KD - 1; j >= 0; j--) {
                double f = w[j];
#pragma unroll
                for (int m = j + 1; m < KD; m++) f -= T[stri<KD>(m, j)] * w[m];
                w[j] = f * dinv[j];
            }
#pragma unroll
            for (int d = 0; d < KD; d++) r[d] = yv[d] - w[d];
        }
        int dj = -1;                  // the most negative passive atom (x_j = t_j / lambda2 <= 0): it leaves
        // ------------------------------------------------------------ dual values of all atoms for the 64 voxels (fp64 MFMA),
        // passive atoms masked out, arg-max in the low mantissa bits (see seed_scan_mfma)
        double best = -inf, best2 = -inf;
        int bj = -1, bj2 = -1;
        {
#pragma unroll
            for (int d = 0; d < KD; d++) Rb[lane * KDP + d] = r[d];
#pragma unroll
            for (int w3 = 0; w3 < 3; w3++) Pb[lane * 3 + w3] = active ? P[w3] : ~0ull;
            double b[4][KS];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) b[nt][ks] = Rb[(16 * nt + c16) * KDP + 4 * ks + q];
            }
            const double ninf = -inf;
            double bv[4] = {ninf, ninf, ninf, ninf};
            // the SAME product holds t_j + lambda1 of the passive atoms: their minimum (kept as the maximum of the negated values, same
            // tag trick) names the atom that leaves -- no second pass over the passive set with per-lane gathers
            double wv[4] = {ninf, ninf, ninf, ninf};
            double b2[4] = {ninf, ninf, ninf, ninf};           // runner-up of bv (two atoms may enter in one trip)
            // (one wavefront per SIMD: unrolled and software-pipelined like seed_scan_mfma -- the products of tile mt + 1 are issued
            //  before the mask / tag / max work on tile mt)
            auto products = [&](int mt, seed_v4d (&acc)[4]) {
                double av[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) av[ks] = Aop[(mt * KS + ks) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[nt] = (seed_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], b[nt][ks], acc[nt], 0, 0, 0);
                }
            };
            seed_v4d cur[4], nxt[PREF2 ? 4 : 1];
            if (PREF2) products(0, cur);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                if (PREF2) { if (mt + 1 < MT) products(mt + 1, reinterpret_cast<seed_v4d (&)[4]>(nxt)); }
                else products(mt, cur);
                // passive bits of the voxels 16 nt + c16 for this tile's atoms (row-shifted), read back per tile instead of
                // living in 24 registers for the whole scan
                unsigned long long pw[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) pw[nt] = Pb[(16 * nt + c16) * 3 + ((16 * mt) >> 6)] >> q;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int bit = 16 * mt + 4 * rr;        // position of this atom in the row-shifted mask
                        const bool pas = (pw[nt] >> (bit & 63)) & 1ull;
                        const double v = cur[nt][rr];
                        const unsigned lo = ((unsigned)__double2loint(v) & 0xffffff00u) | (unsigned)(mt * 4 + rr);
                        const int hi = pas ? (int)0xffe00000 : __double2hiint(v);     // passive: -9e307 (finite whatever the low word is; 0xfff... would be a NaN)
                        const double val = __hiloint2double(hi, (int)lo);
                        b2[nt] = seed_max(b2[nt], seed_min(bv[nt], val));
                        bv[nt] = seed_max(bv[nt], val);
                        const int hn = pas ? (__double2hiint(v) ^ (int)0x80000000) : (int)0xffe00000;
                        wv[nt] = seed_max(wv[nt], __hiloint2double(hn, (int)lo));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (PREF2) {
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) cur[nt] = nxt[PREF2 ? nt : 0];
                }
            }
            double mine = ninf;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const double t = __hiloint2double(__double2hiint(bv[nt]), (int)((unsigned)__double2loint(bv[nt]) | (unsigned)(q << 6)));
                const double m = rows_allmax(t);
                mine = (q == nt) ? m : mine;
            }
            const unsigned code = (unsigned)__double2loint(mine) & 0xffu;
            best = mine - lam1;
            bj = 16 * (int)((code >> 2) & 15u) + 4 * (int)(code & 3u) + (int)(code >> 6);
            // runner-up of the voxel: the row that holds the winner contributes its own runner-up, the other rows their best
            // (the tags make all values distinct, so equality identifies the winner's row)
            double second = ninf;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const double t1 = __hiloint2double(__double2hiint(bv[nt]), (int)((unsigned)__double2loint(bv[nt]) | (unsigned)(q << 6)));
                const double t2 = __hiloint2double(__double2hiint(b2[nt]), (int)((unsigned)__double2loint(b2[nt]) | (unsigned)(q << 6)));
                const double mall = rows_allmax(t1);
                const double m = rows_allmax((t1 == mall) ? t2 : t1);
                second = (q == nt) ? m : second;
            }
            const unsigned codes = (unsigned)__double2loint(second) & 0xffu;
            best2 = second - lam1;
            bj2 = 16 * (int)((codes >> 2) & 15u) + 4 * (int)(codes & 3u) + (int)(codes >> 6);
            double mine2 = ninf;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const double t = __hiloint2double(__double2hiint(wv[nt]), (int)((unsigned)__double2loint(wv[nt]) | (unsigned)(q << 6)));
                const double m = rows_allmax(t);
                mine2 = (q == nt) ? m : mine2;
            }
            const unsigned code2 = (unsigned)__double2loint(mine2) & 0xffu;
            // mine2 = -(smallest passive s_j'r): t_j = -mine2 - lambda1 <= 0 ?  (no passive atom: mine2 = -9e307)
            if (active && mine2 > -1e300 && -mine2 - lam1 <= 0.0)
                dj = 16 * (int)((code2 >> 2) & 15u) + 4 * (int)(code2 & 3u) + (int)(code2 >> 6);
        }
        // ------------------------------------------------------------ up to TWO rank-one changes of the factor per trip: the most
        // negative passive atom leaves and the best atom enters, or -- nothing to remove -- the two best atoms enter (11.9 -> 8.1 trips
        // per voxel in tools/lab/s2_lab.py's emulation, same final supports: the ridge keeps M positive definite whatever enters)
        bool done = false, noseed = false;
        int jj = -1, jj2 = -1;
        double sigma = 0.0, sigma2 = 0.0;
        if (active) {
            const int cnt = __builtin_popcountll(P[0]) + __builtin_popcountll(P[1]) + __builtin_popcountll(P[2]);
            const bool add1 = best > tol && bj < n_wm, add2 = best2 > tol && bj2 < n_wm;
            const int mx = a.max_atoms;            // (20; 26 where a third certificate pass takes supports of up to 24 atoms)
            if (dj >= 0) { jj = dj; sigma = -1.0; if (add1 && cnt < mx) { jj2 = bj; sigma2 = 1.0; } }
            else if (add1) { jj = bj; sigma = 1.0; if (add2 && cnt < mx - 1) { jj2 = bj2; sigma2 = 1.0; } }
            else done = true;
            if (!done && (trips > trip_cap || (sigma > 0.0 && cnt >= mx))) { done = true; noseed = true; jj = -1; sigma = 0.0; jj2 = -1; sigma2 = 0.0; }
        }
#pragma unroll 1
        for (int ch = 0; ch < 2; ch++) {
            const int jc = ch == 0 ? jj : jj2;
            const double sg = ch == 0 ? sigma : sigma2;
            if (ch == 1 && __ballot(jc >= 0) == 0ull) break;
            // v = s_jc (zero for the lanes without a change: every rotation is then the identity, bit for bit)
            double v[KD];
            const double *col = Sl + (jc >= 0 ? jc : 0) * LD;
            double cj = -lam1;
#pragma unroll
            for (int d = 0; d < KD; d++) { v[d] = (jc >= 0) ? col[d] : 0.0; cj += v[d] * yv[d]; }
#pragma unroll
            for (int d = 0; d < KD; d++) g[d] += sg * cj * v[d];
            if (jc >= 0) P[jc >> 6] ^= 1ull << (jc & 63);
#pragma unroll
            for (int j = 0; j < KD; j++) {
                const double al = T[stri<KD>(j, j)], bl = v[j];
                const bool rot = bl != 0.0;
                const double n2 = al * al + sg * bl * bl;
                const double ri = rot ? ((n2 > 0.0) ? inv_sqrt(n2) : 0.0) : dinv[j];
                const double ss = rot ? bl * dinv[j] : 0.0;
                const double cc = rot ? n2 * ri * dinv[j] : 1.0;       // cos
                const double ci = rot ? al * ri : 1.0;                  // 1 / cos
                T[stri<KD>(j, j)] = rot ? n2 * ri : al;
                dinv[j] = ri;
#pragma unroll
                for (int i = j + 1; i < KD; i++) {
                    const double t = (T[stri<KD>(i, j)] + sg * ss * v[i]) * ci;
                    v[i] = cc * v[i] - ss * t;
                    T[stri<KD>(i, j)] = t;
                }
            }
        }
#ifdef SEED2_TRACE
        if (active && pos == 0 && a.trace) { double *tr = a.trace + 8 * trips; tr[0] = trips; tr[1] = jj; tr[2] = sigma; tr[3] = best; tr[4] = bj; tr[5] = dj; tr[6] = r[0]; tr[7] = g[0]; }
#endif
        if (done) {
            unsigned long long *sd = a.seeds + (size_t)pos * 4;
            sd[0] = P[0]; sd[1] = P[1]; sd[2] = P[2]; sd[3] = noseed ? ~0ull : 0ull;
            active = false;
        }
    }
    __syncthreads();                          // every wavefront is through with this chunk's tables
    }
#ifdef AMX_STATS
    if (a.stats && lane == 0) { atomicAdd(&a.stats[0], st_trips); atomicAdd(&a.stats[1], st_used); }
#endif
}

}  // namespace amx
