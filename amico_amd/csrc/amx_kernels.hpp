// amx_kernels.hpp -- gfx950 kernels of the AMICO fit path (included by amx_api.hip only).
//
//   k_dir_to_lut      lut.pyx:316-356 per voxel + histogram of LUT indices
//   k_plan            exclusive scan of the histogram, chunk list (<= CH voxels of ONE
//                     orientation per workgroup, so the dictionary slice is staged once)
//   k_bucket          counting-sort scatter -> permutation grouped by orientation
//   k_build_lut*      one-off re-layout of KERNELS into [ndirs][nS][ldA] tiles
//   k_noddi<STAGE>    models.pyx:902-981 split at its three solver calls
//   k_freewater       models.pyx:1231-1276
//   k_sandi           models.pyx:1567-1619
#pragma once
#include <type_traits>
#include "amx_solver.hpp"
#include "amx_gram_solver.hpp"

namespace amx {

struct Chunk { int dir, start, count, pad; };
constexpr int kSeedKD = 12;      // compressed dimensions of the support-seed problem (amx_seed.hpp)
// rows of a block of the A'y table (k_noddi_gemm, amx_seed.hpp): atoms 0 .. n_atoms - 1, then from aux0 = n_atoms: U'y (12), U2'y (12),
// sum_b0 y, ||y||^2, sum_b0 y^2, t_min = min_dwi y_i / iso_i; padded to whole 16-row MFMA tiles
constexpr int kAuxU = 0, kAuxU2 = 12, kAuxB0 = 24, kAuxYY = 25, kAuxYB = 26, kAuxTmin = 27, kAuxN = 28;
__host__ __device__ constexpr int gemm_rows(int n_atoms) { return ((n_atoms + kAuxN + 15) / 16) * 16; }
constexpr int kScreenLd = 192;   // atoms per row of the float32 screening table [KD][kScreenLd]

enum StatusSlot { ST_ERRVOX = 0, ST_II1 = 1, ST_II2 = 2, ST_OVERFLOW = 3, ST_ITCAP = 4, ST_RERUN = 5, ST_GUARD = 6, ST_GUARDVOX = 7, ST_EXACT = 8, ST_GRAM = 11, ST_ITERS = 14, ST_SEED = 17, ST_LEFT = 92, ST_CLIP = 95, ST_ERRPACK = 96, ST_ERRKIND = 98, ST_WORDS = 100 };   // ST_LEFT + 0..2: voxels the Gram-space certificates of stage 1 / LASSO / stage 3 left to the wavefront-per-voxel kernels; ST_CLIP: voxels whose stage-2 signal was clipped; ST_ERRPACK (two words, 8-byte aligned): first offending voxel << 32 | what it held, ONE 64-bit atomicMin (amx_sync_status decodes it into ST_ERRVOX / ST_II1 / ST_II2 of the host mirror); ST_ERRKIND: 1 = the payload is a dictionary index of the batched solvers

// ------------------------------------------------------------------ shared pieces
struct FitCommon {
    const void *tiles;            // [ndirs][nS][ldA] (float) or [nS][ldA] (double, SANDI)
    const double *y;              // [n_vox][nS]
    const float *y32;             // the same signals as float32 (the *_fit_device_f32 entry points), or null: then `y` is read
    const int *perm;              // voxels grouped by orientation
    const Chunk *chunks;
    const int *n_chunks;
    const int *list;              // LIST mode: voxel ids to re-run
    const int *list_count;
    const int *lutidx;            // [n_vox]
    int *ovf_list;                // voxels whose passive set did not fit MAXP
    int *ovf_count;
    int *status;
    int nS, ldA, n_atoms;
    int tile_stride;              // elements between consecutive orientation tiles (multiple of 4)
    double lam1, lam2;
    unsigned flags;
    double *xdbg;                 // AMX_F_DEBUG_X: coefficient vectors, [n_vox][n_stage][n_atoms] (null = off)
};

template <typename AT>
__device__ __forceinline__ void stage_tile(AT *As, const AT *__restrict__ g, int words, int pad)
{
    // straight copy global -> LDS (tile is already in LDS layout), 16 B per lane per step
    const int nvec = words / (16 / (int)sizeof(AT));
    const uint4 *gv = reinterpret_cast<const uint4 *>(g);
    uint4 *sv = reinterpret_cast<uint4 *>(As);
    for (int k = threadIdx.x; k < nvec; k += blockDim.x) sv[k] = gv[k];
    for (int k = nvec * (16 / (int)sizeof(AT)) + threadIdx.x; k < words; k += blockDim.x) As[k] = g[k];
    for (int k = threadIdx.x; k < pad; k += blockDim.x) As[words + k] = (AT)0;
}

// the same for a tile kept as fp64 in LDS (the fp32 -> fp64 conversions of the tile reads are paid once per chunk instead
// of once per read: 5 % / 9 % of the VALU instructions of the NODDI stage-1 / stage-3 kernels)
__device__ __forceinline__ void stage_tile_widen(double *As, const float *__restrict__ g, int words, int pad)
{
    const int nvec = words / 4;
    const float4 *gv = reinterpret_cast<const float4 *>(g);
    for (int k = threadIdx.x; k < nvec; k += blockDim.x) {
        const float4 t = gv[k];
        As[4 * k] = (double)t.x; As[4 * k + 1] = (double)t.y; As[4 * k + 2] = (double)t.z; As[4 * k + 3] = (double)t.w;
    }
    for (int k = nvec * 4 + threadIdx.x; k < words; k += blockDim.x) As[k] = (double)g[k];
    for (int k = threadIdx.x; k < pad; k += blockDim.x) As[words + k] = 0.0;
}
template <typename AT>
__device__ __forceinline__ void stage_noddi_tile(AT *As, const float *__restrict__ g, int words, int pad)
{
    if constexpr (std::is_same<AT, double>::value) stage_tile_widen(As, g, words, pad);
    else stage_tile<float>(As, g, words, pad);
}

// the voxel's signal row, float64 or float32 storage (FitCommon::y / y32; float32 is what the image holds, core.py:136)
template <int NR>
__device__ __forceinline__ bool load_rows(const FitCommon &c, int vox, int nS, int lane, double (&yr)[NR])
{
    bool finite = true;
    if (c.y32 != nullptr) {
        const float *__restrict__ yv = c.y32 + (size_t)vox * nS;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            yr[rr] = (i < nS) ? (double)yv[i] : 0.0;
            finite = finite && (fabs(yr[rr]) <= 1.79769313486231570e308);
        }
    } else {
        const double *__restrict__ yv = c.y + (size_t)vox * nS;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            yr[rr] = (i < nS) ? yv[i] : 0.0;
            finite = finite && (fabs(yr[rr]) <= 1.79769313486231570e308);
        }
    }
    return ballot64(!finite) == 0ull;
}

template <int NR>
__device__ __forceinline__ bool load_rows(const double *__restrict__ yv, int nS, int lane, double (&yr)[NR])
{
    bool finite = true;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int i = lane + kWave * rr;
        yr[rr] = (i < nS) ? yv[i] : 0.0;
        finite = finite && (fabs(yr[rr]) <= 1.79769313486231570e308);
    }
    return ballot64(!finite) == 0ull;
}

// AMX_F_DEBUG_X: dense coefficient vector of one voxel from slot space (lane s < np: atom idx, value xv); exact zeros
// off the passive set (the contract of cyspams nnls / lasso: `x` fully written).  Diagnosis path: two store phases
// separated by a wait, so the zeros cannot overtake the values.
template <int NQ>
__device__ __forceinline__ void store_x_dense(double *dst, int n_atoms, int lane, int np, int idx, double xv)
{
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int j = lane + kWave * q;
        if (j < n_atoms) dst[j] = 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < np && idx >= 0 && idx < n_atoms) dst[idx] = xv;
}

// ------------------------------------------------------------------ NODDI
// models.pyx:966: ODI = 2/pi * atan2(1, kappa).  Kept out of line: inlined, the constants of ocml's atan2 are
// hoisted out of the voxel loop and stay live across the solver (tens of VGPRs -> scratch spills).
__device__ __attribute__((noinline)) double odi_from_kappa(double k1)
{
    return 2.0 / 3.14159265358979323846 * atan2(1.0, k1);
}

struct NoddiArgs {
    FitCommon c;
    const unsigned char *rowdwi;  // [nS] rows entering stage 2 (scheme.dwi_idx / single_b0 rule)
    const double *colscale;       // [n_atoms] KERNELS['norms'][0][k] (1.0 for iso/dot)
    const float *icvf, *kappa;    // [n_wm]
    int n_wm, is_exvivo, n_maps;
    const double *gram;           // [ndirs][n_atoms][ldG] A'A of every orientation tile (all rows), or null
    const double *gram_dwi;       // same restricted to the stage-2 rows
    int ldG;
    const unsigned long long *seeds;   // support seeds of the NNLS stage being run, bucket order (amx_seed.hpp), or null
    const unsigned long long *seeds2;  // passive-set seeds of the LASSO stage [n_vox][4], bucket order, or null
    int cand_lists;                    // 1: the LASSO certificates left the stage-3 candidate byte lists in seeds2 for the voxels they settled (done == 1)
    const unsigned char *done;         // NNLS stages: [n_vox] bucket order, 1 = settled by k_nnls_gcert (skipped here), or null
    const int *rlist, *rcount;         // NNLS stages after k_nnls_gcert: the chunk list is the second plan's, chunk c works on the
                                       // rcount[c] bucket positions rlist[chunk start ...] (the voxels the Gram certificate left over)
    // dual-value screening of certify_seed (NNLS stages): float32 S [ndirs][12][192], kappa [ndirs], y~ [n_vox][12] (bucket
    // order), fp64 S [ndirs][n_atoms][12]; all null: exact sweep
    const float *scr_S; const double *scr_kappa, *scr_ytil, *scr_Sg;
    const float *scr2_S; const double *scr2_kappa, *scr2_ytil, *scr2_Sg;   // the same for the LASSO stage (n_wm atoms, stage-2 rows)
    double *xiso;                 // [n_vox][2]  x_iso, x_dot after stage 1
    unsigned long long *supp;     // [n_vox][4]  stage-2 support bit set
    double *est, *rmse, *nrmse, *mod;
    int list_is_pos;              // stage 4: the overflow list holds bucket positions (seeded fit), not voxel numbers
    int fork_l2;                  // host side only (round 6, AMX_FORK bit 1): the LASSO left-overs are finished on the side stream
};

// STAGE 1 = NNLS (models.pyx:911), 2 = LASSO by the QR solver, 4 = LASSO by the Gram solver
// (models.pyx:914-926), 3 = debias NNLS + maps (models.pyx:929-967)
template <int STAGE, int NR, int NQ, int MAXP, typename AT = float>
__device__ __forceinline__ void noddi_voxel(const NoddiArgs &a, const AT *As, double *rs, double *rl,
                                            unsigned long long *wmask, int vox, int dir, int lane, int pos = -1, const float *Sf = nullptr)
{
    constexpr bool kLasso = (STAGE == 2 || STAGE == 4);
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_wm = a.n_wm;
    const int iso_atom = n_atoms - 1, dot_atom = a.is_exvivo ? n_atoms - 2 : -1;
    double yr[NR];
    const bool ok = load_rows<NR>(a.c, vox, nS, lane, yr);
    bool rowok[NR];
    double scl[NQ];
    unsigned long long allowed[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int lo = kWave * q;
        const int cnt = (kLasso ? n_wm : n_atoms) - lo;
        allowed[q] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
        scl[q] = 1.0;
    }
    if (!ok) {   // non-finite signal: propagate NaN maps, never iterate (SURVEY 8(b) error convention)
        if (STAGE == 1 && lane < 2) a.xiso[(size_t)vox * 2 + lane] = __builtin_nan("");
        if (STAGE == 3 && lane < a.n_maps) a.est[(size_t)vox * a.n_maps + lane] = __builtin_nan("");
        if (STAGE == 3 && lane == 0) {
            if (a.rmse) a.rmse[vox] = __builtin_nan("");
            if (a.nrmse) a.nrmse[vox] = __builtin_nan("");
            if (a.mod) { a.mod[(size_t)vox * 2] = __builtin_nan(""); a.mod[(size_t)vox * 2 + 1] = __builtin_nan(""); }
        }
        if (kLasso && lane < 4) a.supp[(size_t)vox * 4 + lane] = 0ull;
    }
    if (ok) {
    if (kLasso) {
        // models.pyx:914-925: y2 = max(0, y_dwi - x_iso*iso_dwi (- x_dot)), columns scaled by norms
        const double xiso = a.xiso[(size_t)vox * 2], xdot = a.xiso[(size_t)vox * 2 + 1];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            rowok[rr] = (i < nS) && a.rowdwi[i];
            double t = 0.0;
            if (rowok[rr]) {
                t = yr[rr] - xiso * (double)As[i * ldA + iso_atom];
                if (a.is_exvivo) t -= xdot * 1.0;
                if (t < 0.0) t = 0.0;
            }
            yr[rr] = t;
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int j = lane + kWave * q;
            scl[q] = (j < n_atoms) ? a.colscale[j] : 1.0;
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < NR; rr++) rowok[rr] = (lane + kWave * rr) < nS;
    }
    if (STAGE == 3) {
        // models.pyx:929-936: support of the LASSO solution plus iso (and dot)
#pragma unroll
        for (int q = 0; q < NQ; q++) allowed[q] = a.supp[(size_t)vox * 4 + q];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if ((iso_atom >> 6) == q) allowed[q] |= 1ull << (iso_atom & 63);
            if (dot_atom >= 0 && (dot_atom >> 6) == q) allowed[q] |= 1ull << (dot_atom & 63);
        }
    }

    const double *gm = kLasso ? a.gram_dwi : a.gram;
    const double *gdir = gm ? gm + (size_t)dir * n_atoms * a.ldG : nullptr;
    typename std::conditional<STAGE == 4, GramSolver<NR, NQ, MAXP, AT>, NNSolver<NR, NQ, MAXP, STAGE == 2, AT>>::type S;
    unsigned long long seed = kSeedNone;
    if constexpr (!kLasso) {
        if (a.seeds != nullptr && pos >= 0) seed = a.seeds[pos];
    }
    int st_;
    if constexpr (STAGE == 4) {
        SeedScreen scr;
        if (Sf != nullptr && pos >= 0 && a.seeds2 != nullptr) {
            scr.Sf = Sf; scr.ld = kScreenLd; scr.kappa = a.scr2_kappa[dir];
            scr.ytil = a.scr2_ytil + (size_t)pos * kSeedKD; scr.Sg = a.scr2_Sg + (size_t)dir * n_wm * kSeedKD;
#ifdef AMX_STATS
            scr.count = a.c.status + ST_SEED + 23;
#endif
        }
        st_ = S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane, gdir, a.ldG,
                      (a.seeds2 != nullptr && pos >= 0) ? a.seeds2 + (size_t)pos * 4 : nullptr, scr);
    }
    else {
        SeedScreen scr;
        if (Sf != nullptr && pos >= 0 && seed != kSeedNone) {
            scr.Sf = Sf; scr.ld = kScreenLd; scr.kappa = a.scr_kappa[dir];
            scr.ytil = a.scr_ytil + (size_t)pos * kSeedKD; scr.Sg = a.scr_Sg + (size_t)dir * n_atoms * kSeedKD;
#ifdef AMX_STATS
            scr.count = a.c.status + ST_SEED + 22;
#endif
        }
        st_ = S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, kLasso ? a.c.lam1 : 0.0, kLasso ? a.c.lam2 : 0.0, rs, rl, lane, gdir, a.ldG, seed, scr);
    }
    const int st = __builtin_amdgcn_readfirstlane(st_);
    if (st == kOverflow) {
        if (lane == 0) {
            const int k = atomicAdd(a.c.ovf_count, 1);
            a.c.ovf_list[k] = (STAGE == 4 && a.list_is_pos) ? pos : vox;      // (the re-run kernel wants the seed as well)
        }
    } else {
    if (st == kIterCap && lane == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (st > kIterCap && lane == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
#ifdef AMX_PHASES
    if constexpr (STAGE == 1 || STAGE == 3) {
        if (lane == 0 && S.seeded == 1) {       // (certified voxels only: phases of certify_seed)
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(const_cast<int *>(a.c.n_chunks) + 16) + (STAGE == 1 ? 0 : 8);
            for (int k = 0; k < 8; k++) atomicAdd(&acc[k], (unsigned long long)S.ph[k]);
        }
    }
#endif
#ifdef AMX_STATS
    if (lane == 0) { constexpr int sx = kLasso ? 1 : STAGE - 1; atomicAdd(&a.c.status[ST_EXACT + sx], S.n_exact); atomicAdd(&a.c.status[ST_GRAM + sx], S.n_gram); atomicAdd(&a.c.status[ST_ITERS + sx], S.iters);
        if constexpr (STAGE == 4) { if (S.seeded >= 0) atomicAdd(&a.c.status[ST_SEED + 18], 1); if (S.seeded == 1) atomicAdd(&a.c.status[ST_SEED + 19], 1); }
        if constexpr (!kLasso) { if (S.seeded >= 0) atomicAdd(&a.c.status[ST_SEED + (STAGE == 1 ? 0 : 2)], 1); if (S.seeded == 1) atomicAdd(&a.c.status[ST_SEED + (STAGE == 1 ? 1 : 3)], 1); if (S.seeded == 0 && STAGE == 1) atomicAdd(&a.c.status[ST_SEED + 6 + S.seed_why], 1); } }
#endif
    const bool act = lane < S.np;
    if (a.c.xdbg) {
        // rows: 0 = NNLS over all atoms, 1 = LASSO coefficients (normalised columns) with the stage-1 iso (dot)
        // coefficients kept behind them (the reference reuses one `x` array, models.pyx:911-926), 2 = debiased x
        constexpr int row = (STAGE == 1) ? 0 : (kLasso ? 1 : 2);
        double *dst = a.c.xdbg + ((size_t)vox * 3 + row) * n_atoms;
        store_x_dense<NQ>(dst, n_atoms, lane, S.np, S.idx, S.x);
        if (kLasso && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dst[iso_atom] = a.xiso[(size_t)vox * 2];
            if (dot_atom >= 0) dst[dot_atom] = a.xiso[(size_t)vox * 2 + 1];
        }
    }

    if constexpr (STAGE == 1) {
        // (the coefficient of the slot that holds the atom, if any: a ballot and a broadcast instead of a wavefront sum)
        const unsigned long long mi = ballot64(act && S.idx == iso_atom), md = ballot64(act && S.idx == dot_atom);
        const double xi = mi ? bcast(S.x, __builtin_ctzll(mi)) : 0.0;
        const double xd = md ? bcast(S.x, __builtin_ctzll(md)) : 0.0;
        if (lane == 0) { a.xiso[(size_t)vox * 2] = xi; a.xiso[(size_t)vox * 2 + 1] = xd; }
    } else if constexpr (kLasso) {
        if (lane < 4) wmask[lane] = 0ull;
        if (act && S.x > 0.0) atomicOr(&wmask[S.idx >> 6], 1ull << (S.idx & 63));
        if (lane < 4) a.supp[(size_t)vox * 4 + lane] = wmask[lane];
    } else {
        // error maps first: the factor Q is dead afterwards (keeps it out of the registers of the map arithmetic)
        double rsq = 0.0, ysq = 0.0;
        if (a.c.flags & 3u) {
            S.residual(yr, 0.0);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) { rsq += S.r[rr] * S.r[rr]; ysq += yr[rr] * yr[rr]; }
            rsq = wave_sum(rsq); ysq = wave_sum(ysq);
        }
        // models.pyx:945-967
        const double xs = act ? S.x : 0.0;
        const bool iswm = act && S.idx < n_wm;
        const double sum_atoms = wave_sum(xs) + 1e-16;
        const double sum_wm = wave_sum(iswm ? xs / sum_atoms : 0.0) + 1e-16;
        double f1 = 0.0, f2 = 0.0, k1 = 0.0;
        if (iswm) {
            const float ic = a.icvf[S.idx];
            const double t = xs / sum_atoms / sum_wm;
            f1 = (double)ic * t;
            f2 = (double)((float)(1.0 - (double)ic)) * t;
            k1 = (double)a.kappa[S.idx] * t;
        }
        f1 = wave_sum(f1); f2 = wave_sum(f2); k1 = wave_sum(k1);
        const double ndi = f1 / (f1 + f2 + 1e-16);
        const double odi = odi_from_kappa(k1);
        const double fwf = wave_sum((act && S.idx == iso_atom) ? xs : 0.0) / sum_atoms;
        const double dot = wave_sum((act && S.idx == dot_atom) ? xs : 0.0) / sum_atoms;
        if (lane == 0) {
            double *e = a.est + (size_t)vox * a.n_maps;
            e[0] = ndi; e[1] = odi; e[2] = fwf;
            if (a.is_exvivo) e[3] = dot;
            if (a.rmse) a.rmse[vox] = sqrt(rsq / (double)nS);                       // models.pyx:47-54
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rsq / ysq) : 0.0;      // models.pyx:58-71
            if (a.mod) { const double tf = 1.0 - fwf; a.mod[(size_t)vox * 2] = ndi * tf; a.mod[(size_t)vox * 2 + 1] = odi * tf; }
        }
    }
    }   // st != kOverflow
    }   // ok
}

// ------------------------------------------------------------------ FreeWater
struct FwArgs {
    FitCommon c;
    int n_perp, n_iso, is_mouse, n_maps;
    double *est, *rmse, *nrmse, *ycorr;
    double *cproj;                 // k_fw_project -> k_freewater_refill: c = A'y - lambda1 [N][ldC] (bucket order)
    const double *prep;            // k_fw_orient_prep: per orientation A | H^-1 | H (amx_lut::fw_prep)
    unsigned *p0;                  // passive set after the first block removal (bit j: atom j), bucket order
    int ldC;
    int *queue;                    // global ticket of the persistent solver wavefronts (sub-chunks)
    int sub_per_chunk;
};

template <int NR, int NQ, int MAXP>
__device__ __forceinline__ void fw_voxel(const FwArgs &a, const float *As, double *rs, double *rl, int vox, int lane)
{
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_perp = a.n_perp;
    double yr[NR];
    const bool ok = load_rows<NR>(a.c, vox, nS, lane, yr);
    bool rowok[NR];
    double scl[NQ];
    unsigned long long allowed[NQ];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) rowok[rr] = (lane + kWave * rr) < nS;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int cnt = n_atoms - kWave * q;
        allowed[q] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
        scl[q] = 1.0;
    }
    if (!ok) {
        if (lane < a.n_maps) a.est[(size_t)vox * a.n_maps + lane] = __builtin_nan("");
        if (lane == 0 && a.rmse) a.rmse[vox] = __builtin_nan("");
        if (lane == 0 && a.nrmse) a.nrmse[vox] = __builtin_nan("");
        if (a.ycorr)
            for (int i = lane; i < nS; i += kWave) a.ycorr[(size_t)vox * nS + i] = __builtin_nan("");
    }
    if (ok) {
    NNSolver<NR, NQ, MAXP, true, float> S;
    const int st = __builtin_amdgcn_readfirstlane(S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane));
    if (st == kOverflow) {
        if (lane == 0) { const int k = atomicAdd(a.c.ovf_count, 1); a.c.ovf_list[k] = vox; }
    } else {
    if (st == kIterCap && lane == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (st > kIterCap && lane == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
    const bool act = lane < S.np;
    const double xs = act ? S.x : 0.0;
    if (a.c.xdbg) store_x_dense<NQ>(a.c.xdbg + (size_t)vox * n_atoms, n_atoms, lane, S.np, S.idx, S.x);
    // models.pyx:1241-1256
    const double x_sum = wave_sum(xs) + 1e-16;
    const double v = wave_sum((act && S.idx < n_perp) ? xs : 0.0) / x_sum;
    const double vb = wave_sum((act && S.idx == n_perp) ? xs : 0.0) / x_sum;
    const double vc = wave_sum((act && S.idx == n_perp + 1) ? xs : 0.0) / x_sum;
    double rsq = 0.0, ysq = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) { rsq += S.r[rr] * S.r[rr]; ysq += yr[rr] * yr[rr]; }
    if (a.c.flags & 3u) { rsq = wave_sum(rsq); ysq = wave_sum(ysq); }
    if (lane == 0) {
        double *e = a.est + (size_t)vox * a.n_maps;
        e[0] = v; e[1] = 1.0 - v;
        if (a.is_mouse) { e[2] = vb; e[3] = vc; }
        if (a.rmse) a.rmse[vox] = sqrt(rsq / (double)nS);
        if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rsq / ysq) : 0.0;
    }
    if (a.ycorr) {
        // models.pyx:1264-1274: y - A[:, iso atoms] x_iso, clipped at 0
        double fw[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) fw[rr] = 0.0;
        for (int s = 0; s < S.np; s++) {
            const int at = bcast_i(S.idx, s);
            if (at >= n_perp) {
                const double xv = bcast(S.x, s);
#pragma unroll
                for (int rr = 0; rr < NR; rr++) {
                    const int i = lane + kWave * rr;
                    if (i < nS) fw[rr] += (double)As[i * ldA + at] * xv;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = lane + kWave * rr;
            if (i < nS) { const double yc = yr[rr] - fw[rr]; a.ycorr[(size_t)vox * nS + i] = yc < 0.0 ? 0.0 : yc; }
        }
    }
    }   // st != kOverflow
    }   // ok
}

// ------------------------------------------------------------------ SANDI
struct SandiArgs {
    FitCommon c;
    const double *norms, *Rs, *d_in, *d_isos;
    int n_rs, n_in, n_iso;
    double *est, *rmse, *nrmse;
    const double *tables;          // k_sandi_tables (row-space kernel): T | G | g0 | A
    int n_lin;                     // > 0: no plan, workgroup b takes voxels [256 b, 256 b + 256) of n_lin (row-space kernel: one dictionary, no bucketing)
};

template <int NR, int NQ, int MAXP>
__device__ __forceinline__ void sandi_voxel(const SandiArgs &a, const double *As, double *rs, double *rl, int vox, int lane)
{
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms;
    const int n_rs = a.n_rs, n_in = a.n_in;
    double yr[NR];
    const bool ok = load_rows<NR>(a.c, vox, nS, lane, yr);
    bool rowok[NR];
    double scl[NQ];
    unsigned long long allowed[NQ];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) rowok[rr] = (lane + kWave * rr) < nS;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int cnt = n_atoms - kWave * q;
        allowed[q] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
        scl[q] = 1.0;
    }
    if (!ok) {
        if (lane < 6) a.est[(size_t)vox * 6 + lane] = __builtin_nan("");
        if (lane == 0 && a.rmse) a.rmse[vox] = __builtin_nan("");
        if (lane == 0 && a.nrmse) a.nrmse[vox] = __builtin_nan("");
    }
    if (ok) {
    NNSolver<NR, NQ, MAXP, true, double> S;
    const int st = __builtin_amdgcn_readfirstlane(S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane));
    if (st == kOverflow) {
        if (lane == 0) { const int k = atomicAdd(a.c.ovf_count, 1); a.c.ovf_list[k] = vox; }
    } else {
    if (st == kIterCap && lane == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (st > kIterCap && lane == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
    const bool act = lane < S.np;
    const int at = act ? S.idx : 0;
    const double xs = act ? S.x * a.norms[at] : 0.0;                       // models.pyx:1570-1571
    if (a.c.xdbg) store_x_dense<NQ>(a.c.xdbg + (size_t)vox * n_atoms, n_atoms, lane, S.np, S.idx, xs);
    const bool sph = act && at < n_rs, stk = act && at >= n_rs && at < n_rs + n_in, iso = act && at >= n_rs + n_in;
    const double x_sum = wave_sum(xs) + 1e-16;
    double xsph = wave_sum(sph ? xs : 0.0), xstk = wave_sum(stk ? xs : 0.0), xiso = wave_sum(iso ? xs : 0.0);
    const double Rsoma = wave_sum(sph ? a.Rs[at] * xs : 0.0);
    const double Din = wave_sum(stk ? a.d_in[at - n_rs] * xs : 0.0);
    const double De = wave_sum(iso ? a.d_isos[at - n_rs - n_in] * xs : 0.0);
    if (lane == 0) {
        double *e = a.est + (size_t)vox * 6;
        e[0] = xsph / x_sum; e[1] = xstk / x_sum; e[2] = xiso / x_sum;
        e[3] = 1e6 * Rsoma / (xsph + 1e-16);
        e[4] = 1e3 * Din / (xstk + 1e-16);
        e[5] = 1e3 * De / (xiso + 1e-16);
    }
    if (a.c.flags & 3u) {
        // quirk kept (models.pyx:1571 then 1615): errors use the RESCALED x with the NORMALISED A
        double est[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) est[rr] = 0.0;
        for (int s = 0; s < S.np; s++) {
            const int as = bcast_i(S.idx, s);
            const double xv = bcast(xs, s);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const int i = lane + kWave * rr;
                if (i < nS) est[rr] += As[i * ldA + as] * xv;
            }
        }
        double rsq = 0.0, ysq = 0.0;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) { const double t = yr[rr] - est[rr]; rsq += t * t; ysq += yr[rr] * yr[rr]; }
        rsq = wave_sum(rsq); ysq = wave_sum(ysq);
        if (lane == 0) {
            if (a.rmse) a.rmse[vox] = sqrt(rsq / (double)nS);
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rsq / ysq) : 0.0;
        }
    }
    }   // st != kOverflow
    }   // ok
}

// ------------------------------------------------------------------ CylinderZeppelinBall (models.pyx:526-652)
// A = [cylinders | zeppelins | balls] of the voxel's orientation, lasso with lambda1 = 0, lambda2 = 4 by default
// (models.pyx:439): a strong ridge, so the optimum is dense (22 of 26 atoms on the fixture) and cond(A'A + lambda2 I) is
// tiny -- the Gram-space solver (Cholesky of the passive block in LDS, no register-resident Q) with a passive-set
// capacity that holds every atom.
struct CzbArgs {
    FitCommon c;
    int n_rs, n_perp;
    const double *Rs;             // [n_rs] model.Rs (metres)
    const double *gram;           // [ndirs][n_atoms][ldG] A'A of every orientation tile
    int ldG;
    double *est, *rmse, *nrmse;
};

// QR = true: the A-space solver (thin QR of the passive columns + the ridge rows) for lambda2 too small to bound cond(A'A + lambda2 I)
// -- the reference's lasso accepts any lambda2 >= 0 (models.pyx:439, 615)
template <int NR, int NQ, int MAXP, bool QR = false>
__device__ __forceinline__ void czb_voxel(const CzbArgs &a, const float *As, double *rs, double *rl, int vox, const double *gdir, int lane,
                                          const double *Lf = nullptr, const double *lf_inv = nullptr)
{
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_rs = a.n_rs, n_perp = a.n_perp;
    double yr[NR];
    const bool ok = load_rows<NR>(a.c, vox, nS, lane, yr);
    bool rowok[NR];
    double scl[NQ];
    unsigned long long allowed[NQ];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) rowok[rr] = (lane + kWave * rr) < nS;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int cnt = n_atoms - kWave * q;
        allowed[q] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
        scl[q] = 1.0;
    }
    if (!ok) {
        if (lane < 3) a.est[(size_t)vox * 3 + lane] = __builtin_nan("");
        if (lane == 0 && a.rmse) a.rmse[vox] = __builtin_nan("");
        if (lane == 0 && a.nrmse) a.nrmse[vox] = __builtin_nan("");
    }
    if (ok) {
    typename std::conditional<QR, NNSolver<NR, NQ, MAXP, true, float>, GramSolver<NR, NQ, MAXP, float>>::type S;
    // strong ridge, dense optimum: block principal pivoting from the full set; otherwise (or AMX_COLD_START) Lawson-Hanson
    const bool dense = NQ == 1 && n_atoms <= MAXP && a.c.lam2 >= 1e-2 && !(a.c.flags & 0x80000000u);
    int st_;
    if constexpr (QR) {
        st_ = S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane);
    } else if constexpr (NQ == 1) {
        st_ = dense ? S.solve_dense(As, ldA, nS, n_atoms, yr, a.c.lam1, a.c.lam2, rs, rl, lane, gdir, a.ldG, Lf, lf_inv)
                    : S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane, gdir, a.ldG);
    } else {
        st_ = S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, a.c.lam1, a.c.lam2, rs, rl, lane, gdir, a.ldG);
    }
    const int st = __builtin_amdgcn_readfirstlane(st_);
    if (st == kOverflow) {
        if (lane == 0) { const int k = atomicAdd(a.c.ovf_count, 1); a.c.ovf_list[k] = vox; }
    } else {
    if (st == kIterCap && lane == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (st > kIterCap && lane == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
    const bool act = lane < S.np;
    const int at = act ? S.idx : 0;
    const double xs = act ? S.x : 0.0;
    if (a.c.xdbg) store_x_dense<NQ>(a.c.xdbg + (size_t)vox * n_atoms, n_atoms, lane, S.np, S.idx, xs);
    // models.pyx:616-633
    double f1 = wave_sum((act && at < n_rs) ? xs : 0.0);
    const double f2 = wave_sum((act && at >= n_rs && at < n_rs + n_perp) ? xs : 0.0) + 1e-16;
    const double v = f1 / (f1 + f2 + 1e-16);
    f1 += 1e-16;
    double am = wave_sum((act && at < n_rs) ? a.Rs[at] * xs : 0.0);
    am = 1e6 * 2.0 * am / f1;
    const double d = (4.0 * v) / (3.14159265358979323846 * (am * am) + 1e-16);
    if (lane == 0) { double *e = a.est + (size_t)vox * 3; e[0] = v; e[1] = am; e[2] = d; }
    if (a.c.flags & 3u) {
        double est[NR];
#pragma unroll
        for (int rr = 0; rr < NR; rr++) est[rr] = 0.0;
        for (int s = 0; s < S.np; s++) {
            const int as = bcast_i(S.idx, s);
            const double xv = bcast(xs, s);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const int i = lane + kWave * rr;
                if (i < nS) est[rr] += (double)As[i * ldA + as] * xv;
            }
        }
        double rsq = 0.0, ysq = 0.0;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) { const double t = yr[rr] - est[rr]; rsq += t * t; ysq += yr[rr] * yr[rr]; }
        rsq = wave_sum(rsq); ysq = wave_sum(ysq);
        if (lane == 0) {
            if (a.rmse) a.rmse[vox] = sqrt(rsq / (double)nS);
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rsq / ysq) : 0.0;
        }
    }
    }   // st != kOverflow
    }   // ok
}

// doubles of per-wavefront LDS the solver needs besides the residual scratch:
// QR solver: R and the ridge rows, (MAXP+1)^2 each; Gram solver: two packed triangles
// (without the ridge the QR solver has no augmented rows: R only)
__host__ __device__ constexpr int solver_lds_words(bool gram, int maxp, bool ridge = true)
{
    return gram ? (maxp + 1) * (maxp + 2) : (ridge ? 2 : 1) * (maxp + 1) * (maxp + 1);
}

// XCD-aware block -> chunk map.  Workgroup b runs on XCD b % 8 (observed dispatch order; only a
// speed assumption).  The chunk list is sorted by orientation, so XCD x takes the x-th CONTIGUOUS
// eighth of it: the dictionary tile and the Gram columns of one orientation are then fetched into
// ONE XCD's L2 instead of all eight.
__device__ __forceinline__ int xcd_chunk(int b, int n_chunks)
{
    const int per = (n_chunks + 7) >> 3;
    const int cid = (b & 7) * per + (b >> 3);
    return ((b >> 3) < per && cid < n_chunks) ? cid : -1;
}

// ------------------------------------------------------------------ kernel skeleton
// One workgroup = up to NW wavefronts sharing one dictionary tile in LDS; wavefront w takes the
// voxels w, w+nw, ... of its chunk.  LIST mode re-runs single voxels (large-MAXP variant): tile
// staged per voxel.

// Next voxel ticket of the workgroup.  All 64 lanes execute the LDS atomic (lane 0 adds 1, the others 0): no divergent
// branch around it -- a lane-0-only atomic in this loop made ROCm 7.2's structuriser emit a loop that never exits.
__device__ __forceinline__ int next_ticket(unsigned *ticket, int lane)
{
    const unsigned off = (unsigned)(uintptr_t)ticket;      // LDS offset = low 32 bits of the flat shared address
    const unsigned inc = lane == 0 ? 1u : 0u;
    unsigned old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(off), "v"(inc) : "memory");
    return __builtin_amdgcn_readfirstlane((int)old);
}

// GTv (global tile): the dictionary tile is too large for a CU's LDS (an HCP-style protocol: 288 x 145 float32 = 167 KB) -- the
// solver then reads it where it lies, in HBM / L2 (the chunk's voxels share one orientation, the XCD-aware chunk map keeps it in one
// L2), and LDS holds the per-wavefront blocks only.  A template switch, not a run-time one: the LDS variants keep their ds_read
// addressing.  The tiles array carries kTileSlack floats behind its last tile (lanes of atoms >= n_atoms read past a row's end).
#define AMX_KERNEL_PROLOGUE_GT(AT, NRv, NQv, NWv, RLWv, GTv)                                                \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                              \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                       \
    const int words = a.c.nS * a.c.ldA;                                                               \
    const int words_pad = (GTv) ? 0 : ((words + kWave * NQv + 3) & ~3);                               \
    AT *As = reinterpret_cast<AT *>(smem);                                                            \
    double *rs_all = reinterpret_cast<double *>(smem + (((size_t)words_pad * sizeof(AT) + 15) & ~(size_t)15)); \
    double *rs = rs_all + wave * (NRv * kWave);                                                       \
    const int nw_ = (int)blockDim.x >> 6;   /* wavefronts actually launched (<= NWv) */                  \
    double *rl_all = rs_all + nw_ * NRv * kWave;                                                      \
    double *rl = rl_all + wave * (RLWv);                                                              \
    unsigned long long *wm_all = reinterpret_cast<unsigned long long *>(rl_all + nw_ * (RLWv)); \
    unsigned long long *wmask = wm_all + wave * 4;                                                    \

#define AMX_KERNEL_PROLOGUE(AT, NRv, NQv, NWv, RLWv) AMX_KERNEL_PROLOGUE_GT(AT, NRv, NQv, NWv, RLWv, false)
constexpr int kTileSlack = 512;      // floats (doubles for fp64 dictionaries) readable behind the last tile of a dictionary: 64 * NQ <= 256 atoms per row sweep

// (NW <= 4 with the tile in LDS: the small-call builds of the left-over kernels (round 6) -- float32 tile, few wavefronts, at most 80 KB of LDS and
//  256 registers, so that TWO workgroups share a CU and the ~500 workgroups of a call -- a chunk is an orientation -- run in ONE round)
template <int STAGE, int NR, int NQ, int MAXP, int NW, bool LIST, typename AT = float, bool GT = false>
__global__ void __launch_bounds__(NW * 64, (NW <= 4 && !LIST && !GT) ? 2 : 1) k_noddi(const NoddiArgs a)
{
    static_assert(!GT || std::is_same<AT, float>::value, "the global tile is the float32 dictionary itself");
    using ATs = typename std::conditional<GT, gtile<float>, AT>::type;      // what the solver is told about the tile (amx_solver.hpp: tile_sweep)
    constexpr int RLW = solver_lds_words(STAGE == 4, MAXP, STAGE == 2);
    AMX_KERNEL_PROLOGUE_GT(AT, NR, NQ, NW, RLW, GT)
    const float *tiles = reinterpret_cast<const float *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);     // the 16 spare bytes of fit_lds_bytes
        if ((STAGE == 1 || STAGE == 3 || STAGE == 4) && a.rlist != nullptr) {
            const int left = a.rcount[cid];
            if (left == 0) return;                                  // nothing left over in this chunk
            if (threadIdx.x == 0) atomicAdd(&a.c.status[ST_LEFT + (STAGE == 1 ? 0 : (STAGE == 3 ? 2 : 1))], left);   // (amx_last_seed_stats)
        }
        if (threadIdx.x == 0) { ticket[0] = (unsigned)nw_; ticket[1] = (unsigned)nw_; }
        const AT *At = As;
        if constexpr (GT) At = reinterpret_cast<const AT *>(tiles + (size_t)ck.dir * a.c.tile_stride);
        else stage_noddi_tile<AT>(As, tiles + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
        // NNLS stages with seeds: the float32 compressed dictionary of the orientation for the dual-value screening
        float *Sf = nullptr;
        if ((STAGE == 1 || STAGE == 3) && a.scr_S != nullptr && a.seeds != nullptr) {
            Sf = reinterpret_cast<float *>(ticket + 4);
            const float *src = a.scr_S + (size_t)ck.dir * kSeedKD * kScreenLd;
            for (int e = threadIdx.x; e < kSeedKD * kScreenLd; e += blockDim.x) Sf[e] = src[e];
        }
        if (STAGE == 4 && a.scr2_S != nullptr && a.seeds2 != nullptr) {
            Sf = reinterpret_cast<float *>(ticket + 4);
            const float *src = a.scr2_S + (size_t)ck.dir * kSeedKD * kScreenLd;
            for (int e = threadIdx.x; e < kSeedKD * kScreenLd; e += blockDim.x) Sf[e] = src[e];
        }
        __syncthreads();
#ifdef AMX_STATIC_VOXELS
        for (int k = wave; k < ck.count; k += nw_) {
            noddi_voxel<STAGE, NR, NQ, MAXP, ATs>(a, reinterpret_cast<const ATs *>(At), rs, rl, wmask, a.c.perm[ck.start + k], ck.dir, lane);
        }
#else
        // voxels differ 2-3x in solver iterations: the wavefronts draw the next voxel of the chunk from an LDS ticket
        // (next_ticket keeps the control flow wave-uniform: every lane takes part in the atomic)
        if ((STAGE == 1 || STAGE == 3 || STAGE == 4) && a.rlist != nullptr) {
            // Left-overs of the Gram certificates.  NNLS stages: two walks over the list, the voxels with a wrong or no seed first
            // (done = 0: Lawson-Hanson, ~140 us), then the ones refused for conditioning alone (done = 2: certified on the true
            // residual in ~15 us) -- a long voxel drawn last used to keep eleven wavefronts of the workgroup waiting
            const int cnt = a.rcount[cid];
            const bool two = (STAGE == 1 || STAGE == 3) && a.done != nullptr;
            for (int pass = 0; pass < (two ? 2 : 1); pass++) {
                unsigned *tk = ticket + pass;
                for (int k = wave; k < cnt; k = next_ticket(tk, lane)) {
                    const int pos = a.rlist[ck.start + k];
                    if (two) {
                        const int flag = __builtin_amdgcn_readfirstlane((int)a.done[pos]);
                        if ((flag == 2) != (pass == 1)) continue;
                    }
                    noddi_voxel<STAGE, NR, NQ, MAXP, ATs>(a, reinterpret_cast<const ATs *>(At), rs, rl, wmask, a.c.perm[pos], ck.dir, lane, pos, Sf);
                }
            }
        } else {
            for (int k = wave; k < ck.count; k = next_ticket(ticket, lane)) {
                noddi_voxel<STAGE, NR, NQ, MAXP, ATs>(a, reinterpret_cast<const ATs *>(At), rs, rl, wmask, a.c.perm[ck.start + k], ck.dir, lane, ck.start + k, Sf);
            }
        }
#endif
    } else {
        const int cnt = *a.c.list_count;
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
            // LASSO stage with seeds: the list holds bucket positions, the seed (more atoms than the main pass can hold) is
            // certified here instead of solving from scratch
            const int e = a.c.list[it];
            const int pos = (STAGE == 4 && a.list_is_pos) ? e : -1;
            const int vox = (STAGE == 4 && a.list_is_pos) ? a.c.perm[e] : e;
            if constexpr (GT) {
                noddi_voxel<STAGE, NR, NQ, MAXP, ATs>(a, reinterpret_cast<const ATs *>(tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride), rs, rl, wmask, vox, a.c.lutidx[vox], lane, pos);
            } else {
                __syncthreads();
                stage_noddi_tile<AT>(As, tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride, words, words_pad - words);
                __syncthreads();
                noddi_voxel<STAGE, NR, NQ, MAXP, AT>(a, As, rs, rl, wmask, vox, a.c.lutidx[vox], lane, pos);
            }
        }
    }
}

template <int NR, int NQ, int MAXP, int NW, bool LIST>
__global__ void __launch_bounds__(NW * 64) k_freewater(const FwArgs a)
{
    AMX_KERNEL_PROLOGUE(float, NR, NQ, NW, solver_lds_words(false, MAXP))
    (void)wmask;
    const float *tiles = reinterpret_cast<const float *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);
        if (threadIdx.x == 0) *ticket = (unsigned)nw_;
        stage_tile<float>(As, tiles + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
        __syncthreads();
        for (int k = wave; k < ck.count; k = next_ticket(ticket, lane)) {   // LDS voxel ticket, see k_noddi
            fw_voxel<NR, NQ, MAXP>(a, As, rs, rl, a.c.perm[ck.start + k], lane);
        }
    } else {
        const int cnt = *a.c.list_count;
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
            const int vox = a.c.list[it];
            __syncthreads();
            stage_tile<float>(As, tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride, words, words_pad - words);
            __syncthreads();
            fw_voxel<NR, NQ, MAXP>(a, As, rs, rl, vox, lane);
        }
    }
}

template <int NR, int NQ, int MAXP, int NW, bool LIST>
__global__ void __launch_bounds__(NW * 64) k_sandi(const SandiArgs a)
{
    AMX_KERNEL_PROLOGUE(double, NR, NQ, NW, solver_lds_words(false, MAXP))
    (void)wmask;
    const double *tiles = reinterpret_cast<const double *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);
        if (threadIdx.x == 0) *ticket = (unsigned)nw_;
        stage_tile<double>(As, tiles, words, words_pad - words);
        __syncthreads();
        for (int k = wave; k < ck.count; k = next_ticket(ticket, lane)) {   // LDS voxel ticket, see k_noddi
            sandi_voxel<NR, NQ, MAXP>(a, As, rs, rl, a.c.perm[ck.start + k], lane);
        }
    } else {
        const int cnt = *a.c.list_count;
        stage_tile<double>(As, tiles, words, words_pad - words);
        __syncthreads();
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) sandi_voxel<NR, NQ, MAXP>(a, As, rs, rl, a.c.list[it], lane);
    }
}

template <int NR, int NQ, int MAXP, int NW, bool LIST>
__global__ void __launch_bounds__(NW * 64) k_czb(const CzbArgs a)
{
    AMX_KERNEL_PROLOGUE(float, NR, NQ, NW, solver_lds_words(true, MAXP))
    (void)wmask;
    const float *tiles = reinterpret_cast<const float *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);
        if (threadIdx.x == 0) *ticket = (unsigned)nw_;
        stage_tile<float>(As, tiles + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
        // the Gram matrix of the orientation (n_atoms x ldG doubles: 13 KB for 26 atoms) next to the tile: the solver reads a
        // column per passive atom and step
        double *Gs = reinterpret_cast<double *>(ticket + 4);
        const double *gsrc = a.gram + (size_t)ck.dir * a.c.n_atoms * a.ldG;
        for (int e = threadIdx.x; e < a.c.n_atoms * a.ldG; e += blockDim.x) Gs[e] = gsrc[e];
        __syncthreads();
        // ... and the Cholesky factor of the full passive set (the first step of every voxel's block pivoting), by wavefront 0
        constexpr int kTriF = (MAXP + 1) * (MAXP + 2) / 2;
        double *Hf = Gs + (size_t)a.c.n_atoms * a.ldG, *Lf = Hf + kTriF, *lf_inv = Lf + kTriF;
        const bool dense_ = a.c.n_atoms <= MAXP && a.c.lam2 >= 1e-2 && !(a.c.flags & 0x80000000u);
        if (dense_ && wave == 0) {
            GramSolver<NR, NQ, MAXP, float> F;
            F.factor_full(a.c.n_atoms, a.c.lam2, Hf, Lf, lf_inv, lane, Gs, a.ldG);
        }
        __syncthreads();
        for (int k = wave; k < ck.count; k = next_ticket(ticket, lane)) {   // LDS voxel ticket, see k_noddi
            czb_voxel<NR, NQ, MAXP>(a, As, rs, rl, a.c.perm[ck.start + k], Gs, lane, dense_ ? Lf : nullptr, lf_inv);
        }
    } else {
        const int cnt = *a.c.list_count;
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
            const int vox = a.c.list[it];
            __syncthreads();
            stage_tile<float>(As, tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride, words, words_pad - words);
            __syncthreads();
            czb_voxel<NR, NQ, MAXP>(a, As, rs, rl, vox, a.gram + (size_t)a.c.lutidx[vox] * a.c.n_atoms * a.ldG, lane);
        }
    }
}

// CylinderZeppelinBall with (nearly) no ridge: the A-space QR solver, tile only (no Gram matrix in LDS)
template <int NR, int NQ, int MAXP, int NW, bool LIST>
__global__ void __launch_bounds__(NW * 64) k_czb_qr(const CzbArgs a)
{
    AMX_KERNEL_PROLOGUE(float, NR, NQ, NW, solver_lds_words(false, MAXP))
    (void)wmask;
    const float *tiles = reinterpret_cast<const float *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);
        if (threadIdx.x == 0) *ticket = (unsigned)nw_;
        stage_tile<float>(As, tiles + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
        __syncthreads();
        for (int k = wave; k < ck.count; k = next_ticket(ticket, lane))
            czb_voxel<NR, NQ, MAXP, true>(a, As, rs, rl, a.c.perm[ck.start + k], nullptr, lane);
    } else {
        const int cnt = *a.c.list_count;
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
            const int vox = a.c.list[it];
            __syncthreads();
            stage_tile<float>(As, tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride, words, words_pad - words);
            __syncthreads();
            czb_voxel<NR, NQ, MAXP, true>(a, As, rs, rl, vox, nullptr, lane);
        }
    }
}

// ------------------------------------------------------------------ the solvers themselves, batched (models.pyx:18: `from cyspams.interfaces
// cimport nnls, lasso`): one dictionary per voxel chosen by index, the coefficient vector x (and ||A x - y||) out.  RIDGE = false: nnls
// (lambda1 = lambda2 = 0), true: lasso.  fp64 dictionary tile in LDS, one wavefront per voxel (amx_solver.hpp).
struct BatchedArgs {
    FitCommon c;                  // tiles: double [n_dicts][nS][ldA]; lutidx: the dictionary of each voxel
    double *x;                    // [n_vox][n_atoms]
    double *rnorm;                // [n_vox] or null
};

template <int NR, int NQ, int MAXP, bool RIDGE, typename AT = double>
__device__ __forceinline__ void batched_voxel(const BatchedArgs &a, const AT *As, double *rs, double *rl, int vox, int lane)
{
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms;
    double yr[NR];
    const bool ok = load_rows<NR>(a.c, vox, nS, lane, yr);
    bool rowok[NR];
    double scl[NQ];
    unsigned long long allowed[NQ];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) rowok[rr] = (lane + kWave * rr) < nS;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int cnt = n_atoms - kWave * q;
        allowed[q] = cnt >= 64 ? ~0ull : (cnt > 0 ? ((1ull << cnt) - 1ull) : 0ull);
        scl[q] = 1.0;
    }
    double *dst = a.x + (size_t)vox * n_atoms;
    if (!ok) {                                                // non-finite signal: NaN out, never iterate
        for (int j = lane; j < n_atoms; j += kWave) dst[j] = __builtin_nan("");
        if (lane == 0 && a.rnorm) a.rnorm[vox] = __builtin_nan("");
        return;
    }
    NNSolver<NR, NQ, MAXP, RIDGE, AT> S;
    const int st = __builtin_amdgcn_readfirstlane(S.solve(As, ldA, nS, n_atoms, yr, rowok, scl, allowed, RIDGE ? a.c.lam1 : 0.0, RIDGE ? a.c.lam2 : 0.0, rs, rl, lane));
    if (st == kOverflow) {
        if (lane == 0) { const int k = atomicAdd(a.c.ovf_count, 1); a.c.ovf_list[k] = vox; }
        return;
    }
    if (st == kIterCap && lane == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (st > kIterCap && lane == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
    store_x_dense<NQ>(dst, n_atoms, lane, S.np, S.idx, S.x);
    if (a.rnorm) {
        S.residual(yr, RIDGE ? a.c.lam1 : 0.0);
        double rsq = 0.0;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) rsq += S.r[rr] * S.r[rr];
        rsq = wave_sum(rsq);
        if (lane == 0) a.rnorm[vox] = sqrt(rsq);
    }
}

template <int NR, int NQ, int MAXP, int NW, bool RIDGE, bool LIST, bool GT = false>
__global__ void __launch_bounds__(NW * 64) k_batched(const BatchedArgs a)
{
    AMX_KERNEL_PROLOGUE_GT(double, NR, NQ, NW, solver_lds_words(false, MAXP, RIDGE), GT)
    (void)wmask;
    const double *tiles = reinterpret_cast<const double *>(a.c.tiles);
    if (!LIST) {
        const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
        if (cid < 0) return;
        const Chunk ck = a.c.chunks[cid];
        unsigned *ticket = reinterpret_cast<unsigned *>(wm_all + nw_ * 4);
        if (threadIdx.x == 0) *ticket = (unsigned)nw_;
        using ATs = typename std::conditional<GT, gtile<double>, double>::type;
        const double *At = As;
        if constexpr (GT) At = tiles + (size_t)ck.dir * a.c.tile_stride;
        else stage_tile<double>(As, tiles + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
        __syncthreads();
        for (int k = wave; k < ck.count; k = next_ticket(ticket, lane))
            batched_voxel<NR, NQ, MAXP, RIDGE, ATs>(a, reinterpret_cast<const ATs *>(At), rs, rl, a.c.perm[ck.start + k], lane);
    } else {
        const int cnt = *a.c.list_count;
        for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
            const int vox = a.c.list[it];
            if constexpr (GT) {
                batched_voxel<NR, NQ, MAXP, RIDGE, gtile<double>>(a, reinterpret_cast<const gtile<double> *>(tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride), rs, rl, vox, lane);
            } else {
                __syncthreads();
                stage_tile<double>(As, tiles + (size_t)a.c.lutidx[vox] * a.c.tile_stride, words, words_pad - words);
                __syncthreads();
                batched_voxel<NR, NQ, MAXP, RIDGE>(a, As, rs, rl, vox, lane);
            }
        }
    }
}

template <typename AT>
static inline size_t fit_lds_bytes(int nS, int ldA, int NR, int NQ, int NW, int MAXP, bool gram = false, bool ridge = true, bool global_tile = false)
{
    const size_t words_pad = global_tile ? 0 : (((size_t)nS * ldA + kWave * NQ + 3) & ~(size_t)3);
    size_t b = (words_pad * sizeof(AT) + 15) & ~(size_t)15;
    b += (size_t)NW * NR * kWave * sizeof(double);
    b += (size_t)NW * solver_lds_words(gram, MAXP, ridge) * sizeof(double);
    b += (size_t)NW * 4 * sizeof(unsigned long long);
    b += 16;
    return b;
}

}  // namespace amx
