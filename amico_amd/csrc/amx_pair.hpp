// amx_pair.hpp -- NODDI stage kernels with TWO voxels per wavefront (amx_pair_solver.hpp): the unregularised solves
// of models.pyx:911 (stage 1) and :929-967 (stage 3 + maps).  Same arguments, same intermediates (x_iso, support bit
// set, overflow list) as k_noddi<1/3,...>, which stays the path for protocols beyond 32*NR volumes / 32*NQ atoms and
// re-runs the voxels whose passive set outgrows MAXP.
#pragma once
#include "amx_kernels.hpp"
#include "amx_pair_solver.hpp"

namespace amx {

// a wavefront draws NV consecutive voxels of its chunk (wave-uniform control flow, see next_ticket)
template <int NV>
__device__ __forceinline__ int next_ticket_n(unsigned *ticket, int lane)
{
    const unsigned off = (unsigned)(uintptr_t)ticket;
    const unsigned inc = lane == 0 ? (unsigned)NV : 0u;
    unsigned old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(off), "v"(inc) : "memory");
    return __builtin_amdgcn_readfirstlane((int)old);
}

template <int STAGE, int LPV, int NR, int NQ, int MAXP>
__device__ __forceinline__ void noddi_pair(const NoddiArgs &a, PairNNLS<LPV, NR, NQ, MAXP> &S, const float *As, double *rs,
                                           double *Rl, const Chunk &ck, int k0, int lane)
{
    static_assert(STAGE == 1 || STAGE == 3, "the LASSO stage has its own solver");
    using G_ = Grp<LPV>;
    constexpr int NV = 64 / LPV;                            // voxels per wavefront
    const int l = lane & (LPV - 1), sl = lane & 15, g = lane / LPV;
    const int nS = a.c.nS, ldA = a.c.ldA, n_atoms = a.c.n_atoms, n_wm = a.n_wm;
    const int iso_atom = n_atoms - 1, dot_atom = a.is_exvivo ? n_atoms - 2 : -1;
    const bool has = k0 + g < ck.count;
    const int vox = has ? a.c.perm[ck.start + k0 + g] : 0;
    double yr[NR];
    bool finite = true;
    {
        const double *yv = a.c.y + (size_t)vox * nS;
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int i = l + LPV * rr;
            yr[rr] = (has && i < nS) ? yv[i] : 0.0;
            finite = finite && (fabs(yr[rr]) <= 1.79769313486231570e308);
        }
    }
    const bool bad = has && G_::any(!finite, lane);
    const bool ok = has && !bad;
    if (bad) {   // non-finite signal: NaN maps, never iterate (SURVEY 8(b) error convention)
        const double nan = __builtin_nan("");
        if (STAGE == 1 && l < 2) a.xiso[(size_t)vox * 2 + l] = nan;
        if (STAGE == 3 && l < a.n_maps) a.est[(size_t)vox * a.n_maps + l] = nan;
        if (STAGE == 3 && l == 0) {
            if (a.rmse) a.rmse[vox] = nan;
            if (a.nrmse) a.nrmse[vox] = nan;
            if (a.mod) { a.mod[(size_t)vox * 2] = nan; a.mod[(size_t)vox * 2 + 1] = nan; }
        }
    }
    unsigned allowed = 0u;
    if (STAGE == 1) {
#pragma unroll
        for (int q = 0; q < NQ; q++) allowed |= ((l + LPV * q) < n_atoms ? 1u : 0u) << q;
    } else {
        // models.pyx:929-936: support of the LASSO solution plus iso (and dot); atom j = l + LPV q is bit j & 63 of word j >> 6
        const unsigned long long *sp = a.supp + (size_t)vox * 4;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const unsigned long long w = ok ? sp[(LPV * q) >> 6] : 0ull;
            unsigned b = (unsigned)(w >> (l + ((LPV * q) & 63))) & 1u;
            const int j = l + LPV * q;
            if (j == iso_atom || (dot_atom >= 0 && j == dot_atom)) b = 1u;
            allowed |= (j < n_atoms ? b : 0u) << q;
        }
    }
    const double *gdir = a.gram ? a.gram + (size_t)ck.dir * n_atoms * a.ldG : nullptr;
    S.solve(As, ldA, nS, n_atoms, yr, allowed, ok, rs, Rl, lane, gdir, a.ldG);

    const int st = S.status;
    if (ok && st == kOverflow) {
        if (l == 0) { const int k = atomicAdd(a.c.ovf_count, 1); a.c.ovf_list[k] = vox; }
    }
    const bool fin = ok && st != kOverflow;
    if (fin && st == kIterCap && l == 0) atomicAdd(&a.c.status[ST_ITCAP], 1);
    if (fin && st > kIterCap && l == 0) { atomicAdd(&a.c.status[ST_GUARD], 1); a.c.status[ST_GUARDVOX] = vox * 8 + st; }
#ifdef AMX_STATS
    if (lane == 0) { atomicAdd(&a.c.status[ST_EXACT + STAGE - 1], S.n_exact); atomicAdd(&a.c.status[ST_GRAM + STAGE - 1], S.n_gram); }
    if (fin && l == 0) atomicAdd(&a.c.status[ST_ITERS + STAGE - 1], S.iters);
#endif
    const bool act = sl < S.np;
    if (a.c.xdbg && fin) {
        double *dst = a.c.xdbg + ((size_t)vox * 3 + (STAGE == 1 ? 0 : 2)) * n_atoms;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int j = l + LPV * q;
            if (j < n_atoms) dst[j] = 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (act && S.idx >= 0) dst[S.idx] = S.x;       // (with two rows per voxel both store the same value)
    }
    if constexpr (STAGE == 1) {
        const double xi = slot_sum((act && S.idx == iso_atom) ? S.x : 0.0);
        const double xd = slot_sum((act && S.idx == dot_atom) ? S.x : 0.0);
        if (fin && l == 0) { a.xiso[(size_t)vox * 2] = xi; a.xiso[(size_t)vox * 2 + 1] = xd; }
    } else {
        double rsq = 0.0, ysq = 0.0;
        if (a.c.flags & 3u) {
            double r[NR];
            S.residual(yr, r);
#pragma unroll
            for (int rr = 0; rr < NR; rr++) { rsq += r[rr] * r[rr]; ysq += yr[rr] * yr[rr]; }
            rsq = G_::sum(rsq); ysq = G_::sum(ysq);
        }
        // models.pyx:945-967
        const double xs = act ? S.x : 0.0;
        const bool iswm = act && S.idx < n_wm;
        const double sum_atoms = slot_sum(xs) + 1e-16;
        const double sum_wm = slot_sum(iswm ? xs / sum_atoms : 0.0) + 1e-16;
        double f1 = 0.0, f2 = 0.0, k1 = 0.0;
        if (iswm) {
            const float ic = a.icvf[S.idx];
            const double t = xs / sum_atoms / sum_wm;
            f1 = (double)ic * t;
            f2 = (double)((float)(1.0 - (double)ic)) * t;
            k1 = (double)a.kappa[S.idx] * t;
        }
        f1 = slot_sum(f1); f2 = slot_sum(f2); k1 = slot_sum(k1);
        const double ndi = f1 / (f1 + f2 + 1e-16);
        const double odi = odi_from_kappa(k1);
        const double fwf = slot_sum((act && S.idx == iso_atom) ? xs : 0.0) / sum_atoms;
        const double dot = slot_sum((act && S.idx == dot_atom) ? xs : 0.0) / sum_atoms;
        if (fin && l == 0) {
            double *e = a.est + (size_t)vox * a.n_maps;
            e[0] = ndi; e[1] = odi; e[2] = fwf;
            if (a.is_exvivo) e[3] = dot;
            if (a.rmse) a.rmse[vox] = sqrt(rsq / (double)nS);                       // models.pyx:47-54
            if (a.nrmse) a.nrmse[vox] = (ysq > 1e-16) ? sqrt(rsq / ysq) : 0.0;      // models.pyx:58-71
            if (a.mod) { const double tf = 1.0 - fwf; a.mod[(size_t)vox * 2] = ndi * tf; a.mod[(size_t)vox * 2 + 1] = odi * tf; }
        }
    }
}

template <int LPV, int NR, int NQ, int MAXP>
static inline size_t pair_lds_bytes(int nS, int ldA, int NW)
{
    const size_t words_pad = ((size_t)nS * ldA + LPV * NQ + 3) & ~(size_t)3;
    size_t b = (words_pad * sizeof(float) + 15) & ~(size_t)15;
    b += (size_t)NW * (64 / LPV) * (PairNNLS<LPV, NR, NQ, MAXP>::kRsWords + PairNNLS<LPV, NR, NQ, MAXP>::kRlWords) * sizeof(double);
    return b + 16;
}

template <int STAGE, int LPV, int NR, int NQ, int MAXP, int NW>
__global__ void __launch_bounds__(NW * 64) k_noddi_pair(const NoddiArgs a)
{
    using Solver = PairNNLS<LPV, NR, NQ, MAXP>;
    constexpr int NV = 64 / LPV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int words = a.c.nS * a.c.ldA;
    const int words_pad = (words + LPV * NQ + 3) & ~3;
    float *As = reinterpret_cast<float *>(smem_p);
    double *ws = reinterpret_cast<double *>(smem_p + (((size_t)words_pad * sizeof(float) + 15) & ~(size_t)15));
    double *rs = ws + (size_t)(wave * NV + lane / LPV) * (Solver::kRsWords + Solver::kRlWords);
    double *Rl = rs + Solver::kRsWords;
    unsigned *ticket = reinterpret_cast<unsigned *>(ws + (size_t)nw * NV * (Solver::kRsWords + Solver::kRlWords));
    const int cid = xcd_chunk((int)blockIdx.x, *a.c.n_chunks);
    if (cid < 0) return;
    const Chunk ck = a.c.chunks[cid];
    if (threadIdx.x == 0) *ticket = (unsigned)(nw * NV);
    stage_tile<float>(As, reinterpret_cast<const float *>(a.c.tiles) + (size_t)ck.dir * a.c.tile_stride, words, words_pad - words);
    __syncthreads();
    Solver S;
    S.init_once();
    for (int k = wave * NV; k < ck.count; k = next_ticket_n<NV>(ticket, lane)) noddi_pair<STAGE, LPV, NR, NQ, MAXP>(a, S, As, rs, Rl, ck, k, lane);
}

}  // namespace amx
