// amx_batched.hip -- the solvers the reference binds (cyspams.interfaces.nnls / lasso, models.pyx:18), batched over voxels
#include "amx_launch.hpp"
using namespace amx;

template <int NR, bool RIDGE>
static int go(amx_ctx *ctx, BatchedArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 3, MP = 16, MB = 48;      // passive set of the main pass / of the one-wavefront re-run pass (48: its triangular factors still fit next to a 99 x 145 fp64 tile)
    constexpr int NW = 8;
    return launch_pair<NW>(ctx, a, pl, s, k_batched<NR, NQ, MP, NW, RIDGE, false>, k_batched<NR, NQ, MB, 1, RIDGE, true>,
                           [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, RIDGE); },
                           fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, RIDGE), 0, 2);
}

// dictionaries beyond the LDS variants (m > 256 samples, n > 192 atoms, or an fp64 tile larger than a CU's LDS): the same solver with
// the tile read where it lies (k_batched<..., GT = true>: 8 rows / 4 atoms per lane: m <= 512, n <= 256)
template <bool RIDGE>
static int go_global(amx_ctx *ctx, BatchedArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NR = 8, NQ = 4, MP = 16, MB = 48, NW = 4;
    return launch_pair<NW>(ctx, a, pl, s, k_batched<NR, NQ, MP, NW, RIDGE, false, true>, k_batched<NR, NQ, MB, 1, RIDGE, true, true>,
                           [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, RIDGE, true); },
                           fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, RIDGE, true), 0, 2);
}

bool amx_batched_tile_global(int m, int ldA, int n)
{
    if (m > 256 || n > 192) return true;
    return fit_lds_bytes<double>(m, ldA, m <= 128 ? 2 : 4, 3, 1, 48, false, true) > kLdsPerCU;
}

int amx_launch_batched(amx_ctx *ctx, BatchedArgs &a, const Plan &pl, hipStream_t s, bool ridge)
{
    if (amx_batched_tile_global(a.c.nS, a.c.ldA, a.c.n_atoms)) return ridge ? go_global<true>(ctx, a, pl, s) : go_global<false>(ctx, a, pl, s);
    if (ridge) return a.c.nS <= 128 ? go<2, true>(ctx, a, pl, s) : go<4, true>(ctx, a, pl, s);
    return a.c.nS <= 128 ? go<2, false>(ctx, a, pl, s) : go<4, false>(ctx, a, pl, s);
}
