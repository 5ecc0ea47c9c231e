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

int amx_launch_batched(amx_ctx *ctx, BatchedArgs &a, const Plan &pl, hipStream_t s, bool ridge)
{
    if (ridge) return a.c.nS <= 128 ? go<2, true>(ctx, a, pl, s) : go<4, true>(ctx, a, pl, s);
    return a.c.nS <= 128 ? go<2, false>(ctx, a, pl, s) : go<4, false>(ctx, a, pl, s);
}
