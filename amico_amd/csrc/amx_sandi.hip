// amx_sandi.hip -- SANDI solver kernel (models.pyx:1567-1619)
#include "amx_launch.hpp"
using namespace amx;

template <int NR>
static int go(amx_ctx *ctx, SandiArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 1, MP = 16, MB = 64;
    constexpr int NW = 8;
    return launch_pair<NW>(ctx, a, pl, s, k_sandi<NR, NQ, MP, NW, false>, k_sandi<NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, MP); }, fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, 1, MB), 0, 2, "k_sandi (wavefront per voxel)");
}

int amx_launch_sandi(amx_ctx *ctx, SandiArgs &a, const Plan &pl, hipStream_t s)
{
    if (amx_use_lane_solver(ctx, a.c.n_atoms, a.c.lam2)) return amx_launch_sandi_small(ctx, a, pl, s);
    return a.c.nS <= 64 ? go<1>(ctx, a, pl, s) : go<2>(ctx, a, pl, s);
}
