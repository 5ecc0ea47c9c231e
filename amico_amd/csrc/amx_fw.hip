// amx_fw.hip -- FreeWater solver kernel (models.pyx:1231-1276)
#include "amx_launch.hpp"
using namespace amx;

template <int NR>
static int go(amx_ctx *ctx, FwArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 1, MP = 16, MB = 64;
    constexpr int NW = 8;
    return launch_pair<NW>(ctx, a, pl, s, k_freewater<NR, NQ, MP, NW, false>, k_freewater<NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP); }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB), 0, 2, "k_freewater (wavefront per voxel)");
}

int amx_launch_fw(amx_ctx *ctx, FwArgs &a, const Plan &pl, hipStream_t s)
{
    if (amx_use_lane_solver(ctx, a.c.n_atoms, a.c.lam2)) return amx_launch_fw_small(ctx, a, pl, s);
    // (8 rows per lane: protocols of up to 512 volumes -- the <= 64-atom tile still fits the LDS)
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : (a.c.nS <= 256 ? go<4>(ctx, a, pl, s) : go<8>(ctx, a, pl, s));
}
