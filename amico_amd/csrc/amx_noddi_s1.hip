// amx_noddi_s1.hip -- NODDI solver stage 1 (models.pyx:911)
#include "amx_launch.hpp"
using namespace amx;

template <int NR>
static int go(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
#ifndef AMX_S1_MP
#define AMX_S1_MP 8
#endif
    constexpr int NQ = 3, MP = AMX_S1_MP, MB = 32;
#ifndef AMX_S1_NW
#define AMX_S1_NW 16
#endif
    constexpr int NW = AMX_S1_NW; // wavefronts per workgroup: as many as the register budget of this stage allows
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<1, NR, NQ, MP, NW, false>, k_noddi<1, NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP); }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB),
                       0, 2);
}

int amx_launch_noddi_s1(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : go<4>(ctx, a, pl, s);
}
