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
    if (a.c.lam2 < 1e-6) return a.c.nS <= 128 ? go_qr<2>(ctx, a, pl, s) : go_qr<4>(ctx, a, pl, s);
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : go<4>(ctx, a, pl, s);
}
