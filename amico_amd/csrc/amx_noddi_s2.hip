// amx_noddi_s2.hip -- NODDI solver stage 2, the LASSO (models.pyx:914-926)
#include "amx_launch.hpp"
using namespace amx;

// QR (A-space) variant: any lambda2 > 0
template <int NR>
static int go_qr(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 3, MP = 20, MB = 64;
    constexpr int NW = 8;   // wavefronts per workgroup: as many as the register budget of this stage allows
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<2, NR, NQ, MP, NW, false>, k_noddi<2, NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP); }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB),
                       1, 4, a.rlist ? "k_noddi<4|2> (left-overs of k_lasso_gcert)" : "k_noddi<4|2> (all voxels)");
}

// Gram-space variant (amx_gram_solver.hpp): needs the Gram matrices and a ridge that bounds cond(H)
template <int NR>
static int go_gram(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NQ = 3, MP = 20, MB = 64;
    constexpr int NW = AMX_S2_NW;
    const size_t scr = (a.scr2_S && a.seeds2) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;   // screening table (amx_gram_solver.hpp)
    if (a.rlist != nullptr && a.seeds2 != nullptr) {
        // left-over lists of the Gram certificates: few voxels, mostly seeds of more than 16 atoms -- half the wavefronts with
        // room for 32 atoms each, so that practically nothing is left for the (slow, one wavefront per voxel) re-run kernel
        constexpr int MPL = 32, NWL = AMX_S2_NW / 2;
        // small calls: two workgroups per CU (amx_noddi_s1.hip) -- two wavefronts with room for 32 atoms each and no screening table (the
        // certificate of a seed then takes the exact sweep of the dual vector: same decisions, 9 KB of LDS less): 78 KB
        if ((long long)pl.n < ctx->opt_left_small[1] && 2 * fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 2, MPL, true) <= kLdsPerCU) {
            NoddiArgs b = a;
            b.scr2_S = nullptr;
            const int rc = launch_pair<2>(ctx, b, pl, s, k_noddi<4, NR, NQ, MPL, 2, false>, k_noddi<4, NR, NQ, MB, 1, true>,
                                          [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MPL, true); }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, true),
                                          1, 4, "k_noddi<4> (left-overs of k_lasso_gcert; small-call build: two workgroups per CU)", true);
            return rc;
        }
        return launch_pair<NWL>(ctx, a, pl, s, k_noddi<4, NR, NQ, MPL, NWL, false>, k_noddi<4, NR, NQ, MB, 1, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MPL, true) + scr; }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, true),
                           1, 4, a.rlist ? "k_noddi<4|2> (left-overs of k_lasso_gcert)" : "k_noddi<4|2> (all voxels)", true);
    }
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<4, NR, NQ, MP, NW, false>, k_noddi<4, NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, true) + scr; }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, true),
                       1, 4, a.rlist ? "k_noddi<4|2> (left-overs of k_lasso_gcert)" : "k_noddi<4|2> (all voxels)", true);
}

// shapes beyond the LDS variants (see amx_noddi_s1.hip): the tile read where it lies; passive sets of up to 32 atoms in the main
// pass, 48 in the re-run pass
static int go_global(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s, bool gram)
{
    constexpr int NR = 8, NQ = 4, NW = 4;
    if (gram) {
        constexpr int MP = 32, MB = 64;
        const size_t scr = (a.scr2_S && a.seeds2) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;
        return launch_pair<NW>(ctx, a, pl, s, k_noddi<4, NR, NQ, MP, NW, false, float, true>, k_noddi<4, NR, NQ, MB, 1, true, float, true>,
                               [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, true, true, true) + scr; },
                               fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, true, true, true), 1, 4, a.rlist ? "k_noddi<4|2> (left-overs of k_lasso_gcert)" : "k_noddi<4|2> (all voxels)", true);
    }
    constexpr int MP = 20, MB = 32;       // (A-space QR: the factor lives in registers -- 32 x 8 rows per lane is what a wavefront holds)
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<2, NR, NQ, MP, NW, false, float, true>, k_noddi<2, NR, NQ, MB, 1, true, float, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, true, true); },
                           fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, true, true), 1, 4, a.rlist ? "k_noddi<4|2> (left-overs of k_lasso_gcert)" : "k_noddi<4|2> (all voxels)");
}

int amx_launch_noddi_s2(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    const bool gram = a.gram_dwi != nullptr && a.c.lam2 >= 1e-5 && !ctx->opt_lasso_qr;
    int rc;
    // lambda1 = 0 (a pure ridge: set_solver(lambda1=0, ...)): the optimum holds most of the dictionary -- every voxel would walk Lawson-Hanson
    // to 20, then 64 atoms and overflow twice on its way to the solver that can hold it: straight there (AMX_BIG_ALL=0: the long way, diagnosis)
    if (gram && a.c.lam1 == 0.0 && a.rlist == nullptr && a.n_wm > 64 && !ctx->opt_no_big_all)
        return amx_launch_noddi_big(ctx, a, pl, s, nullptr, nullptr, (int)pl.n);
    if (amx_noddi_tile_global(a.c.nS, a.c.ldA, a.c.n_atoms)) rc = go_global(ctx, a, pl, s, gram);
    else if (gram) rc = a.c.nS <= 128 ? go_gram<2>(ctx, a, pl, s) : go_gram<4>(ctx, a, pl, s);
    else return a.c.nS <= 128 ? go_qr<2>(ctx, a, pl, s) : go_qr<4>(ctx, a, pl, s);
    if (rc || !gram) return rc;
    // supports of more than 64 atoms (a weak lambda1: the optimum is dense): the slow exact solver, from the re-run kernel's own overflow list
    return amx_launch_noddi_big(ctx, a, pl, s, pl.ovf_list + (size_t)6 * pl.n, pl.ovf_count + 10, 0);
}
