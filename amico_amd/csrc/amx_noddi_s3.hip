// amx_noddi_s3.hip -- NODDI solver stage 3 (models.pyx:929-967)
#include "amx_launch.hpp"
using namespace amx;

template <int NR>
static int go(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
#ifndef AMX_S3_MP
#define AMX_S3_MP 8
#endif
    constexpr int NQ = 3, MP = AMX_S3_MP, MB = 32;
#ifndef AMX_S3_NW
#define AMX_S3_NW 12      // (measured with the seeded path: 16 wavefronts -> 128 VGPRs and 223 spilled registers, 5.9 ms; 12 -> 168 VGPRs, 3.7 ms)
#endif
    constexpr int NW = AMX_S3_NW; // wavefronts per workgroup: as many as the register budget of this stage allows
    const size_t scr = (a.scr_S && a.seeds) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;   // screening table (amx_solver.hpp)
    // small calls: two workgroups per CU (amx_noddi_s1.hip)
    if (a.rlist != nullptr && (long long)pl.n < ctx->opt_left_small[2] &&
        2 * (fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 4, MP, false, false) + scr) <= kLdsPerCU)
        return launch_pair<4>(ctx, a, pl, s, k_noddi<3, NR, NQ, MP, 4, false, float>, k_noddi<3, NR, NQ, MB, 1, true>,
                               [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; },
                               fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 2, 6, "k_noddi<3> (left-overs; small-call build: two workgroups per CU)");
    // protocols of 129 .. 256 volumes: eight wavefronts on the float32 tile instead of twelve with 697 spilled registers (amx_noddi_s1.hip)
    if constexpr (NR == 4) {
        if (a.rlist != nullptr && !ctx->opt_no_nr4_nw8 && fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 8, MP, false, false) + scr <= kLdsPerCU &&
            !(fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, NW, MP, false, false) + scr <= kLdsPerCU))
            return launch_pair<8>(ctx, a, pl, s, k_noddi<3, NR, NQ, MP, 8, false, float>, k_noddi<3, NR, NQ, MB, 1, true>,
                                   [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; },
                                   fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 2, 6, "k_noddi<3> (left-overs; 8 wavefronts, float32 tile)");
    }
    // fp64 tile in LDS when it fits next to the per-wavefront blocks (99 x 145: 115 KB + 16 x 2.3 KB of 160 KB): the
    // fp32 -> fp64 conversions of the tile reads are then paid once per chunk.  AMX_TILE_F32=1: the fp32 tile.
    {
        if (fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, NW, MP, false, false) + scr <= kLdsPerCU && !ctx->opt_tile_f32)
            return launch_pair<NW>(ctx, a, pl, s, k_noddi<3, NR, NQ, MP, NW, false, double>, k_noddi<3, NR, NQ, MB, 1, true>,
                                   [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; },
                                   fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 2, 6, a.rlist ? "k_noddi<3> (left-overs)" : "k_noddi<3> (all voxels)");
    }
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<3, NR, NQ, MP, NW, false>, k_noddi<3, NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false),
                       2, 6, a.rlist ? "k_noddi<3> (left-overs)" : "k_noddi<3> (all voxels)");
}

// shapes beyond the LDS variants (see amx_noddi_s1.hip): the tile read where it lies
static int go_global(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NR = 8, NQ = 4, MP = 8, MB = 32, NW = 4;
    const size_t scr = (a.scr_S && a.seeds) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<3, NR, NQ, MP, NW, false, float, true>, k_noddi<3, NR, NQ, MB, 1, true, float, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false, true) + scr; },
                           fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false, true), 2, 6, a.rlist ? "k_noddi<3> (left-overs)" : "k_noddi<3> (all voxels)");
}

int amx_launch_noddi_s3(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    if (amx_noddi_tile_global(a.c.nS, a.c.ldA, a.c.n_atoms)) return go_global(ctx, a, pl, s);
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : go<4>(ctx, a, pl, s);
}
