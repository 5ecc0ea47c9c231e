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
    const size_t scr = (a.scr_S && a.seeds) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;   // screening table (amx_solver.hpp)
    // Left-overs of the Gram certificates (seeded chain): few voxels per chunk, and the ones that get here are the hard ones --
    // room for 12 passive atoms and 12 wavefronts (168 VGPRs, 16 spilled, against 128 / 64 with 8 atoms and 16 wavefronts): nothing
    // overflows into the one-wavefront re-run kernel any more (it cost 0.16 ms for five voxels per million), 1 M voxels 0.82 -> 0.65 ms
    // SMALL calls (round 6): a chunk is an orientation, so even a 50 000-voxel call launches ~500 workgroups here, each with a handful of
    // left-over voxels; the build above holds a whole CU's LDS (154 KB: the fp64 tile) -- 500 workgroups on 256 CUs are TWO rounds, each as long
    // as one hard voxel's Lawson-Hanson (~140 us).  With the float32 tile and 4 wavefronts a workgroup takes 77 KB: two per CU, one round.
    // Same arithmetic (the tile's values are widened when read instead of when staged): bit-identical maps.  AMX_LEFT_SMALL=a,b,c: call sizes
    // (voxels) below which stage 1 / LASSO / stage 3 take their small builds.
    if (a.rlist != nullptr && (long long)pl.n < ctx->opt_left_small[0] &&
        2 * (fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 4, 12, false, false) + scr) <= kLdsPerCU)
        return launch_pair<4>(ctx, a, pl, s, k_noddi<1, NR, NQ, 12, 4, false, float>, k_noddi<1, NR, NQ, MB, 1, true>,
                               [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, 12, false, false) + scr; },
                               fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 0, 2, "k_noddi<1> (left-overs of k_nnls_gcert<1>; small-call build: two workgroups per CU)");
    // Protocols of 129 .. 256 volumes (four signal rows per lane; the fp64 tile of 150 x 145 does not fit the LDS): the left-over lists used to
    // fall through to the 16-wavefront build below -- 128 registers, 203 of them spilled.  Eight wavefronts (two per SIMD, 256 registers: no
    // spills) on the float32 tile (round 6; AMX_LEFT_NR4_NW8=0: the old build)
    if constexpr (NR == 4) {
        if (a.rlist != nullptr && !ctx->opt_no_nr4_nw8 && fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 8, 12, false, false) + scr <= kLdsPerCU &&
            !(fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, 12, 12, false, false) + scr <= kLdsPerCU))
            return launch_pair<8>(ctx, a, pl, s, k_noddi<1, NR, NQ, 12, 8, false, float>, k_noddi<1, NR, NQ, MB, 1, true>,
                                   [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, 12, false, false) + scr; },
                                   fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 0, 2, "k_noddi<1> (left-overs of k_nnls_gcert<1>; 8 wavefronts, float32 tile)");
    }
    if (a.rlist != nullptr && fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, 12, 12, false, false) + scr <= kLdsPerCU && !ctx->opt_tile_f32)
        return launch_pair<12>(ctx, a, pl, s, k_noddi<1, NR, NQ, 12, 12, false, double>, k_noddi<1, NR, NQ, MB, 1, true>,
                               [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, 12, false, false) + scr; },
                               fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 0, 2, a.rlist ? "k_noddi<1> (left-overs of k_nnls_gcert<1>)" : "k_noddi<1> (all voxels)");
    // fp64 tile in LDS when it fits next to the per-wavefront blocks (99 x 145: 115 KB + 16 x 2.3 KB of 160 KB): the
    // fp32 -> fp64 conversions of the tile reads are then paid once per chunk.  AMX_TILE_F32=1: the fp32 tile.
    {
        if (fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, NW, MP, false, false) + scr <= kLdsPerCU && !ctx->opt_tile_f32)
            return launch_pair<NW>(ctx, a, pl, s, k_noddi<1, NR, NQ, MP, NW, false, double>, k_noddi<1, NR, NQ, MB, 1, true>,
                                   [&](int nw) { return fit_lds_bytes<double>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; },
                                   fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false), 0, 2, a.rlist ? "k_noddi<1> (left-overs of k_nnls_gcert<1>)" : "k_noddi<1> (all voxels)");
    }
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<1, NR, NQ, MP, NW, false>, k_noddi<1, NR, NQ, MB, 1, true>,
                       [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false) + scr; }, fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false),
                       0, 2, a.rlist ? "k_noddi<1> (left-overs of k_nnls_gcert<1>)" : "k_noddi<1> (all voxels)");
}

// Any shape the reference's loop takes (models.pyx:825-861: any nS, any n_wm): protocols of more than 256 volumes, dictionaries of more
// than 192 atoms, and tiles that do not fit a CU's LDS (288 x 145 float32 = 167 KB) run the SAME solver with the tile read where it
// lies (k_noddi<..., GT = true>: 8 rows and 4 atoms per lane: nS <= 512, n_atoms <= 256), room for 12 passive atoms, 4 wavefronts.
static int go_global(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    constexpr int NR = 8, NQ = 4, MP = 12, MB = 32, NW = 4;
    const size_t scr = (a.scr_S && a.seeds) ? (size_t)kSeedKD * kScreenLd * sizeof(float) : 0;
    return launch_pair<NW>(ctx, a, pl, s, k_noddi<1, NR, NQ, MP, NW, false, float, true>, k_noddi<1, NR, NQ, MB, 1, true, float, true>,
                           [&](int nw) { return fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, nw, MP, false, false, true) + scr; },
                           fit_lds_bytes<float>(a.c.nS, a.c.ldA, NR, NQ, 1, MB, false, false, true), 0, 2, a.rlist ? "k_noddi<1> (left-overs of k_nnls_gcert<1>)" : "k_noddi<1> (all voxels)");
}

int amx_launch_noddi_s1(amx_ctx *ctx, NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    if (amx_noddi_tile_global(a.c.nS, a.c.ldA, a.c.n_atoms)) return go_global(ctx, a, pl, s);
    return a.c.nS <= 128 ? go<2>(ctx, a, pl, s) : go<4>(ctx, a, pl, s);
}
