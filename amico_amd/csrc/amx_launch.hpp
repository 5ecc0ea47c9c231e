// amx_launch.hpp -- launch helpers used by the per-model units.  Every unit is compiled on
// its own (make -j) and carries its own gfx950 code object; nothing is linked device-side.
#pragma once
#include "amx_host.hpp"

template <typename K>
static int set_lds(amx_ctx *ctx, K kern, size_t bytes)
{
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return AMX_OK;
}

// main pass over the orientation chunks + re-run of the voxels whose passive set overflowed
constexpr size_t kLdsPerCU = 160 * 1024;

// NODDI wavefront-per-voxel kernels: does this shape run the global-tile variants (amx_kernels.hpp: GT)?  The LDS variants hold
// <= 256 rows (4 per lane), <= 192 atoms (3 per lane) and need the float32 tile + the blocks of at least two wavefronts in LDS
// (the widest: the LASSO stage's Gram solver with 32 passive atoms; the re-run kernels with 64)
static inline bool amx_noddi_tile_global(int nS, int ldA, int n_atoms)
{
    if (nS > 256 || n_atoms > 192) return true;
    const int NR = nS <= 128 ? 2 : 4;
    return amx::fit_lds_bytes<float>(nS, ldA, NR, 3, 2, 32, true) + (size_t)amx::kSeedKD * amx::kScreenLd * sizeof(float) > kLdsPerCU ||
           amx::fit_lds_bytes<float>(nS, ldA, NR, 3, 1, 64, true) > kLdsPerCU;
}

// LDSF(nw) -> dynamic LDS bytes of the main kernel with nw wavefronts per workgroup
template <int NW, typename Args, typename KM, typename KL, typename LDSF>
static int launch_pair(amx_ctx *ctx, Args &a, const Plan &pl, hipStream_t s, KM km, KL kl, LDSF ldsf,
                       size_t lds_list, int slot, int ev, const char *name = "wavefront-per-voxel solver", bool big_after = false)
{
    amx_note(ctx, name);
    int rc;
    int nw = NW;                               // fewer wavefronts per workgroup for long protocols
    while (nw > 1 && ldsf(nw) > kLdsPerCU) nw >>= 1;
    const size_t lds_main = ldsf(nw);
    if (lds_main > kLdsPerCU || lds_list > kLdsPerCU) {
        ctx->err = "dictionary tile does not fit the 160 KB LDS of a CU (nS x n_atoms too large)";
        return AMX_E_BADARG;
    }
    // overflow lists: slot 0 .. 2 = the three stages (counts at misc[4 + slot], lists of n ints each), second level behind them (misc[12], list 3).
    // A launch on the side stream of a forked fit (ctx->side_launch; round 6) must not share a list with the main stream's kernels that may
    // run beside it: its stage-3 pass takes slot 3 (misc[7], list 4) and its re-run kernels count into misc[13] (list 5)
    if (ctx->side_launch && slot == 2) slot = 3;
    a.c.ovf_count = pl.ovf_count + slot;
    a.c.ovf_list = pl.ovf_list + (size_t)(slot < 3 ? slot : 4) * pl.n;
    a.c.list = a.c.ovf_list;
    a.c.list_count = a.c.ovf_count;
    if ((rc = set_lds(ctx, km, lds_main))) return rc;
    if ((rc = set_lds(ctx, kl, lds_list))) return rc;
    rec(ctx, ev, s);
    hipLaunchKernelGGL(km, dim3(((pl.max_chunks + 7) / 8) * 8), dim3(nw * 64), lds_main, s, a);   // see xcd_chunk()
    AMX_TRACE(ctx, s, "solver main pass");
    Args b = a;
    b.c.ovf_count = pl.ovf_count + (ctx->side_launch ? 9 : 8);
    b.c.ovf_list = pl.ovf_list + (size_t)(ctx->side_launch ? 5 : 3) * pl.n;
    // (NODDI's Gram-space LASSO stage: what does not fit the re-run kernel's 64 atoms either is not an error -- k_noddi_lasso_big, amx_big.hip,
    //  takes it from a list of its own: misc[14], list 6)
    if (big_after) { b.c.ovf_count = pl.ovf_count + 10; b.c.ovf_list = pl.ovf_list + (size_t)6 * pl.n; }
    hipLaunchKernelGGL(kl, dim3(kListGrid), dim3(64), lds_list, s, b);
    AMX_TRACE(ctx, s, "solver re-run pass");
    rec(ctx, ev + 1, s);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}
