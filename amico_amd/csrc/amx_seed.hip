// amx_seed.hip -- support seeds of the NODDI NNLS stages (amx_seed.hpp): basis per orientation, projection, seed solver
#include "amx_launch.hpp"
#include "amx_seed.hpp"
using namespace amx;

// basis of the dominant column space of every orientation tile (once per dictionary upload)
int amx_build_basis(amx_ctx *ctx, amx_lut *lut)
{
    if (lut->n_atoms > 160 || lut->n_wm > 144) return AMX_OK;          // no seeds for this shape (the scans hold 160 atoms)
    // deflated columns in LDS when they fit next to Q, else in a global scratch block per workgroup (<= 4096 orientations per launch)
    const size_t lds_full = ((size_t)lut->nS * lut->ldA + (size_t)kSeedKD * lut->nS + 256) * sizeof(double);
    const bool in_lds = lds_full <= kLdsPerCU;
    const size_t lds = in_lds ? lds_full : ((size_t)kSeedKD * lut->nS + 256) * sizeof(double);
    const int batch = in_lds ? lut->ndirs : (lut->ndirs < 4096 ? lut->ndirs : 4096);
    double *Rg = nullptr;
    if (!in_lds) HIPCHK(ctx, hipMalloc((void **)&Rg, (size_t)batch * lut->nS * lut->ldA * sizeof(double) + 64));
    const size_t ub = (size_t)lut->ndirs * lut->nS * kSeedKD * sizeof(double);
    const size_t sb = (size_t)lut->ndirs * lut->n_atoms * kSeedKD * sizeof(double);
    HIPCHK(ctx, hipMalloc((void **)&lut->basis_U, ub + 64));
    HIPCHK(ctx, hipMalloc((void **)&lut->basis_S, sb + 64));
    HIPCHK(ctx, hipMalloc((void **)&lut->screen_S, (size_t)lut->ndirs * kSeedKD * kScreenLd * sizeof(float) + 64));
    HIPCHK(ctx, hipMemset(lut->screen_S, 0, (size_t)lut->ndirs * kSeedKD * kScreenLd * sizeof(float)));
    HIPCHK(ctx, hipMalloc((void **)&lut->screen_kappa, (size_t)lut->ndirs * sizeof(double) + 64));
    HIPCHK(ctx, hipMalloc((void **)&lut->screen_kappa0, (size_t)lut->ndirs * sizeof(double) + 64));
    int rc;
    if ((rc = set_lds(ctx, k_build_basis, lds))) return rc;
    for (int d0 = 0; d0 < lut->ndirs; d0 += batch) {
        const int nb = lut->ndirs - d0 < batch ? lut->ndirs - d0 : batch;
        hipLaunchKernelGGL(k_build_basis, dim3(nb), dim3(256), lds, nullptr, (const float *)lut->tiles, lut->tile_stride, lut->nS,
                           lut->ldA, lut->n_atoms, (const unsigned char *)nullptr, (const double *)nullptr, lut->basis_U, lut->basis_S, kSeedKD,
                           lut->screen_S, lut->screen_kappa, lut->screen_kappa0, Rg, d0);
    }
    // the LASSO stage's dictionary: DWI rows, column-normalised wm atoms (models.pyx:917-921), rank 8
    if (lut->gram_dwi) {
        HIPCHK(ctx, hipMalloc((void **)&lut->basis2_U, (size_t)lut->ndirs * lut->nS * kSeed2Ld * sizeof(double) + 64));
        HIPCHK(ctx, hipMalloc((void **)&lut->basis2_S, (size_t)lut->ndirs * lut->n_wm * kSeed2Ld * sizeof(double) + 64));
        HIPCHK(ctx, hipMalloc((void **)&lut->screen2_S, (size_t)lut->ndirs * kSeedKD * kScreenLd * sizeof(float) + 64));
        HIPCHK(ctx, hipMemset(lut->screen2_S, 0, (size_t)lut->ndirs * kSeedKD * kScreenLd * sizeof(float)));
        HIPCHK(ctx, hipMalloc((void **)&lut->screen2_kappa, (size_t)lut->ndirs * sizeof(double) + 64));
        HIPCHK(ctx, hipMalloc((void **)&lut->screen2_kappa0, (size_t)lut->ndirs * sizeof(double) + 64));
        for (int d0 = 0; d0 < lut->ndirs; d0 += batch) {
            const int nb = lut->ndirs - d0 < batch ? lut->ndirs - d0 : batch;
            hipLaunchKernelGGL(k_build_basis, dim3(nb), dim3(256), lds, nullptr, (const float *)lut->tiles, lut->tile_stride, lut->nS,
                               lut->ldA, lut->n_wm, (const unsigned char *)lut->rowdwi, (const double *)lut->colscale, lut->basis2_U, lut->basis2_S, kSeed2Ld,
                               lut->screen2_S, lut->screen2_kappa, lut->screen2_kappa0, Rg, d0);
        }
        // U2'iso: what x_iso takes out of the projected stage-2 signal (k_s2_prep, k_lasso_gcert)
        HIPCHK(ctx, hipMalloc((void **)&lut->u2iso, (size_t)lut->ndirs * kSeedKD * sizeof(double) + 64));
        hipLaunchKernelGGL(k_u2iso, dim3(lut->ndirs), dim3(64), 0, nullptr, (const float *)lut->tiles, lut->tile_stride, lut->nS, lut->ldA,
                           lut->n_atoms - 1, (const double *)lut->basis2_U, lut->u2iso);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    if (Rg) HIPCHK(ctx, hipFree(Rg));
    return AMX_OK;
}

// LASSO stage: y2~ = U2'y2 and the passive-set seeds (after stage 1: y2 needs x_iso)
int amx_launch_noddi_seed2(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s, bool have_ytil2)
{
    Seed2Args sa;
    memset(&sa, 0, sizeof sa);
    sa.y = a.c.y; sa.y32 = a.c.y32; sa.perm = pl.perm; sa.chunks = pl.chunks; sa.n_chunks = pl.n_chunks;
    sa.schunks = pl.schunks; sa.n_schunks = pl.n_chunks + 1;
    sa.tiles = (const float *)lut->tiles; sa.tile_stride = lut->tile_stride; sa.ldA = lut->ldA;
    sa.rowdwi = lut->rowdwi; sa.xiso = a.xiso; sa.Ub = lut->basis2_U; sa.Sb = lut->basis2_S;
    sa.ytil = (double *)ctx->ytil2.p; sa.seeds = (unsigned long long *)ctx->seeds2.p;
    sa.nS = lut->nS; sa.n_wm = lut->n_wm; sa.iso_atom = lut->n_atoms - 1; sa.is_exvivo = lut->is_exvivo;
    sa.lam1 = a.c.lam1; sa.lam2 = a.c.lam2;
    sa.trip_cap = ctx->opt_seed_tripcap[1];
    // supports of up to 24 atoms are certified lane-per-voxel where the third pass runs: the seed solver goes on to 26 atoms there
    // (and gets 2 atoms per trip: the trip cap grows with it); elsewhere 20 (the second pass ends at 18)
    sa.max_atoms = 20;
    if (ctx->opt_seed2_maxatoms > 0) sa.max_atoms = ctx->opt_seed2_maxatoms;
    else if (have_ytil2 && amx_gcert2_third(ctx, lut, !ctx->opt_no_gcert_wide)) { sa.max_atoms = 26; sa.trip_cap += 6; }
#ifdef AMX_STATS
    sa.stats = a.c.status + ST_SEED + 20;
#endif
#ifdef SEED2_TRACE
    sa.trace = (double *)ctx->seeds.p; hipMemsetAsync(ctx->seeds.p, 0, 8 * 8 * 80, s);
#endif
    const dim3 grid(((pl.max_chunks + 7) / 8) * 8);
    if (!have_ytil2) {
        if (lut->nS <= 128) hipLaunchKernelGGL(k_noddi_project2<2>, grid, dim3(1024), 0, s, sa);
        else if (lut->nS <= 256) hipLaunchKernelGGL(k_noddi_project2<4>, grid, dim3(1024), 0, s, sa);
        else hipLaunchKernelGGL(k_noddi_project2<8>, grid, dim3(512), 0, s, sa);
        AMX_TRACE(ctx, s, "projection of the clipped signals");
    }
    const size_t lds = ((size_t)lut->n_wm * (kSeed2KD + 1) + 8 + (size_t)9 * (kSeed2KD / 4) * 64 + (size_t)4 * (64 * (kSeed2KD + 1) + 64 * 3)) * sizeof(double);
    int rc;
    sa.gcount = pl.feed_set(FEED_SEED2); sa.n_gcount = pl.max_schunks;
    if (pl.seed2_occ2) {
        if ((rc = set_lds(ctx, k_lasso_seed<true>, lds))) return rc;
        hipLaunchKernelGGL(k_lasso_seed<true>, dim3(((pl.max_schunks + 7) / 8) * 8), dim3(64 * pl.seed2_waves), lds, s, sa);
    } else {
        if ((rc = set_lds(ctx, k_lasso_seed<false>, lds))) return rc;
        hipLaunchKernelGGL(k_lasso_seed<false>, dim3(((pl.max_schunks + 7) / 8) * 8), dim3(64 * pl.seed2_waves), lds, s, sa);
    }
    amx_note(ctx, pl.seed2_occ2 ? "k_lasso_seed<occ2>" : "k_lasso_seed");
    AMX_TRACE(ctx, s, "LASSO seed solver");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

static void fill(SeedArgs &sa, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, amx_ctx *ctx)
{
    memset(&sa, 0, sizeof sa);
    sa.y = a.c.y; sa.y32 = a.c.y32; sa.perm = pl.perm; sa.chunks = pl.chunks; sa.n_chunks = pl.n_chunks;
    sa.schunks = pl.schunks; sa.n_schunks = pl.n_chunks + 1;
    sa.Ub = lut->basis_U; sa.Sb = lut->basis_S;
    sa.ytil = (double *)ctx->ytil.p; sa.seeds = (unsigned long long *)ctx->seeds.p;
    sa.nS = lut->nS; sa.n_atoms = lut->n_atoms; sa.iso_atom = lut->n_atoms - 1;
    sa.dot_atom = lut->is_exvivo ? lut->n_atoms - 2 : -1;
    sa.trip_cap = ctx->opt_seed_tripcap[0];
#ifdef AMX_STATS
    sa.stats = a.c.status + ST_SEED + 4;
#endif
}

// LDS of k_noddi_gemm for a dictionary shape and K-steps: float32 tiles of atoms only, fp64 tiles for the rest, iso table, column scales
static size_t gemm_lds(int n_cols, int rows, int ks)
{
    const int mtf = n_cols / 16, mt = rows / 16;
    return (size_t)mtf * ks * 64 * sizeof(float) + ((size_t)(mt - mtf) * ks * 64 + 4 * ks + rows + 2) * sizeof(double);   // (+ the workgroup's next chunk)
}

// can the table kernels take this dictionary?  (K-steps of 4 samples: 25 or 40 per voxel; the scans of the seed solvers and
// certificates hold 160 atoms; one workgroup's operands must fit a CU's LDS)
// Protocols of more than 160 volumes take several launches, each over a window of <= 160 samples of every voxel (GemmArgs::k0, k1):
// 288 volumes = 2 windows of 144, 512 = 4 of 128.
static int gemm_passes(int nS) { return (nS + 159) / 160; }
static int gemm_window(int nS) { const int np = gemm_passes(nS); return ((nS + np - 1) / np + 3) & ~3; }   // samples per window, a multiple of 4
int amx_gemm_ksteps(const amx_lut *lut)
{
    if (lut->n_atoms > 160 || lut->n_wm > 144 || lut->nS > 512) return 0;
    // (every windowed launch is the KS = 40 build, whatever its window: the LDS size and this fit check must be those of the launched template --
    //  161 .. 200 volumes have windows of 84 .. 100 samples and were sized for 25 K-steps until round 6: operands beyond the allocation)
    const int ks = (gemm_passes(lut->nS) > 1 || gemm_window(lut->nS) > 100) ? 40 : 25;
    if (gemm_lds(lut->n_atoms, gemm_rows(lut->n_atoms), ks) > kLdsPerCU) return 0;
    return ks;
}

// C = [A | U | U2 | 1_b0]'Y of every voxel (k_noddi_gemm): block-wise table in ctx->cgemm; lasso: the clipped voxels' stage-2
// problem (y2, U2, scaled rows), compact, in ctx->cgemm2
int amx_launch_noddi_gemm(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s, bool lasso)
{
    GemmArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.y = a.c.y; ga.y32 = a.c.y32; ga.perm = pl.perm; ga.schunks = pl.schunks; ga.n_schunks = pl.n_chunks + 1;
    ga.tiles = (const float *)lut->tiles; ga.tile_stride = lut->tile_stride; ga.ldA = lut->ldA; ga.nS = lut->nS; ga.n_atoms = lut->n_atoms;
    ga.rows = gemm_rows(lut->n_atoms); ga.aux0 = lut->n_atoms;
    ga.Ub = lasso ? lut->basis2_U : lut->basis_U; ga.U2b = lasso ? nullptr : lut->basis2_U;
    ga.Cb = (double *)(lasso ? ctx->cgemm2.p : ctx->cgemm.p); ga.ytil = (double *)(lasso ? ctx->ytil2.p : ctx->ytil.p);
    ga.xiso = a.xiso; ga.rowdwi = lut->rowdwi; ga.colscale = lut->colscale; ga.iso_atom = lut->n_atoms - 1; ga.is_exvivo = lut->is_exvivo; ga.n_wm = lut->n_wm;
    if (lasso) { ga.clist = (const int *)ctx->clip.p; ga.ccount = pl.zcount(ZC_CLIP); }
    else { ga.gcount = pl.feed_set(FEED_GEMM); ga.n_gcount = pl.max_schunks; }
    const int ks = amx_gemm_ksteps(lut);
    const size_t lds = gemm_lds(lasso ? lut->n_wm : lut->n_atoms, ga.rows, ks);
    int rc;
    const dim3 grid(((pl.max_schunks + 7) / 8) * 8);
#define AMX_GEMM_GO(L, K, ...)                                                                     \
    do {                                                                                           \
        if ((rc = set_lds(ctx, (k_noddi_gemm<L, K, ##__VA_ARGS__>), lds))) return rc;              \
        hipLaunchKernelGGL((k_noddi_gemm<L, K, ##__VA_ARGS__>), grid, dim3(512), lds, s, ga);      \
    } while (0)
    const int mtf = (lasso ? lut->n_wm : lut->n_atoms) / 16;
    const int n_pass = gemm_passes(lut->nS), win = gemm_window(lut->nS);
    for (int p = 0; p < n_pass; p++) {
        ga.k0 = p * win; ga.k1 = (p + 1) * win < lut->nS ? (p + 1) * win : lut->nS;
        ga.accumulate = p > 0 ? 1 : 0; ga.last = p == n_pass - 1 ? 1 : 0;
        // (every window walks all voxels: its launch starts from fresh chunk counters)
        if (p > 0 && !lasso) HIPCHK(ctx, hipMemsetAsync(ga.gcount, 0, (size_t)(pl.max_schunks + 8) * sizeof(int), s));
        if (n_pass > 1) { if (lasso) AMX_GEMM_GO(true, 40, 0, true); else AMX_GEMM_GO(false, 40, 0, true); }      // windows: always the K-steps 40 build (amx_gemm_ksteps sizes the LDS for it)
        else if (ks == 25 && mtf == 9 && !lasso) AMX_GEMM_GO(false, 25, 9);          // the default dictionary: unrolled tile loop
        else if (ks == 25) { if (lasso) AMX_GEMM_GO(true, 25); else AMX_GEMM_GO(false, 25); }
        else if (ks == 40) { if (lasso) AMX_GEMM_GO(true, 40); else AMX_GEMM_GO(false, 40); }
        else return amx_bad(ctx, "k_noddi_gemm: unsupported dictionary shape");
    }
#undef AMX_GEMM_GO
    amx_note(ctx, lasso ? "k_noddi_gemm<lasso> (clipped voxels)" : (n_pass > 1 ? "k_noddi_gemm<windows>" : (ks == 25 && mtf == 9 ? "k_noddi_gemm<false,25,9>" : (ks == 25 ? "k_noddi_gemm<false,25>" : "k_noddi_gemm<false,40>"))));
    AMX_TRACE(ctx, s, lasso ? "stage-2 products of the clipped voxels on the matrix cores" : "A'y of every voxel on the matrix cores");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

// after stage 1: y2~ of every voxel from the table, the clipped voxels compacted (k_s2_prep), then their exact products
int amx_launch_noddi_s2prep(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    S2PrepArgs p;
    memset(&p, 0, sizeof p);
    p.perm = pl.perm; p.schunks = pl.schunks; p.n_schunks = pl.n_chunks + 1;
    p.Cb = (const double *)ctx->cgemm.p; p.rows = gemm_rows(lut->n_atoms); p.aux0 = lut->n_atoms;
    p.xiso = a.xiso; p.u2iso = lut->u2iso; p.ytil2 = (double *)ctx->ytil2.p;
    p.clist = (int *)ctx->clip.p; p.cslot = p.clist + pl.n; p.ccount = pl.zcount(ZC_CLIP);      // (counts: cleared with the plan)
    p.status = a.c.status; p.force_all = (!lut->s2_derive || ctx->opt_s2_exact) ? 1 : 0;
    hipLaunchKernelGGL(k_s2_prep, dim3(((pl.max_schunks + 7) / 8) * 8), dim3(256), 0, s, p);
    amx_note(ctx, "k_s2_prep");
    AMX_TRACE(ctx, s, "stage-2 signals from the table, clipped voxels compacted");
    HIPCHK(ctx, hipGetLastError());
    return amx_launch_noddi_gemm(ctx, lut, a, pl, s, true);
}

size_t amx_gcert2_leftover_offset(const Plan &pl, bool wide, bool third)
{
    return (wide && !third) ? amx_rlist_half(pl) : 0;      // (an even number of passes after the first ends in the first half again)
}
const int *amx_gcert2_leftover_counts(const Plan &pl, bool wide, bool third)
{
    return pl.zcount(!wide ? ZC_CERT2 : (third ? ZC_CERT2W3 : ZC_CERT2W));
}
// A third pass (supports of 19 .. 24 atoms, the triangle mostly in scratch) pays where a left-over voxel is expensive: shapes whose
// wavefront-per-voxel kernels read their tile from L2 (a 288-volume protocol leaves 6.8 % of the voxels after two passes: 6.5 of that
// fit's 24 ms went to k_noddi<4, .., GT>).  At 99 volumes it is a wash (round 3: 10.34 against 10.33 ms).  AMX_GCERT2_THIRD=0 / 1 forces.
// Round 6: the pass is launched for EVERY shape and decides per chunk (Gcert2Args::min_items): a chunk that still holds a block's worth of
// such supports is worth its tables, one that holds a dozen is handed on as it is.  AMX_GCERT2_THIRD=0: never (the round-5 default for tiles in
// LDS), 1: every chunk whatever it holds (the round-5 behaviour for global tiles, still their default).
// lambda1 is fixed while ||A2'y2|| grows with the number of stage-2 rows: beyond the default protocol's 90 the LASSO supports are denser (105
// volumes = 100 rows: 41 000 of 1 M voxels beyond 18 atoms, against 6 300) -- there the pass runs too, and decides per chunk.
bool amx_gcert2_third(const amx_ctx *ctx, const amx_lut *lut, bool wide)
{
    if (!wide || !(kGcert2Wide3 > kGcert2Wide)) return false;
    if (ctx->opt_gcert2_third >= 0) return ctx->opt_gcert2_third != 0;
    // (... from 600 000 voxels: the pass's kernel holds its 24 x 24 triangle mostly in scratch (2.6 KB per lane) and a block costs it ~0.4 ms whatever
    //  the call's size -- 105 volumes, tools/r06/a12.sh: 300 000 voxels 3.58 ms without, 3.70 with it (0.50 ms of certificates to save 0.38 of
    //  left-over kernel); 1 M voxels 8.59 -> 8.01 (0.72 to save 1.27))
    // (... and up to 128 volumes: at 150 the pass costs 0.5 ms of certificates to save 0.13 of left-over kernel -- most of what is left there holds
    //  more than 24 atoms, or fails for other reasons)
    return amx_noddi_tile_global(lut->nS, lut->ldA, lut->n_atoms) || (lut->n_dwi > 95 && lut->nS <= 128 && ctx->call_vox >= 600000);
}
int amx_gcert2_third_min_items(const amx_ctx *ctx, const amx_lut *lut)
{
    if (ctx->opt_gcert2_third == 1 || amx_noddi_tile_global(lut->nS, lut->ldA, lut->n_atoms)) return 0;
    return ctx->opt_gcert2_third_min;
}

// Gram-space certificates of the LASSO seeds (k_lasso_gcert): support bits of the voxels it settles, left-over lists for k_noddi<4>
int amx_launch_noddi_gcert2(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s, bool wide)
{
    const bool third = amx_gcert2_third(ctx, lut, wide);
    Gcert2Args g;
    memset(&g, 0, sizeof g);
    g.perm = pl.perm; g.schunks = pl.schunks; g.n_schunks = pl.n_chunks + 1;
    g.seeds2 = (const unsigned long long *)ctx->seeds2.p; g.cand8 = (unsigned long long *)ctx->seeds2.p; g.Cb = (const double *)ctx->cgemm.p; g.Cb2 = (const double *)ctx->cgemm2.p;
    g.cslot = (const int *)ctx->clip.p + pl.n; g.u2iso = lut->u2iso; g.rows = gemm_rows(lut->n_atoms); g.aux0 = lut->n_atoms;
    g.gram = lut->gram_dwi; g.colscale = lut->colscale; g.ldG = lut->ldG; g.n_atoms = lut->n_atoms; g.n_wm = lut->n_wm;
    g.iso_atom = lut->n_atoms - 1; g.dot_atom = lut->is_exvivo ? lut->n_atoms - 2 : -1;
    g.Sb = lut->basis2_S; g.kappa0 = lut->screen2_kappa0; g.lam1 = a.c.lam1; g.lam2 = a.c.lam2;
    g.supp = a.supp; g.xiso = a.xiso; g.done = (unsigned char *)ctx->done.p;
    g.rlist = (int *)ctx->rlist.p; g.rcount = pl.zcount(ZC_CERT2);
    g.gcount = pl.feed_set(FEED_CERT2); g.n_gcount = pl.max_schunks;
    if (a.c.xdbg) g.xdbg = a.c.xdbg;
#ifdef AMX_STATS
    g.stats = a.c.status + ST_SEED + 36;
#endif
    const size_t lds = ((size_t)lut->n_wm * kSeedLd + 2 + 2 * ((lut->n_wm + 1) & ~1) + (size_t)9 * (kSeedKD / 4) * 64 + (size_t)4 * 64 * 16) * sizeof(double);
    int rc;
    if ((rc = set_lds(ctx, (k_lasso_gcert<kGcert2Max, false>), lds))) return rc;
    hipLaunchKernelGGL((k_lasso_gcert<kGcert2Max, false>), dim3(((pl.max_schunks + 7) / 8) * 8), dim3(64 * pl.seed_waves), lds, s, g);
    amx_note(ctx, "k_lasso_gcert<11>");
    AMX_TRACE(ctx, s, "Gram-space certificates of the LASSO seeds");
    HIPCHK(ctx, hipGetLastError());
    if (wide) {
        // second pass: supports of 12 .. 18 atoms from the left-over lists; its own left-overs in the second half of the buffer
        g.rlist_in = g.rlist; g.rcount_in = g.rcount;
        g.rlist = (int *)ctx->rlist.p + amx_rlist_half(pl); g.rcount = pl.zcount(ZC_CERT2W);
        if (third) { g.qcount = pl.zcount(ZC_CERT2Q); g.q_hi = kGcert2Wide3; }
#ifdef AMX_STATS
        g.stats = a.c.status + ST_SEED + 48;
#endif
        if ((rc = set_lds(ctx, (k_lasso_gcert<kGcert2Wide, true>), lds))) return rc;
        hipLaunchKernelGGL((k_lasso_gcert<kGcert2Wide, true>), dim3(((pl.max_schunks + 7) / 8) * 8), dim3(64 * pl.seed_waves), lds, s, g);
        amx_note(ctx, "k_lasso_gcert<18,wide>");
        AMX_TRACE(ctx, s, "Gram-space certificates of the LASSO seeds, supports of 12 .. 18 atoms");
        HIPCHK(ctx, hipGetLastError());
        if (third) {
            // third pass: supports beyond the second pass from the second pass's left-overs, back into the first half of the buffer
            g.rlist_in = g.rlist; g.rcount_in = g.rcount;
            g.rlist = (int *)ctx->rlist.p; g.rcount = pl.zcount(ZC_CERT2W3);
            g.min_items = amx_gcert2_third_min_items(ctx, lut);
            g.qcount_in = g.min_items > 0 ? pl.zcount(ZC_CERT2Q) : nullptr; g.qcount = nullptr;
#ifdef AMX_STATS
            g.stats = nullptr;
#endif
            if ((rc = set_lds(ctx, (k_lasso_gcert<kGcert2Wide3, true, kGcert2Wide>), lds))) return rc;
            hipLaunchKernelGGL((k_lasso_gcert<kGcert2Wide3, true, kGcert2Wide>), dim3(((pl.max_schunks + 7) / 8) * 8), dim3(64 * pl.seed_waves), lds, s, g);
            amx_note(ctx, "k_lasso_gcert<24,wide,18>");
            AMX_TRACE(ctx, s, "Gram-space certificates of the LASSO seeds, supports beyond the second pass");
            HIPCHK(ctx, hipGetLastError());
        }
    }
    return AMX_OK;
}

// Gram-space certificates of the NNLS seeds, one voxel per lane (k_nnls_gcert): done[pos] = 1 for the voxels it settles
// (+ the rescue pass over its left-over lists; *list_off: where in ctx->rlist the lists for the wavefront-per-voxel kernel are)
int amx_launch_noddi_gcert(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s, int stage, size_t *list_off, const int **count_out)
{
    *list_off = 0;
    GcertArgs g;
    memset(&g, 0, sizeof g);
    g.perm = pl.perm; g.schunks = pl.schunks; g.n_schunks = pl.n_chunks + 1;
    g.seeds = (const unsigned long long *)ctx->seeds.p; g.Cb = (const double *)ctx->cgemm.p;
    g.rows = gemm_rows(lut->n_atoms); g.aux0 = lut->n_atoms;
    g.gram = lut->gram; g.ldG = lut->ldG; g.n_atoms = lut->n_atoms; g.n_wm = lut->n_wm; g.nS = lut->nS;
    g.iso_atom = lut->n_atoms - 1; g.dot_atom = lut->is_exvivo ? lut->n_atoms - 2 : -1; g.n_maps = a.n_maps;
    g.Sb = lut->basis_S; g.kappa0 = lut->screen_kappa0; g.supp = a.supp; g.icvf = lut->icvf; g.kappa = lut->kappa;
    // (forked fit, stage 3: the LASSO stage's left-over lists -- either half of the buffer -- are still being read on the side stream:
    //  this stage's lists go behind them)
    const size_t lbase = (stage == 3 && a.fork_l2) ? 2 * amx_rlist_half(pl) : 0;
    g.fork_skip = (stage == 3 && a.fork_l2) ? 1 : 0;
    g.done = (unsigned char *)ctx->done.p; g.rlist = (int *)ctx->rlist.p + lbase; g.rcount = pl.zcount(stage == 1 ? ZC_CERT1 : ZC_CERT3);
    *list_off = lbase;
    *count_out = g.rcount;
    g.gcount = pl.feed_set(stage == 1 ? FEED_CERT1 : FEED_CERT3); g.n_gcount = pl.max_schunks;
    g.xiso = a.xiso; g.est = a.est; g.rmse = a.rmse; g.nrmse = a.nrmse; g.mod = a.mod;
    if (a.c.xdbg) g.xdbg = a.c.xdbg;
#ifdef AMX_STATS
    g.stats = a.c.status + ST_SEED + 24 + (stage == 1 ? 0 : 6);
#endif
    const size_t lds = ((size_t)lut->n_atoms * kSeedLd + 2 + (size_t)10 * (kSeedKD / 4) * 64 + (size_t)4 * 64 * 16) * sizeof(double);
    const dim3 grid(((pl.max_schunks + 7) / 8) * 8);
    int rc;
    // a second look at the seeds a lane can mend itself (k_nnls_gcert<.., REPAIR>): where a left-over voxel is expensive and the seeds are
    // wrong more often -- the shapes whose tile is read from L2 (AMX_GCERT_REPAIR=0 / 1 forces)
    const bool repair = ctx->opt_gcert_repair >= 0 ? ctx->opt_gcert_repair != 0 : amx_noddi_tile_global(lut->nS, lut->ldA, lut->n_atoms);
    if (stage == 1 && repair) {
        if ((rc = set_lds(ctx, (k_nnls_gcert<1, false, true>), lds))) return rc;
        hipLaunchKernelGGL((k_nnls_gcert<1, false, true>), grid, dim3(64 * pl.seed_waves), lds, s, g);
    } else if (stage == 1) {
        if ((rc = set_lds(ctx, k_nnls_gcert<1>, lds))) return rc;
        hipLaunchKernelGGL(k_nnls_gcert<1>, grid, dim3(64 * pl.seed_waves), lds, s, g);
    } else if (repair) {
        if ((rc = set_lds(ctx, (k_nnls_gcert<3, false, true>), lds))) return rc;
        hipLaunchKernelGGL((k_nnls_gcert<3, false, true>), grid, dim3(64 * pl.seed_waves), lds, s, g);
    } else {
        if ((rc = set_lds(ctx, k_nnls_gcert<3>, lds))) return rc;
        hipLaunchKernelGGL(k_nnls_gcert<3>, grid, dim3(64 * pl.seed_waves), lds, s, g);
    }
    amx_note(ctx, stage == 1 ? (repair ? "k_nnls_gcert<1,repair>" : "k_nnls_gcert<1>") : (repair ? "k_nnls_gcert<3,repair>" : "k_nnls_gcert<3>"));
    AMX_TRACE(ctx, s, "Gram-space certificates");
    HIPCHK(ctx, hipGetLastError());
    // (ex-vivo dictionaries leave 7 % of the voxels instead of 5 and 2.5 % -- the dot atom makes more supports ill-conditioned -- and
    //  gain from 1 M voxels: 108 -> 112 M voxels/s; deciding on the device from the first pass's count was tried: the launch that only
    //  hands the lists on costs every other call 1 %)
    // (batches of one host-buffer call all take the path the whole call's size asks for, as make_plan does for the seed solvers' builds:
    //  host and device entry points settle the same voxels with the same arithmetic)
    // (shapes whose tile does not fit the LDS -- an HCP-style protocol -- run it at every size: a left-over voxel costs their
    //  wavefront-per-voxel kernels ten times what it costs the LDS variants, and the lane that corrects a support reads 8 atoms, not the tile)
    // (round 6: ... and protocols of more than 128 volumes, whose wavefront-per-voxel kernels hold four signal rows per lane -- a left-over voxel
    //  costs them 37 ns against 14 at 99 volumes: 150 volumes, 1 M voxels 11.23 -> 10.65 ms with the pass, stage-3 left-overs 38 219 -> 6 429;
    //  at 99 - 105 volumes it still loses below 2 M voxels, SNR 50 included: profiles/r06_protocols_ab.txt)
    if ((ctx->in_host_fit ? ctx->host_total_vox : (int64_t)pl.n) >= (lut->is_exvivo ? ctx->opt_rescue_from / 4 : ctx->opt_rescue_from) ||
        (lut->nS > 128 && ctx->opt_rescue_from > 0 && !ctx->opt_rescue_from_set) ||
        amx_noddi_tile_global(lut->nS, lut->ldA, lut->n_atoms)) {
        // second pass (large calls: below ~2 M voxels the launch costs more than the wavefront-per-voxel kernel saves -- 1 M voxels
        // 8.08 -> 8.24 ms with it, 4 M 24.99 -> 24.38): the supports refused for conditioning, corrected with the signal itself;
        // what is left goes to the second half
        g.rlist_in = g.rlist; g.rcount_in = g.rcount;
        g.rlist = (int *)ctx->rlist.p + lbase + amx_rlist_half(pl); g.rcount = pl.zcount(stage == 1 ? ZC_RESC1 : ZC_RESC3);
        *count_out = g.rcount;
        g.y = a.c.y; g.y32 = a.c.y32; g.tiles = (const float *)lut->tiles; g.tile_stride = lut->tile_stride; g.ldA = lut->ldA;
        const size_t tile = (size_t)lut->nS * lut->ldA * sizeof(float);
        g.tile_in_lds = lds + tile <= kLdsPerCU ? 1 : 0;
        const size_t lds2 = lds + (g.tile_in_lds ? tile : 0);
        if (stage == 1) {
            if ((rc = set_lds(ctx, (k_nnls_gcert<1, true>), lds2))) return rc;
            hipLaunchKernelGGL((k_nnls_gcert<1, true>), grid, dim3(64 * pl.seed_waves), lds2, s, g);
        } else {
            if ((rc = set_lds(ctx, (k_nnls_gcert<3, true>), lds2))) return rc;
            hipLaunchKernelGGL((k_nnls_gcert<3, true>), grid, dim3(64 * pl.seed_waves), lds2, s, g);
        }
        amx_note(ctx, stage == 1 ? "k_nnls_gcert<1,rescue>" : "k_nnls_gcert<3,rescue>");
        AMX_TRACE(ctx, s, "Gram-space certificates, rescue pass");
        HIPCHK(ctx, hipGetLastError());
        *list_off = lbase + amx_rlist_half(pl);
    }
    return AMX_OK;
}

int amx_launch_noddi_project(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s)
{
    SeedArgs sa; fill(sa, lut, a, pl, ctx);
    const dim3 grid(((pl.max_chunks + 7) / 8) * 8);
    if (lut->nS <= 128) hipLaunchKernelGGL(k_noddi_project<2>, grid, dim3(1024), 0, s, sa);
    else if (lut->nS <= 256) hipLaunchKernelGGL(k_noddi_project<4>, grid, dim3(1024), 0, s, sa);
    else hipLaunchKernelGGL(k_noddi_project<8>, grid, dim3(512), 0, s, sa);
    AMX_TRACE(ctx, s, "projection onto the orientation bases");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

int amx_launch_noddi_seed(amx_ctx *ctx, const amx_lut *lut, const NoddiArgs &a, const Plan &pl, hipStream_t s, int stage)
{
    SeedArgs sa; fill(sa, lut, a, pl, ctx);
    sa.supp = stage == 3 ? a.supp : nullptr;
    if (stage == 3 && a.cand_lists && a.seeds2 != nullptr) { sa.cand8 = a.seeds2; sa.cdone = (const unsigned char *)ctx->done.p; sa.fork_skip = a.fork_l2; }
    sa.trip_cap = ctx->opt_seed_tripcap[stage == 1 ? 0 : 2];
    // S for the per-lane gathers + ticket; stage 1 adds S in MFMA operand order (10 x 3 x 64) and a residual block per wavefront
    const size_t lds = (size_t)lut->n_atoms * kSeedLd * sizeof(double) + 64 +
                       (stage == 1 ? ((size_t)10 * (kSeedKD / 4) * 64 + (size_t)4 * 64 * (kSeedKD + 1)) * sizeof(double)
                                   : (size_t)256 * kSeed3ListRow);                  // stage 3: the lanes' candidate byte lists
    const dim3 grid(((pl.max_schunks + 7) / 8) * 8);
    int rc;
    sa.gcount = pl.feed_set(stage == 1 ? FEED_SEED1 : FEED_SEED3); sa.n_gcount = pl.max_schunks;
    if (stage == 1 && pl.seed_occ2) {
        if ((rc = set_lds(ctx, (k_nnls_seed<1, 8, true>), lds))) return rc;
        hipLaunchKernelGGL((k_nnls_seed<1, 8, true>), grid, dim3(64 * pl.seed1_waves), lds, s, sa);
    } else if (stage == 1) {
        if ((rc = set_lds(ctx, (k_nnls_seed<1, 8>), lds))) return rc;
        hipLaunchKernelGGL((k_nnls_seed<1, 8>), grid, dim3(64 * pl.seed1_waves), lds, s, sa);
    } else {
        if ((rc = set_lds(ctx, (k_nnls_seed<3, 6>), lds))) return rc;
        hipLaunchKernelGGL((k_nnls_seed<3, 6>), grid, dim3(64 * pl.seed_waves), lds, s, sa);
    }
    amx_note(ctx, stage == 1 ? (pl.seed_occ2 ? "k_nnls_seed<1,8,occ2>" : "k_nnls_seed<1,8>") : "k_nnls_seed<3,6>");
    AMX_TRACE(ctx, s, "seed solver");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}
