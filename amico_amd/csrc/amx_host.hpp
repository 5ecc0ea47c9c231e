// amx_host.hpp -- host-side context shared by amx_api.hip and the per-model launch units.
#pragma once
#include "../../include/amico_amd.h"
#include "amx_kernels.hpp"
#include "amx_stage.hpp"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef AMX_CHUNK
#define AMX_CHUNK 256
#endif
constexpr int kChunk = AMX_CHUNK;        // voxels of one orientation per workgroup
constexpr int kListGrid = 512;     // workgroups of the large-MAXP re-run pass
constexpr int kEv = 20;           // event pairs: 0 whole call, 1-3 NODDI stage kernels, 4 small-model solver, 5-7 NODDI GEMM + seed + certificate kernels of stages 1 / 2 / 3, 8 k_nnls_seed<1> alone, 9 k_lasso_seed alone

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
};

struct amx_ctx {
    int device = 0;
    int n_cu = 256;                // compute units of the device (persistent-grid launches)
    std::string err;
    // stream-ordered workspace (grow-only)
    DevBuf big;                    // factor blocks of k_noddi_lasso_big for dictionaries of more than 176 candidate atoms (amx_big.hip)
    DevBuf lutidx, perm, counts, dir_start, cursor, chunks, misc, xiso, supp, ovf, cproj, ytil, seeds, schunks, ytil2, seeds2, cgemm, done, rlist, cgemm2, clip, feed;
    DevBuf hy, hdirs, hest, hrmse, hnrmse, hextra;   // staging for the host-pointer entry points
    int *status_d = nullptr;       // ST_WORDS ints
    int *status_h = nullptr;       // pinned mirror (+16 words: copy of the misc counters)
    int profiling = 0;             // 0 off, 1 every event pair of a call, 2 + w: pair w only (amx_set_profiling)
    int64_t host_total_vox = 0;    // voxels of the whole host-buffer call while its batches are enqueued
    int64_t call_vox = 0;          // voxels of the call being enqueued (the whole host-buffer call for its batches): size-dependent path choices made below noddi_fit_dev
    int64_t call_total_vox = 0;    // amx_set_call_voxels: the host-buffer calls on this ctx are shards of a call of this many voxels (0: they are the call)
    bool in_host_fit = false;      // the host-buffer entry points report progress per batch themselves
    hipEvent_t ev[kEv];
    bool ev_valid[kEv];
    int64_t stats[4] = {0, 0, 0, 0};
    int64_t seed_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // amx_last_seed_stats
    int64_t uncert_vox[3] = {0, 0, 0};   // ... of which a stage ran without its Gram-space certificate (shape gate, AMX_NO_GCERT)
    int64_t seeded_vox = 0;        // voxels enqueued on the seed -> certificate chain since the last amx_sync_status
    int64_t vox_base = 0;          // index of the first voxel of the batch being enqueued (chunked host entry points)
    double *dbg_x = nullptr;       // AMX_F_DEBUG_X destination (caller-owned device buffer, amx_set_debug_x)
    void (*progress)(int64_t, int64_t, void *) = nullptr;   // amx_set_progress
    void *progress_user = nullptr;
    DevBuf wy;                     // float64 copy of float32 device signals for the lane kernels that read float64 only (amx_*_fit_device_f32)
    DevBuf hy32;                   // float32 signals of the *_fit_f32 entry points (and of float64 host signals that are float32 values: amx_stage.hpp)
    amx_stage::Pool *stage = nullptr;  // host threads + pinned slots of the lossless float64 -> float32 transport (made at the first large float64 host call)
    std::thread stage_thread;      // makes the pool beside the dictionary upload (prefetch_stage_pool); joined by the first host-buffer fit that needs it
    amx_stage::Pool *stage_bg = nullptr;
    bool stage_bg_started = false;
    bool host_native32 = false;    // set by a model's host entry point whose kernels read float32 signals in place (fit_host then skips k_widen)
    const float *host_y32 = nullptr;   // the current batch's float32 signals in HBM (fit_host -> the model's enqueue), or null: float64 in the staging buffer
    bool opt_host_no_native32 = false; // AMX_HOST_NATIVE32=0: float32 batches are widened on the device first (the round-5 behaviour)
    bool stage_failed = false;     // the pool could not be made: host signals are copied as they are
    int host_narrowed = 0;         // batches of the last host-buffer call that travelled as float32 (amx_last_host_narrowed)
    hipStream_t hs = nullptr;      // non-blocking compute streams of the chunked host entry points: batches alternate
    hipStream_t hs2 = nullptr;     // between the two, so the tail of one batch's kernels is filled by the next batch's
    hipEvent_t hev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t up_ev = nullptr;    // recorded on the null stream behind a batch's uploads; the batch's compute stream waits for it
    // second workspace set for the batch in flight on the other stream (swap_work exchanges it with the named buffers)
    DevBuf alt[22];
    void swap_work()
    {
        DevBuf *named[22] = {&lutidx, &perm, &counts, &dir_start, &cursor, &chunks, &misc, &xiso, &supp, &ovf, &cproj, &ytil, &seeds, &schunks, &ytil2, &seeds2, &cgemm, &done, &rlist, &cgemm2, &clip, &feed};
        for (int i = 0; i < 22; i++) { DevBuf t = *named[i]; *named[i] = alt[i]; alt[i] = t; }
        work_idx ^= 1;
    }
    // switches read ONCE, at amx_ctx_create (environment): diagnosis / A-B only
    // switches below: environment variables read ONCE, at amx_ctx_create (diagnosis / A-B tools; the defaults are the product path)
    bool opt_no_gram = false;          // AMX_NO_GRAM=1: no Gram matrices at upload (the solvers sweep the tile instead)
    bool opt_lasso_qr = false;         // AMX_LASSO_QR=1: NODDI stage 2 by the A-space QR solver whatever lambda2
    bool opt_cold_start = false;       // AMX_COLD_START=1: FreeWater / SANDI / CZB start from the empty passive set
    bool opt_host_one_shot = false;    // AMX_HOST_ONE_SHOT=1: host-buffer entry points upload everything, then fit
    bool opt_host_one_stream = false;  // AMX_HOST_ONE_STREAM=1: pipelined host path on one stream
    bool opt_host_late_results = false; // AMX_HOST_LATE_RESULTS=1: the results of a host-buffer call go home in one copy after the last batch (diagnosis)
    long long opt_host_pipeline_from = 393217;   // AMX_HOST_PIPELINE_FROM: host-buffer calls of fewer voxels upload everything, then fit (>= 262144).  From 3 x 131 072 voxels a call has two batches -- the first short -- and the second copy hides behind the first fit: 400 000 voxels 7.63 -> 6.85 ms (float32 signals 6.94 -> 6.08), 500 000 8.66 -> 7.92 (8.30 -> 7.51); was 524 288 while a float64 copy took twice as long (profiles/r05c_host_transport.txt, section 9)
    bool opt_host_no_narrow = false;   // AMX_HOST_NARROW=0: float64 host signals are always copied as they are (amx_stage.hpp)
    int opt_host_threads = 12;         // AMX_HOST_THREADS: host threads that narrow + send float64 host signals (1 .. 64; at most half the logical CPUs)
    long long opt_host_ramp = 131072;  // AMX_HOST_RAMP: voxels of the first pipelined batch (its copy is the only one nothing hides; 0 = equal batches)
    long long opt_host_batch = 393216; // AMX_HOST_BATCH: voxels per pipelined batch (>= 131072, multiple of 4)
    bool opt_tile_f32 = false;         // AMX_TILE_F32=1: NNLS stages keep the float32 tile in LDS
    bool opt_fw_proj_valu = false;     // AMX_FW_PROJ_VALU=1: FreeWater projection without the matrix cores
    bool opt_fw_no_fuse = false;       // AMX_FW_NO_FUSE=1: FreeWater by the projection + solver kernel pair instead of k_freewater_fused
    bool opt_sandi_atom_space = false; // AMX_SANDI_ATOM_SPACE=1: SANDI 6 x 15 by the atom-space lane kernel
    bool opt_prep_scalar = false;      // AMX_PREP_SCALAR=1: the streaming preparation kernel with one voxel per lane (4-byte loads) instead of four
    bool opt_prep_no_direct = false;   // AMX_PREP_NO_DIRECT=1: k_prep_gather stages the planes through registers (32 loads in flight) instead of global -> LDS loads
    bool opt_prep_tile = false;        // AMX_PREP_TILE=1: signal preparation always through the transposition tile
    bool opt_lut_regs = false;         // AMX_LUT_REGS=1: LUT resampling with register operands
    bool opt_no_refill = false;        // AMX_NO_REFILL=1: FreeWater by k_freewater_lane (one solve per lane and pass)
    bool opt_wave_per_voxel = false;   // AMX_WAVE_PER_VOXEL=1: small models by the wavefront-per-voxel kernels
    int opt_refill_chunk = 0;          // AMX_REFILL_CHUNK: voxels per workgroup of k_freewater_refill (0 = by problem size)
    bool opt_no_chunk_order = false; // AMX_NO_CHUNK_ORDER=1: the chunks of the second plan stay in orientation order (default: longest first)
    int opt_seed2_maxatoms = 0;     // AMX_SEED2_MAXATOMS=n: atoms at which the LASSO seed solver gives a voxel up (default 20, 26 with a third certificate pass; <= 30)
    int opt_gcert_repair = -1;      // AMX_GCERT_REPAIR=0 / 1: never / always the NNLS certificates' second look at a mendable seed (default: where the tile is read from L2)
    int opt_gcert2_third_min = 16;  // AMX_GCERT2_THIRD_MIN: list entries a chunk must hold for the third LASSO certificate pass to work on it (tiles in LDS; 0 with global tiles)
    bool opt_rescue_from_set = false;   // AMX_RESCUE_FROM given: the caller's threshold alone decides
    int opt_gcert2_third = -1;      // AMX_GCERT2_THIRD=0 / 1: never / always a third LASSO certificate pass (default: where the tile is read from L2)
    bool opt_no_gcert_wide = false; // AMX_NO_GCERT_WIDE=1: no second Gram-certificate pass for LASSO supports of 13 .. 16 atoms
    int64_t opt_rescue_from = 2000000;   // AMX_RESCUE_FROM=n: calls of n voxels and more run the rescue pass of the NNLS certificates (k_nnls_gcert<., true>)
    bool opt_no_gcert = false;     // AMX_NO_GCERT=1: every seed is certified by the wavefront-per-voxel kernels (true residual)
    bool opt_no_screen = false;    // AMX_NO_SCREEN=1: certify seeds with the full exact sweep of the dual vector
    bool opt_s2_exact = false;     // AMX_S2_EXACT=1: every voxel's stage-2 products by the exact pass (k_noddi_gemm<true>), none derived from the stage-1 table
    bool opt_no_seed = false;      // AMX_NO_SEED=1: Lawson-Hanson from the empty set in the NNLS stages (the round-2 path)
    long long opt_seed_min_voxels = 22528; // AMX_SEED_MIN_VOXELS: smaller calls run the wavefront-per-voxel kernels on all voxels (the seeded chain of ~16 kernels has a floor of ~1.5 ms; measured, tools/r04/round8.sh: 20 000 voxels 1.58 against 1.49 ms, 25 000 voxels 1.60 against 1.78, 40 000 1.70 against 2.34)
    long long opt_seed_occ2_from = 65536; // AMX_SEED_OCC2_FROM: calls of at least this many voxels run k_nnls_seed<1> at two wavefronts per SIMD (a third more time per trip, twice the wavefronts: wins when the kernel is throughput bound -- 1 M voxels 2.80 -> 2.14 ms --, loses when the longest voxel's path bounds it: 50 000 voxels 0.54 -> 0.73 ms; with four wavefronts per workgroup the crossover sits between 200 000 and 300 000 voxels)
    long long opt_seed2_occ2_from = 65536; // AMX_SEED2_OCC2_FROM: the same for k_lasso_seed (200 000 voxels: 0.50 -> 0.42 ms, 1 M: 1.67 -> 1.27 ms)
    // AMX_SEED_TRIPCAP=a,b,c: trips after which k_nnls_seed<1> / k_lasso_seed / k_nnls_seed<3> give a voxel up (no seed: it goes
    // to the left-over kernels).  A lane kernel lasts as long as its slowest voxel, and the slowest are a handful: of 1 M bench
    // voxels 12 need more than 32 stage-1 trips (mean ~10), yet with the old cap of 64 they held the kernel 0.4 ms longer.
    // Measured (tools/r04/tripcap2.sh; 50 000 / 200 000 / 1 M voxels, fit in ms): 64,64,64 2.09 / 3.31 / 8.17; 28,24,12 1.74 / 2.94 /
    // 7.66; below 20 / 18 / 8 the left-over kernels get more voxels than the shorter tails are worth
    int opt_seed_tripcap[3] = {20, 20, 10};     // (28, 24, 12 before the normalised entering rule shortened the paths, 24, 24, 10 until the stage-1 solver handed its support on and the LASSO left-over solver started from the seed: profiles/r05b_tripcaps.txt)
    bool opt_no_hard_first = false; // AMX_NO_HARD_FIRST=1: the left-over kernels of the NNLS stages walk their lists in the order the certificates wrote them
    int opt_seed_waves = 0;        // AMX_SEED_WAVES: wavefronts per workgroup of the lane kernels (0 = by the number of chunks, make_plan)
    int opt_seed_stages = 7;       // AMX_SEED_STAGES: bit 0 = seed stage 1, bit 1 = seed stage 3, bit 2 = seed the LASSO stage
    // AMX_FORK (round 6; bit 0 / bit 1): the left-over kernels of stage 1 / of the LASSO stage run on a SIDE stream beside the next stage's
    // lane kernels (noddi_fit_dev).  Bit 1 is a correct fit (the forked voxels skip the stage-3 lane kernels and end in k_noddi<3> on the side
    // stream); bit 0 is a TIMING PROBE only (the stage-2 lane kernels read the x_iso the previous call left for those voxels).
    int opt_fork = 0;
    int opt_fork_cus = 0;           // AMX_FORK_CUS=n: the side stream may use n compute units only (hipExtStreamCreateWithCUMask; 0 = no mask)
    int opt_fork_prio = 0;          // AMX_FORK_PRIO=1: the side stream at the highest priority
    hipStream_t fork_s[2] = {nullptr, nullptr};          // (one per workspace set: work_idx)
    hipEvent_t fork_ev[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    int work_idx = 0;               // which of the two workspace sets the named buffers are (swap_work)
    // AMX_LEFT_SMALL=a,b,c: calls of fewer voxels run the left-over kernels of stage 1 / LASSO / stage 3 in their small-call builds (float32 tile, 4
    // wavefronts, two workgroups per CU: amx_noddi_s1.hip).  Measured (tools/r06/a05.sh; 50 000 / 100 000 / 200 000 / 300 000 / 500 000 / 1 M voxels, fit
    // in ms, round-5 builds 1.441 / 1.817 / 2.386 / 2.885 / 4.000 / 6.421): stage 1 alone 1.421 / 1.793 / 2.394 / 2.921 / 4.072 / 6.591; stage 3 alone
    // 1.415 / 1.797 / 2.371 / 2.849 / 3.989 / 6.506 (its 12-wavefront build spills 117 registers, this one none); the LASSO stage's (two wavefronts,
    // no screening table) loses at every size (1.460 / 1.905 / 2.511 / ...): off.  The gain is ~20 us per kernel, not the ~100 the "one round of
    // workgroups instead of two" arithmetic promised: a left-over kernel lasts as long as its longest voxel's Lawson-Hanson, whatever the rounds.
    long long opt_left_small[3] = {150000, 0, 600000};
    bool opt_no_nr4_nw8 = false;    // AMX_LEFT_NR4_NW8=0: protocols of 129 .. 256 volumes run their left-over lists on the round-5 builds (12 / 16 wavefronts, spilling)
    bool opt_no_big_all = false;    // AMX_BIG_ALL=0: lambda1 = 0 fits take the fast kernels first and reach k_noddi_lasso_big through the overflow lists
    bool side_launch = false;       // transient: the launch being enqueued goes to the side stream (launch_pair picks its own overflow lists)
    std::string path;               // kernels of the last fit enqueued on this ctx, in launch order (amx_last_path)
    int opt_seed_chunk = 0;        // AMX_SEED_CHUNK (0 = by the call's size, make_plan): voxels of one orientation per workgroup of the seed solvers (lanes refill from the chunk: the more voxels per lane, the smaller the share of the tail; 1 M voxels: 1024 -> 7.2 ms, 2048 -> 7.3, 4096 -> 5.5 for stage 1)
};

struct amx_lut {
    amx_ctx *ctx = nullptr;
    int model = 0;                 // 1 NODDI, 2 FreeWater, 3 SANDI, 4 CylinderZeppelinBall
    int nS = 0, ldA = 0, n_atoms = 0, ndirs = 0, tile_stride = 0;
    int n_wm = 0, is_exvivo = 0;   // NODDI
    int n_dwi = 0;                 // NODDI: rows of the stage-2 problem (scheme.dwi_idx; the single-b0 rule may add one)
    int n_perp = 0, n_iso = 0;     // FreeWater
    int n_rs = 0, n_in = 0, n_isos = 0;   // SANDI
    void *tiles = nullptr;
    double *gram = nullptr, *gram_dwi = nullptr;   // per-orientation Gram matrices (NODDI)
    double *basis_U = nullptr, *basis_S = nullptr; // per-orientation compressed basis and dictionary (amx_seed.hpp), NODDI
    double *screen2_kappa0 = nullptr;
    double *u2iso = nullptr;       // [ndirs][12] U2'iso (amx_build_basis)
    int s2_derive = 0;             // b0 rows of every atom are exactly 1.0 and iso > 0 on the stage-2 rows: stage-2 products derive from the stage-1 table
    float *screen2_S = nullptr; double *screen2_kappa = nullptr; // the same for the LASSO stage's dictionary
    double *screen_kappa0 = nullptr;                             // max ||(I - U U') a_j|| per orientation (k_nnls_gcert)
    float *screen_S = nullptr; double *screen_kappa = nullptr;   // float32 S [ndirs][12][192] + kappa [ndirs]: dual-value screening
    double *basis2_U = nullptr, *basis2_S = nullptr;   // the same for the LASSO stage's dictionary (DWI rows, normalised atoms)
    int ldG = 0;
    short *htable = nullptr;
    unsigned char *rowdwi = nullptr;
    double *colscale = nullptr;
    float *icvf = nullptr, *kappa = nullptr;
    double *norms = nullptr, *Rs = nullptr, *d_in = nullptr, *d_isos = nullptr;
    // FreeWater, per orientation and for one lambda2 (k_fw_orient_prep, rebuilt when lambda2 changes): fp64 dictionary
    // A [nS][NP], H^-1 [N][NP], H = A'A + lambda2 I [N][N]
    mutable double *fw_prep = nullptr;
    mutable double fw_lam2 = -1.0;
    mutable int fw_N = 0;
    mutable hipEvent_t fw_ready = nullptr;
    // CylinderZeppelinBall fast path (amx_czb.hip), per orientation and for one lambda2: M = (A'A + lambda2 I)^-1, M 1, M A'
    mutable double *czb_prep = nullptr;
    mutable double czb_lam2 = -1.0;
    mutable hipEvent_t czb_ready = nullptr;
    // SANDI row-space solver tables (k_sandi_tables) for one (lambda1, lambda2)
    mutable double *sandi_prep = nullptr;
    mutable double sandi_lam1 = -1.0, sandi_lam2 = -1.0;
    mutable hipEvent_t sandi_ready = nullptr;
};

// dictionaries of the batched solver entry points (amx_nnls_batched / amx_lasso_batched)
struct amx_dict {
    amx_ctx *ctx = nullptr;
    int m = 0, n = 0, ldA = 0, tile_stride = 0, n_dicts = 0;
    double *tiles = nullptr;       // device f64 [n_dicts][m][ldA] (row-major, odd leading dimension)
};

// principal-direction estimator of one acquisition scheme (amx_signal.hip)
struct amx_dti {
    amx_ctx *ctx = nullptr;
    int nS = 0;
    double min_signal = 0.0;
    double *wt = nullptr;          // device f64[nS][6]: transposed first six rows of pinv(design matrix)
};

// signal preparation plan of one (image geometry, mask, scheme, options) combination (amx_volume.hip)
struct amx_prep {
    amx_ctx *ctx = nullptr;
    long long d[3] = {0, 0, 0};    // spatial extents, d[0] = axis that is fastest in the image's memory
    long long s[3] = {0, 0, 0};    // element strides of those axes in the image
    long long c[3] = {0, 0, 0};    // strides of those axes in a C-ordered [X][Y][Z] volume
    long long sv = 0;              // element stride of the volume axis
    long long extent = 0;          // elements spanned by the image (largest offset + 1)
    long long n_total = 0, n_vox = 0;
    int nS = 0, n_out = 0, n_b0 = 0, n_gidx = 0, identity = 0, inplace = 0, hazard = 0, layout = 0;   // layout: 1 planar (s[0]==1), 2 interleaved (sv==1), 0 generic
    int *rank = nullptr;           // device int32[d2][d1][d0]: index in the masked list or -1
    long long *cidx = nullptr;     // device int64[n_vox]: C-order linear index of the masked voxels
    int *gptr = nullptr, *gidx = nullptr, *b0idx = nullptr;
    // tiles of 64 voxels along the fastest axis that hold at least one masked voxel, made once with the plan: the wavefronts of
    // k_prep_gather draw from this list through a counter of `tile_counter` (a ring: launch k of the plan zeroes and uses entry
    // k mod kCounterRing on its own stream, so up to kCounterRing gathers of one plan may be in flight together): an
    // image is half background, and a strided walk over ALL tiles left some wavefronts with seven full tiles and others with none
    static constexpr int kCounterRing = 16;
    int *live64 = nullptr, *tile_counter = nullptr;
    mutable unsigned launch_seq = 0;
    long long n_live64 = 0;
};

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[512];                                                                         \
            snprintf(b_, sizeof b_, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, \
                     __LINE__, #call);                                                            \
            (ctx)->err = b_;                                                                      \
            return AMX_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)

// amx_last_path: every launch site names its kernel (the first batch of a host-buffer call only)
static inline void amx_note(amx_ctx *ctx, const char *kernel)
{
    if ((ctx->in_host_fit && ctx->vox_base > 0) || ctx->path.size() > 1500) return;
    if (!ctx->path.empty()) ctx->path += " -> ";
    ctx->path += kernel;
}

static inline int amx_bad(amx_ctx *ctx, const char *msg)
{
    if (ctx) ctx->err = msg;
    return AMX_E_BADARG;
}

// grow-only device workspace
static inline int amx_ensure(amx_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return AMX_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return AMX_OK;
}

// misc buffer layout (ints): [0] n_chunks, [4..6] overflow counters of the 3 stages,
// [12] voxels that did not fit the large-MAXP variant either
struct Plan {
    int *lutidx, *perm, *counts, *dir_start, *cursor, *n_chunks, *ovf_count, *ovf_list;
    amx::Chunk *chunks;
    int max_chunks;
    size_t n;
    amx::Chunk *schunks = nullptr;  // larger chunks of the seed solver (count at n_chunks[1]); null: none
    int max_schunks = 0;
    int seed_chunk = 4096;         // voxels of one orientation per workgroup of the lane kernels (second plan)
    int seed1_waves = 4;           // the same for k_nnls_seed<1> (one wavefront per SIMD: with few voxels per chunk two wavefronts per workgroup keep more lanes busy)
    bool seed_occ2 = false, seed2_occ2 = false;   // k_nnls_seed<1> / k_lasso_seed in their two-wavefronts-per-SIMD builds (large calls)
    int seed2_waves = 4;           // wavefronts per workgroup of k_lasso_seed
    int seed_waves = 4;            // wavefronts per workgroup of the lane-per-voxel NODDI kernels (one workgroup per chunk of the second plan)
    int *feed = nullptr;           // kFeedSets sets of max_schunks + 8 chunk counters, one per kernel that shares its chunks (zeroed with the plan)
    int *feed_set(int k) const { return feed + (size_t)k * (max_schunks + 8); }
    // per-chunk counts of the lists the kernels of the chain compact (left-over lists of every certificate pass, clipped lists): one
    // array per pass, all in the arena behind the feed sets and cleared by the ONE memset that clears those -- each pass used to clear
    // its own (ten 5 us fill kernels per fit: 3 % of a 50 000-voxel call)
    int *zcount(int k) const { return feed + (size_t)(kFeedSetsN + k) * (max_schunks + 8); }
    static constexpr int kFeedSetsN = 7;
};
enum { FEED_SEED1 = 0, FEED_SEED2, FEED_SEED3, FEED_GEMM, FEED_CERT1, FEED_CERT2, FEED_CERT3, kFeedSets };
enum { ZC_CERT1 = 0, ZC_RESC1, ZC_CLIP, ZC_CERT2, ZC_CERT2W, ZC_CERT2W3, ZC_CERT3, ZC_RESC3, ZC_CERT2Q, kZCounts };   // ZC_CERT2Q: per chunk, what the second LASSO certificate pass leaves that a third could settle
static_assert(kFeedSets == Plan::kFeedSetsN, "Plan::zcount sits behind the feed sets");

// AMX_DEBUG=1: synchronise after every launch and trace progress on stderr
static inline bool amx_debug() { static int d = -1; if (d < 0) { const char *e = getenv("AMX_DEBUG"); d = (e && *e && *e != '0') ? 1 : 0; } return d == 1; }
#define AMX_TRACE(ctx, s, what)                                                                   \
    do {                                                                                          \
        if (amx_debug()) {                                                                        \
            fprintf(stderr, "[amx] %s ...", what); fflush(stderr);                                \
            hipError_t e_ = hipStreamSynchronize(s);                                              \
            fprintf(stderr, " %s\n", hipGetErrorString(e_)); fflush(stderr);                      \
        }                                                                                         \
    } while (0)

static inline void rec(amx_ctx *ctx, int k, hipStream_t s)
{
    // (an event in the stream is a packet of its own: ~5 us of a call's time each -- twenty of them 70 us of a 1.3 ms fit)
    if (ctx->profiling == 1 || (ctx->profiling >= 2 && (k < 2 ? 0 : k / 2) == ctx->profiling - 2)) { (void)hipEventRecord(ctx->ev[k], s); ctx->ev_valid[k] = true; }
}

// defined in the per-model launch units (amx_noddi.hip, amx_fw.hip, amx_sandi.hip)
int amx_build_basis(amx_ctx *ctx, amx_lut *lut);
int amx_launch_noddi_project(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_noddi_seed(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, int stage);
int amx_launch_noddi_gcert(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, int stage, size_t *list_off, const int **count_out);
int amx_launch_noddi_gemm(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, bool lasso);
int amx_launch_noddi_s2prep(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s);
int amx_gemm_ksteps(const amx_lut *lut);   // K-steps of the table kernels for this dictionary (25 / 40), 0 = shape not supported
int amx_launch_noddi_gcert2(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, bool wide);
bool amx_gcert2_third(const amx_ctx *ctx, const amx_lut *lut, bool wide);
int amx_gcert2_third_min_items(const amx_ctx *ctx, const amx_lut *lut);   // list entries a chunk must hold for that pass to work on it   // a third LASSO certificate pass for this dictionary? (amx_seed.hip)
size_t amx_gcert2_leftover_offset(const Plan &pl, bool wide, bool third);      // which half of ctx->rlist the LASSO certificate passes end in (amx_seed.hip)
const int *amx_gcert2_leftover_counts(const Plan &pl, bool wide, bool third);  // ... and the per-chunk counts of those lists (Plan::zcount)
static inline size_t amx_rlist_half(const Plan &pl) { return (size_t)pl.n + pl.max_schunks + 64; }   // ints per left-over list + counts
int amx_launch_noddi_seed2(amx_ctx *ctx, const amx_lut *lut, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, bool have_ytil2);
int amx_launch_noddi_big(amx_ctx *ctx, const amx::NoddiArgs &a, const Plan &pl, hipStream_t s, const int *list, const int *count, int n_all);   // amx_big.hip: LASSO stage, any support size
int amx_launch_noddi_s1(amx_ctx *ctx, amx::NoddiArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_noddi_s2(amx_ctx *ctx, amx::NoddiArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_noddi_s3(amx_ctx *ctx, amx::NoddiArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_fw(amx_ctx *ctx, amx::FwArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_sandi(amx_ctx *ctx, amx::SandiArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_czb(amx_ctx *ctx, amx::CzbArgs &a, const Plan &pl, hipStream_t s);
int amx_launch_batched(amx_ctx *ctx, amx::BatchedArgs &a, const Plan &pl, hipStream_t s, bool ridge);
int amx_czb_prepare(amx_ctx *ctx, const amx_lut *lut, double lam2, hipStream_t s);
int amx_launch_czb_fast(amx_ctx *ctx, const amx_lut *lut, amx::CzbArgs &a, const Plan &pl, hipStream_t s);
// lane-per-voxel variants for dictionaries of <= 16 atoms (amx_small.hip)
int amx_launch_fw_small(amx_ctx *ctx, amx::FwArgs &a, const Plan &pl, hipStream_t s);
int amx_fw_prepare(amx_ctx *ctx, const amx_lut *lut, amx::FwArgs &a, hipStream_t s);
int amx_sandi_prepare(amx_ctx *ctx, const amx_lut *lut, amx::SandiArgs &a, hipStream_t s);   // before amx_launch_fw when the refill path runs
int amx_launch_sandi_small(amx_ctx *ctx, amx::SandiArgs &a, const Plan &pl, hipStream_t s);
// Lane-per-voxel solvers: start the active set from ALL atoms and drop the non-positive ones in blocks (unique optimum
// with lambda2 > 0, so the path is free; dense optima are reached in 3-4 factorisations).  AMX_COLD_START=1: the
// Lawson-Hanson start from the empty set.  Needs a ridge that keeps the full system well conditioned.
__host__ __device__ static inline bool amx_warm_start(double lam2, unsigned flags) { return lam2 >= 1e-5 && !(flags & 0x80000000u); }
// FreeWater with lanes that never idle (k_freewater_refill, amx_small.hip): maps only (the error maps / corrected DWI
// need the signal again and stay with k_freewater_lane), <= 12 atoms; chunks of up to 4096 voxels per workgroup
// voxels of one orientation per workgroup of the refill kernel: large enough to keep the lanes fed (the buffer needs a
// pool to draw from), small enough for ~3 rounds of workgroups over the chip (measured on 2 M voxels: 512 -> 1.83 ms,
// 1024 -> 1.79, 2048 -> 1.93, 4096 -> 2.54)
static inline int amx_refill_chunk(const amx_ctx *ctx, long long n_vox)
{
    const int v = ctx->opt_refill_chunk;
    if (v >= AMX_CHUNK) return v;                       // (make_plan sizes the chunk list for kChunk: smaller chunks would overrun it)
    const long long c = n_vox / 1536;
    return (int)(c < 512 ? 512 : (c > 2048 ? 2048 : c));
}
// (the projection + block-pivoting kernels assume the warm start: with lambda2 < 1e-5, or AMX_COLD_START=1, the fit goes to the
//  Lawson-Hanson lane kernels -- single exchanges from the empty set, which is all block pivoting could do there, ran into
//  the iteration cap on 15 % of the voxels at lambda2 = 1e-6)
static inline bool amx_fw_use_refill(const amx_ctx *ctx, int n_atoms, int nS, unsigned flags, double lam2)
{
    if (!amx_warm_start(lam2, flags) || ctx->opt_cold_start) return false;
    if (ctx->opt_no_refill || ctx->opt_wave_per_voxel) return false;
    return n_atoms <= 12 && (flags & (AMX_F_RMSE | AMX_F_NRMSE | AMX_F_CORRECTED)) == 0 &&
           ((size_t)nS * 12 + 144 + 4 * (16 * 65 + 12 * 64 + 32)) * sizeof(double) + 16 <= 80 * 1024;
}
// float32 signals in HBM are read natively by the NODDI kernels, by the wavefront-per-voxel kernels of every model (load_rows) and by
// FreeWater's matrix-core projection; the other lane kernels get a float64 copy made on the device first (k_widen)
static inline bool amx_fw_native_f32(const amx_ctx *ctx, int n_atoms, int nS, unsigned flags, double lam2);
// lane-per-voxel solvers work on H = A'A + lambda2 I (Gram space): they need the ridge to bound cond(H); with
// lambda2 (nearly) 0 the problem goes to the QR solver in A-space (wavefront per voxel), like the reference's lasso,
// which accepts any lambda2 >= 0
static inline bool amx_use_lane_solver(const amx_ctx *ctx, int n_atoms, double lam2) { return n_atoms <= 16 && lam2 >= 1e-9 && !ctx->opt_wave_per_voxel; }
static inline bool amx_fw_native_f32(const amx_ctx *ctx, int n_atoms, int nS, unsigned flags, double lam2)
{
    if (!amx_use_lane_solver(ctx, n_atoms, lam2)) return true;                     // wavefront per voxel: load_rows
    return amx_fw_use_refill(ctx, n_atoms, nS, flags, lam2) && nS <= 96 && !ctx->opt_fw_proj_valu;
}
