// amx_api.hip -- C ABI (include/amico_amd.h) of the MI355X-native AMICO fitter.
// Host side only drives HIP: workspace, dictionary upload, kernel launches, status.
#include "amx_host.hpp"
#include "amx_prep.hpp"
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>

using namespace amx;

namespace {

int ensure(amx_ctx *ctx, DevBuf &b, size_t bytes) { return amx_ensure(ctx, b, bytes); }

template <typename T>
int upload(amx_ctx *ctx, T **dst, const T *src, size_t n)
{
    HIPCHK(ctx, hipMalloc((void **)dst, n * sizeof(T) + 16));
    HIPCHK(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

int reset_status(amx_ctx *ctx, hipStream_t s)
{
    HIPCHK(ctx, hipMemsetAsync(ctx->status_d, 0, ST_WORDS * sizeof(int), s));
    HIPCHK(ctx, hipMemsetAsync(ctx->status_d + ST_ERRPACK, 0x7f, 2 * sizeof(int), s));
    return AMX_OK;
}

int make_plan(amx_ctx *ctx, int64_t n, int ndirs, Plan &pl, bool seeds = false, int table_rows = 0, int blocks_chunk = 0)
{
    int rc;
    const int max_chunks = (int)(n / kChunk) + ndirs + 1;
    pl.n = (size_t)n;
    if (seeds) {
        // Chunk of the second plan: whole orientations wherever possible (a lane then walks many voxels and the tail of the chunk
        // is a small share), i.e. about twice the mean population -- 4 M voxels, ndirs 500: 4096 -> 103 M voxels/s (every
        // orientation cut in two or three), 8192 -> 83 M, 16384 -> 123 M; never below 4096 (1 M voxels: 2048 -> 72 M, 4096 -> 99 M).
        int sc = ctx->opt_seed_chunk;
        if (sc <= 0) {
            const long long want = 2 * (long long)n / (ndirs > 512 ? ndirs : 512);
            sc = (int)(want < 4096 ? 4096 : (want > 65536 ? 65536 : ((want + 63) & ~63LL)));
        }
        pl.seed_chunk = sc;
        pl.max_schunks = (int)(n / sc) + ndirs + 1;
        pl.seed_waves = ctx->opt_seed_waves ? ctx->opt_seed_waves : 4;
        // (measured, ndirs = 500: 100 000 / 200 000 / 400 000 / 1 M voxels -> stage-1 group 1.11 / 1.55 / 2.31 / 4.66 ms with two
        //  wavefronts per workgroup against 1.33 / 1.80 / 2.39 / 4.14 ms with four; every other lane kernel is best with four)
        pl.seed1_waves = ctx->opt_seed_waves ? ctx->opt_seed_waves : ((double)n / (double)pl.max_schunks < 640.0 ? 2 : 4);
        const long long call_vox = ctx->in_host_fit ? ctx->host_total_vox : n;      // (batches of one host call all take the same build)
        pl.seed_occ2 = call_vox >= ctx->opt_seed_occ2_from;
        pl.seed2_occ2 = call_vox >= ctx->opt_seed2_occ2_from;
        pl.seed2_waves = pl.seed1_waves;
        if (!ctx->opt_seed_waves) { if (pl.seed_occ2) pl.seed1_waves = 4; if (pl.seed2_occ2) pl.seed2_waves = 4; }
        if ((rc = ensure(ctx, ctx->schunks, (size_t)pl.max_schunks * sizeof(Chunk)))) return rc;
        if ((rc = ensure(ctx, ctx->ytil, (size_t)n * amx::kSeedKD * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->seeds, (size_t)n * sizeof(unsigned long long)))) return rc;
        // the A'y table of all voxels, and the compact table of the voxels whose stage-2 signal clips (sized for all of them: a
        // dictionary whose b0 rows are not ones sends every voxel there), the clipped lists / counts / slots of k_s2_prep
        if ((rc = ensure(ctx, ctx->cgemm, ((size_t)n / 64 + ndirs + 1) * table_rows * 64 * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->cgemm2, ((size_t)n / 64 + ndirs + 1) * table_rows * 64 * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->clip, ((size_t)2 * n + pl.max_schunks + 64) * sizeof(int)))) return rc;
        if ((rc = ensure(ctx, ctx->feed, (size_t)(kFeedSets + kZCounts) * (pl.max_schunks + 8) * sizeof(int)))) return rc;      // chunk counters of the kernels that share their chunks (SeedFeed, BlockFeed) + the list counts of every pass (Plan::zcount)
        if ((rc = ensure(ctx, ctx->done, (size_t)n + 64))) return rc;
        if ((rc = ensure(ctx, ctx->rlist, 4 * amx_rlist_half(pl) * sizeof(int)))) return rc;      // (two halves per stage's certificate passes; a forked fit's stage 3 takes the third and fourth)
        if ((rc = ensure(ctx, ctx->ytil2, (size_t)n * amx::kSeedKD * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->seeds2, (size_t)n * 4 * sizeof(unsigned long long)))) return rc;
        pl.schunks = (Chunk *)ctx->schunks.p;
        pl.feed = (int *)ctx->feed.p;
    }
    if (!seeds && blocks_chunk > 0) {
        // second plan only (chunks of whole 64-voxel blocks) + a block-wise table of table_rows rows: CylinderZeppelinBall's fast path
        pl.seed_chunk = blocks_chunk;
        pl.max_schunks = (int)(n / blocks_chunk) + ndirs + 1;
        if ((rc = ensure(ctx, ctx->schunks, (size_t)pl.max_schunks * sizeof(Chunk)))) return rc;
        if ((rc = ensure(ctx, ctx->cgemm, ((size_t)n / 64 + ndirs + 1) * table_rows * 64 * sizeof(double)))) return rc;
        pl.schunks = (Chunk *)ctx->schunks.p;
    }
    if ((rc = ensure(ctx, ctx->lutidx, n * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->perm, n * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->counts, (size_t)(ndirs + 1) * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->dir_start, (size_t)(ndirs + 1) * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->cursor, (size_t)(ndirs + 1) * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->chunks, (size_t)max_chunks * sizeof(Chunk)))) return rc;
    if ((rc = ensure(ctx, ctx->misc, 64 * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, ctx->ovf, (size_t)7 * n * sizeof(int)))) return rc;      // (lists 0 .. 2: the stages' overflow, 3: second level; 4, 5: the same for a forked fit's side stream; 6: what k_noddi_lasso_big takes; amx_launch.hpp)
    pl.lutidx = (int *)ctx->lutidx.p; pl.perm = (int *)ctx->perm.p; pl.counts = (int *)ctx->counts.p;
    pl.dir_start = (int *)ctx->dir_start.p; pl.cursor = (int *)ctx->cursor.p;
    pl.chunks = (Chunk *)ctx->chunks.p; pl.n_chunks = (int *)ctx->misc.p;
    pl.ovf_count = (int *)ctx->misc.p + 4; pl.ovf_list = (int *)ctx->ovf.p;
    pl.max_chunks = max_chunks;
    pl.n = (size_t)n;
    return AMX_OK;
}

int enqueue_bucketing(amx_ctx *ctx, const amx_lut *lut, const double *d_dirs, int64_t n, Plan &pl, hipStream_t s, int chunk = kChunk,
                      double *zero_rows = nullptr, int zero_cols = 0)
{
    // (four launches: the counters cleared in one, the chunk order in k_plan's tail; they were three memsets and five kernels --
    //  ~11 us a node in a small call, profiles/r06_launch_nodes.txt)
    const int n_feed = pl.feed ? (kFeedSets + kZCounts) * (pl.max_schunks + 8) : 0;
    hipLaunchKernelGGL(k_clear3, dim3(n_feed > 4096 ? 8 : 1), dim3(1024), 0, s, pl.counts, lut->ndirs + 1, (int *)ctx->misc.p, 64, pl.feed, n_feed);
    const int span = prep_span(n);
    const int nb = (int)((n + span - 1) / span);
    const int use_lds = lut->ndirs <= 8192 ? 1 : 0;          // LDS histograms: 2 * ndirs ints
    hipLaunchKernelGGL(k_dir_to_lut, dim3(nb), dim3(1024), use_lds ? (size_t)lut->ndirs * sizeof(int) : 0, s, d_dirs,
                       (int)n, lut->htable, lut->ndirs, pl.lutidx, pl.counts, ctx->status_d, use_lds, (int)ctx->vox_base, span, zero_rows, zero_cols);
    AMX_TRACE(ctx, s, "k_dir_to_lut");
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, pl.counts, lut->ndirs, chunk, pl.dir_start,
                       pl.cursor, pl.chunks, pl.n_chunks, pl.schunks ? pl.seed_chunk : 0, pl.schunks, (pl.schunks && !ctx->opt_no_chunk_order) ? 1 : 0);
    AMX_TRACE(ctx, s, "k_plan");
    hipLaunchKernelGGL(k_bucket, dim3(nb), dim3(1024), use_lds ? (size_t)2 * lut->ndirs * sizeof(int) : 0, s, pl.lutidx,
                       (int)n, lut->ndirs, pl.dir_start, pl.cursor, pl.perm, use_lds, span);
    AMX_TRACE(ctx, s, "k_bucket");
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

// wavefront primitives exercised on the device (tests/test_gpu_parity.py::test_wave_primitives)
// per-call counters (misc, cleared by the next call) -> status words that accumulate until amx_sync_status
__global__ void k_fold_counters(const int *misc, int *status)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        atomicAdd(&status[ST_RERUN], misc[4] + misc[5] + misc[6] + misc[7]);      // (two batches may fold concurrently: fit_host)
        atomicAdd(&status[ST_OVERFLOW], misc[12] + misc[13]);
    }
}

__global__ void k_selftest(double *out)
{
    const int lane = threadIdx.x & 63;
    const double v = (double)(lane * lane) - 100.5 * lane + 3.25;     // distinct, sign-changing values
    out[0 * 64 + lane] = wave_sum(v);
    out[1 * 64 + lane] = wave_max(v);
    out[2 * 64 + lane] = wave_min(v);
    out[3 * 64 + lane] = bcast(v, 37);
    out[4 * 64 + lane] = from_next_lane(v);
    out[5 * 64 + lane] = (double)__builtin_popcountll(ballot64(v > 0.0));
    out[6 * 64 + lane] = (double)bcast_i(lane * 3, 21);
    out[7 * 64 + lane] = v;
    double q4[4] = {v, v * v, 1.0 / (1.0 + lane), (double)(lane & 7) - v};
    wave_sum4(q4, lane);
    out[8 * 64 + lane] = q4[0]; out[9 * 64 + lane] = q4[1]; out[10 * 64 + lane] = q4[2]; out[11 * 64 + lane] = q4[3];
}

int bad(amx_ctx *ctx, const char *msg) { return amx_bad(ctx, msg); }

// a profiled call starts with no event pair valid: amx_last_kernel_ms of a group this call does not run is an error, not the
// timing of an earlier call (the dti / prep / lut entry points record slot 4 only and clear it themselves)
void clear_events(amx_ctx *ctx)
{
    if (ctx->profiling) for (int k = 0; k < kEv; k++) ctx->ev_valid[k] = false;
}

// float32 signals (the image dtype of the reference, core.py:136) -> the float64 rows the solvers read: exact
__global__ void k_widen(const float *__restrict__ src, double *__restrict__ dst, size_t n)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    // (16-byte loads only from a 16-byte aligned source: a row slice of a float32 tensor may start at any multiple of 4 bytes)
    if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i0);
        dst[i0] = (double)v.x; dst[i0 + 1] = (double)v.y; dst[i0 + 2] = (double)v.z; dst[i0 + 3] = (double)v.w;
    } else {
        for (size_t i = i0; i < n; i++) dst[i] = (double)src[i];
    }
}

void progress(amx_ctx *ctx, int64_t done, int64_t total)
{
    if (ctx->progress) ctx->progress(done, total, ctx->progress_user);
}

// Device-pointer entry points: the fit is only ENQUEUED when the call returns, so the callback is a host function on the
// stream (hipLaunchHostFunc) -- it runs on a runtime thread once everything enqueued before it has finished.
struct ProgressTick { amx_ctx *ctx; int64_t done, total; };
void progress_host_fn(void *p)
{
    ProgressTick *t = static_cast<ProgressTick *>(p);
    if (t->ctx->progress) t->ctx->progress(t->done, t->total, t->ctx->progress_user);
    delete t;
}
void progress_tick(amx_ctx *ctx, hipStream_t s, int64_t done, int64_t total)
{
    if (!ctx->progress || ctx->in_host_fit) return;
    ProgressTick *t = new ProgressTick{ctx, done, total};
    if (hipLaunchHostFunc(s, progress_host_fn, t) != hipSuccess) { (void)hipGetLastError(); delete t; }
}

// The host threads + pinned ring of the float32 transport (amx_stage.hpp).  The threads stay on the NUMA node the DEVICE hangs on, one stripe of
// physical cores each: the pinned ring they write lives there and the copies leave from there; the caller's buffer is read across the socket
// link if it lives on the other node.  Measured on the two-socket box, host_trace.py and bench.py, two processes each: 12.6 - 12.9 ms per 1 M
// voxels in all four; on the calling thread's node 15.5 - 19 ms when that is the far socket (AMX_HOST_PIN = gpu | caller | 0: diagnosis;
// profiles/r05c_host_transport.txt, section 8)
amx_stage::Pool *make_stage_pool(amx_ctx *ctx)
{
    const unsigned hw = std::thread::hardware_concurrency();
    int nt = ctx->opt_host_threads;
    if (hw >= 2 && nt > (int)(hw / 2)) nt = (int)(hw / 2);
    int node = -1;
    const char *pe = getenv("AMX_HOST_PIN");
    if (!pe || pe[0] == 'g') node = amx_stage::device_node(ctx->device);
    if ((pe && pe[0] == 'c') || ((!pe || pe[0] == 'g') && node < 0)) {
        const int cpu = sched_getcpu();
        cpu_set_t cs;
        for (int nd = 0; nd < 16 && cpu >= 0; nd++)
            if (amx_stage::node_cpus(nd, &cs) && CPU_ISSET(cpu, &cs)) { node = nd; break; }
    }
    // the other devices of that node have pools of their own (this process's other contexts, or other ranks): disjoint shares
    int sib_i = 0, sib_n = 1;
    const char *se = getenv("AMX_HOST_SIBLINGS");          // "i/n" forces (diagnosis, tests)
    if (se && sscanf(se, "%d/%d", &sib_i, &sib_n) == 2 && sib_n >= 1 && sib_i >= 0) { }
    else { sib_i = 0; sib_n = 1; if (!pe || pe[0] == 'g') amx_stage::device_siblings(ctx->device, node, &sib_i, &sib_n); }
    return amx_stage::Pool::create(nt < 1 ? 1 : nt, node, sib_i, sib_n);
}

// Round 6: the pool is made WHILE the dictionary is uploaded.  Pinning its 64 MB ring and starting its threads takes ~15 ms, and it used to
// happen inside the first host-buffer fit of a process -- the one fit a subject gets (core.py:465-466: one model.fit per Evaluation) --
// behind 10 ms of dictionary tables built on the GPU and a digest of KERNELS on the host: a helper thread makes the pool beside those, the
// first fit adopts it (NODDI().fit(evaluation), first call of a process, 1 M voxels: 41.6 ms -> see profiles/r06_first_call.txt).
void prefetch_stage_pool(amx_ctx *ctx)
{
    if (ctx->stage || ctx->stage_failed || ctx->stage_bg_started || ctx->opt_host_no_narrow) return;
    const char *pe = getenv("AMX_HOST_PIN"), *pf = getenv("AMX_HOST_PREFETCH");
    if ((pe && pe[0] == 'c') || (pf && pf[0] == '0')) return;          // (the caller's node is the CALLING thread's: made where it is needed)
    ctx->stage_bg_started = true;
    try {
        ctx->stage_thread = std::thread([ctx] {
            if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return; }
            ctx->stage_bg = make_stage_pool(ctx);
        });
    } catch (...) { ctx->stage_bg_started = false; }
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int amx_version(void) { return 100; }

int amx_device_count(void)
{
    int ndev = 0, n = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (int d = 0; d < ndev; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) n = d + 1;     // (device numbers are HIP's)
    }
    return n;
}

int amx_ctx_create(int device, amx_ctx **out)
{
    if (!out) return AMX_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AMX_E_NODEVICE;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return AMX_E_NODEVICE; }
    if (device >= ndev) return AMX_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return AMX_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return AMX_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AMX_E_NODEVICE;   // code objects are gfx950 only
    amx_ctx *ctx = new amx_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc((void **)&ctx->status_d, ST_WORDS * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->status_h, (ST_WORDS + 16) * sizeof(int), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {   // (k_status_home writes it from the device)
        delete ctx;
        return AMX_E_HIP;
    }
    for (int k = 0; k < kEv; k++) { hipEventCreate(&ctx->ev[k]); ctx->ev_valid[k] = false; }
    {
        const char *e = getenv("AMX_NO_SEED");
        ctx->opt_no_seed = e && *e && *e != '0';
        e = getenv("AMX_NO_GCERT");
        ctx->opt_no_gcert = e && *e && *e != '0';
        e = getenv("AMX_S2_EXACT");
        ctx->opt_s2_exact = e && *e && *e != '0';
        auto on = [](const char *name) { const char *v = getenv(name); return v && *v && *v != '0'; };
        ctx->opt_no_gram = on("AMX_NO_GRAM"); ctx->opt_lasso_qr = on("AMX_LASSO_QR"); ctx->opt_cold_start = on("AMX_COLD_START");
        { const char *m2 = getenv("AMX_SEED2_MAXATOMS"); if (m2 && *m2) { const int v = atoi(m2); ctx->opt_seed2_maxatoms = v < 8 ? 8 : (v > 30 ? 30 : v); } }
        { const char *rp = getenv("AMX_GCERT_REPAIR"); if (rp && *rp) ctx->opt_gcert_repair = atoi(rp) != 0 ? 1 : 0; }
        { const char *t3 = getenv("AMX_GCERT2_THIRD"); if (t3 && *t3) ctx->opt_gcert2_third = atoi(t3) != 0 ? 1 : 0; }
        ctx->opt_host_one_shot = on("AMX_HOST_ONE_SHOT"); ctx->opt_host_one_stream = on("AMX_HOST_ONE_STREAM"); ctx->opt_host_late_results = on("AMX_HOST_LATE_RESULTS");
        e = getenv("AMX_HOST_PIPELINE_FROM");
        if (e && atoll(e) >= 262144) ctx->opt_host_pipeline_from = atoll(e);       // (a pipelined call has a first batch of 131 072 voxels and a second one at least as long)
        e = getenv("AMX_HOST_NATIVE32");
        ctx->opt_host_no_native32 = e && *e == '0';
        e = getenv("AMX_HOST_NARROW");
        ctx->opt_host_no_narrow = e && *e == '0';
        e = getenv("AMX_HOST_THREADS");
        if (e && atoi(e) >= 1) ctx->opt_host_threads = atoi(e) > 64 ? 64 : atoi(e);
        ctx->opt_tile_f32 = on("AMX_TILE_F32"); ctx->opt_fw_proj_valu = on("AMX_FW_PROJ_VALU"); ctx->opt_sandi_atom_space = on("AMX_SANDI_ATOM_SPACE");
        ctx->opt_prep_tile = on("AMX_PREP_TILE"); ctx->opt_prep_scalar = on("AMX_PREP_SCALAR"); ctx->opt_lut_regs = on("AMX_LUT_REGS"); ctx->opt_no_refill = on("AMX_NO_REFILL");
        ctx->opt_wave_per_voxel = on("AMX_WAVE_PER_VOXEL"); ctx->opt_fw_no_fuse = on("AMX_FW_NO_FUSE");
        e = getenv("AMX_REFILL_CHUNK");
        if (e) ctx->opt_refill_chunk = atoi(e);
        e = getenv("AMX_HOST_RAMP");
        if (e && *e) { const long long v = atoll(e); ctx->opt_host_ramp = v <= 0 ? 0 : (v > 131072 ? 131072 : ((v + 3) & ~3LL)); }
        e = getenv("AMX_HOST_BATCH");
        // (a multiple of 4: k_widen reads float4; at least the largest ramp batch, 131072: the ramp batches are written into slots of this size)
        if (e && atol(e) >= 131072) ctx->opt_host_batch = ((long long)atol(e) + 3) & ~3LL;
        e = getenv("AMX_SEED_WAVES");
        if (e && *e) { const int v = atoi(e); ctx->opt_seed_waves = (v == 1 || v == 2 || v == 4) ? v : 0; }
        e = getenv("AMX_SEED_MIN_VOXELS");
        if (e && *e) ctx->opt_seed_min_voxels = atoll(e);
        e = getenv("AMX_SEED_OCC2_FROM");
        if (e && *e) ctx->opt_seed_occ2_from = atoll(e);
        e = getenv("AMX_SEED2_OCC2_FROM");
        if (e && *e) ctx->opt_seed2_occ2_from = atoll(e);
        ctx->opt_no_chunk_order = on("AMX_NO_CHUNK_ORDER");
        ctx->opt_no_hard_first = on("AMX_NO_HARD_FIRST");
        ctx->opt_prep_no_direct = on("AMX_PREP_NO_DIRECT");
        e = getenv("AMX_NO_GCERT_WIDE");
        ctx->opt_no_gcert_wide = e && *e && *e != '0';
        e = getenv("AMX_RESCUE_FROM");
        if (e && *e) { ctx->opt_rescue_from = atoll(e); ctx->opt_rescue_from_set = true; }
        e = getenv("AMX_GCERT2_THIRD_MIN");
        if (e && *e) ctx->opt_gcert2_third_min = atoi(e) < 0 ? 0 : atoi(e);
        e = getenv("AMX_NO_SCREEN");
        ctx->opt_no_screen = e && *e && *e != '0';
        e = getenv("AMX_SEED_STAGES");
        if (e && *e) ctx->opt_seed_stages = atoi(e) & 7;
        e = getenv("AMX_SEED_TRIPCAP");
        if (e && *e) {
            int c[3] = {ctx->opt_seed_tripcap[0], ctx->opt_seed_tripcap[1], ctx->opt_seed_tripcap[2]};
            sscanf(e, "%d,%d,%d", &c[0], &c[1], &c[2]);
            for (int k = 0; k < 3; k++) ctx->opt_seed_tripcap[k] = c[k] < 4 ? 4 : c[k];
        }
        e = getenv("AMX_LEFT_SMALL");
        if (e && *e) {
            long long c[3] = {ctx->opt_left_small[0], ctx->opt_left_small[1], ctx->opt_left_small[2]};
            sscanf(e, "%lld,%lld,%lld", &c[0], &c[1], &c[2]);
            for (int k = 0; k < 3; k++) ctx->opt_left_small[k] = c[k] < 0 ? 0 : c[k];
        }
        e = getenv("AMX_LEFT_NR4_NW8");
        ctx->opt_no_nr4_nw8 = e && *e == '0';
        e = getenv("AMX_BIG_ALL");
        ctx->opt_no_big_all = e && *e == '0';
        e = getenv("AMX_FORK");
        if (e && *e) ctx->opt_fork = atoi(e) & 3;
        e = getenv("AMX_FORK_CUS");
        if (e && *e) ctx->opt_fork_cus = atoi(e) < 0 ? 0 : atoi(e);
        ctx->opt_fork_prio = on("AMX_FORK_PRIO") ? 1 : 0;
        e = getenv("AMX_SEED_CHUNK");
        // (never below kChunk: the left-over passes size their grid by the FIRST plan's chunk count, n / kChunk + ndirs + 1)
        if (e && atoi(e) >= kChunk) ctx->opt_seed_chunk = (atoi(e) + 63) & ~63;
    }
    reset_status(ctx, nullptr);
    hipStreamSynchronize(nullptr);
    *out = ctx;
    return AMX_OK;
}

void amx_ctx_destroy(amx_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    DevBuf *bufs[] = {&ctx->lutidx, &ctx->perm, &ctx->counts, &ctx->dir_start, &ctx->cursor, &ctx->chunks,
                      &ctx->misc, &ctx->xiso, &ctx->supp, &ctx->ovf, &ctx->cproj, &ctx->hy, &ctx->hdirs, &ctx->hest,
                      &ctx->hrmse, &ctx->hnrmse, &ctx->hextra, &ctx->big, &ctx->hy32, &ctx->wy, &ctx->ytil, &ctx->seeds, &ctx->schunks, &ctx->ytil2, &ctx->seeds2, &ctx->cgemm, &ctx->done, &ctx->rlist, &ctx->cgemm2, &ctx->clip, &ctx->feed};
    for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
    for (DevBuf &b : ctx->alt) if (b.p) hipFree(b.p);
    if (ctx->status_d) hipFree(ctx->status_d);
    if (ctx->status_h) hipHostFree(ctx->status_h);
    for (int k = 0; k < kEv; k++) (void)hipEventDestroy(ctx->ev[k]);
    if (ctx->up_ev) (void)hipEventDestroy(ctx->up_ev);
    if (ctx->hs) { (void)hipStreamDestroy(ctx->hs); (void)hipStreamDestroy(ctx->hs2); for (hipEvent_t e : ctx->hev) (void)hipEventDestroy(e); }
    if (ctx->stage_thread.joinable()) ctx->stage_thread.join();
    delete ctx->stage_bg;
    for (int w = 0; w < 2; w++) if (ctx->fork_s[w]) { (void)hipStreamDestroy(ctx->fork_s[w]); for (hipEvent_t e : ctx->fork_ev[w]) if (e) (void)hipEventDestroy(e); }
    delete ctx->stage;             // (joins the host threads of the float32 transport)
    delete ctx;
}

const char *amx_last_error(amx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

void amx_lut_destroy(amx_lut *lut)
{
    if (!lut) return;
    if (lut->ctx) hipSetDevice(lut->ctx->device);
    void *ps[] = {lut->u2iso, lut->screen2_kappa0, lut->screen_kappa0, lut->screen2_S, lut->screen2_kappa, lut->screen_S, lut->screen_kappa, lut->basis_U, lut->basis_S, lut->basis2_U, lut->basis2_S, lut->gram, lut->gram_dwi, lut->tiles, lut->htable, lut->rowdwi, lut->colscale, lut->icvf, lut->kappa,
                  lut->norms, lut->Rs, lut->d_in, lut->d_isos, lut->fw_prep, lut->sandi_prep, lut->czb_prep};
    for (void *p : ps) if (p) hipFree(p);
    if (lut->fw_ready) (void)hipEventDestroy(lut->fw_ready);
    if (lut->sandi_ready) (void)hipEventDestroy(lut->sandi_ready);
    if (lut->czb_ready) (void)hipEventDestroy(lut->czb_ready);
    delete lut;
}

static int build_tiles(amx_ctx *ctx, amx_lut *lut, const float *src, size_t src_n, const float *fix,
                       size_t fix_n, const std::vector<int> &fix_ones, int n_lut)
{
    float *d_src = nullptr, *d_fix = nullptr; int *d_ones = nullptr;
    int rc;
    if ((rc = upload(ctx, &d_src, src, src_n))) return rc;
    if ((rc = upload(ctx, &d_fix, fix, fix_n ? fix_n : 1))) return rc;
    if ((rc = upload(ctx, &d_ones, fix_ones.data(), fix_ones.size()))) return rc;
    const size_t bytes = ((size_t)lut->ndirs * lut->tile_stride + kTileSlack) * sizeof(float) + 64;   // (slack: the global-tile kernels' row sweeps read past the last row's end)
    HIPCHK(ctx, hipMalloc(&lut->tiles, bytes));
    HIPCHK(ctx, hipMemset(lut->tiles, 0, bytes));
    hipLaunchKernelGGL(k_build_lut, dim3(2048), dim3(256), 0, nullptr, d_src, d_fix, d_ones, n_lut,
                       (int)fix_ones.size(), lut->ndirs, lut->nS, lut->ldA, lut->tile_stride, (float *)lut->tiles);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    hipFree(d_src); hipFree(d_fix); hipFree(d_ones);
    return AMX_OK;
}

int amx_lut_upload_noddi(amx_ctx *ctx, const float *wm, const float *iso, const double *norms,
                         const float *icvf, const float *kappa, const int16_t *htable,
                         const int64_t *dwi_idx, int n_wm, int ndirs, int nS, int dwi_count,
                         int is_exvivo, amx_lut **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!wm || !iso || !norms || !icvf || !kappa || !htable || !dwi_idx || !out || n_wm <= 0 || ndirs <= 0 ||
        nS <= 0 || dwi_count < 0 || dwi_count > nS)
        return bad(ctx, "amx_lut_upload_noddi: bad argument");
    const int n_atoms = n_wm + 1 + (is_exvivo ? 1 : 0);
    // any shape models.pyx:825-861 would run, up to what a wavefront's lanes hold: 8 rows / 4 atoms per lane
    if (n_atoms > 256 || nS > 512) return bad(ctx, "amx_lut_upload_noddi: unsupported size (n_atoms <= 256, nS <= 512)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prefetch_stage_pool(ctx);            // (the host threads of the float32 transport are made beside this upload)
    amx_lut *lut = new amx_lut();
    lut->ctx = ctx; lut->model = 1; lut->nS = nS; lut->n_atoms = n_atoms; lut->ndirs = ndirs;
    lut->n_wm = n_wm; lut->is_exvivo = is_exvivo;
    lut->ldA = (n_atoms & 1) ? n_atoms : n_atoms + 1;         // odd: conflict-free LDS columns
    lut->tile_stride = (nS * lut->ldA + 3) & ~3;
    int rc;
    std::vector<int> ones;
    std::vector<float> fix;
    if (is_exvivo) { ones.push_back(1); fix.insert(fix.end(), nS, 1.0f); }   // models.pyx:843-844
    ones.push_back(0); fix.insert(fix.end(), iso, iso + nS);
    if ((rc = build_tiles(ctx, lut, wm, (size_t)n_wm * ndirs * nS, fix.data(), fix.size(), ones, n_wm))) { amx_lut_destroy(lut); return rc; }
    // rows of stage 2 (models.pyx:820, 917-921): j+1 if nS == 1+dwi_count ("single_b0") else dwi_idx[j]
    std::vector<unsigned char> rowdwi(nS, 0);
    const bool single_b0 = (nS == 1 + dwi_count);
    for (int j = 0; j < dwi_count; j++) {
        const int64_t row = single_b0 ? j + 1 : dwi_idx[j];
        if (row < 0 || row >= nS) { amx_lut_destroy(lut); return bad(ctx, "amx_lut_upload_noddi: dwi_idx out of range"); }
        rowdwi[row] = 1;
    }
    // Are the rows outside stage 2 (the b0 volumes) exactly 1.0 in every atom, as resample_kernel leaves them (lut.pyx:298, 305)?
    // Then the stage-2 products of an unclipped voxel derive from the stage-1 table (k_noddi_gemm); otherwise every voxel takes
    // the exact pass.
    lut->n_dwi = dwi_count;
    lut->s2_derive = is_exvivo ? 0 : 1;
    for (int i = 0; i < nS && lut->s2_derive; i++) {
        if (rowdwi[i]) { if (!(iso[i] > 1e-30f) || !(iso[i] <= 3.0e38f)) lut->s2_derive = 0; continue; }
        if (iso[i] != 1.0f) lut->s2_derive = 0;
        for (size_t kd = 0; kd < (size_t)n_wm * ndirs && lut->s2_derive; kd++) if (wm[kd * nS + i] != 1.0f) lut->s2_derive = 0;
    }
    std::vector<double> colscale(n_atoms, 1.0);
    for (int k = 0; k < n_wm; k++) colscale[k] = dwi_count > 0 ? norms[k] : 1.0;   // rows of norms are identical
    std::vector<short> ht(htable, htable + 181 * 181);
    if ((rc = upload(ctx, &lut->rowdwi, rowdwi.data(), rowdwi.size())) ||
        (rc = upload(ctx, &lut->colscale, colscale.data(), colscale.size())) ||
        (rc = upload(ctx, &lut->icvf, icvf, (size_t)n_wm)) || (rc = upload(ctx, &lut->kappa, kappa, (size_t)n_wm)) ||
        (rc = upload(ctx, &lut->htable, ht.data(), ht.size()))) { amx_lut_destroy(lut); return rc; }
    // Gram matrices of every orientation (all rows for the NNLS stages, stage-2 rows for the LASSO):
    // they let the solver update the dual vector without sweeping the tile (amx_solver.hpp)
    {
        if (!ctx->opt_no_gram) {
            lut->ldG = n_atoms <= 192 ? 192 : 256;            // (>= 64 atoms per lane-row of the solvers' column reads)
            const size_t gbytes = (size_t)ndirs * n_atoms * lut->ldG * sizeof(double);
            const size_t lds_tile = (size_t)nS * lut->ldA * sizeof(float);
            const int in_lds = lds_tile <= 160 * 1024 ? 1 : 0;
            const size_t lds = in_lds ? lds_tile : 0;
            HIPCHK(ctx, hipMalloc((void **)&lut->gram, gbytes));
            HIPCHK(ctx, hipMalloc((void **)&lut->gram_dwi, gbytes));
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_build_gram),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_build_gram, dim3(ndirs), dim3(512), lds, nullptr, (const float *)lut->tiles,
                               lut->tile_stride, nS, lut->ldA, n_atoms, (const unsigned char *)nullptr, lut->ldG, lut->gram, in_lds);
            hipLaunchKernelGGL(k_build_gram, dim3(ndirs), dim3(512), lds, nullptr, (const float *)lut->tiles,
                               lut->tile_stride, nS, lut->ldA, n_atoms, (const unsigned char *)lut->rowdwi, lut->ldG, lut->gram_dwi, in_lds);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipDeviceSynchronize());
            // compressed basis of every orientation: support seeds of the NNLS stages (amx_seed.hpp)
            // (ex-vivo dictionaries too: the dot atom -- a column of ones -- is one more atom; their stage-2 products always take the exact
            //  pass, s2_derive = 0: y2 = y - x_iso iso - x_dot is not a function of x_iso alone)
            if ((rc = amx_build_basis(ctx, lut))) { amx_lut_destroy(lut); return rc; }
        }
    }
    *out = lut;
    return AMX_OK;
}

int amx_lut_upload_freewater(amx_ctx *ctx, const float *D, const float *CSF, const int16_t *htable,
                             int n_perp, int n_iso, int ndirs, int nS, amx_lut **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!D || !CSF || !htable || !out || n_perp <= 0 || n_iso <= 0 || ndirs <= 0 || nS <= 0)
        return bad(ctx, "amx_lut_upload_freewater: bad argument");
    const int n_atoms = n_perp + n_iso;
    if (n_atoms > 64 || nS > 512) return bad(ctx, "amx_lut_upload_freewater: unsupported size (n_atoms <= 64, nS <= 512)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prefetch_stage_pool(ctx);            // (the host threads of the float32 transport are made beside this upload)
    amx_lut *lut = new amx_lut();
    lut->ctx = ctx; lut->model = 2; lut->nS = nS; lut->n_atoms = n_atoms; lut->ndirs = ndirs;
    lut->n_perp = n_perp; lut->n_iso = n_iso;
    lut->ldA = (n_atoms & 1) ? n_atoms : n_atoms + 1;
    lut->tile_stride = (nS * lut->ldA + 3) & ~3;
    int rc;
    std::vector<int> ones(n_iso, 0);
    if ((rc = build_tiles(ctx, lut, D, (size_t)n_perp * ndirs * nS, CSF, (size_t)n_iso * nS, ones, n_perp))) { amx_lut_destroy(lut); return rc; }
    std::vector<short> ht(htable, htable + 181 * 181);
    if ((rc = upload(ctx, &lut->htable, ht.data(), ht.size()))) { amx_lut_destroy(lut); return rc; }
    *out = lut;
    return AMX_OK;
}

int amx_lut_upload_sandi(amx_ctx *ctx, const double *signal, const double *norms, const double *Rs,
                         const double *d_in, const double *d_isos, int nS, int n_rs, int n_in,
                         int n_iso, amx_lut **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!signal || !norms || !Rs || !d_in || !d_isos || !out || nS <= 0 || n_rs < 0 || n_in < 0 || n_iso < 0)
        return bad(ctx, "amx_lut_upload_sandi: bad argument");
    const int n_atoms = n_rs + n_in + n_iso;
    if (n_atoms <= 0 || n_atoms > 64 || nS > 128) return bad(ctx, "amx_lut_upload_sandi: unsupported size (n_atoms <= 64, nS <= 128)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prefetch_stage_pool(ctx);            // (the host threads of the float32 transport are made beside this upload)
    amx_lut *lut = new amx_lut();
    lut->ctx = ctx; lut->model = 3; lut->nS = nS; lut->n_atoms = n_atoms; lut->ndirs = 1;
    lut->n_rs = n_rs; lut->n_in = n_in; lut->n_isos = n_iso;
    lut->ldA = (n_atoms & 1) ? n_atoms : n_atoms + 1;
    lut->tile_stride = (nS * lut->ldA + 3) & ~3;
    std::vector<double> tile((size_t)lut->tile_stride + 8, 0.0);
    for (int j = 0; j < n_atoms; j++)
        for (int i = 0; i < nS; i++) tile[(size_t)i * lut->ldA + j] = signal[(size_t)j * nS + i];   // col-major in
    int rc;
    double *dt = nullptr;
    if ((rc = upload(ctx, &dt, tile.data(), tile.size())) || (rc = upload(ctx, &lut->norms, norms, (size_t)n_atoms)) ||
        (rc = upload(ctx, &lut->Rs, Rs, (size_t)(n_rs ? n_rs : 1))) || (rc = upload(ctx, &lut->d_in, d_in, (size_t)(n_in ? n_in : 1))) ||
        (rc = upload(ctx, &lut->d_isos, d_isos, (size_t)(n_iso ? n_iso : 1)))) { lut->tiles = dt; amx_lut_destroy(lut); return rc; }
    lut->tiles = dt;
    *out = lut;
    return AMX_OK;
}

int amx_lut_upload_czb(amx_ctx *ctx, const float *wmr, const float *wmh, const float *iso, const double *Rs,
                       const int16_t *htable, int n_rs, int n_perp, int n_iso, int ndirs, int nS, amx_lut **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!wmr || !wmh || !iso || !Rs || !htable || !out || n_rs <= 0 || n_perp <= 0 || n_iso <= 0 || ndirs <= 0 || nS <= 0)
        return bad(ctx, "amx_lut_upload_czb: bad argument");
    const int n_atoms = n_rs + n_perp + n_iso;
    if (n_atoms > 64 || nS > 512) return bad(ctx, "amx_lut_upload_czb: unsupported size (n_atoms <= 64, nS <= 512)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prefetch_stage_pool(ctx);            // (the host threads of the float32 transport are made beside this upload)
    amx_lut *lut = new amx_lut();
    lut->ctx = ctx; lut->model = 4; lut->nS = nS; lut->n_atoms = n_atoms; lut->ndirs = ndirs;
    lut->n_rs = n_rs; lut->n_perp = n_perp; lut->n_iso = n_iso;
    lut->ldA = (n_atoms & 1) ? n_atoms : n_atoms + 1;
    lut->tile_stride = (nS * lut->ldA + 3) & ~3;
    int rc;
    // columns: cylinders, zeppelins (both per orientation), balls (models.pyx:608-610)
    std::vector<float> rot((size_t)(n_rs + n_perp) * ndirs * nS);
    memcpy(rot.data(), wmr, (size_t)n_rs * ndirs * nS * sizeof(float));
    memcpy(rot.data() + (size_t)n_rs * ndirs * nS, wmh, (size_t)n_perp * ndirs * nS * sizeof(float));
    std::vector<int> ones(n_iso, 0);
    if ((rc = build_tiles(ctx, lut, rot.data(), rot.size(), iso, (size_t)n_iso * nS, ones, n_rs + n_perp))) { amx_lut_destroy(lut); return rc; }
    std::vector<short> ht(htable, htable + 181 * 181);
    if ((rc = upload(ctx, &lut->htable, ht.data(), ht.size())) || (rc = upload(ctx, &lut->Rs, Rs, (size_t)n_rs))) { amx_lut_destroy(lut); return rc; }
    // Gram matrices of every orientation: the solver works on A'A + lambda2 I (amx_gram_solver.hpp)
    lut->ldG = 64;
    const size_t gbytes = (size_t)ndirs * n_atoms * lut->ldG * sizeof(double);
    const size_t lds = (size_t)nS * lut->ldA * sizeof(float);
    if (hipMalloc((void **)&lut->gram, gbytes) != hipSuccess) { amx_lut_destroy(lut); return bad(ctx, "amx_lut_upload_czb: out of device memory"); }
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_build_gram), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_build_gram, dim3(ndirs), dim3(512), lds, nullptr, (const float *)lut->tiles, lut->tile_stride, nS,
                       lut->ldA, n_atoms, (const unsigned char *)nullptr, lut->ldG, lut->gram);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    *out = lut;
    return AMX_OK;
}

// status words home (the pinned mirror, written from the device) and cleared for the next call: one launch where a copy and two
// memsets were three nodes of the stream (~11 us each at the end of every call)
__global__ void k_status_home(int *__restrict__ st, int *__restrict__ home)
{
    const int i = threadIdx.x;
    if (i < ST_WORDS) {
        home[i] = st[i];
        st[i] = (i == ST_ERRPACK || i == ST_ERRPACK + 1) ? 0x7f7f7f7f : 0;
    }
    __threadfence_system();
}

int amx_sync_status(amx_ctx *ctx, void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    static_assert(ST_WORDS <= 128, "k_status_home: one thread per status word");
    hipLaunchKernelGGL(k_status_home, dim3(1), dim3(128), 0, s, ctx->status_d, ctx->status_h);
#ifdef AMX_PHASES
    unsigned long long ph_[16];
    if (ctx->misc.p) HIPCHK(ctx, hipMemcpyAsync(ph_, (int *)ctx->misc.p + 16, sizeof ph_, hipMemcpyDeviceToHost, s));
#endif
    int rc = AMX_OK;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(s));
#ifdef AMX_PHASES
    if (ctx->misc.p) {
        static const char *nm[8] = {"decode", "columns+gram", "cholesky", "solve+residual", "refinement", "norms", "screening", "exact-dots"};   // certified voxels: phases of certify_seed
        for (int st_ = 0; st_ < 2; st_++) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; k++) tot += ph_[st_ * 8 + k];
            fprintf(stderr, "[amx] NNLS stage %d phases:", st_ == 0 ? 1 : 3);
            for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f%%", nm[k], tot ? 100.0 * ph_[st_ * 8 + k] / tot : 0.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    const int *st = ctx->status_h;
    ctx->stats[0] = st[ST_RERUN];
    ctx->stats[1] = st[ST_ITCAP];
    ctx->stats[2] = st[ST_OVERFLOW];
    ctx->stats[3] = ((int64_t)st[ST_GUARD] << 32) | (unsigned)st[ST_GUARDVOX];
    ctx->seed_stats[0] = ctx->seeded_vox; ctx->seeded_vox = 0;
    for (int k = 0; k < 3; k++) { ctx->seed_stats[1 + k] = st[ST_LEFT + k] + ctx->uncert_vox[k]; ctx->uncert_vox[k] = 0; }
    ctx->seed_stats[4] = st[ST_CLIP];
    if (amx_debug()) fprintf(stderr, "[amx] dual-vector evaluations per stage: exact %d %d %d  gram %d %d %d  inner iterations %d %d %d\n", st[ST_EXACT], st[ST_EXACT + 1], st[ST_EXACT + 2], st[ST_GRAM], st[ST_GRAM + 1], st[ST_GRAM + 2], st[ST_ITERS], st[ST_ITERS + 1], st[ST_ITERS + 2]);
    if (amx_debug()) fprintf(stderr, "[amx] seeds: stage 1 tried %d certified %d, stage 3 tried %d certified %d; seed solver trips %d lane-trips used %d; stage-1 refusals: malformed %d pivot %d refinement %d x<=0 %d dual %d\n", st[ST_SEED], st[ST_SEED + 1], st[ST_SEED + 2], st[ST_SEED + 3], st[ST_SEED + 4], st[ST_SEED + 5], st[ST_SEED + 7], st[ST_SEED + 8], st[ST_SEED + 9], st[ST_SEED + 10], st[ST_SEED + 11]);
    if (amx_debug()) fprintf(stderr, "[amx] screened certificates: %d exact dot products (NNLS stages), %d (LASSO stage)\n", st[ST_SEED + 22], st[ST_SEED + 23]);
    if (amx_debug()) fprintf(stderr, "[amx] Gram certificates stage 1: %d voxels, %d certified (pivot ratio %d, x <= 0 %d, dual %d), %d dual values; stage 3: %d voxels, %d certified (pivot %d, x <= 0 %d, dual %d), %d dual values\n",
                             st[ST_SEED + 24], st[ST_SEED + 25], st[ST_SEED + 26], st[ST_SEED + 27], st[ST_SEED + 28], st[ST_SEED + 29], st[ST_SEED + 30], st[ST_SEED + 31], st[ST_SEED + 32], st[ST_SEED + 33], st[ST_SEED + 34], st[ST_SEED + 35]);
    if (amx_debug()) fprintf(stderr, "[amx] Gram certificates LASSO: %d voxels, %d certified (more than 12 atoms %d, x <= 0 %d, dual %d), %d dual values\n",
                             st[ST_SEED + 36], st[ST_SEED + 37], st[ST_SEED + 38], st[ST_SEED + 39], st[ST_SEED + 40], st[ST_SEED + 41]);
    if (amx_debug()) fprintf(stderr, "[amx] Gram certificates LASSO, second pass: %d voxels, %d certified (more than 18 atoms %d, x <= 0 %d, dual %d), %d dual values\n",
                             st[ST_SEED + 48], st[ST_SEED + 49], st[ST_SEED + 50], st[ST_SEED + 51], st[ST_SEED + 52], st[ST_SEED + 53]);
    if (amx_debug()) fprintf(stderr, "[amx] LASSO seeds: tried %d certified %d; seed solver trips %d lane-trips used %d\n", st[ST_SEED + 18], st[ST_SEED + 19], st[ST_SEED + 20], st[ST_SEED + 21]);
    if (amx_debug()) fprintf(stderr, "[amx] seed solver kcycles (wave sums / 1024): take %d solve+drop %d residual %d scan %d append %d store %d\n", st[ST_SEED + 12], st[ST_SEED + 13], st[ST_SEED + 14], st[ST_SEED + 15], st[ST_SEED + 16], st[ST_SEED + 17]);
    if (amx_debug()) fprintf(stderr, "[amx] Gram certificate kcycles (decode | gather+factor+solve | screening | exact duals | output): stage 1 %d %d %d %d %d, stage 3 %d %d %d %d %d\n",
                             st[ST_SEED + 60], st[ST_SEED + 61], st[ST_SEED + 62], st[ST_SEED + 63], st[ST_SEED + 64], st[ST_SEED + 65], st[ST_SEED + 66], st[ST_SEED + 67], st[ST_SEED + 68], st[ST_SEED + 69]);
    if (amx_debug()) fprintf(stderr, "[amx] LASSO Gram certificate kcycles (decode | gather+factor+solve | screening | exact duals | output): %d %d %d %d %d\n",
                             st[ST_SEED + 70], st[ST_SEED + 71], st[ST_SEED + 72], st[ST_SEED + 73], st[ST_SEED + 74]);
    if (amx_debug()) fprintf(stderr, "[amx] stage-3 seed solver kcycles: take %d solve+drop %d residual %d scan %d append %d store %d\n", st[ST_SEED + 54], st[ST_SEED + 55], st[ST_SEED + 56], st[ST_SEED + 57], st[ST_SEED + 58], st[ST_SEED + 59]);
    if (amx_debug() && st[ST_GRAM + 1] > 0) fprintf(stderr, "[amx] k_noddi_lasso_big: %d voxels, %.1f pivoting steps per voxel\n", st[ST_GRAM + 1], (double)st[ST_ITERS + 1] / st[ST_GRAM + 1]);
    {
        // first offending voxel and what it held, as the kernels' one 64-bit atomicMin left them
        int *sth = ctx->status_h;
        const unsigned lo = (unsigned)sth[ST_ERRPACK], hi = (unsigned)sth[ST_ERRPACK + 1];
        sth[ST_ERRVOX] = (int)hi;
        if (hi != 0x7f7f7f7fu) {
            if (sth[ST_ERRKIND] == 1) sth[ST_II1] = (int)lo;                                   // (ST_II2 = number of dictionaries, stored by the kernel)
            else { sth[ST_II1] = (int)(lo >> 16) - 1; sth[ST_II2] = (int)(lo & 0xffffu) - 1; }
        }
    }
    if (st[ST_ERRVOX] != 0x7f7f7f7f) {
        char b[256];
        snprintf(b, sizeof b, "\"amico.lut.dir_to_lut_idx\" index out of bounds (%d, %d) [voxel %d]", st[ST_II1], st[ST_II2], st[ST_ERRVOX]);
        ctx->err = b;
        return AMX_E_DIR_OOB;
    }
    if (st[ST_OVERFLOW] > 0) {
        char b[256];
        snprintf(b, sizeof b, "%d voxel(s) exceeded the largest supported active set", st[ST_OVERFLOW]);
        ctx->err = b;
        return AMX_E_OVERFLOW;
    }
    return AMX_OK;
}

int amx_set_progress(amx_ctx *ctx, void (*callback)(int64_t done, int64_t total, void *user), void *user)
{
    if (!ctx) return AMX_E_BADARG;
    ctx->progress = callback;
    ctx->progress_user = user;
    return AMX_OK;
}

// diagnosis / tests: copy a workspace buffer of the LAST fit (0 perm int32[n], 1 y~ f64[n][12], 2 seeds u64[n]) or a
// dictionary table (10 basis U f64[ndirs][nS][12], 11 compressed dictionary S f64[ndirs][n_atoms][12]) to the host
int amx_debug_fetch(amx_ctx *ctx, const amx_lut *lut, int which, void *dst, size_t bytes)
{
    if (!ctx || !dst) return AMX_E_BADARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    const void *src = nullptr;
    switch (which) {
    case 0: src = ctx->perm.p; break;
    case 1: src = ctx->ytil.p; break;
    case 2: src = ctx->seeds.p; break;
    case 3: src = ctx->ytil2.p; break;
    case 4: src = ctx->seeds2.p; break;
    case 5: src = ctx->cgemm.p; break;
    case 6: src = ctx->schunks.p; break;
    case 7: src = ctx->misc.p; break;
    case 10: src = lut ? lut->basis_U : nullptr; break;
    case 11: src = lut ? lut->basis_S : nullptr; break;
    case 12: src = lut ? lut->basis2_U : nullptr; break;
    case 13: src = lut ? lut->basis2_S : nullptr; break;
    default: break;
    }
    if (!src) return bad(ctx, "amx_debug_fetch: no such buffer");
    HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return AMX_OK;
}

int amx_set_debug_x(amx_ctx *ctx, double *d_x)
{
    if (!ctx) return AMX_E_BADARG;
    ctx->dbg_x = d_x;
    return AMX_OK;
}

int amx_selftest(amx_ctx *ctx, double *out512)
{
    if (!ctx || !out512) return AMX_E_BADARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->hest, 768 * sizeof(double)))) return rc;
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, nullptr, (double *)ctx->hest.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpy(out512, ctx->hest.p, 768 * sizeof(double), hipMemcpyDeviceToHost));
    return AMX_OK;
}

int amx_set_profiling(amx_ctx *ctx, int enable)
{
    if (!ctx) return AMX_E_BADARG;
    ctx->profiling = (enable >= 0 && enable <= 11) ? enable : 1;
    for (int k = 0; k < kEv; k++) ctx->ev_valid[k] = false;
    return AMX_OK;
}

int amx_last_kernel_ms(amx_ctx *ctx, int which, float *out_ms)
{
    if (!ctx || !out_ms || which < 0 || which > 9) return AMX_E_BADARG;
    const int a = which == 0 ? 0 : 2 * which, b = which == 0 ? 1 : 2 * which + 1;
    if (!ctx->ev_valid[a] || !ctx->ev_valid[b]) return bad(ctx, "amx_last_kernel_ms: no profiled call");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[b]));
    HIPCHK(ctx, hipEventElapsedTime(out_ms, ctx->ev[a], ctx->ev[b]));
    return AMX_OK;
}

int amx_last_stats(amx_ctx *ctx, int64_t out[4])
{
    if (!ctx || !out) return AMX_E_BADARG;
    for (int k = 0; k < 4; k++) out[k] = ctx->stats[k];
    return AMX_OK;
}

int amx_last_host_narrowed(amx_ctx *ctx) { return ctx ? ctx->host_narrowed : 0; }

int amx_set_call_voxels(amx_ctx *ctx, int64_t total)
{
    if (!ctx || total < 0) return AMX_E_BADARG;
    ctx->call_total_vox = total;
    return AMX_OK;
}

int amx_host_pool_info(amx_ctx *ctx, int out[4])
{
    if (!ctx || !out) return AMX_E_BADARG;
    out[0] = ctx->stage ? ctx->stage->threads() : 0;
    out[1] = ctx->stage ? ctx->stage->share_first() : -1;
    out[2] = ctx->stage ? ctx->stage->share_cores() : 0;
    out[3] = ctx->device;
    return AMX_OK;
}

int amx_last_path(amx_ctx *ctx, char *buf, int cap)
{
    if (!ctx || !buf || cap <= 0) return AMX_E_BADARG;
    snprintf(buf, (size_t)cap, "%s", ctx->path.c_str());
    return AMX_OK;
}

int amx_last_seed_stats(amx_ctx *ctx, int64_t out[8])
{
    if (!ctx || !out) return AMX_E_BADARG;
    for (int k = 0; k < 8; k++) out[k] = ctx->seed_stats[k];
    return AMX_OK;
}

// ------------------------------------------------------------------ NODDI
// side stream + events of a forked fit (AMX_FORK), one set per workspace set (fit_host alternates two: swap_work)
static int fork_ready(amx_ctx *ctx)
{
    const int w = ctx->work_idx;
    if (ctx->fork_s[w]) return AMX_OK;
    int lo = 0, hi = 0;
    if (ctx->opt_fork_prio) (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (ctx->opt_fork_cus > 0) {
        // n compute units for the side stream, spread evenly over the mask's bits (one bit per CU)
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int total = ctx->n_cu > 256 ? 256 : ctx->n_cu, n = ctx->opt_fork_cus > total ? total : ctx->opt_fork_cus;
        for (int k = 0; k < n; k++) { const int b = (int)((long long)k * total / n); mask[b >> 5] |= 1u << (b & 31); }
        HIPCHK(ctx, hipExtStreamCreateWithCUMask(&ctx->fork_s[w], (uint32_t)((total + 31) / 32), mask));
    } else {
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->fork_s[w], hipStreamNonBlocking, ctx->opt_fork_prio ? hi : lo));
    }
    for (int k = 0; k < 4; k++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->fork_ev[w][k], hipEventDisableTiming));
    return AMX_OK;
}

static int noddi_fit_dev(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const float *d_y32, const double *d_dirs,
                         int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                         double *d_estimates, double *d_rmse, double *d_nrmse, double *d_mod,
                         void *hip_stream)
{
    if (ctx && !(ctx->in_host_fit && ctx->vox_base > 0)) ctx->path.clear();
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 1 || lut->ctx != ctx) return bad(ctx, "amx_noddi_fit: not a NODDI dictionary of this ctx");
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, "amx_noddi_fit: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if ((!d_y && !d_y32) || !d_dirs || !d_estimates) return bad(ctx, "amx_noddi_fit: null buffer");
    if (((flags & AMX_F_RMSE) && !d_rmse) || ((flags & AMX_F_NRMSE) && !d_nrmse) || ((flags & AMX_F_MODULATED) && !d_mod))
        return bad(ctx, "amx_noddi_fit: flag set but output buffer is null");
    if (!(lambda2 >= 0.0) || !(lambda1 >= 0.0)) return bad(ctx, "amx_noddi_fit: need lambda1 >= 0 and lambda2 >= 0");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Plan pl; int rc;
    bool joined1 = false, fork2 = false;
    ctx->call_vox = ctx->in_host_fit ? ctx->host_total_vox : n_vox;
    const bool seeds = lut->basis_S != nullptr && lut->gram != nullptr && !ctx->opt_no_seed &&
                       (ctx->in_host_fit ? ctx->host_total_vox : n_vox) >= ctx->opt_seed_min_voxels;   // (batches of one host call all take the same path: bit-identical to the one-shot call)
    const int gemm_ks = seeds ? amx_gemm_ksteps(lut) : 0;                        // 0: no table kernels for this shape (seeds certified on the true residual only)
    if ((rc = make_plan(ctx, n_vox, lut->ndirs, pl, seeds, gemm_rows(lut->n_atoms)))) return rc;
    if ((rc = ensure(ctx, ctx->xiso, (size_t)n_vox * 2 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->supp, (size_t)n_vox * 4 * sizeof(unsigned long long)))) return rc;
    clear_events(ctx);
    rec(ctx, 0, s);
    // (voxels with an out-of-bounds direction are skipped: k_dir_to_lut gives them defined (zero) maps; every other voxel's maps are
    //  written by the kernel that settles its stage 3 -- tests/test_gpu_parity.py::test_noddi_fit_writes_every_voxel)
    if ((rc = enqueue_bucketing(ctx, lut, d_dirs, n_vox, pl, s, kChunk, d_estimates, 3 + (lut->is_exvivo ? 1 : 0)))) return rc;
    NoddiArgs a;
    memset(&a, 0, sizeof a);
    a.c.tiles = lut->tiles; a.c.y = d_y; a.c.y32 = d_y32; a.c.perm = pl.perm; a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
    a.c.lutidx = pl.lutidx; a.c.status = ctx->status_d; a.c.nS = lut->nS; a.c.ldA = lut->ldA;
    a.c.n_atoms = lut->n_atoms; a.c.tile_stride = lut->tile_stride; a.c.lam1 = lambda1; a.c.lam2 = lambda2; a.c.flags = flags;
    a.rowdwi = lut->rowdwi; a.colscale = lut->colscale; a.icvf = lut->icvf; a.kappa = lut->kappa;
    a.n_wm = lut->n_wm; a.is_exvivo = lut->is_exvivo; a.n_maps = 3 + (lut->is_exvivo ? 1 : 0);
    a.gram = lut->gram; a.gram_dwi = lut->gram_dwi; a.ldG = lut->ldG;
    if (flags & AMX_F_DEBUG_X) {
        if (!ctx->dbg_x) return bad(ctx, "amx_noddi_fit: AMX_F_DEBUG_X without a buffer (amx_set_debug_x)");
        a.c.xdbg = ctx->dbg_x + (size_t)ctx->vox_base * 3 * lut->n_atoms;
    }
    a.xiso = (double *)ctx->xiso.p; a.supp = (unsigned long long *)ctx->supp.p;
    a.est = d_estimates; a.rmse = (flags & AMX_F_RMSE) ? d_rmse : nullptr;
    a.nrmse = (flags & AMX_F_NRMSE) ? d_nrmse : nullptr; a.mod = (flags & AMX_F_MODULATED) ? d_mod : nullptr;
    if (seeds) {
        ctx->seeded_vox += n_vox;
        // y~ = U'y once; the seed solver proposes the stage's support, the stage kernel certifies it (amx_seed.hpp)
        rec(ctx, 10, s);
        const bool gcert = !ctx->opt_no_gcert && gemm_ks > 0;
        if (gcert && (rc = amx_launch_noddi_gemm(ctx, lut, a, pl, s, false))) return rc;
        if (!gcert && (rc = amx_launch_noddi_project(ctx, lut, a, pl, s))) return rc;      // (the GEMM writes y~ as well)
        if (!ctx->opt_no_screen) { a.scr_S = lut->screen_S; a.scr_kappa = lut->screen_kappa; a.scr_ytil = (const double *)ctx->ytil.p; a.scr_Sg = lut->basis_S; }
        if (ctx->opt_seed_stages & 1) {
            a.seeds = (const unsigned long long *)ctx->seeds.p;
            rec(ctx, 16, s);
            if ((rc = amx_launch_noddi_seed(ctx, lut, a, pl, s, 1))) return rc;
            rec(ctx, 17, s);
            if (!gcert) ctx->uncert_vox[0] += n_vox;
            if (gcert) {
                size_t off = 0; const int *cnt = nullptr;
                if ((rc = amx_launch_noddi_gcert(ctx, lut, a, pl, s, 1, &off, &cnt))) return rc;
                a.done = ctx->opt_no_hard_first ? nullptr : (const unsigned char *)ctx->done.p;
                a.rlist = (const int *)ctx->rlist.p + off; a.rcount = cnt;
                a.c.chunks = pl.schunks; a.c.n_chunks = pl.n_chunks + 1;      // the stage kernel walks the left-over lists of the second plan
            }
        }
        rec(ctx, 11, s);
    }
    // AMX_FORK bit 0 (TIMING PROBE, not a fit): the stage-1 left-over kernel on the side stream beside the LASSO seed solver, which reads
    // the x_iso the previous call left for those voxels; joined before the LASSO certificates
    const bool fork1 = (ctx->opt_fork & 1) && a.rlist != nullptr;
    hipStream_t fs = nullptr;
    if (ctx->opt_fork) { if ((rc = fork_ready(ctx))) return rc; fs = ctx->fork_s[ctx->work_idx]; }
    hipEvent_t *fev = ctx->fork_ev[ctx->work_idx];
    if (fork1) {
        HIPCHK(ctx, hipEventRecord(fev[0], s));
        HIPCHK(ctx, hipStreamWaitEvent(fs, fev[0], 0));
        ctx->side_launch = true;
        rc = amx_launch_noddi_s1(ctx, a, pl, fs);
        ctx->side_launch = false;
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(fev[1], fs));
    } else if ((rc = amx_launch_noddi_s1(ctx, a, pl, s))) return rc;
    progress_tick(ctx, s, n_vox / 3, n_vox);                       // (three stages: a third of the work each, roughly)
    a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks; a.rlist = nullptr; a.rcount = nullptr; a.done = nullptr;
    // the LASSO seeds need x_iso: Gram-space solver only (lambda2 >= 1e-5), with the default dictionary shape
    if (seeds && (ctx->opt_seed_stages & 4) && lut->basis2_S != nullptr && lambda2 >= 1e-5 && (gemm_ks > 0 || lut->nS <= 128) && !ctx->opt_lasso_qr &&
        (lambda1 > 0.0 || ctx->opt_no_big_all || lut->n_wm <= 64)) {       // (lambda1 = 0: a dense optimum -- no seeds to propose, amx_launch_noddi_s2 goes to k_noddi_lasso_big)
        const bool gcert2 = !ctx->opt_no_gcert && gemm_ks > 0 && lut->screen2_kappa0 != nullptr && lut->u2iso != nullptr;
        rec(ctx, 12, s);
        // y2~ of every voxel and c2 = A2'y2, ||y2||^2 of the unclipped ones derive from the stage-1 table; the clipped voxels' exactly
        if (gcert2 && (rc = amx_launch_noddi_s2prep(ctx, lut, a, pl, s))) return rc;
        rec(ctx, 18, s);
        if ((rc = amx_launch_noddi_seed2(ctx, lut, a, pl, s, gcert2))) return rc;
        rec(ctx, 19, s);
        a.seeds2 = (const unsigned long long *)ctx->seeds2.p;
        a.list_is_pos = 1;
        if (!ctx->opt_no_screen && lut->screen2_S) { a.scr2_S = lut->screen2_S; a.scr2_kappa = lut->screen2_kappa; a.scr2_ytil = (const double *)ctx->ytil2.p; a.scr2_Sg = lut->basis2_S; }
        if (!gcert2) ctx->uncert_vox[1] += n_vox;
        if (fork1) { HIPCHK(ctx, hipStreamWaitEvent(s, fev[1], 0)); joined1 = true; }
        if (gcert2) {
            const bool wide = !ctx->opt_no_gcert_wide;
            if ((rc = amx_launch_noddi_gcert2(ctx, lut, a, pl, s, wide))) return rc;
            a.cand_lists = 1;       // (k_lasso_gcert: the candidate lists of stage 3 wait in seeds2 for the voxels it settled)
            const bool third = amx_gcert2_third(ctx, lut, wide);
            a.rlist = (const int *)ctx->rlist.p + amx_gcert2_leftover_offset(pl, wide, third); a.rcount = amx_gcert2_leftover_counts(pl, wide, third);   // (two wide passes end in the first half again)
            a.c.chunks = pl.schunks; a.c.n_chunks = pl.n_chunks + 1;
            // AMX_FORK bit 1: the voxels these certificates left over (0.6 %) do not come back to the lane kernels -- k_noddi<4> and then
            // k_noddi<3> (no seed: Lawson-Hanson on the support it has just found, plus iso) finish them, a wavefront per voxel, on the side
            // stream, while k_nnls_seed<3> / k_nnls_gcert<3> work on everybody else (they skip the voxels whose certificate flag is not 1)
            fork2 = (ctx->opt_fork & 2) && (ctx->opt_seed_stages & 2) && !ctx->opt_no_gcert && gemm_ks > 0;
        }
        rec(ctx, 13, s);
    }
    if (fork1 && !joined1) HIPCHK(ctx, hipStreamWaitEvent(s, fev[1], 0));
    if (fork2) {
        HIPCHK(ctx, hipEventRecord(fev[2], s));
        HIPCHK(ctx, hipStreamWaitEvent(fs, fev[2], 0));
        ctx->side_launch = true;
        NoddiArgs b = a;
        rc = amx_launch_noddi_s2(ctx, b, pl, fs);
        if (!rc) {
            b = a;      // (same left-over lists, same chunks: now stage 3 without seeds)
            b.seeds = nullptr; b.done = nullptr; b.seeds2 = nullptr; b.cand_lists = 0;
            rc = amx_launch_noddi_s3(ctx, b, pl, fs);
        }
        ctx->side_launch = false;
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(fev[3], fs));
        a.fork_l2 = 1;
    }
    if (fork2 || !(rc = amx_launch_noddi_s2(ctx, a, pl, s))) {
        progress_tick(ctx, s, 2 * (n_vox / 3), n_vox);
        a.seeds = nullptr; a.done = nullptr; a.rlist = nullptr; a.rcount = nullptr;
        a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
        if (seeds && (ctx->opt_seed_stages & 2)) {
            a.seeds = (const unsigned long long *)ctx->seeds.p;
            rec(ctx, 14, s);
            rc = amx_launch_noddi_seed(ctx, lut, a, pl, s, 3);
            const bool gcert3 = !ctx->opt_no_gcert && gemm_ks > 0;
            if (!gcert3) ctx->uncert_vox[2] += n_vox;
            if (!rc && gcert3) {
                size_t off = 0; const int *cnt = nullptr;
                rc = amx_launch_noddi_gcert(ctx, lut, a, pl, s, 3, &off, &cnt);
                a.done = ctx->opt_no_hard_first ? nullptr : (const unsigned char *)ctx->done.p;
                a.rlist = (const int *)ctx->rlist.p + off; a.rcount = cnt;
                a.c.chunks = pl.schunks; a.c.n_chunks = pl.n_chunks + 1;
            }
            rec(ctx, 15, s);
        }
        if (!rc) rc = amx_launch_noddi_s3(ctx, a, pl, s);
    }
    if (fork2) HIPCHK(ctx, hipStreamWaitEvent(s, fev[3], 0));      // the side stream's voxels are part of this fit
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, (const int *)ctx->misc.p, ctx->status_d);
    rec(ctx, 1, s);
    if (!rc) progress_tick(ctx, s, n_vox, n_vox);
    return rc;
}

// ------------------------------------------------------------------ FreeWater
static int freewater_fit_dev(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const float *d_y32,
                             const double *d_dirs, int64_t n_vox, double lambda1, double lambda2,
                             int is_mouse, unsigned flags, double *d_estimates, double *d_rmse,
                             double *d_nrmse, double *d_ycorr, void *hip_stream)
{
    if (ctx && !(ctx->in_host_fit && ctx->vox_base > 0)) ctx->path.clear();
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 2 || lut->ctx != ctx) return bad(ctx, "amx_freewater_fit: not a FreeWater dictionary of this ctx");
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, "amx_freewater_fit: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if ((!d_y && !d_y32) || !d_dirs || !d_estimates) return bad(ctx, "amx_freewater_fit: null buffer");
    if (((flags & AMX_F_RMSE) && !d_rmse) || ((flags & AMX_F_NRMSE) && !d_nrmse) || ((flags & AMX_F_CORRECTED) && !d_ycorr))
        return bad(ctx, "amx_freewater_fit: flag set but output buffer is null");
    if (!(lambda2 >= 0.0) || !(lambda1 >= 0.0)) return bad(ctx, "amx_freewater_fit: need lambda1 >= 0 and lambda2 >= 0");
    if (is_mouse && lut->n_iso < 2) return bad(ctx, "amx_freewater_fit: Mouse needs two isotropic atoms");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Plan pl; int rc;
    if ((rc = make_plan(ctx, n_vox, lut->ndirs, pl))) return rc;
    clear_events(ctx);
    rec(ctx, 0, s);
    const bool refill = amx_use_lane_solver(ctx, lut->n_atoms, lambda2) && amx_fw_use_refill(ctx, lut->n_atoms, lut->nS, flags, lambda2);
    // (no memset of the maps: skipped voxels are zeroed by k_dir_to_lut, every other voxel is written -- test_freewater_fit_writes_every_voxel)
    if ((rc = enqueue_bucketing(ctx, lut, d_dirs, n_vox, pl, s, refill ? amx_refill_chunk(ctx, n_vox) : kChunk, d_estimates, is_mouse ? 4 : 2))) return rc;
    FwArgs a;
    memset(&a, 0, sizeof a);
    a.c.tiles = lut->tiles; a.c.y = d_y; a.c.y32 = d_y32; a.c.perm = pl.perm; a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
    a.c.lutidx = pl.lutidx; a.c.status = ctx->status_d; a.c.nS = lut->nS; a.c.ldA = lut->ldA;
    a.c.n_atoms = lut->n_atoms; a.c.tile_stride = lut->tile_stride; a.c.lam1 = lambda1; a.c.lam2 = lambda2; a.c.flags = flags;
    a.n_perp = lut->n_perp; a.n_iso = lut->n_iso; a.is_mouse = is_mouse; a.n_maps = is_mouse ? 4 : 2;
    if (ctx->opt_cold_start) a.c.flags |= 0x80000000u;
    if (flags & AMX_F_DEBUG_X) {
        if (!ctx->dbg_x) return bad(ctx, "amx_freewater_fit: AMX_F_DEBUG_X without a buffer (amx_set_debug_x)");
        a.c.xdbg = ctx->dbg_x + (size_t)ctx->vox_base * lut->n_atoms;
    }
    a.est = d_estimates; a.rmse = (flags & AMX_F_RMSE) ? d_rmse : nullptr;
    a.nrmse = (flags & AMX_F_NRMSE) ? d_nrmse : nullptr; a.ycorr = (flags & AMX_F_CORRECTED) ? d_ycorr : nullptr;
    if (refill && (rc = amx_fw_prepare(ctx, lut, a, s))) return rc;
    rc = amx_launch_fw(ctx, a, pl, s);
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, (const int *)ctx->misc.p, ctx->status_d);
    rec(ctx, 1, s);
    if (!rc) progress_tick(ctx, s, n_vox, n_vox);
    return rc;
}

// ------------------------------------------------------------------ SANDI
static int sandi_fit_dev(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const float *d_y32, int64_t n_vox,
                         double lambda1, double lambda2, unsigned flags, double *d_estimates,
                         double *d_rmse, double *d_nrmse, void *hip_stream)
{
    if (ctx && !(ctx->in_host_fit && ctx->vox_base > 0)) ctx->path.clear();
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 3 || lut->ctx != ctx) return bad(ctx, "amx_sandi_fit: not a SANDI dictionary of this ctx");
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, "amx_sandi_fit: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if ((!d_y && !d_y32) || !d_estimates) return bad(ctx, "amx_sandi_fit: null buffer");
    if (((flags & AMX_F_RMSE) && !d_rmse) || ((flags & AMX_F_NRMSE) && !d_nrmse))
        return bad(ctx, "amx_sandi_fit: flag set but output buffer is null");
    if (!(lambda2 >= 0.0) || !(lambda1 >= 0.0)) return bad(ctx, "amx_sandi_fit: need lambda1 >= 0 and lambda2 >= 0");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Plan pl; int rc;
    if ((rc = make_plan(ctx, n_vox, 1, pl))) return rc;
    clear_events(ctx);
    rec(ctx, 0, s);
    SandiArgs a;
    memset(&a, 0, sizeof a);
    a.c.tiles = lut->tiles; a.c.y = d_y; a.c.y32 = d_y32; a.c.perm = pl.perm; a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
    a.c.lutidx = pl.lutidx; a.c.status = ctx->status_d; a.c.nS = lut->nS; a.c.ldA = lut->ldA;
    a.c.n_atoms = lut->n_atoms; a.c.tile_stride = lut->tile_stride; a.c.lam1 = lambda1; a.c.lam2 = lambda2; a.c.flags = flags;
    a.norms = lut->norms; a.Rs = lut->Rs; a.d_in = lut->d_in; a.d_isos = lut->d_isos;
    a.n_rs = lut->n_rs; a.n_in = lut->n_in; a.n_iso = lut->n_isos;
    if (ctx->opt_cold_start) a.c.flags |= 0x80000000u;
    if (flags & AMX_F_DEBUG_X) {
        if (!ctx->dbg_x) return bad(ctx, "amx_sandi_fit: AMX_F_DEBUG_X without a buffer (amx_set_debug_x)");
        a.c.xdbg = ctx->dbg_x + (size_t)ctx->vox_base * lut->n_atoms;
    }
    a.est = d_estimates; a.rmse = (flags & AMX_F_RMSE) ? d_rmse : nullptr; a.nrmse = (flags & AMX_F_NRMSE) ? d_nrmse : nullptr;
    if ((rc = amx_sandi_prepare(ctx, lut, a, s))) return rc;
    // the row-space kernel (default protocol) takes the voxels in order and counts straight into the status words: one launch
    // per fit; the other SANDI kernels walk the (trivial) plan and use the per-call counters
    const bool rows = a.tables && amx_use_lane_solver(ctx, a.c.n_atoms, a.c.lam2) && !ctx->opt_sandi_atom_space;
    if (rows) a.n_lin = (int)n_vox;
    else {
        HIPCHK(ctx, hipMemsetAsync(ctx->misc.p, 0, 64 * sizeof(int), s));
        const int nb = (int)((n_vox + 255) / 256);
        hipLaunchKernelGGL(k_plan_linear, dim3(nb), dim3(256), 0, s, (int)n_vox, kChunk, pl.chunks, pl.n_chunks, pl.perm);
    }
    rc = amx_launch_sandi(ctx, a, pl, s);
    if (!rows) hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, (const int *)ctx->misc.p, ctx->status_d);
    rec(ctx, 1, s);
    if (!rc) progress_tick(ctx, s, n_vox, n_vox);
    return rc;
}

// ------------------------------------------------------------------ CylinderZeppelinBall
static int czb_fit_dev(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const float *d_y32, const double *d_dirs, int64_t n_vox,
                       double lambda1, double lambda2, unsigned flags, double *d_estimates, double *d_rmse,
                       double *d_nrmse, void *hip_stream)
{
    if (ctx && !(ctx->in_host_fit && ctx->vox_base > 0)) ctx->path.clear();
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 4 || lut->ctx != ctx) return bad(ctx, "amx_czb_fit: not a CylinderZeppelinBall dictionary of this ctx");
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, "amx_czb_fit: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if ((!d_y && !d_y32) || !d_dirs || !d_estimates) return bad(ctx, "amx_czb_fit: null buffer");
    if (((flags & AMX_F_RMSE) && !d_rmse) || ((flags & AMX_F_NRMSE) && !d_nrmse))
        return bad(ctx, "amx_czb_fit: flag set but output buffer is null");
    // (the Gram-space solver needs a ridge -- models.pyx:439 default: 4.0; lambda2 < 1e-6 runs the thin-QR solver in A-space)
    if (!(lambda2 >= 0.0) || !(lambda1 >= 0.0)) return bad(ctx, "amx_czb_fit: need lambda1 >= 0 and lambda2 >= 0");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Plan pl; int rc;
    // the default problem (strong ridge, <= 32 atoms, maps only): complementary form, one voxel per lane (amx_czb.hip)
    const bool fast = lut->n_atoms <= 32 && lambda2 >= 1e-2 && !(flags & (AMX_F_RMSE | AMX_F_NRMSE)) && !ctx->opt_cold_start &&
                      !ctx->opt_wave_per_voxel && lut->nS <= 160 && lut->gram != nullptr;
    if ((rc = make_plan(ctx, n_vox, lut->ndirs, pl, false, 64, fast ? 2048 : 0))) return rc;
    clear_events(ctx);
    rec(ctx, 0, s);
    if ((rc = enqueue_bucketing(ctx, lut, d_dirs, n_vox, pl, s, kChunk, d_estimates, 3))) return rc;      // (no memset of the maps: tests/test_gpu_czb.py::test_czb_fit_writes_every_voxel)
    CzbArgs a;
    memset(&a, 0, sizeof a);
    a.c.tiles = lut->tiles; a.c.y = d_y; a.c.y32 = d_y32; a.c.perm = pl.perm; a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
    a.c.lutidx = pl.lutidx; a.c.status = ctx->status_d; a.c.nS = lut->nS; a.c.ldA = lut->ldA;
    a.c.n_atoms = lut->n_atoms; a.c.tile_stride = lut->tile_stride; a.c.lam1 = lambda1; a.c.lam2 = lambda2; a.c.flags = flags;
    a.n_rs = lut->n_rs; a.n_perp = lut->n_perp; a.Rs = lut->Rs; a.gram = lut->gram; a.ldG = lut->ldG;
    if (ctx->opt_cold_start) a.c.flags |= 0x80000000u;
    if (flags & AMX_F_DEBUG_X) {
        if (!ctx->dbg_x) return bad(ctx, "amx_czb_fit: AMX_F_DEBUG_X without a buffer (amx_set_debug_x)");
        a.c.xdbg = ctx->dbg_x + (size_t)ctx->vox_base * lut->n_atoms;
    }
    a.est = d_estimates; a.rmse = (flags & AMX_F_RMSE) ? d_rmse : nullptr; a.nrmse = (flags & AMX_F_NRMSE) ? d_nrmse : nullptr;
    if (fast) { if (!(rc = amx_czb_prepare(ctx, lut, lambda2, s))) rc = amx_launch_czb_fast(ctx, lut, a, pl, s); }
    else rc = amx_launch_czb(ctx, a, pl, s);
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, (const int *)ctx->misc.p, ctx->status_d);
    rec(ctx, 1, s);
    if (!rc) progress_tick(ctx, s, n_vox, n_vox);
    return rc;
}

}  // extern "C"

#define AMX_H2D(buf, src, bytes)                                                     \
    if ((rc = ensure(ctx, buf, bytes))) return rc;                                   \
    HIPCHK(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, nullptr));

// ------------------------------------------------------------------ the solvers themselves, batched
// histogram of the caller's dictionary indices (+ range check: the first bad voxel is reported like a bad direction)
__global__ void k_idx_hist(const int *__restrict__ idx, int n, int n_dicts, int *__restrict__ lutidx, int *__restrict__ counts, int *__restrict__ status)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int d = idx ? idx[v] : 0;
    if (d < 0 || d >= n_dicts) {
        // first bad voxel and ITS index in one 64-bit atomic (two plain stores after an atomicMin on the voxel alone could pair the
        // smallest voxel with another voxel's index); k_fold_counters unpacks it into ST_ERRVOX / ST_II1
        atomicMin(reinterpret_cast<unsigned long long *>(status + ST_ERRPACK), ((unsigned long long)(unsigned)v << 32) | (unsigned)d);
        status[ST_II2] = n_dicts; status[ST_ERRKIND] = 1;      // (the same values from every lane)
        d = -1;
    } else {
        atomicAdd(&counts[d], 1);
    }
    lutidx[v] = d;
}

static int batched_dev(amx_ctx *ctx, const amx_dict *dict, const int32_t *d_idx, const double *d_y, int64_t n_vox, double lambda1, double lambda2,
                       bool ridge, double *d_x, double *d_rnorm, void *hip_stream, const char *who)
{
    if (!ctx) return AMX_E_BADARG;
    if (!dict || dict->ctx != ctx) return bad(ctx, "amx_*_batched: not a dictionary of this ctx");
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, "amx_*_batched: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if (!d_y || !d_x) return bad(ctx, "amx_*_batched: null buffer");
    if (ridge && (!(lambda1 >= 0.0) || !(lambda2 >= 0.0))) return bad(ctx, "amx_lasso_batched: need lambda1 >= 0 and lambda2 >= 0");
    if (!d_idx && dict->n_dicts != 1) return bad(ctx, "amx_*_batched: dict_idx may only be NULL for a single dictionary");
    (void)who;
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Plan pl; int rc;
    if ((rc = make_plan(ctx, n_vox, dict->n_dicts, pl))) return rc;
    clear_events(ctx);
    rec(ctx, 0, s);
    HIPCHK(ctx, hipMemsetAsync(pl.counts, 0, (size_t)(dict->n_dicts + 1) * sizeof(int), s));
    HIPCHK(ctx, hipMemsetAsync(ctx->misc.p, 0, 64 * sizeof(int), s));
    hipLaunchKernelGGL(k_idx_hist, dim3((unsigned)((n_vox + 255) / 256)), dim3(256), 0, s, (const int *)d_idx, (int)n_vox, dict->n_dicts, pl.lutidx, pl.counts, ctx->status_d);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, pl.counts, dict->n_dicts, kChunk, pl.dir_start, pl.cursor, pl.chunks, pl.n_chunks, 0, (Chunk *)nullptr, 0);
    const int nb = (int)((n_vox + kPrepSpan - 1) / kPrepSpan);
    const int use_lds = dict->n_dicts <= 8192 ? 1 : 0;
    hipLaunchKernelGGL(k_bucket, dim3(nb), dim3(1024), use_lds ? (size_t)2 * dict->n_dicts * sizeof(int) : 0, s, pl.lutidx, (int)n_vox, dict->n_dicts,
                       pl.dir_start, pl.cursor, pl.perm, use_lds, kPrepSpan);
    HIPCHK(ctx, hipGetLastError());
    BatchedArgs a;
    memset(&a, 0, sizeof a);
    a.c.tiles = dict->tiles; a.c.y = d_y; a.c.perm = pl.perm; a.c.chunks = pl.chunks; a.c.n_chunks = pl.n_chunks;
    a.c.lutidx = pl.lutidx; a.c.status = ctx->status_d; a.c.nS = dict->m; a.c.ldA = dict->ldA; a.c.n_atoms = dict->n;
    a.c.tile_stride = dict->tile_stride; a.c.lam1 = lambda1; a.c.lam2 = lambda2;
    a.x = d_x; a.rnorm = d_rnorm;
    // (voxels with a bad dictionary index are skipped: defined zeros)
    HIPCHK(ctx, hipMemsetAsync(d_x, 0, (size_t)n_vox * dict->n * sizeof(double), s));
    rc = amx_launch_batched(ctx, a, pl, s, ridge);
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, (const int *)ctx->misc.p, ctx->status_d);
    rec(ctx, 1, s);
    return rc;
}

extern "C" {

int amx_dict_upload(amx_ctx *ctx, const double *A, int m, int n, int n_dicts, amx_dict **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!A || !out || m <= 0 || n <= 0 || n_dicts <= 0) return bad(ctx, "amx_dict_upload: bad argument");
    // (dictionaries that fit a CU's LDS as fp64 are staged there; larger ones -- up to 512 samples x 256 atoms, what a wavefront's lanes
    //  hold -- are read where they lie: amx_batched.hip)
    if (n > 256 || m > 512) return bad(ctx, "amx_dict_upload: unsupported size (n <= 256 atoms, m <= 512 samples)");
    const int ldA = (n & 1) ? n : n + 1;
    const int tile_stride = (m * ldA + 3) & ~3;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<double> t((size_t)n_dicts * tile_stride + kTileSlack, 0.0);
    for (int d = 0; d < n_dicts; d++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < m; i++) t[(size_t)d * tile_stride + (size_t)i * ldA + j] = A[((size_t)d * n + j) * m + i];     // column-major in, ld = m
    amx_dict *h = new amx_dict();
    h->ctx = ctx; h->m = m; h->n = n; h->ldA = ldA; h->tile_stride = tile_stride; h->n_dicts = n_dicts;
    int rc;
    if ((rc = upload(ctx, &h->tiles, t.data(), t.size()))) { delete h; return rc; }
    *out = h;
    return AMX_OK;
}

void amx_dict_destroy(amx_dict *h)
{
    if (!h) return;
    if (h->ctx) hipSetDevice(h->ctx->device);
    if (h->tiles) hipFree(h->tiles);
    delete h;
}

int amx_nnls_batched_device(amx_ctx *ctx, const amx_dict *dict, const int32_t *d_dict_idx, const double *d_y, int64_t n_vox, double *d_x,
                            double *d_rnorm, void *hip_stream)
{
    return batched_dev(ctx, dict, d_dict_idx, d_y, n_vox, 0.0, 0.0, false, d_x, d_rnorm, hip_stream, "amx_nnls_batched");
}

int amx_lasso_batched_device(amx_ctx *ctx, const amx_dict *dict, const int32_t *d_dict_idx, const double *d_y, int64_t n_vox, double lambda1,
                             double lambda2, double *d_x, void *hip_stream)
{
    return batched_dev(ctx, dict, d_dict_idx, d_y, n_vox, lambda1, lambda2, true, d_x, nullptr, hip_stream, "amx_lasso_batched");
}

static int batched_host(amx_ctx *ctx, const amx_dict *dict, const int32_t *idx, const double *y, int64_t n_vox, double lambda1, double lambda2, bool ridge,
                        double *x, double *rnorm)
{
    if (!ctx) return AMX_E_BADARG;
    const std::string who = ridge ? "amx_lasso_batched" : "amx_nnls_batched";
    if (!dict || dict->ctx != ctx) return bad(ctx, (who + ": not a dictionary of this ctx").c_str());
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || n_vox > INT_MAX / 4) return bad(ctx, (who + ": bad n_vox").c_str());      // (before anything is sized from it)
    if (!y || !x) return bad(ctx, (who + ": null buffer").c_str());
    if (ridge && (!(lambda1 >= 0.0) || !(lambda2 >= 0.0))) return bad(ctx, "amx_lasso_batched: need lambda1 >= 0 and lambda2 >= 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    AMX_H2D(ctx->hy, y, (size_t)n_vox * dict->m * sizeof(double))
    if (idx) { AMX_H2D(ctx->hdirs, idx, (size_t)n_vox * sizeof(int32_t)) }
    if ((rc = ensure(ctx, ctx->hest, (size_t)n_vox * dict->n * sizeof(double)))) return rc;
    if (rnorm && (rc = ensure(ctx, ctx->hrmse, (size_t)n_vox * sizeof(double)))) return rc;
    if ((rc = batched_dev(ctx, dict, idx ? (const int32_t *)ctx->hdirs.p : nullptr, (const double *)ctx->hy.p, n_vox, lambda1, lambda2, ridge,
                          (double *)ctx->hest.p, rnorm ? (double *)ctx->hrmse.p : nullptr, nullptr, "amx_*_batched"))) return rc;
    const int rcs = amx_sync_status(ctx, nullptr);
    if (rcs == AMX_E_DIR_OOB) {
        const int *st = ctx->status_h;
        char b[256];
        snprintf(b, sizeof b, "%s: dict_idx out of range (%d, dictionaries: %d) [voxel %d]", who.c_str(), st[ST_II1], st[ST_II2], st[ST_ERRVOX]);
        ctx->err = b;
    } else if (rcs) return rcs;
    // (a bad dict_idx: every other voxel is solved, the offending ones hold zeros -- the caller gets those results with the error code,
    //  as include/amico_amd.h says)
    HIPCHK(ctx, hipMemcpy(x, ctx->hest.p, (size_t)n_vox * dict->n * sizeof(double), hipMemcpyDeviceToHost));
    if (rnorm) HIPCHK(ctx, hipMemcpy(rnorm, ctx->hrmse.p, (size_t)n_vox * sizeof(double), hipMemcpyDeviceToHost));
    return rcs;
}

int amx_nnls_batched(amx_ctx *ctx, const amx_dict *dict, const int32_t *dict_idx, const double *y, int64_t n_vox, double *x, double *rnorm)
{
    return batched_host(ctx, dict, dict_idx, y, n_vox, 0.0, 0.0, false, x, rnorm);
}

int amx_lasso_batched(amx_ctx *ctx, const amx_dict *dict, const int32_t *dict_idx, const double *y, int64_t n_vox, double lambda1, double lambda2, double *x)
{
    return batched_host(ctx, dict, dict_idx, y, n_vox, lambda1, lambda2, true, x, nullptr);
}

}  // extern "C"

extern "C" {

// ---- public device-pointer entry points: float64 signals, or the float32 the image holds (core.py:136; lossless).  float32 is
// read in place by the NODDI kernels, by every wavefront-per-voxel kernel and by FreeWater's matrix-core projection; the other
// lane kernels get a float64 copy made on the device first.
static int widen_on_device(amx_ctx *ctx, const float *d_y32, size_t nel, hipStream_t s, const double **out)
{
    int rc;
    if ((rc = ensure(ctx, ctx->wy, nel * sizeof(double)))) return rc;
    hipLaunchKernelGGL(k_widen, dim3((unsigned)((nel / 4 + 256) / 256)), dim3(256), 0, s, d_y32, (double *)ctx->wy.p, nel);
    HIPCHK(ctx, hipGetLastError());
    *out = (const double *)ctx->wy.p;
    return AMX_OK;
}

int amx_noddi_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                         double lambda2, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, double *d_mod, void *hip_stream)
{
    return noddi_fit_dev(ctx, lut, d_y, nullptr, d_dirs, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, d_mod, hip_stream);
}

int amx_noddi_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                             double lambda2, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, double *d_mod, void *hip_stream)
{
    return noddi_fit_dev(ctx, lut, nullptr, d_y, d_dirs, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, d_mod, hip_stream);
}

int amx_freewater_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                             double lambda2, int is_mouse, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse,
                             double *d_ycorr, void *hip_stream)
{
    return freewater_fit_dev(ctx, lut, d_y, nullptr, d_dirs, n_vox, lambda1, lambda2, is_mouse, flags, d_estimates, d_rmse, d_nrmse, d_ycorr, hip_stream);
}

int amx_freewater_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                                 double lambda2, int is_mouse, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse,
                                 double *d_ycorr, void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 2 || !d_y || n_vox <= 0 || amx_fw_native_f32(ctx, lut->n_atoms, lut->nS, flags | (ctx->opt_cold_start ? 0x80000000u : 0u), lambda2))
        return freewater_fit_dev(ctx, lut, nullptr, d_y, d_dirs, n_vox, lambda1, lambda2, is_mouse, flags, d_estimates, d_rmse, d_nrmse, d_ycorr, hip_stream);
    const double *wide; int rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = widen_on_device(ctx, d_y, (size_t)n_vox * lut->nS, (hipStream_t)hip_stream, &wide))) return rc;
    return freewater_fit_dev(ctx, lut, wide, nullptr, d_dirs, n_vox, lambda1, lambda2, is_mouse, flags, d_estimates, d_rmse, d_nrmse, d_ycorr, hip_stream);
}

int amx_sandi_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, int64_t n_vox, double lambda1, double lambda2,
                         unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream)
{
    return sandi_fit_dev(ctx, lut, d_y, nullptr, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, hip_stream);
}

int amx_sandi_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, int64_t n_vox, double lambda1, double lambda2,
                             unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 3 || !d_y || n_vox <= 0 || !amx_use_lane_solver(ctx, lut->n_atoms, lambda2))
        return sandi_fit_dev(ctx, lut, nullptr, d_y, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, hip_stream);
    const double *wide; int rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = widen_on_device(ctx, d_y, (size_t)n_vox * lut->nS, (hipStream_t)hip_stream, &wide))) return rc;
    return sandi_fit_dev(ctx, lut, wide, nullptr, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, hip_stream);
}

int amx_czb_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                       double lambda2, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream)
{
    return czb_fit_dev(ctx, lut, d_y, nullptr, d_dirs, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, hip_stream);
}

int amx_czb_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs, int64_t n_vox, double lambda1,
                           double lambda2, unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream)
{
    return czb_fit_dev(ctx, lut, nullptr, d_y, d_dirs, n_vox, lambda1, lambda2, flags, d_estimates, d_rmse, d_nrmse, hip_stream);   // k_czb: load_rows
}

}  // extern "C"

// ------------------------------------------------------------------ host-pointer entry points

// Host buffers in, host buffers out, for all three models and both signal dtypes (float64 = evaluation.y of the
// reference; float32 = the dtype the image has before core.py:451 casts it -- lossless, half the PCIe bytes).
// Large inputs travel in batches: while the GPU fits batches c-1 and c-2 (on the two non-blocking streams of the
// context, each with its own workspace set) the blocking host-to-device copy of batch c is already running, so the PCIe
// time hides behind the solver instead of preceding it.  Three signal buffers; results leave in one copy per output at
// the end.  The progress
// callback (amx_set_progress; models.pyx:28-43, 981 keep a per-thread counter for the same purpose) is called as
// batches complete.
// (largest batch; measured on 1 M NODDI voxels: 131072 -> 38.3 ms, 262144 -> 37.2 ms, 393216 -> 36.3 ms per call)

struct HostOut { void *dst; DevBuf *buf; size_t cols; bool on; };

// enqueue(y_dev, dirs_dev, count, est, rmse, nrmse, extra, stream) -> the model's *_fit_device
template <typename T, typename Enqueue>
static int fit_host(amx_ctx *ctx, const T *y, const double *dirs, int64_t n_vox, int nS, HostOut (&outs)[4], Enqueue enqueue)
{
    int rc;
    const int64_t kHostBatch = ctx->opt_host_batch;
    constexpr bool kF32 = sizeof(T) == 4;
    const bool pipelined = n_vox >= ctx->opt_host_pipeline_from && !ctx->opt_host_one_shot;
    constexpr int kBufs = 3;                      // staging buffers: batch c uploads while c-1 and c-2 are being solved
    const int64_t cap = pipelined ? kBufs * kHostBatch : n_vox;
    // float64 signals that are float32 values (evaluation.y always is: core.py:136, 451) cross the link as float32 (amx_stage.hpp)
    constexpr size_t kNarrowFrom = 2u << 20;      // elements of a batch from which the host threads are worth waking (16 MB: 0.3 ms of link)
    bool narrow = false;
    if (!kF32 && !ctx->opt_host_no_narrow && !ctx->stage_failed && (size_t)n_vox * nS >= kNarrowFrom) {
        if (!ctx->stage) {
            if (ctx->stage_thread.joinable()) { ctx->stage_thread.join(); ctx->stage = ctx->stage_bg; ctx->stage_bg = nullptr; }      // made beside the dictionary upload
            if (!ctx->stage) ctx->stage = make_stage_pool(ctx);
            if (!ctx->stage) ctx->stage_failed = true;
        }
        narrow = ctx->stage != nullptr;
    }
    ctx->host_narrowed = 0;
    // (a model whose kernels read float32 signals in place -- NODDI: ctx->host_native32 -- needs the float64 staging buffer only for a batch that
    //  could not travel as float32: made when that happens.  934 MB less to allocate in a process's first call, no k_widen pass per batch)
    const bool native32 = ctx->host_native32 && !ctx->opt_host_no_native32;
    if (!(native32 && (kF32 || narrow)) && (rc = ensure(ctx, ctx->hy, (size_t)cap * nS * sizeof(double)))) return rc;
    if ((kF32 || narrow) && (rc = ensure(ctx, ctx->hy32, (size_t)cap * nS * sizeof(float)))) return rc;
    if (dirs && (rc = ensure(ctx, ctx->hdirs, (size_t)cap * 3 * sizeof(double)))) return rc;
    for (HostOut &o : outs)
        if (o.on && (rc = ensure(ctx, *o.buf, (size_t)n_vox * o.cols * sizeof(double)))) return rc;
    if (!ctx->up_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->up_ev, hipEventDisableTiming));
    if (!ctx->hs) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->hs, hipStreamNonBlocking));
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->hs2, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->hev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIPCHK(ctx, hipStreamSynchronize(nullptr));                      // earlier default-stream work on these buffers
    const int was_profiling = ctx->profiling;
    if (pipelined) ctx->profiling = 0;
    // (a shard of a larger call -- amx_set_call_voxels: one of several contexts that share a fit -- takes the paths the whole call's size asks for,
    //  as the batches of one call do: every context settles its voxels with the arithmetic the single-context call would use)
    ctx->in_host_fit = true; ctx->host_total_vox = ctx->call_total_vox > n_vox ? ctx->call_total_vox : n_vox;
    struct HostFitScope { amx_ctx *c; ~HostFitScope() { c->in_host_fit = false; } } host_scope{ctx};
    // Batch c runs on stream c & 1 with workspace set c & 1: the kernels of consecutive batches overlap, so the idle tail
    // of every launch (and the one-wavefront re-run kernels) is filled by the other batch instead of adding up six times.
    const bool two_streams = pipelined && !ctx->opt_host_one_stream;
    hipStream_t s = pipelined ? ctx->hs : nullptr;
    int64_t off = 0, done_before[kBufs] = {0, 0, 0};         // voxels complete once the event of that buffer has fired
    // AMX_HOST_TRACE=1 (diagnosis): wall-clock timeline of the call on stderr -- per batch the wait for its buffer, its copy, its enqueue
    static const bool trace = getenv("AMX_HOST_TRACE") != nullptr;
    auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tr0 = trace ? wall() : 0.0;
    int64_t reported = 0;                                    // progress is reported once per batch, in order
    auto report = [&](int64_t v) { if (v > reported) { progress(ctx, v, n_vox); reported = v; } };
    auto batch_cnt = [&](int c, int64_t at) -> int64_t {
        const int64_t rem = n_vox - at, ramp = ctx->opt_host_ramp;
        const int64_t parts = (rem + kHostBatch - 1) / kHostBatch;
        return !pipelined ? n_vox : ((c < 1 && ramp > 0 && rem > 3 * ramp) ? ramp : (rem + parts - 1) / parts);
    };
    // the call's batches are known up front, and with them the chunks the host threads narrow ahead of the copies (amx_stage.hpp)
    std::vector<amx_stage::Chunk> chunks;
    std::vector<int> first_chunk;                  // of batch c
    if (narrow) {
        int64_t at = 0;
        for (int c = 0; at < n_vox; c++) {
            const int64_t cnt = batch_cnt(c, at);
            const size_t nel = (size_t)cnt * nS;
            if (nel < kNarrowFrom) { narrow = false; break; }         // (a batch too small to be worth it: SANDI's six values per voxel)
            first_chunk.push_back((int)chunks.size());
            for (size_t o = 0; o < nel; o += amx_stage::Pool::kChunkEl)
                chunks.push_back({(size_t)at * nS + o, nel - o < amx_stage::Pool::kChunkEl ? nel - o : amx_stage::Pool::kChunkEl});
            at += cnt;
        }
        first_chunk.push_back((int)chunks.size());
    }
    struct NarrowScope { amx_stage::Pool *p; ~NarrowScope() { if (p) p->end(); } } narrow_scope{narrow ? ctx->stage : nullptr};   // (its threads read the caller's buffer)
    if (narrow) ctx->stage->begin(reinterpret_cast<const double *>(y), chunks);
    for (int c = 0; off < n_vox; c++) {
        // the first copy is the only one the solver cannot hide: one short batch (131 072 voxels; shorter ones cost more in the
        // ~2.4 ms floor of the seeded kernel chain than their copy saves), then the rest in equal batches of <= kHostBatch voxels
        const int64_t cnt = batch_cnt(c, off);
        int b = c % kBufs;
        const double tr1 = trace ? wall() : 0.0;
        if (pipelined && c >= kBufs) {
            HIPCHK(ctx, hipEventSynchronize(ctx->hev[b]));           // batch c-3 has released this buffer
            report(done_before[b]);                                  // batches 0 .. c-3 are complete
        }
        if (two_streams) { s = (c & 1) ? ctx->hs2 : ctx->hs; if (c) ctx->swap_work(); }
        // (the uploads below are blocking hipMemcpy calls on the null stream -- from the caller's pageable memory, or from the pinned slots of the
        //  float32 transport -- and the consumers run on the non-blocking streams hs / hs2: hipMemcpy returns when the data has landed for pageable
        //  sources; for the pinned ones that is the runtime's behaviour, not its contract, so the batch's stream WAITS for an event recorded behind
        //  the batch's last copy (round 6, ADVICE r05: two API calls, ~3 us per batch))
        double *yb = ctx->hy.p ? (double *)ctx->hy.p + (size_t)b * kHostBatch * nS : nullptr;
        double *db = dirs ? (double *)ctx->hdirs.p + (size_t)b * kHostBatch * 3 : nullptr;
        const double tr2 = trace ? wall() : 0.0;
        ctx->host_y32 = nullptr;
        if (kF32) {
            float *y32 = (float *)ctx->hy32.p + (size_t)b * kHostBatch * nS;
            const size_t nel = (size_t)cnt * nS;
            HIPCHK(ctx, hipMemcpy(y32, y + (size_t)off * nS, nel * sizeof(float), hipMemcpyHostToDevice));
            if (native32) ctx->host_y32 = y32;
            else hipLaunchKernelGGL(k_widen, dim3((unsigned)((nel / 4 + 256) / 256)), dim3(256), 0, s, y32, yb, nel);
        } else {
            bool sent = false;
            const size_t nel = (size_t)cnt * nS;
            if (narrow) {
                float *y32 = (float *)ctx->hy32.p + (size_t)b * kHostBatch * nS;
                const size_t base_el = (size_t)off * nS;
                sent = true;
                for (int j = first_chunk[c]; j < first_chunk[c + 1]; j++) {
                    // not float32 values (nothing to gain for the rest of the call either) or a failed copy: this batch and the rest go the plain way
                    if (!ctx->stage->ready(j) ||
                        hipMemcpy(y32 + (chunks[j].off - base_el), ctx->stage->slot(j), chunks[j].n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                        (void)hipGetLastError();
                        ctx->stage->end(); narrow = false; sent = false;
                        break;
                    }
                    ctx->stage->consumed(j);
                }
                if (sent) {
                    if (native32) ctx->host_y32 = y32;
                    else hipLaunchKernelGGL(k_widen, dim3((unsigned)((nel / 4 + 256) / 256)), dim3(256), 0, s, y32, yb, nel);
                    ctx->host_narrowed++;
                }
            }
            if (!sent) {
                if (native32) {      // (the float64 staging buffer of a native-float32 call: needed after all -- and an earlier, smaller call's may be too small)
                    if ((rc = ensure(ctx, ctx->hy, (size_t)cap * nS * sizeof(double)))) { ctx->profiling = was_profiling; return rc; }
                    yb = (double *)ctx->hy.p + (size_t)b * kHostBatch * nS;
                }
                HIPCHK(ctx, hipMemcpy(yb, y + (size_t)off * nS, (size_t)cnt * nS * sizeof(double), hipMemcpyHostToDevice));
            }
        }
        if (dirs) HIPCHK(ctx, hipMemcpy(db, dirs + (size_t)off * 3, (size_t)cnt * 3 * sizeof(double), hipMemcpyHostToDevice));
        if (ctx->up_ev) { HIPCHK(ctx, hipEventRecord(ctx->up_ev, nullptr)); HIPCHK(ctx, hipStreamWaitEvent(s, ctx->up_ev, 0)); }
        const double tr3 = trace ? wall() : 0.0;
        // the copy above took a while: has the previous batch finished meanwhile? (a query, never a wait)
        // (batches c-1 and c-2 run on different streams: both must have fired before batch c-1's count is reported)
        if (pipelined && c >= 1 && ctx->progress && hipEventQuery(ctx->hev[(c - 1) % kBufs]) == hipSuccess &&
            (c < 2 || hipEventQuery(ctx->hev[(c - 2) % kBufs]) == hipSuccess))
            report(done_before[(c - 1) % kBufs]);
        ctx->vox_base = off;
        rc = enqueue(yb, db, cnt, (double *)outs[0].buf->p + (size_t)off * outs[0].cols,
                     outs[1].on ? (double *)outs[1].buf->p + off : nullptr, outs[2].on ? (double *)outs[2].buf->p + off : nullptr,
                     outs[3].on ? (double *)outs[3].buf->p + (size_t)off * outs[3].cols : nullptr, s);
        ctx->vox_base = 0; ctx->host_y32 = nullptr;
        if (rc) { ctx->profiling = was_profiling; return rc; }
        off += cnt;
        if (pipelined) { HIPCHK(ctx, hipEventRecord(ctx->hev[b], s)); done_before[b] = off; }
        if (trace) fprintf(stderr, "amx host trace: batch %d  %lld voxels  at %.2f ms: buffer wait %.2f, copy %.2f (%.1f GB/s), enqueue %.2f\n", c, (long long)cnt,
                           tr1 - tr0, tr2 - tr1, tr3 - tr2, ((double)cnt * nS * sizeof(T) + (dirs ? cnt * 24.0 : 0.0)) / (tr3 - tr2) * 1e-6, wall() - tr3);
    }
    ctx->profiling = was_profiling;
    const double tr4 = trace ? wall() : 0.0;
    // The results of the batches that are through go home while the last ones are still being solved (the calling thread has nothing else to
    // do during the solver's tail): the voxels [0, upto) of every output are final once every batch that ends at or before `upto` has fired
    // its event -- batches older than the kBufs last ones were waited for when their staging buffer was taken again.
    int64_t copied = 0;
    auto results_upto = [&](int64_t upto) -> int {
        if (upto <= copied) return AMX_OK;
        for (HostOut &o : outs)
            if (o.on) HIPCHK(ctx, hipMemcpy((char *)o.dst + (size_t)copied * o.cols * sizeof(double), (const char *)o.buf->p + (size_t)copied * o.cols * sizeof(double),
                                            (size_t)(upto - copied) * o.cols * sizeof(double), hipMemcpyDeviceToHost));
        copied = upto;
        return AMX_OK;
    };
    const bool early_results = pipelined && !ctx->opt_host_late_results;
    if (pipelined && (ctx->progress || early_results)) {
        // the batches still in flight, in submission order: one callback as each of them completes (the queries above only
        // catch a batch that finished while the next one was being copied)
        int64_t order[kBufs]; int idx[kBufs];
        for (int b = 0; b < kBufs; b++) { order[b] = done_before[b]; idx[b] = b; }
        for (int i = 0; i < kBufs; i++) for (int j = i + 1; j < kBufs; j++) if (order[j] < order[i]) { std::swap(order[i], order[j]); std::swap(idx[i], idx[j]); }
        for (int i = 0; i < kBufs; i++) {
            if (order[i] <= 0 || order[i] >= n_vox) continue;
            HIPCHK(ctx, hipEventSynchronize(ctx->hev[idx[i]]));
            if (two_streams && i > 0 && order[i - 1] > 0) HIPCHK(ctx, hipEventSynchronize(ctx->hev[idx[i - 1]]));
            report(order[i]);
            if (early_results && (rc = results_upto(order[i]))) return rc;
        }
    }
    if (two_streams) { HIPCHK(ctx, hipStreamSynchronize(s == ctx->hs ? ctx->hs2 : ctx->hs)); }
    rc = amx_sync_status(ctx, s);
    if (rc) return rc;
    const double tr5 = trace ? wall() : 0.0;
    if ((rc = results_upto(n_vox))) return rc;
    if (trace) fprintf(stderr, "amx host trace: last enqueue at %.2f ms, solver tail %.2f, results to the host %.2f, call %.2f ms, %d batches as float32\n", tr4 - tr0, tr5 - tr4, wall() - tr5, wall() - tr0, ctx->host_narrowed);
    progress(ctx, n_vox, n_vox);
    return AMX_OK;
}

template <typename T>
static int noddi_fit_any(amx_ctx *ctx, const amx_lut *lut, const T *y, const double *dirs, int64_t n_vox, double lambda1,
                         double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse, double *out_mod)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 1) return bad(ctx, "amx_noddi_fit: not a NODDI dictionary");
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || !y || !dirs || !out_estimates) return bad(ctx, "amx_noddi_fit: bad argument");
    if (((flags & AMX_F_RMSE) && !out_rmse) || ((flags & AMX_F_NRMSE) && !out_nrmse) || ((flags & AMX_F_MODULATED) && !out_mod))
        return bad(ctx, "amx_noddi_fit: flag set but output buffer is null");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostOut outs[4] = {{out_estimates, &ctx->hest, (size_t)(3 + (lut->is_exvivo ? 1 : 0)), true},
                       {out_rmse, &ctx->hrmse, 1, (flags & AMX_F_RMSE) != 0}, {out_nrmse, &ctx->hnrmse, 1, (flags & AMX_F_NRMSE) != 0},
                       {out_mod, &ctx->hextra, 2, (flags & AMX_F_MODULATED) != 0}};
    // (every NODDI kernel reads float32 signals in place -- the table GEMM, the wavefront-per-voxel kernels' load_rows, the rescue pass --: a batch
    //  that crossed the link as float32 is fitted as float32, no widened copy; bit-identical maps: amx_noddi_fit_device_f32)
    ctx->host_native32 = true;
    struct Native32Scope { amx_ctx *c; ~Native32Scope() { c->host_native32 = false; c->host_y32 = nullptr; } } n32{ctx};
    return fit_host<T>(ctx, y, dirs, n_vox, lut->nS, outs,
                       [&](double *yb, double *db, int64_t cnt, double *e, double *r, double *nr, double *x, hipStream_t s) {
                           if (ctx->host_y32) return amx_noddi_fit_device_f32(ctx, lut, ctx->host_y32, db, cnt, lambda1, lambda2, flags, e, r, nr, x, s);
                           return amx_noddi_fit_device(ctx, lut, yb, db, cnt, lambda1, lambda2, flags, e, r, nr, x, s);
                       });
}

template <typename T>
static int freewater_fit_any(amx_ctx *ctx, const amx_lut *lut, const T *y, const double *dirs, int64_t n_vox, double lambda1,
                             double lambda2, int is_mouse, unsigned flags, double *out_estimates, double *out_rmse,
                             double *out_nrmse, double *out_ycorr)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 2) return bad(ctx, "amx_freewater_fit: not a FreeWater dictionary");
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || !y || !dirs || !out_estimates) return bad(ctx, "amx_freewater_fit: bad argument");
    if (((flags & AMX_F_RMSE) && !out_rmse) || ((flags & AMX_F_NRMSE) && !out_nrmse) || ((flags & AMX_F_CORRECTED) && !out_ycorr))
        return bad(ctx, "amx_freewater_fit: flag set but output buffer is null");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostOut outs[4] = {{out_estimates, &ctx->hest, (size_t)(is_mouse ? 4 : 2), true},
                       {out_rmse, &ctx->hrmse, 1, (flags & AMX_F_RMSE) != 0}, {out_nrmse, &ctx->hnrmse, 1, (flags & AMX_F_NRMSE) != 0},
                       {out_ycorr, &ctx->hextra, (size_t)lut->nS, (flags & AMX_F_CORRECTED) != 0}};
    return fit_host<T>(ctx, y, dirs, n_vox, lut->nS, outs,
                       [&](double *yb, double *db, int64_t cnt, double *e, double *r, double *nr, double *x, hipStream_t s) {
                           return amx_freewater_fit_device(ctx, lut, yb, db, cnt, lambda1, lambda2, is_mouse, flags, e, r, nr, x, s);
                       });
}

template <typename T>
static int sandi_fit_any(amx_ctx *ctx, const amx_lut *lut, const T *y, int64_t n_vox, double lambda1, double lambda2,
                         unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 3) return bad(ctx, "amx_sandi_fit: not a SANDI dictionary");
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || !y || !out_estimates) return bad(ctx, "amx_sandi_fit: bad argument");
    if (((flags & AMX_F_RMSE) && !out_rmse) || ((flags & AMX_F_NRMSE) && !out_nrmse))
        return bad(ctx, "amx_sandi_fit: flag set but output buffer is null");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostOut outs[4] = {{out_estimates, &ctx->hest, 6, true}, {out_rmse, &ctx->hrmse, 1, (flags & AMX_F_RMSE) != 0},
                       {out_nrmse, &ctx->hnrmse, 1, (flags & AMX_F_NRMSE) != 0}, {nullptr, &ctx->hextra, 1, false}};
    return fit_host<T>(ctx, y, (const double *)nullptr, n_vox, lut->nS, outs,
                       [&](double *yb, double *, int64_t cnt, double *e, double *r, double *nr, double *, hipStream_t s) {
                           return amx_sandi_fit_device(ctx, lut, yb, cnt, lambda1, lambda2, flags, e, r, nr, s);
                       });
}

template <typename T>
static int czb_fit_any(amx_ctx *ctx, const amx_lut *lut, const T *y, const double *dirs, int64_t n_vox, double lambda1,
                       double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || lut->model != 4) return bad(ctx, "amx_czb_fit: not a CylinderZeppelinBall dictionary");
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || !y || !dirs || !out_estimates) return bad(ctx, "amx_czb_fit: bad argument");
    if (((flags & AMX_F_RMSE) && !out_rmse) || ((flags & AMX_F_NRMSE) && !out_nrmse))
        return bad(ctx, "amx_czb_fit: flag set but output buffer is null");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostOut outs[4] = {{out_estimates, &ctx->hest, 3, true}, {out_rmse, &ctx->hrmse, 1, (flags & AMX_F_RMSE) != 0},
                       {out_nrmse, &ctx->hnrmse, 1, (flags & AMX_F_NRMSE) != 0}, {nullptr, &ctx->hextra, 1, false}};
    return fit_host<T>(ctx, y, dirs, n_vox, lut->nS, outs,
                       [&](double *yb, double *db, int64_t cnt, double *e, double *r, double *nr, double *, hipStream_t s) {
                           return amx_czb_fit_device(ctx, lut, yb, db, cnt, lambda1, lambda2, flags, e, r, nr, s);
                       });
}

extern "C" {

int amx_noddi_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs, int64_t n_vox, double lambda1,
                  double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse, double *out_mod)
{
    return noddi_fit_any<double>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse, out_mod);
}

int amx_noddi_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs, int64_t n_vox, double lambda1,
                      double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse, double *out_mod)
{
    return noddi_fit_any<float>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse, out_mod);
}

int amx_freewater_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs, int64_t n_vox, double lambda1,
                      double lambda2, int is_mouse, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse,
                      double *out_ycorr)
{
    return freewater_fit_any<double>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, is_mouse, flags, out_estimates, out_rmse, out_nrmse, out_ycorr);
}

int amx_freewater_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs, int64_t n_vox, double lambda1,
                          double lambda2, int is_mouse, unsigned flags, double *out_estimates, double *out_rmse,
                          double *out_nrmse, double *out_ycorr)
{
    return freewater_fit_any<float>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, is_mouse, flags, out_estimates, out_rmse, out_nrmse, out_ycorr);
}

int amx_sandi_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, int64_t n_vox, double lambda1, double lambda2,
                  unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    return sandi_fit_any<double>(ctx, lut, y, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse);
}

int amx_sandi_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, int64_t n_vox, double lambda1, double lambda2,
                      unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    return sandi_fit_any<float>(ctx, lut, y, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse);
}

int amx_czb_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs, int64_t n_vox, double lambda1,
                double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    return czb_fit_any<double>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse);
}

int amx_czb_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs, int64_t n_vox, double lambda1,
                    double lambda2, unsigned flags, double *out_estimates, double *out_rmse, double *out_nrmse)
{
    return czb_fit_any<float>(ctx, lut, y, dirs, n_vox, lambda1, lambda2, flags, out_estimates, out_rmse, out_nrmse);
}

int amx_dir_to_lut_idx(amx_ctx *ctx, const amx_lut *lut, const double *dirs, int64_t n, int32_t *out_idx)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lut || !lut->htable) return bad(ctx, "amx_dir_to_lut_idx: dictionary has no hash table");
    if (n == 0) return AMX_OK;
    if (n < 0 || n > INT_MAX / 4 || !dirs || !out_idx) return bad(ctx, "amx_dir_to_lut_idx: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    AMX_H2D(ctx->hdirs, dirs, (size_t)n * 3 * sizeof(double))
    if ((rc = ensure(ctx, ctx->lutidx, (size_t)n * sizeof(int)))) return rc;
    hipLaunchKernelGGL(k_dir_to_lut, dim3((unsigned)((n + kPrepSpan - 1) / kPrepSpan)), dim3(1024), 0, nullptr,
                       (const double *)ctx->hdirs.p, (int)n, lut->htable, lut->ndirs, (int *)ctx->lutidx.p, (int *)nullptr,
                       ctx->status_d, 0, 0, kPrepSpan, (double *)nullptr, 0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out_idx, ctx->lutidx.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, nullptr));
    return amx_sync_status(ctx, nullptr);
}

}  // extern "C"

#ifdef AMX_PEEK
// diagnosis only (never built into the shipped library): read back the stage intermediates of one voxel
extern "C" int amx_peek(amx_ctx *ctx, int64_t vox, double *xiso2, unsigned long long *supp4)
{
    HIPCHK(ctx, hipMemcpy(xiso2, (double *)ctx->xiso.p + vox * 2, 2 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(supp4, (unsigned long long *)ctx->supp.p + vox * 4, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return AMX_OK;
}
#endif
