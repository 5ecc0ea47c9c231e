// amx_signal.hip -- the steps either side of model.fit (SURVEY section 8 f): principal directions from the
// log-linear tensor fit (core.py:431-436, 456-458) -- streaming, HBM-bound kernels.
#include "amx_host.hpp"
#include "amx_tensor.hpp"

using namespace amx;

namespace amx {

constexpr int kDtiVox = 64;                 // voxels per tile
constexpr int kDtiLanes = 8;                // lanes sharing one voxel's contraction
constexpr int kDtiThreads = kDtiLanes * kDtiVox;
constexpr int kDtiBatch = 4;                // tiles whose tensors are diagonalised together (one per lane, 4 wavefronts)

// y f64[n][nS] -> dirs f64[n][3].  Tiles of `tv` voxels (tv * nS <= kDtiPre * 2 * kDtiThreads doubles):
// (1) the tile is streamed with 16-byte lane loads into registers one tile AHEAD of its use, so that the HBM
// latency is covered by the arithmetic of the previous tile; (2) log(max(y, min_signal)) (TensorModel.fit +
// ols_fit_tensor, dipy/reconst/dti.py) goes to LDS; (3) kDtiLanes lanes per voxel contract the log-signal with the
// first six rows of pinv(design matrix) (LDS, [nS][6]); (4) every kDtiBatch tiles, wavefronts 0..3 (one per SIMD)
// diagonalise the batch's tensors, one per lane -- the Jacobi sweeps are a serial chain, so they are batched to
// full wavefronts and spread over the SIMDs instead of being left to one wavefront per tile.
constexpr int kDtiPre = 8;                  // double2 registers per thread holding the tile in flight

__device__ __forceinline__ void dti_prefetch(double2 (&pre)[kDtiPre], const double *__restrict__ src, int cnt, int tid)
{
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) {
        const int e = 2 * (tid + i * kDtiThreads);
        if (e + 1 < cnt) pre[i] = *reinterpret_cast<const double2 *>(src + e);
        else if (e < cnt) pre[i] = make_double2(src[e], 1.0);
    }
}
// float32 signals (the image's own dtype, core.py:136): two 4-byte loads per pair (a tile starts at any multiple of 4 bytes)
__device__ __forceinline__ void dti_prefetch(double2 (&pre)[kDtiPre], const float *__restrict__ src, int cnt, int tid)
{
    // (loads under their guards, conversions outside them: converted inside its guard, every load was waited for before the next one left)
    float r0[kDtiPre], r1[kDtiPre];
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) {
        const int e = 2 * (tid + i * kDtiThreads);
        r0[i] = 1.0f; r1[i] = 1.0f;
        if (e < cnt) r0[i] = src[e];
        if (e + 1 < cnt) r1[i] = src[e + 1];
    }
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) {
        const int e = 2 * (tid + i * kDtiThreads);
        if (e < cnt) pre[i] = make_double2((double)r0[i], (double)r1[i]);
    }
}

template <typename YT>
__global__ __launch_bounds__(kDtiThreads, 4) void k_dti_dirs(const YT *__restrict__ y, const double *__restrict__ wt,
                                                         int nS, int ldl, int tv, long long n, double min_signal,
                                                         double *__restrict__ dirs)
{
    extern __shared__ double sm[];
    double *wl = sm;                                   // nS * 6
    double *yl = sm + ((nS * 6 + 1) & ~1);             // tv * ldl
    double *dl = yl + tv * ldl;                        // kDtiBatch * tv * 7 (6 tensor entries, odd stride)
    const int tid = threadIdx.x;
    for (int i = tid; i < nS * 6; i += kDtiThreads) wl[i] = wt[i];
    const long long n_tiles = (n + tv - 1) / tv;
    const int step = 2 * kDtiThreads;
    const int step_vol = step % nS, step_adr = (step / nS) * ldl + step_vol;
    const int vox0 = (2 * tid) / nS, vol0 = 2 * tid - vox0 * nS, adr0 = vox0 * ldl + vol0;
    const int eslot = tid / tv, evox = tid - eslot * tv;      // tensor this thread diagonalises in a batch
    double2 pre[kDtiPre];
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) pre[i] = make_double2(1.0, 1.0);
    long long tile = blockIdx.x;
    if (tile < n_tiles) {
        const long long v0 = tile * tv;
        dti_prefetch(pre, y + v0 * nS, (int)((n - v0) < tv ? (n - v0) : tv) * nS, tid);
    }
    long long batch_tile = tile;                       // first tile of the batch being collected
    int slot = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const long long v0 = tile * tv;
        const int nv = (int)((n - v0) < tv ? (n - v0) : tv);
        const int cnt = nv * nS;
#pragma unroll
        for (int i = 0; i < kDtiPre; i++) {            // independent chains: no guards, so that they interleave
            pre[i].x = fast_log(fmax(pre[i].x, min_signal));
            pre[i].y = fast_log(fmax(pre[i].y, min_signal));
        }
        int vol = vol0, adr = adr0;
#pragma unroll
        for (int i = 0; i < kDtiPre; i++) {
            const int e = 2 * (tid + i * kDtiThreads);
            if (e < cnt) yl[adr] = pre[i].x;
            if (e + 1 < cnt) yl[vol + 1 == nS ? adr + 1 + ldl - nS : adr + 1] = pre[i].y;
            vol += step_vol; adr += step_adr;
            if (vol >= nS) { vol -= nS; adr += ldl - nS; }
        }
        __syncthreads();                               // log-signals of this tile are in LDS
        const long long nt = tile + gridDim.x;
        if (nt < n_tiles) {
            const long long w0 = nt * tv;
            dti_prefetch(pre, y + w0 * nS, (int)((n - w0) < tv ? (n - w0) : tv) * nS, tid);
        }
        const int vox = tid / kDtiLanes, q = tid % kDtiLanes;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (vox < nv) {
            const double *yr = yl + vox * ldl;
            for (int v = q; v < nS; v += kDtiLanes) {
                const double ly = yr[v];
                const double *w = wl + v * 6;
#pragma unroll
                for (int k = 0; k < 6; k++) acc[k] = fma(w[k], ly, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
#pragma unroll
            for (int m = 1; m < kDtiLanes; m <<= 1) acc[k] += __shfl_xor(acc[k], m);
        }
        if (q == 0 && vox < nv) {
#pragma unroll
            for (int k = 0; k < 6; k++) dl[(slot * tv + vox) * 7 + k] = acc[k];
        }
        slot++;
        __syncthreads();                               // tensors in LDS; the log-signal rows may be overwritten
        if (slot == kDtiBatch || nt >= n_tiles) {
            if (eslot < slot) {
                const long long e0 = (batch_tile + (long long)eslot * gridDim.x) * tv;
                if (e0 + evox < n) {
                    double d[6], o[3];
#pragma unroll
                    for (int k = 0; k < 6; k++) d[k] = dl[(eslot * tv + evox) * 7 + k];
                    principal_direction(d, o);
                    double *dst = dirs + (e0 + evox) * 3;
                    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
                }
            }
            slot = 0;
            batch_tile = nt;
        }
    }
}

}  // namespace amx

extern "C" {

int amx_dti_create(amx_ctx *ctx, const double *inv_design, int nS, double min_signal, amx_dti **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!inv_design || !out || nS < 7 || nS > 2048) return amx_bad(ctx, "amx_dti_create: need inv_design f64[7][nS], 7 <= nS <= 2048");
    if (!(min_signal > 0.0)) return amx_bad(ctx, "amx_dti_create: min_signal must be positive");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<double> wt((size_t)nS * 6);
    for (int v = 0; v < nS; v++)
        for (int k = 0; k < 6; k++) wt[(size_t)v * 6 + k] = inv_design[(size_t)k * nS + v];
    amx_dti *h = new amx_dti;
    h->ctx = ctx; h->nS = nS; h->min_signal = min_signal;
    hipError_t e = hipMalloc((void **)&h->wt, wt.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->wt, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->wt) (void)hipFree(h->wt);
        delete h;
        ctx->err = std::string("amx_dti_create: ") + hipGetErrorString(e);
        return AMX_E_HIP;
    }
    *out = h;
    return AMX_OK;
}

void amx_dti_destroy(amx_dti *h)
{
    if (!h) return;
    if (h->ctx) (void)hipSetDevice(h->ctx->device);
    if (h->wt) (void)hipFree(h->wt);
    delete h;
}

}  // extern "C"

template <typename YT>
static int dti_directions_dev(amx_ctx *ctx, const amx_dti *h, const YT *d_y, int64_t n_vox, double *d_dirs, void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    if (!h || h->ctx != ctx) return amx_bad(ctx, "amx_dti_directions: not an estimator of this ctx");
    if (n_vox < 0) return amx_bad(ctx, "amx_dti_directions: bad n_vox");
    if (n_vox == 0) return AMX_OK;
    if (!d_y || !d_dirs) return amx_bad(ctx, "amx_dti_directions: null buffer");
    if (sizeof(YT) == 8 && ((uintptr_t)d_y & 15) != 0) return amx_bad(ctx, "amx_dti_directions: y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nS = h->nS;
    int ldl = (nS + 3) & ~3;                  // row stride = 4 * odd doubles: the 16 quads of a wavefront read
    if (((ldl >> 2) & 1) == 0) ldl += 4;      // conflict-free LDS rows
    int tv = (kDtiPre * 2 * kDtiThreads) / nS;   // voxels per tile: what the prefetch registers hold, even, <= kDtiVox
    tv = tv > kDtiVox ? kDtiVox : (tv & ~1);
    const size_t lds = ((size_t)((nS * 6 + 1) & ~1) + (size_t)tv * ldl + (size_t)kDtiBatch * tv * 7) * sizeof(double);
    if (tv < 2 || lds > 160 * 1024) return amx_bad(ctx, "amx_dti_directions: scheme too long for the LDS tile");
    static bool attr_set[64];              // (per instantiation of this function template AND device: the attribute belongs to the pair)
    if (!attr_set[ctx->device & 63]) {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_dti_dirs<YT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device & 63] = true;
    }
    const long long n_tiles = (n_vox + tv - 1) / tv;
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    long long grid = 256LL * (per_cu > 8 ? 8 : per_cu);
    if (grid > n_tiles) grid = n_tiles;
    rec(ctx, 8, s);
    hipLaunchKernelGGL(k_dti_dirs<YT>, dim3((unsigned)grid), dim3(kDtiThreads), lds, s, d_y, h->wt, nS, ldl, tv,
                       (long long)n_vox, h->min_signal, d_dirs);
    HIPCHK(ctx, hipGetLastError());
    rec(ctx, 9, s);
    return AMX_OK;
}

extern "C" {

int amx_dti_directions_device(amx_ctx *ctx, const amx_dti *h, const double *d_y, int64_t n_vox, double *d_dirs, void *hip_stream)
{
    return dti_directions_dev<double>(ctx, h, d_y, n_vox, d_dirs, hip_stream);
}

int amx_dti_directions_device_f32(amx_ctx *ctx, const amx_dti *h, const float *d_y, int64_t n_vox, double *d_dirs, void *hip_stream)
{
    return dti_directions_dev<float>(ctx, h, d_y, n_vox, d_dirs, hip_stream);
}

int amx_dti_directions(amx_ctx *ctx, const amx_dti *h, const double *y, int64_t n_vox, double *out_dirs)
{
    if (!ctx) return AMX_E_BADARG;
    if (!h || h->ctx != ctx) return amx_bad(ctx, "amx_dti_directions: not an estimator of this ctx");
    if (n_vox == 0) return AMX_OK;
    if (n_vox < 0 || !y || !out_dirs) return amx_bad(ctx, "amx_dti_directions: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t yb = (size_t)n_vox * h->nS * sizeof(double), db = (size_t)n_vox * 3 * sizeof(double);
    if ((rc = amx_ensure(ctx, ctx->hy, yb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hdirs, db))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->hy.p, y, yb, hipMemcpyHostToDevice, nullptr));
    if ((rc = amx_dti_directions_device(ctx, h, (const double *)ctx->hy.p, n_vox, (double *)ctx->hdirs.p, nullptr))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_dirs, ctx->hdirs.p, db, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

}  // extern "C"
