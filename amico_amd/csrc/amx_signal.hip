// amx_signal.hip -- the steps either side of model.fit (SURVEY section 8 f): principal directions from the
// log-linear tensor fit (core.py:431-436, 456-458) -- streaming, HBM-bound kernels.
#include "amx_host.hpp"

using namespace amx;

namespace amx {

constexpr int kDtiVox = 64;                 // voxels per tile
constexpr int kDtiLanes = 8;                // lanes sharing one voxel's contraction
constexpr int kDtiThreads = kDtiLanes * kDtiVox;
constexpr int kDtiBatch = 4;                // tiles whose tensors are diagonalised together (one per lane, 4 wavefronts)

// 1/x and 1/sqrt(x) to double precision from the hardware estimates + Newton steps (x normal, > 0)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    y = y * (1.5 - 0.5 * a * y * y);
    y = y * (1.5 - 0.5 * a * y * y);
    return y;
}

// log(x) for x > 0 in the normal range, <= 2 ulp: x = 2^e m with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716 -> 11 odd terms.  (ocml's log costs ~6x more VALU
// instructions, and this kernel is bound by them: 99 logarithms per voxel.)
__device__ __forceinline__ double fast_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? 2.0 * m : m;
    e = lo ? e - 1 : e;
    const double num = m - 1.0, den = m + 1.0;
    const double r = fast_rcp(den);
    double s = num * r;
    s = fma(fma(-s, den, num), r, s);
    const double z = s * s;
    double p = 2.0 / 23.0;
    p = fma(p, z, 2.0 / 21.0); p = fma(p, z, 2.0 / 19.0); p = fma(p, z, 2.0 / 17.0); p = fma(p, z, 2.0 / 15.0);
    p = fma(p, z, 2.0 / 13.0); p = fma(p, z, 2.0 / 11.0); p = fma(p, z, 2.0 / 9.0); p = fma(p, z, 2.0 / 7.0);
    p = fma(p, z, 2.0 / 5.0); p = fma(p, z, 2.0 / 3.0);
    const double ed = (double)e;
    return fma(ed, 0x1.62e42fee00000p-1, 2.0 * s + fma(s * z, p, ed * 0x1.a39ef35793c76p-33));
}

// One Jacobi rotation annihilating a[P][Q] of the symmetric 3x3 matrix a; v accumulates the eigenvectors (columns).
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&a)[3][3], double (&v)[3][3])
{
    constexpr int R = 3 - P - Q;
    const double apq = a[P][Q];
    // t = tan(rotation angle), the smaller root of t^2 + 2 t theta - 1 = 0 with theta = (aqq - app) / (2 apq)
    const double w = a[Q][Q] - a[P][P];
    const double h2 = fma(w, w, 4.0 * apq * apq);
    const bool rot = h2 > 1e-290 && apq != 0.0;
    const double h = rot ? h2 * fast_rsqrt(h2) : 1.0;
    double t = rot ? 2.0 * apq * fast_rcp(fabs(w) + h) : 0.0;
    t = w < 0.0 ? -t : t;
    const double c = fast_rsqrt(fma(t, t, 1.0)), s = t * c;
    a[P][P] -= t * apq;
    a[Q][Q] += t * apq;
    a[P][Q] = a[Q][P] = 0.0;
    const double arp = a[R][P], arq = a[R][Q];
    a[R][P] = a[P][R] = c * arp - s * arq;
    a[R][Q] = a[Q][R] = s * arp + c * arq;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double vip = v[i][P], viq = v[i][Q];
        v[i][P] = c * vip - s * viq;
        v[i][Q] = s * vip + c * viq;
    }
}

// Eigenvector of the largest eigenvalue of the symmetric tensor (lower-triangular order Dxx Dxy Dyy Dxz Dyz Dzz):
// what `decompose_tensor` (dipy/reconst/dti.py) returns as evecs[:, 0] after sorting eigh's output in descending
// order -- up to the sign, which LAPACK leaves unspecified and dir_to_lut_idx folds away (lut.pyx:335-338).
__device__ inline void principal_direction(const double d[6], double out[3])
{
    // scale to max |entry| = 1 (the eigenvectors do not change): keeps the squares of the rotations in range
    double mx = fmax(fmax(fabs(d[0]), fabs(d[1])), fmax(fabs(d[2]), fabs(d[3])));
    mx = fmax(mx, fmax(fabs(d[4]), fabs(d[5])));
    const double sc = mx > 1e-290 ? fast_rcp(mx) : 0.0;
    double a[3][3] = {{d[0] * sc, d[1] * sc, d[3] * sc}, {d[1] * sc, d[2] * sc, d[4] * sc}, {d[3] * sc, d[4] * sc, d[5] * sc}};
    double v[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
#pragma unroll 1
    for (int sweep = 0; sweep < 6; sweep++) {     // cyclic Jacobi converges quadratically: 6 sweeps >> fp64 for 3x3
        jacobi_rotate<0, 1>(a, v);
        jacobi_rotate<0, 2>(a, v);
        jacobi_rotate<1, 2>(a, v);
    }
    const bool c1 = a[1][1] > a[0][0];
    double best = c1 ? a[1][1] : a[0][0];
    double x = c1 ? v[0][1] : v[0][0], y = c1 ? v[1][1] : v[1][0], z = c1 ? v[2][1] : v[2][0];
    const bool c2 = a[2][2] > best;
    x = c2 ? v[0][2] : x; y = c2 ? v[1][2] : y; z = c2 ? v[2][2] : z;
    const double inv = fast_rsqrt(x * x + y * y + z * z);
    out[0] = x * inv; out[1] = y * inv; out[2] = z * inv;
}

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
    float2 raw[kDtiPre];
    bool got[kDtiPre];
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) {
        // (the loads under their guards, the conversions outside them: converted inside, every load was waited for before the next one left)
        const int e = 2 * (tid + i * kDtiThreads);
        float2 t = make_float2(1.0f, 1.0f);
        bool any = false;
        if (e + 1 < cnt) { t = *reinterpret_cast<const float2 *>(src + e); any = true; }
        else if (e < cnt) { t.x = src[e]; any = true; }
        raw[i] = t; got[i] = any;
    }
#pragma unroll
    for (int i = 0; i < kDtiPre; i++) if (got[i]) pre[i] = make_double2((double)raw[i].x, (double)raw[i].y);
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
    static bool attr_set = false;          // (one flag per instantiation of this function template)
    if (!attr_set) {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_dti_dirs<YT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
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
