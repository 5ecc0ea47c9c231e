// amx_volume.hip -- the bandwidth kernels either side of the fit (SURVEY section 8 f, rows 2 and 3):
// 4-D float32 image -> masked, b0-normalised, (b0-merged | shell-averaged), clipped float64 signals, and the
// scatter of the per-voxel results back into float32 volumes.  Reference: core.py:209-223 (normalisation),
// 225-268 (b0 merge, directional average), 451-452 (mask gather + clip), 472-498 (scatter).
#include "amx_host.hpp"
#include "amx_tensor.hpp"

using namespace amx;

namespace amx {

// the lanes of a wavefront exchange data through LDS: order the accesses, no workgroup barrier needed
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

constexpr int kDirectLd = 65;              // words per volume row of the direct-load tile (k_prep_gather)
constexpr int kPrepWaves = 2;               // wavefronts per workgroup, one 64-voxel tile each

struct PrepArgs {
    const float *img; const int *rank; double *y; float *y32; float *mean_b0;     // y32: float32 output instead of y (lossless: the values ARE float32)
    long long d0, d1, d2, s0, s1, s2, sv;
    long long n_tiles, tiles_per_row;
    int nS, n_out, n_b0, n_gidx, ldt, inplace, layout, normalize, direct;
    const int *gptr, *gidx, *b0idx;
    float thr;
    const double *wt; double min_signal; double *dirs;   // DIRS: tensor fit along the way -- pinv(design) f64[n_out][6], clamp, out f64[n_vox][3]
    const int *live;              // tiles with masked voxels (amx_prep::live64), n_live of them
    int *counter;                 // next entry of `live` to hand out (zeroed before the launch)
    long long n_live;
};

// Next tile of the plan's live list for this wavefront (one device-wide atomic per tile, broadcast from lane 0).  A tile is ~25 us
// of a wavefront's time and a wavefront sees ~7 of them, so the hand-out has to be this fine: batches of four tiles per ticket
// measured 0.253 -> 0.299 ms, as slow as the static walk over all tiles.
__device__ __forceinline__ long long next_live_tile(const PrepArgs &a, int lane)
{
    int k = 0;
    if (lane == 0) k = atomicAdd(a.counter, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    return k < a.n_live ? (long long)a.live[k] : -1ll;
}

// One wavefront per tile of 64 voxels that are consecutive along the image's fastest spatial axis.
//  (1) the tile's nS values per voxel go to LDS T[voxel][volume] (odd row stride) with coalesced loads in either
//      memory layout: planar (x fastest, one 256-byte run per volume) or interleaved (volume fastest, one row per voxel);
//  (2) lane = voxel: mean of the b0 volumes (float32, summed in index order like numpy reduces the fancy-indexed
//      array of core.py:213), norm factor (core.py:216-220); when volumes are grouped (IDENTITY = false) the row is
//      scaled in place (core.py:221-222) and every output volume = float32 mean of its group in index order
//      (core.py:225-227 / 236-252); with identity groups the single multiplication is applied on the way out;
//  (3) rows are written to y[rank][:] as float64 with negative values clipped (core.py:451-452), coalesced per row.
// The plan's index lists live in LDS (P): scalar loads from global memory would serialise phase 2.
// DIRS (round 5): the principal direction of the voxel's diffusion tensor (k_dti_dirs' arithmetic: log-linear fit, Jacobi) is taken
// while the voxel's nS values sit in the tile -- lane = voxel walks them once more --, so the device pipeline reads the image once
// and y is not read back for the tensor fit.
template <bool IDENTITY, bool DIRECT, bool DIRS = false>
__global__ __launch_bounds__(64 * kPrepWaves) void k_prep_gather(PrepArgs a)
{
    extern __shared__ float smf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // IDENTITY groups on a planar image: the plane runs go global -> LDS directly (no registers in between, so ALL volumes of the
    // tile are in flight at once instead of 32 at a time) as T[volume][voxel] with a row stride of kDirectLd = 65 words; the other
    // paths keep T[voxel][volume] (odd stride ldt).  Both are conflict-free for the lane = voxel and the lane = volume accesses.
    constexpr bool direct = IDENTITY && DIRECT;
    const int per_wave = IDENTITY ? (direct ? a.nS * kDirectLd : 64 * a.ldt) : 64 * a.ldt * (a.inplace ? 1 : 2);
    int *P = reinterpret_cast<int *>(smf + (size_t)kPrepWaves * per_wave);
    int *Pb0 = P, *Pgp = P + a.n_b0, *Pgi = Pgp + a.n_out + 1;
    for (int i = threadIdx.x; i < a.n_b0; i += blockDim.x) Pb0[i] = a.b0idx[i];
    if (!IDENTITY) {
        for (int i = threadIdx.x; i <= a.n_out; i += blockDim.x) Pgp[i] = a.gptr[i];
        for (int i = threadIdx.x; i < a.n_gidx; i += blockDim.x) Pgi[i] = a.gidx[i];
    }
    __syncthreads();
    float *T = smf + (size_t)wave * per_wave;
    float *O = a.inplace ? T : T + 64 * a.ldt;
    for (long long t = next_live_tile(a, lane); t >= 0; t = next_live_tile(a, lane)) {
        const long long row = t / a.tiles_per_row;
        const long long x0 = (t - row * a.tiles_per_row) * 64;
        const long long i2 = row / a.d1, i1 = row - i2 * a.d1;
        const long long x = x0 + lane;
        const int r = x < a.d0 ? a.rank[(i2 * a.d1 + i1) * a.d0 + x] : -1;
        const unsigned long long live = __ballot(r >= 0);
        if (live == 0ull) continue;
        const long long base = x0 * a.s0 + i1 * a.s1 + i2 * a.s2;
        float *row_l = T + lane * a.ldt;
        if (a.layout == 2) {
            // interleaved: lanes run over the volumes of one voxel, every live row goes global -> LDS directly (two loads per 99-volume
            // row, all rows of the tile in flight at once; through registers it was sixteen rows at a time)
            for (int k = 0; k < 64; k++) {
                if (!((live >> k) & 1ull)) continue;
                const float *rowp = a.img + base + k * a.s0;
                for (int v0 = 0; v0 < a.nS; v0 += 64) {
                    if (v0 + lane < a.nS)
                        __builtin_amdgcn_global_load_lds(rowp + v0 + lane, (__attribute__((address_space(3))) void *)(T + k * a.ldt + v0), 4, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (direct) {
            if (r >= 0) {
                const float *src = a.img + base + lane * a.s0;
                int v = 0;
                for (; v + 8 <= a.nS; v += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        __builtin_amdgcn_global_load_lds(src + (long long)(v + u) * a.sv, (__attribute__((address_space(3))) void *)(T + (v + u) * kDirectLd), 4, 0, 0);
                }
                for (; v < a.nS; v++)
                    __builtin_amdgcn_global_load_lds(src + (long long)v * a.sv, (__attribute__((address_space(3))) void *)(T + v * kDirectLd), 4, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (r >= 0) {
            // planar (or generic strides): lanes run over the voxels of one volume; 32 volumes in flight
            const float *src = a.img + base + lane * a.s0;
            int v = 0;
            for (; v + 32 <= a.nS; v += 32) {
                float tv[32];
#pragma unroll
                for (int u = 0; u < 32; u++) tv[u] = src[(long long)(v + u) * a.sv];
#pragma unroll
                for (int u = 0; u < 32; u++) row_l[v + u] = tv[u];
            }
            for (; v + 4 <= a.nS; v += 4) {
                float tv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) tv[u] = src[(long long)(v + u) * a.sv];
#pragma unroll
                for (int u = 0; u < 4; u++) row_l[v + u] = tv[u];
            }
            for (; v < a.nS; v++) row_l[v] = src[(long long)v * a.sv];
        }
        WAVE_SYNC();
        float f = 1.0f;
        if (r >= 0) {
            if (a.normalize) {
                float m = 0.0f;
                if (direct) { for (int i = 0; i < a.n_b0; i++) m = m + T[Pb0[i] * kDirectLd + lane]; }
                else { for (int i = 0; i < a.n_b0; i++) m = m + row_l[Pb0[i]]; }
                m = m / (float)a.n_b0;
                if (a.mean_b0) a.mean_b0[r] = m;
                f = (m <= a.thr) ? 0.0f : 1.0f / m;                  // norm_factor[idx] = 0, else 1 / mean_b0
            }
            if (!IDENTITY) {
                if (a.normalize)
                    for (int v = 0; v < a.nS; v++) row_l[v] = row_l[v] * f;
                float *out_l = O + lane * a.ldt;
                for (int j = 0; j < a.n_out; j++) {
                    const int g0 = Pgp[j], g1 = Pgp[j + 1];
                    float acc = row_l[Pgi[g0]];
                    if (g1 - g0 > 1) {
                        for (int g = g0 + 1; g < g1; g++) acc = acc + row_l[Pgi[g]];
                        acc = acc / (float)(g1 - g0);
                    }
                    out_l[j] = acc;
                }
            }
        }
        if (!IDENTITY) WAVE_SYNC();
        const bool scale = IDENTITY && a.normalize;
        if (DIRS && r >= 0) {
            // y[r][j] as the write-out below makes it (scaled, clipped, float32), log(max(y, min_signal)) contracted with the six rows
            // of the design's pseudo-inverse (wave-uniform addresses: scalar loads), then the 3 x 3 eigen-problem -- all in this lane
            using CD = const __attribute__((address_space(4))) double;
            CD *W = (CD *)a.wt;
            double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j = 0; j < a.n_out; j++) {
                float val = direct ? T[j * kDirectLd + lane] : O[lane * a.ldt + j];
                if (scale) val = val * f;
                val = val < 0.0f ? 0.0f : val;
                const double ly = fast_log(fmax((double)val, a.min_signal));
#pragma unroll
                for (int k = 0; k < 6; k++) acc[k] = fma(W[j * 6 + k], ly, acc[k]);
            }
            double o[3];
            principal_direction(acc, o);
            double *dd = a.dirs + (long long)r * 3;
            dd[0] = o[0]; dd[1] = o[1]; dd[2] = o[2];
        }
        // (four rows per step -- their LDS reads in flight together -- measured slower: 0.254 -> 0.268 ms)
        for (int k = 0; k < 64; k++) {
            if (!((live >> k) & 1ull)) continue;
            const int rk = __builtin_amdgcn_readlane(r, k);
            const float fk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), k));
            double *dst = a.y + (long long)rk * a.n_out;
            float *dst32 = a.y32 + (long long)rk * a.n_out;
            for (int j = lane; j < a.n_out; j += 64) {
                float val = direct ? T[j * kDirectLd + k] : O[k * a.ldt + j];
                if (scale) val = val * fk;
                val = val < 0.0f ? 0.0f : val;
                if (a.y32) dst32[j] = val;
                else dst[j] = (double)val;
            }
        }
        WAVE_SYNC();
    }
}

// Streaming variant for GROUPED outputs on a planar image (x fastest) -- the shell average of SANDI, the b0 merge:
// lane = voxel, no transposition tile.  Each volume plane is read with coalesced 256-byte wavefront loads, the group
// sums are accumulated in index order in a register (float32, exactly as k_prep_gather does), and the few output
// values of a voxel are stored directly.  Without the 25-80 KB LDS tile per wavefront the occupancy is limited by
// registers only, which is what a 1.2 KB-in / 48 B-out reduction needs to approach the HBM rate.  Used when no group
// reads a volume index that an earlier output occupies (plan flag `hazard` clear), i.e. when order does not matter.
// img[..., i] *= norm_factor is a float32 operation of its own in the reference: the product must be rounded before
// it enters a sum, so it is hidden from the compiler's fma contraction behind an empty asm
__device__ __forceinline__ float scaled(float v, float f)
{
    float p = v * f;
    asm volatile("" : "+v"(p));
    return p;
}

__global__ __launch_bounds__(256) void k_prep_stream(PrepArgs a)
{
    extern __shared__ int P[];
    int *Pb0 = P, *Pgp = P + a.n_b0, *Pgi = Pgp + a.n_out + 1;
    for (int i = threadIdx.x; i < a.n_b0; i += blockDim.x) Pb0[i] = a.b0idx[i];
    for (int i = threadIdx.x; i <= a.n_out; i += blockDim.x) Pgp[i] = a.gptr[i];
    for (int i = threadIdx.x; i < a.n_gidx; i += blockDim.x) Pgi[i] = a.gidx[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const long long wave_id = (long long)blockIdx.x * nwv + wave, n_waves = (long long)gridDim.x * nwv;
    for (long long t = wave_id; t < a.n_tiles; t += n_waves) {
        const long long row = t / a.tiles_per_row;
        const long long x0 = (t - row * a.tiles_per_row) * 64;
        const long long i2 = row / a.d1, i1 = row - i2 * a.d1;
        const long long x = x0 + lane;
        const int r = x < a.d0 ? a.rank[(i2 * a.d1 + i1) * a.d0 + x] : -1;
        if (__ballot(r >= 0) == 0ull) continue;
        if (r < 0) continue;
        const float *src = a.img + x * a.s0 + i1 * a.s1 + i2 * a.s2;
        float f = 1.0f;
        if (a.normalize) {
            float m = 0.0f;
            for (int i = 0; i < a.n_b0; i++) m = m + src[(long long)Pb0[i] * a.sv];
            m = m / (float)a.n_b0;
            if (a.mean_b0) a.mean_b0[r] = m;
            f = (m <= a.thr) ? 0.0f : 1.0f / m;
        }
        double *dst = a.y + (long long)r * a.n_out;
        float *dst32 = a.y32 + (long long)r * a.n_out;
        for (int j = 0; j < a.n_out; j++) {
            const int g0 = Pgp[j], g1 = Pgp[j + 1];
            float acc = 0.0f;
            int g = g0;
            for (; g + 8 <= g1; g += 8) {               // eight loads in flight, summed in index order
                float t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = src[(long long)Pgi[g + u] * a.sv];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float v = scaled(t8[u], f);
                    acc = (g + u == g0) ? v : acc + v;
                }
            }
            for (; g < g1; g++) { const float v = scaled(src[(long long)Pgi[g] * a.sv], f); acc = (g == g0) ? v : acc + v; }
            if (g1 - g0 > 1) acc = acc / (float)(g1 - g0);
            acc = acc < 0.0f ? 0.0f : acc;
            if (a.y32) dst32[j] = acc;
            else dst[j] = (double)acc;
        }
    }
}

// The same with FOUR consecutive voxels per lane and 16-byte loads (round 4): a wavefront load covers 1 KB of a volume plane instead
// of 256 bytes -- a quarter of the load instructions and address computations for the same bytes, which is what a 1.2 KB-in /
// 48 B-out reduction is made of.  Needs the three spatial axes contiguous in memory (s0 = 1, s1 = d0, s2 = d0 d1: the layout nibabel
// hands out), the volume stride a multiple of 4 and a 16-byte aligned image; the tile is a run of 256 voxels of the LINEAR spatial
// index.  Same float32 operations per voxel in the same order: bit-exact with k_prep_stream and the numpy statements.
__global__ __launch_bounds__(256) void k_prep_stream4(PrepArgs a)
{
    extern __shared__ int P[];
    int *Pb0 = P, *Pgp = P + a.n_b0, *Pgi = Pgp + a.n_out + 1;
    for (int i = threadIdx.x; i < a.n_b0; i += blockDim.x) Pb0[i] = a.b0idx[i];
    for (int i = threadIdx.x; i <= a.n_out; i += blockDim.x) Pgp[i] = a.gptr[i];
    for (int i = threadIdx.x; i < a.n_gidx; i += blockDim.x) Pgi[i] = a.gidx[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const long long wave_id = (long long)blockIdx.x * nwv + wave, n_waves = (long long)gridDim.x * nwv;
    const long long n_lin = a.d0 * a.d1 * a.d2, n_tiles4 = (n_lin + 255) / 256;
    for (long long t = wave_id; t < n_tiles4; t += n_waves) {
        const long long m = t * 256 + 4 * lane;
        int r[4] = {-1, -1, -1, -1};
        if (m + 3 < n_lin) { const int4 rr = *reinterpret_cast<const int4 *>(a.rank + m); r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; r[3] = rr.w; }
        else { for (int u = 0; u < 4; u++) if (m + u < n_lin) r[u] = a.rank[m + u]; }
        const bool any = r[0] >= 0 || r[1] >= 0 || r[2] >= 0 || r[3] >= 0;
        if (__ballot(any) == 0ull) continue;
        if (!any || m + 3 >= n_lin + 3) continue;
        const bool full = m + 3 < n_lin;                 // (the very last lane of the image may hold fewer than four voxels)
        const float *src = a.img + m;
        auto load4 = [&](long long off, float (&v)[4]) {
            if (full) { const float4 q = *reinterpret_cast<const float4 *>(src + off); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
            else { for (int u = 0; u < 4; u++) v[u] = (m + u < n_lin) ? src[off + u] : 0.0f; }
        };
        float f[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (a.normalize) {
            float mm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // (batches of eight loads in flight, the tail of a list as a clamped batch: a loop of single loads is a chain of HBM round
            //  trips -- 6 b0 volumes + 4 left-over volumes in each of 5 shells were 26 of a tile's 64 trips)
            for (int i0 = 0; i0 < a.n_b0; i0 += 8) {
                float t4[8][4];
#pragma unroll
                for (int w = 0; w < 8; w++) load4((long long)Pb0[i0 + w < a.n_b0 ? i0 + w : a.n_b0 - 1] * a.sv, t4[w]);
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    if (i0 + w < a.n_b0) {
#pragma unroll
                        for (int u = 0; u < 4; u++) mm[u] = mm[u] + t4[w][u];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                mm[u] = mm[u] / (float)a.n_b0;
                if (a.mean_b0 && r[u] >= 0) a.mean_b0[r[u]] = mm[u];
                f[u] = (mm[u] <= a.thr) ? 0.0f : 1.0f / mm[u];
            }
        }
        // (round 5) few outputs per voxel -- the shell means of a direction-averaged acquisition: 6 -- wait in registers and leave as ONE
        // run of stores per voxel: written one group at a time, 8 bytes per voxel every ~50 volumes, they cost 4.8 x their bytes in HBM
        // writes (WRITE_SIZE 264 MB against 55 MB for 1.16 M voxels: every store a partial sector of its own)
        constexpr int kHold = 8;
        const bool hold = a.n_out <= kHold;
        float held[kHold][4];
        for (int j = 0; j < a.n_out; j++) {
            const int g0 = Pgp[j], g1 = Pgp[j + 1];
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            int g = g0;
            for (; g + 8 <= g1; g += 8) {               // eight 16-byte loads in flight (8 KB per wavefront), summed in index order
                float t4[8][4];
#pragma unroll
                for (int w = 0; w < 8; w++) load4((long long)Pgi[g + w] * a.sv, t4[w]);
#pragma unroll
                for (int w = 0; w < 8; w++) {
#pragma unroll
                    for (int u = 0; u < 4; u++) { const float v = scaled(t4[w][u], f[u]); acc[u] = (g + w == g0) ? v : acc[u] + v; }
                }
            }
            if (g < g1) {                               // the list's tail: one clamped batch, same order of the sums
                float t4[8][4];
#pragma unroll
                for (int w = 0; w < 8; w++) load4((long long)Pgi[g + w < g1 ? g + w : g1 - 1] * a.sv, t4[w]);
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    if (g + w < g1) {
#pragma unroll
                        for (int u = 0; u < 4; u++) { const float v = scaled(t4[w][u], f[u]); acc[u] = (g + w == g0) ? v : acc[u] + v; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float v = acc[u];
                if (g1 - g0 > 1) v = v / (float)(g1 - g0);
                v = v < 0.0f ? 0.0f : v;
                if (hold) {
#pragma unroll
                    for (int jj = 0; jj < kHold; jj++) held[jj][u] = (jj == j) ? v : held[jj][u];
                } else if (r[u] >= 0) {
                    if (a.y32) a.y32[(long long)r[u] * a.n_out + j] = v;
                    else a.y[(long long)r[u] * a.n_out + j] = (double)v;
                }
            }
        }
        if (hold) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (r[u] >= 0) {
#pragma unroll
                    for (int jj = 0; jj < kHold; jj++) {
                        if (jj < a.n_out) {
                            if (a.y32) a.y32[(long long)r[u] * a.n_out + jj] = held[jj][u];
                            else a.y[(long long)r[u] * a.n_out + jj] = (double)held[jj][u];
                        }
                    }
                }
            }
        }
    }
}

// float32 mean of the b0 volumes of EVERY voxel (self.mean_b0s, core.py:213), written in C order [X][Y][Z]
__global__ void k_mean_b0(const float *img, long long d0, long long d1, long long d2, long long s0, long long s1,
                          long long s2, long long sv, long long c0, long long c1, long long c2, const int *b0idx,
                          int n_b0, float *out)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= d0 * d1 * d2) return;
    const long long i0 = m % d0, i1 = (m / d0) % d1, i2 = m / (d0 * d1);
    const float *src = img + i0 * s0 + i1 * s1 + i2 * s2;
    float acc = 0.0f;
    for (int i = 0; i < n_b0; i++) acc = acc + src[(long long)b0idx[i] * sv];
    out[i0 * c0 + i1 * c1 + i2 * c2] = acc / (float)n_b0;
}

// RESULTS[...][mask == 1, :] = values (core.py:472-498): f64[n][k] -> float32 volume [X][Y][Z][k] (C order, zeroed by the host)
__global__ void k_scatter(const double *src, const long long *cidx, long long n, int k, float *vol)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const long long v = i / k;
    vol[cidx[v] * k + (i - v * k)] = (float)src[i];
}

}  // namespace amx

namespace {

template <typename T>
int dev_copy(amx_ctx *ctx, T **dst, const T *src, size_t n)
{
    HIPCHK(ctx, hipMalloc((void **)dst, (n ? n : 1) * sizeof(T)));
    if (n) HIPCHK(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

}  // namespace

extern "C" {

void amx_prep_destroy(amx_prep *p)
{
    if (!p) return;
    if (p->ctx) (void)hipSetDevice(p->ctx->device);
    void *ps[] = {p->rank, p->cidx, p->gptr, p->gidx, p->b0idx, p->live64, p->tile_counter};
    for (void *q : ps) if (q) (void)hipFree(q);
    delete p;
}

int amx_prep_create(amx_ctx *ctx, const int64_t dims[3], const int64_t strides[4], int nS, const int32_t *rank,
                    int64_t n_vox, const int32_t *group_ptr, const int32_t *group_idx, int n_out,
                    const int32_t *b0_idx, int n_b0, int overwrite_in_order, amx_prep **out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!dims || !strides || !rank || !group_ptr || !group_idx || !out) return amx_bad(ctx, "amx_prep_create: null argument");
    if (nS < 1 || n_out < 1 || n_out > nS || n_b0 < 0 || (n_b0 > 0 && !b0_idx) || n_vox < 0)
        return amx_bad(ctx, "amx_prep_create: bad sizes (need 1 <= n_out <= nS)");
    for (int k = 0; k < 3; k++) if (dims[k] < 1 || strides[k] < 1) return amx_bad(ctx, "amx_prep_create: dims and strides must be positive");
    if (strides[3] < 1) return amx_bad(ctx, "amx_prep_create: strides must be positive");
    const long long total = (long long)dims[0] * dims[1] * dims[2];
    if (total > INT_MAX || n_vox > total) return amx_bad(ctx, "amx_prep_create: volume too large");
    if (group_ptr[0] != 0) return amx_bad(ctx, "amx_prep_create: group_ptr[0] must be 0");
    int inplace = 1, hazard = 0;       // hazard: some group reads a volume index that an earlier output occupies
    for (int j = 0; j < n_out; j++) {
        if (group_ptr[j + 1] <= group_ptr[j]) return amx_bad(ctx, "amx_prep_create: empty output group");
        for (int g = group_ptr[j]; g < group_ptr[j + 1]; g++) {
            if (group_idx[g] < 0 || group_idx[g] >= nS) return amx_bad(ctx, "amx_prep_create: group index out of range");
            // output j would overwrite an input a later group still needs -- unless that is what the caller asks
            // for (the shell average of core.py:231-245 writes into a VIEW of the image it keeps reading from)
            if (group_idx[g] < j) hazard = 1;
            if (group_idx[g] < j && !overwrite_in_order) inplace = 0;
        }
    }
    for (int i = 0; i < n_b0; i++) if (b0_idx[i] < 0 || b0_idx[i] >= nS) return amx_bad(ctx, "amx_prep_create: b0 index out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // spatial axes sorted by stride: ax[0] is the fastest one in memory
    int ax[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (strides[ax[j]] < strides[ax[i]]) { const int t = ax[i]; ax[i] = ax[j]; ax[j] = t; }
    const long long cstride[3] = {(long long)dims[1] * dims[2], (long long)dims[2], 1};
    amx_prep *p = new amx_prep;
    p->ctx = ctx; p->nS = nS; p->n_out = n_out; p->n_b0 = n_b0; p->inplace = inplace;
    p->n_gidx = group_ptr[n_out];
    p->hazard = hazard;
    p->identity = n_out == nS;
    for (int j = 0; j < n_out && p->identity; j++)
        if (group_ptr[j + 1] != j + 1 || group_idx[j] != j) p->identity = 0;
    p->n_total = total; p->n_vox = n_vox; p->sv = strides[3];
    for (int k = 0; k < 3; k++) { p->d[k] = dims[ax[k]]; p->s[k] = strides[ax[k]]; p->c[k] = cstride[ax[k]]; }
    p->layout = p->s[0] == 1 ? 1 : (p->sv == 1 ? 2 : 0);
    p->extent = 1 + (long long)(nS - 1) * strides[3];
    for (int k = 0; k < 3; k++) p->extent += (long long)(dims[k] - 1) * strides[k];
    // rank in the image's memory-axis order, C-order positions of the masked voxels
    std::vector<int> rank_mem((size_t)total);
    std::vector<long long> cidx((size_t)n_vox, -1);
    long long seen = 0;
    bool ok = true;
    for (long long i2 = 0; i2 < p->d[2]; i2++)
        for (long long i1 = 0; i1 < p->d[1]; i1++)
            for (long long i0 = 0; i0 < p->d[0]; i0++) {
                const long long c = i0 * cstride[ax[0]] + i1 * cstride[ax[1]] + i2 * cstride[ax[2]];
                const int r = rank[c];
                rank_mem[(size_t)((i2 * p->d[1] + i1) * p->d[0] + i0)] = r;
                if (r >= 0) {
                    if (r >= n_vox || cidx[(size_t)r] != -1) ok = false;
                    else { cidx[(size_t)r] = c; seen++; }
                }
            }
    if (!ok || seen != n_vox) { delete p; return amx_bad(ctx, "amx_prep_create: rank must number the masked voxels 0..n_vox-1 exactly once"); }
    // tiles with at least one masked voxel (see amx_prep): 64 voxels along the fastest axis
    std::vector<int> live64;
    {
        const long long tpr = (p->d[0] + 63) / 64;
        for (long long row = 0; row < p->d[1] * p->d[2]; row++)
            for (long long xt = 0; xt < tpr; xt++) {
                const long long x1 = std::min<long long>(p->d[0], xt * 64 + 64);
                bool any = false;
                for (long long x = xt * 64; x < x1 && !any; x++) any = rank_mem[(size_t)(row * p->d[0] + x)] >= 0;
                if (any) live64.push_back((int)(row * tpr + xt));
            }
    }
    p->n_live64 = (long long)live64.size();
    int rc = dev_copy(ctx, &p->rank, rank_mem.data(), rank_mem.size());
    if (!rc) rc = dev_copy(ctx, &p->live64, live64.data(), live64.size());
    if (!rc) { const int zero[amx_prep::kCounterRing] = {0}; rc = dev_copy(ctx, &p->tile_counter, zero, amx_prep::kCounterRing); }
    if (!rc) rc = dev_copy(ctx, &p->cidx, cidx.data(), cidx.size());
    if (!rc) rc = dev_copy(ctx, &p->gptr, group_ptr, (size_t)n_out + 1);
    if (!rc) rc = dev_copy(ctx, &p->gidx, group_idx, (size_t)group_ptr[n_out]);
    if (!rc) rc = dev_copy(ctx, &p->b0idx, b0_idx, (size_t)n_b0);
    if (rc) { amx_prep_destroy(p); return rc; }
    *out = p;
    return AMX_OK;
}

static int prep_gather_dev(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize, float b0_threshold,
                           double *d_y, float *d_y32, float *d_mean_b0, void *hip_stream, const amx_dti *dti = nullptr, double *d_dirs = nullptr);

int amx_prep_gather_device(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize, float b0_threshold,
                           double *d_y, float *d_mean_b0, void *hip_stream)
{
    return prep_gather_dev(ctx, p, d_img, normalize, b0_threshold, d_y, nullptr, d_mean_b0, hip_stream);
}

int amx_prep_gather_device_f32(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize, float b0_threshold,
                               float *d_y, float *d_mean_b0, void *hip_stream)
{
    return prep_gather_dev(ctx, p, d_img, normalize, b0_threshold, nullptr, d_y, d_mean_b0, hip_stream);
}

// the gather with the tensor fit's principal directions taken along (core.py:209-268, 451-452 and 431-436, 456-458 in one pass over
// the image): what amx_prep_gather_device[_f32] followed by amx_dti_directions_device[_f32] computes
int amx_prep_gather_directions_device(amx_ctx *ctx, const amx_prep *p, const amx_dti *h, const float *d_img, int normalize,
                                      float b0_threshold, double *d_y, float *d_mean_b0, double *d_dirs, void *hip_stream)
{
    if (!h) return ctx ? amx_bad(ctx, "amx_prep_gather_directions: null tensor helper") : AMX_E_BADARG;
    return prep_gather_dev(ctx, p, d_img, normalize, b0_threshold, d_y, nullptr, d_mean_b0, hip_stream, h, d_dirs);
}

int amx_prep_gather_directions_device_f32(amx_ctx *ctx, const amx_prep *p, const amx_dti *h, const float *d_img, int normalize,
                                          float b0_threshold, float *d_y, float *d_mean_b0, double *d_dirs, void *hip_stream)
{
    if (!h) return ctx ? amx_bad(ctx, "amx_prep_gather_directions: null tensor helper") : AMX_E_BADARG;
    return prep_gather_dev(ctx, p, d_img, normalize, b0_threshold, nullptr, d_y, d_mean_b0, hip_stream, h, d_dirs);
}

static int prep_gather_dev(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize, float b0_threshold,
                           double *d_y, float *d_y32, float *d_mean_b0, void *hip_stream, const amx_dti *dti, double *d_dirs)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_gather: not a plan of this ctx");
    if (normalize && p->n_b0 == 0) return amx_bad(ctx, "amx_prep_gather: no b0 volume to normalize signal with");   // core.py:214-215
    if (p->n_vox == 0) return AMX_OK;
    if (!d_img || (!d_y && !d_y32)) return amx_bad(ctx, "amx_prep_gather: null buffer");
    if (dti != nullptr) {
        if (dti->ctx != ctx || !d_dirs) return amx_bad(ctx, "amx_prep_gather_directions: not a tensor helper of this ctx / null buffer");
        if (dti->nS != p->n_out) return amx_bad(ctx, "amx_prep_gather_directions: the tensor helper's scheme does not match the plan's output volumes");
    }
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PrepArgs a;
    memset(&a, 0, sizeof a);
    a.img = d_img; a.rank = p->rank; a.y = d_y; a.y32 = d_y32; a.mean_b0 = normalize ? d_mean_b0 : nullptr;
    a.d0 = p->d[0]; a.d1 = p->d[1]; a.d2 = p->d[2]; a.s0 = p->s[0]; a.s1 = p->s[1]; a.s2 = p->s[2]; a.sv = p->sv;
    a.tiles_per_row = (p->d[0] + 63) / 64;
    a.n_tiles = a.tiles_per_row * p->d[1] * p->d[2];
    a.nS = p->nS; a.n_out = p->n_out; a.n_b0 = p->n_b0; a.ldt = p->nS | 1; a.inplace = p->inplace;
    a.layout = p->layout; a.normalize = normalize ? 1 : 0;
    a.gptr = p->gptr; a.gidx = p->gidx; a.b0idx = p->b0idx; a.thr = b0_threshold;
    a.n_gidx = p->n_gidx;
    if (dti != nullptr) { a.wt = dti->wt; a.min_signal = dti->min_signal; a.dirs = d_dirs; }
    const bool identity = p->identity != 0;
    a.direct = (identity && p->layout != 2 && !ctx->opt_prep_no_direct) ? 1 : 0;
    const size_t per_wave = identity ? (a.direct ? (size_t)a.nS * kDirectLd : (size_t)64 * a.ldt) : (size_t)64 * a.ldt * (a.inplace ? 1 : 2);
    const size_t lds = ((size_t)kPrepWaves * per_wave + (size_t)p->n_b0 + p->n_out + 1 + p->n_gidx) * sizeof(float);
    if (lds > 160 * 1024) return amx_bad(ctx, "amx_prep_gather: scheme too long for the LDS tile");
    static bool attr_set[64];                                        // per device: the attribute belongs to (function, device)
    if (!attr_set[ctx->device & 63]) {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)k_prep_gather<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device & 63] = true;
    }
    const int per_cu = (int)((160 * 1024) / lds) > 8 ? 8 : (int)((160 * 1024) / lds);
    long long grid = 256LL * per_cu;               // what is resident at once: the wavefronts draw their tiles from the live list
    const long long need = (p->n_live64 + kPrepWaves - 1) / kPrepWaves;
    if (grid > need) grid = need;
    // every launch draws its tiles from ITS OWN counter (a ring of kCounterRing, taken in launch order): two gathers of one plan may
    // be in flight together (two streams, double-buffered images sharing a mask) without sharing or resetting each other's ticket
    int *const my_counter = p->tile_counter + (p->launch_seq++ % amx_prep::kCounterRing);
    a.live = p->live64; a.n_live = p->n_live64; a.counter = my_counter;
    if (!identity && !p->hazard && p->layout == 1 && !ctx->opt_prep_tile) {
        // grouped outputs on a planar image: streaming kernel, no transposition tile
        const size_t lds_s = ((size_t)p->n_b0 + p->n_out + 1 + p->n_gidx) * sizeof(int);
        long long g2 = (a.n_tiles + 3) / 4;
        if (g2 > 256LL * 16) g2 = 256LL * 16;
        rec(ctx, 8, s);
        const bool contig = p->s[0] == 1 && p->s[1] == p->d[0] && p->s[2] == p->d[0] * p->d[1] && (p->sv & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(d_img) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->rank) & 15) == 0;
        if (contig && !ctx->opt_prep_scalar) {
            long long g4 = ((p->d[0] * p->d[1] * p->d[2] + 255) / 256 + 3) / 4;
            if (g4 > 256LL * 16) g4 = 256LL * 16;
            hipLaunchKernelGGL(k_prep_stream4, dim3((unsigned)g4), dim3(256), lds_s, s, a);
        } else {
            hipLaunchKernelGGL(k_prep_stream, dim3((unsigned)g2), dim3(256), lds_s, s, a);
        }
        HIPCHK(ctx, hipGetLastError());
        rec(ctx, 9, s);
        // (the streaming kernels hold no tile a tensor fit could ride on: the directions come from the rows they wrote)
        if (dti != nullptr)
            return d_y32 ? amx_dti_directions_device_f32(ctx, dti, d_y32, p->n_vox, d_dirs, hip_stream)
                         : amx_dti_directions_device(ctx, dti, d_y, p->n_vox, d_dirs, hip_stream);
        return AMX_OK;
    }
    HIPCHK(ctx, hipMemsetAsync(my_counter, 0, sizeof(int), s));
    rec(ctx, 8, s);
    if (dti != nullptr) {
        if (identity && a.direct) hipLaunchKernelGGL((k_prep_gather<true, true, true>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
        else if (identity) hipLaunchKernelGGL((k_prep_gather<true, false, true>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
        else hipLaunchKernelGGL((k_prep_gather<false, false, true>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
    }
    else if (identity && a.direct) hipLaunchKernelGGL((k_prep_gather<true, true>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
    else if (identity) hipLaunchKernelGGL((k_prep_gather<true, false>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
    else hipLaunchKernelGGL((k_prep_gather<false, false>), dim3((unsigned)grid), dim3(64 * kPrepWaves), lds, s, a);
    HIPCHK(ctx, hipGetLastError());
    rec(ctx, 9, s);
    return AMX_OK;
}

int amx_prep_mean_b0_device(amx_ctx *ctx, const amx_prep *p, const float *d_img, float *d_mean_b0_volume, void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_mean_b0: not a plan of this ctx");
    if (p->n_b0 == 0) return amx_bad(ctx, "amx_prep_mean_b0: no b0 volume to normalize signal with");
    if (!d_img || !d_mean_b0_volume) return amx_bad(ctx, "amx_prep_mean_b0: null buffer");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long long blocks = (p->n_total + 255) / 256;
    hipLaunchKernelGGL(k_mean_b0, dim3((unsigned)blocks), dim3(256), 0, s, d_img, p->d[0], p->d[1], p->d[2], p->s[0],
                       p->s[1], p->s[2], p->sv, p->c[0], p->c[1], p->c[2], p->b0idx, p->n_b0, d_mean_b0_volume);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

int amx_prep_scatter_device(amx_ctx *ctx, const amx_prep *p, const double *d_values, int n_cols, float *d_volume,
                            void *hip_stream)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_scatter: not a plan of this ctx");
    if (n_cols < 1 || !d_volume || (p->n_vox > 0 && !d_values)) return amx_bad(ctx, "amx_prep_scatter: bad argument");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(d_volume, 0, (size_t)p->n_total * n_cols * sizeof(float), s));     // np.zeros(...)
    if (p->n_vox == 0) return AMX_OK;
    const long long blocks = (p->n_vox * n_cols + 255) / 256;
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)blocks), dim3(256), 0, s, d_values, p->cidx, p->n_vox, n_cols, d_volume);
    HIPCHK(ctx, hipGetLastError());
    return AMX_OK;
}

// ---- host-buffer variants (H2D + kernel + D2H, blocking)
int amx_prep_gather(amx_ctx *ctx, const amx_prep *p, const float *img, int normalize, float b0_threshold,
                    double *out_y, float *out_mean_b0)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_gather: not a plan of this ctx");
    if (p->n_vox == 0) return AMX_OK;
    if (!img || !out_y) return amx_bad(ctx, "amx_prep_gather: null buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t ib = (size_t)p->extent * sizeof(float), yb = (size_t)p->n_vox * p->n_out * sizeof(double);
    if ((rc = amx_ensure(ctx, ctx->hextra, ib))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hy, yb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hrmse, (size_t)p->n_vox * sizeof(float)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->hextra.p, img, ib, hipMemcpyHostToDevice, nullptr));
    if ((rc = amx_prep_gather_device(ctx, p, (const float *)ctx->hextra.p, normalize, b0_threshold, (double *)ctx->hy.p,
                                     (float *)ctx->hrmse.p, nullptr))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_y, ctx->hy.p, yb, hipMemcpyDeviceToHost, nullptr));
    if (normalize && out_mean_b0)
        HIPCHK(ctx, hipMemcpyAsync(out_mean_b0, ctx->hrmse.p, (size_t)p->n_vox * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

int amx_prep_mean_b0(amx_ctx *ctx, const amx_prep *p, const float *img, float *out_mean_b0_volume)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_mean_b0: not a plan of this ctx");
    if (!img || !out_mean_b0_volume) return amx_bad(ctx, "amx_prep_mean_b0: null buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t ib = (size_t)p->extent * sizeof(float), ob = (size_t)p->n_total * sizeof(float);
    if ((rc = amx_ensure(ctx, ctx->hextra, ib))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hrmse, ob))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->hextra.p, img, ib, hipMemcpyHostToDevice, nullptr));
    if ((rc = amx_prep_mean_b0_device(ctx, p, (const float *)ctx->hextra.p, (float *)ctx->hrmse.p, nullptr))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_mean_b0_volume, ctx->hrmse.p, ob, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

int amx_prep_scatter(amx_ctx *ctx, const amx_prep *p, const double *values, int n_cols, float *out_volume)
{
    if (!ctx) return AMX_E_BADARG;
    if (!p || p->ctx != ctx) return amx_bad(ctx, "amx_prep_scatter: not a plan of this ctx");
    if (n_cols < 1 || !out_volume || (p->n_vox > 0 && !values)) return amx_bad(ctx, "amx_prep_scatter: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t vb = (size_t)p->n_vox * n_cols * sizeof(double), ob = (size_t)p->n_total * n_cols * sizeof(float);
    if ((rc = amx_ensure(ctx, ctx->hy, vb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hextra, ob))) return rc;
    if (vb) HIPCHK(ctx, hipMemcpyAsync(ctx->hy.p, values, vb, hipMemcpyHostToDevice, nullptr));
    if ((rc = amx_prep_scatter_device(ctx, p, (const double *)ctx->hy.p, n_cols, (float *)ctx->hextra.p, nullptr))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_volume, ctx->hextra.p, ob, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

}  // extern "C"

// ============================================================================ LUT resampling (SURVEY section 8 f, row 4)
// lut.pyx:274-311 `resample_kernel`:  KR = ones(ndirs, nS);  KR[i, idx_out] = dot(Ylm_out, KRlm[i, :])  for every
// orientation i -- batched over all atoms as ONE float32 GEMM  C[M x N] = L[M x K] * Ylm^T[K x N]  (M = atoms * ndirs rows
// of rotated SH coefficients, K = nSH * shells, N = number of DWI volumes) on the matrix cores.
namespace amx {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_fill_ones(float *out, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = 1.0f;
}

// Shapes whose SH -> signal operator fits the LDS (K <= 256 reduction indices; NODDI / FreeWater: K = 182, N = 90).
// One wavefront = 32 rows of L against all N columns, 32 at a time, with v_mfma_f32_32x32x2_f32.  The reduction index is
// split in two halves, one per half-wavefront (lane l works on k = (l / 32) * 2 KH2 + j): every lane owns a contiguous
// half row of L, held in registers for all column tiles, and a half row of Ylm in LDS (N x K floats staged once per
// workgroup, read with ds_read_b64; the row stride has an odd number of 8-byte words: 32 rows hit all 64 banks).
// KH2 = pairs per half-wavefront, a compile-time bound: the loops carry no conditions (indices beyond K read zeros).
// The order of a sum does not matter to the GEMM.  Measured for 72 000 x 182 x 90 (profiles/): 0.093 ms = 25 TFLOP/s, 16 %
// of the f32 matrix peak (0.35 ms before); variants that did NOT help: the rows of L through a coalesced per-wavefront
// LDS tile (0.13 ms at one workgroup per CU), three independent accumulators (0.10 ms, one wavefront per SIMD).
constexpr int kLutKhMax = 64;

// Fused mode (Z != nullptr, amx_lut_rotate_resample): row (atom, orientation) of L is never materialised --
// L[atom * ndirs + dir][k] = Z[atom][k] * R[dir][k mod n_sh]  (rotate_kernel, lut.pyx:262-264: const * Klm[idx_m0] * Ylm_rot).
template <int KH2>
__global__ __launch_bounds__(256) void k_lut_resample(const float *__restrict__ L, const float *__restrict__ Y,
                                                      const int *__restrict__ idx_out, long long M, int K, int N, int nS,
                                                      float *__restrict__ out, const float *__restrict__ Z = nullptr,
                                                      const float *__restrict__ R = nullptr, int ndirs = 1, int n_sh = 1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
    constexpr int Kp = 4 * KH2;                                      // padded K
    constexpr int Ks = Kp + 2;                                       // Ylm row stride (floats): Ks / 2 odd
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int Npad = (N + 31) & ~31;
    float *Ys = reinterpret_cast<float *>(smem_l);                    // [Npad][Ks], zero padded
    // (rows by wavefront, columns by lane, eight rows of loads in flight)
    for (int nb = wave * 8; nb < Npad; nb += 32) {
        for (int k = lane; k < Ks; k += 64) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int n = nb + u;
                t[u] = (n < N && k < K) ? Y[(long long)n * K + k] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (nb + u < Npad) Ys[(size_t)(nb + u) * Ks + k] = t[u];
        }
    }
    __syncthreads();
    const bool even = (K & 1) == 0;                                  // rows of L start 8-byte aligned
    for (long long m0 = ((long long)blockIdx.x * 4 + wave) * 32; m0 < M; m0 += (long long)gridDim.x * 128) {
        // this lane's half row of L, loaded once with 8-byte loads (rows beyond M repeat the last one; masked at the store)
        const long long row = m0 + l32 < M ? m0 + l32 : M - 1;
        float2 a2[KH2];
        if (Z == nullptr) {
            const float *lrow = L + row * K;
#pragma unroll
            for (int j = 0; j < KH2; j++) {
                const int k = 2 * half * KH2 + 2 * j;
                if (even && 2 * KH2 + 2 * j + 1 < K) {               // (holds for both halves: no condition per lane)
                    a2[j] = *reinterpret_cast<const float2 *>(lrow + k);
                } else {
                    a2[j].x = (k < K) ? lrow[k] : 0.0f;
                    a2[j].y = (k + 1 < K) ? lrow[k + 1] : 0.0f;
                }
            }
        } else {
            // rotation by the addition theorem, formed in registers: per-(l, m) factor of the atom x basis value of the orientation
            const float *zrow = Z + (row / ndirs) * K, *rrow = R + (row % ndirs) * n_sh;
            int c = (2 * half * KH2) % n_sh;                          // position inside the shell's block of coefficients
#pragma unroll
            for (int j = 0; j < KH2; j++) {
                const int k = 2 * half * KH2 + 2 * j;
                const int c1 = (c + 1 == n_sh) ? 0 : c + 1;
                a2[j].x = (k < K) ? zrow[k] * rrow[c] : 0.0f;
                a2[j].y = (k + 1 < K) ? zrow[k + 1] * rrow[c1] : 0.0f;
                c = (c1 + 1 == n_sh) ? 0 : c1 + 1;
            }
        }
        for (int n0 = 0; n0 < N; n0 += 32) {
            const float2 *yrow = reinterpret_cast<const float2 *>(Ys + (size_t)(n0 + l32) * Ks + 2 * half * KH2);
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
            for (int j = 0; j < KH2; j++) {
                const float2 b2 = yrow[j];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].x, b2.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j].y, b2.y, acc, 0, 0, 0);
            }
            const int n = n0 + l32;
            if (n < N) {
                const int col = idx_out[n];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long long r = m0 + 8 * (i / 4) + 4 * half + (i % 4);
                    if (r < M) out[r * nS + col] = acc[i];
                }
            }
        }
    }
}

// The same GEMM with BOTH operands in LDS (plain `lm` input, K even, operator + four row tiles <= 160 KB): what limited
// k_lut_resample was not the matrix pipe but its left operand -- every lane walking its own 728-byte row of L with 8-byte
// loads (64 cache lines per load instruction: 59 % of the wavefronts' time, matrix pipe 18 % busy).  32 consecutive rows of L
// are ONE contiguous 23 KB block, so a wavefront streams its tile global -> LDS with direct 16-byte loads (23 instructions,
// all in flight at once) and reads the MFMA operand from there; the row stride K = 182 floats puts 32 consecutive rows into
// 32 different even banks, for the tile and for the operator alike, so neither needs padding or a swizzle.  A row is read up
// to 4 KH2 - K floats beyond its end (the next row: finite values); the left operand is forced to zero there.
#ifdef AMX_LUT_PHASES
__device__ unsigned long long g_lut_ph[8];
#define LPH_T() __builtin_readcyclecounter()
#define LPH_ADD(k, t0) ph[k] += __builtin_readcyclecounter() - (t0)
#else
#define LPH_T() 0ull
#define LPH_ADD(k, t0) (void)(t0)
#endif
template <int KH2>
__global__ __launch_bounds__(256) void k_lut_resample_tile(const float *__restrict__ L, const float *__restrict__ Y,
                                                           const int *__restrict__ idx_out, long long M, int K, int N, int nS,
                                                           float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l32 = lane & 31;
#ifdef AMX_LUT_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_all = LPH_T();
#endif
    unsigned long long t0 = LPH_T();
    // LDS: Ys [N][K] (+ a zeroed tail of 1056 B) | per wavefront: Lt [32][K]
    float *Ys = reinterpret_cast<float *>(smem_t);
    const long long ybytes = (long long)N * K * 4;
    const long long yzone = ((ybytes + 1023) & ~1023LL) + 2048;      // the copy's last piece ends inside; rows N .. N + 31 start inside or behind
    const long long tile_bytes = 32LL * K * 4, tile_stride = (tile_bytes + 1023) & ~1023LL;      // whole 1 KB pieces per wavefront
    float *Lt = reinterpret_cast<float *>(smem_t + yzone + wave * tile_stride);
    {
        const int pieces = (int)((ybytes + 1023) >> 10);
        for (int p = wave; p < pieces; p += 4) {
            long long off = ((long long)p << 10) + lane * 16;
            if (off > ybytes - 16) off = ybytes - 16;                 // (the tail lanes of the last piece repeat its last 16 bytes)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char *>(Y) + off,
                                             (__attribute__((address_space(3))) void *)(smem_t + ((long long)p << 10)), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // behind the last row of the operator: zeros (the over-read of row N - 1 and the rows N .. of the last column block)
        for (long long pos = ybytes / 4 + threadIdx.x; pos < yzone / 4; pos += blockDim.x) Ys[pos] = 0.0f;
        __syncthreads();
    }
    LPH_ADD(0, t0);
    // the tile of rows m0 .. m0 + 31 -> this wavefront's LDS block (whole 1 KB pieces; the tail lanes of the last piece repeat
    // its last 16 bytes, which land in the slack behind the tile)
    auto load_tile = [&](long long m0) {
        const long long rows = (M - m0) < 32 ? (M - m0) : 32;
        const long long bytes = rows * K * 4;
        const char *src = reinterpret_cast<const char *>(L + m0 * K);
        const int pieces = (int)(tile_stride >> 10);
#pragma unroll 4
        for (int p = 0; p < pieces; p++) {
            long long off = ((long long)p << 10) + lane * 16;
            if (off > bytes - 16) off = bytes - 16;
            __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(Lt) + ((long long)p << 10)),
                                             16, 0, 0);
        }
    };
    // (measured alternatives, both slower than this plain order: issuing the next tile's loads behind the last block's MFMAs so
    //  that they overlap the stores -- 0.064 instead of 0.053 ms; advancing all column blocks together with one read of the left
    //  operand per K-step -- 0.060 ms: the stores of a block then no longer overlap the MFMAs of the next one)
    auto process = [&](long long m0, int c_begin, int c_end) {
        unsigned long long t1 = LPH_T();
        load_tile(m0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (also: the previous tile's stores are out)
        LPH_ADD(1, t1);
        const float2 *arow = reinterpret_cast<const float2 *>(Lt + (size_t)l32 * K + 2 * half * KH2);
        for (int c = c_begin; c < c_end; c++) {
            t1 = LPH_T();
            const int n0 = 32 * c;
            const float2 *yrow = reinterpret_cast<const float2 *>(Ys + (size_t)(n0 + l32) * K + 2 * half * KH2);
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
            for (int j = 0; j < KH2; j++) {
                const int k = 2 * half * KH2 + 2 * j;
                float2 a2 = arow[j];
                const float2 b2 = yrow[j];
                if (2 * KH2 + 2 * j >= K) {                          // (compile-time position: only the last columns of the second half)
                    a2.x = (k < K) ? a2.x : 0.0f;
                    a2.y = (k + 1 < K) ? a2.y : 0.0f;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2.y, acc, 0, 0, 0);
            }
            LPH_ADD(2, t1); t1 = LPH_T();
            const int n = n0 + l32;
            if (n < N) {
                const int col = idx_out[n];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long long r = m0 + 8 * (i / 4) + 4 * half + (i % 4);
                    if (r < M) out[r * nS + col] = acc[i];
                }
            }
            LPH_ADD(3, t1);
        }
#ifdef AMX_LUT_PHASES
        ph[5] += 1;
#endif
    };
    // Whole tiles first, the same number for every wavefront; the tiles that are left over (2250 tiles over 1024 wavefronts: 202)
    // are cut into their column blocks, so that the launch ends after 2 1/3 tiles' worth of work instead of 3.
    const int nb = (N + 31) >> 5;
    const long long T = (M + 31) >> 5, W = (long long)gridDim.x * 4, wid = (long long)blockIdx.x * 4 + wave;
    const long long F = T / W;
    for (long long f = 0; f < F; f++) process((wid + f * W) * 32, 0, nb);
    for (long long u = wid; u < (T - F * W) * nb; u += W) process((F * W + u / nb) * 32, (int)(u % nb), (int)(u % nb) + 1);
#ifdef AMX_LUT_PHASES
    ph[7] = LPH_T() - t_all;
    if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lut_ph[k], ph[k]);
#endif
}

// Generic shapes (more than 256 reduction indices, or an SH -> signal operator beyond the LDS: SANDI's 5 shells x 91
// coefficients x 300 volumes): one wavefront = 32 rows of L against 32 columns at a time, operands straight from L2.  The reduction
// index is split in two halves, one per half-wavefront (lane l works on k = (l / 32) * Kh + j), so that every lane
// streams a contiguous piece of its row of L and of its row of Ylm; the order of a sum does not matter to the GEMM.
__global__ __launch_bounds__(256) void k_lut_resample_generic(const float *__restrict__ L, const float *__restrict__ Y,
                                                      const int *__restrict__ idx_out, long long M, int K, int N, int nS,
                                                      float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int Kh = (K + 1) / 2;
    const long long m0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (m0 >= M) return;
    const long long row = m0 + l32;
    const float *lrow = L + (row < M ? row : M - 1) * K + (long long)half * Kh;
    const int kcnt = half == 0 ? Kh : K - Kh;                        // elements of this lane's half
    for (int n0 = 0; n0 < N; n0 += 32) {
        const int n = n0 + l32;
        const float *yrow = Y + (long long)(n < N ? n : N - 1) * K + (long long)half * Kh;
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        for (int j = 0; j < Kh; j++) {
            const float av = (j < kcnt && row < M) ? lrow[j] : 0.0f;
            const float bv = (j < kcnt && n < N) ? yrow[j] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        if (n < N) {
            const int col = idx_out[n];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const long long r = m0 + 8 * (i / 4) + 4 * half + (i % 4);
                if (r < M) out[r * nS + col] = acc[i];
            }
        }
    }
}

}  // namespace amx

// shared by amx_lut_resample (lm given) and amx_lut_rotate_resample (zonal factors + basis values given): device buffers
// d_lm or (d_z, d_r) -> ctx->hextra (f32 [n_rows][nS], ones outside idx_out)
static int lut_gemm(amx_ctx *ctx, const float *d_lm, const float *d_z, const float *d_r, int ndirs, int n_sh1, int64_t n_rows,
                    int n_sh, const float *d_ylm, const int *d_idx, int n_out, int nS)
{
    int rc = AMX_OK;
    hipLaunchKernelGGL(amx::k_fill_ones, dim3(2048), dim3(256), 0, nullptr, (float *)ctx->hextra.p, (long long)n_rows * nS);
    const int kh2 = (n_sh + 3) / 4;
    const int kh2t = kh2 <= 23 ? 23 : (kh2 <= 46 ? 46 : 64);         // compile-time reduction lengths: 1 / 2 shells of lmax 12, <= 256
    const size_t lds = (size_t)((n_out + 31) & ~31) * (4 * kh2t + 2) * sizeof(float);
    // both operands in LDS: plain input, even K, operator zone + four row tiles within the 160 KB of a CU
    const size_t yzone = ((((size_t)n_out * n_sh * 4) + 1023) & ~(size_t)1023) + 2048;
    const size_t lds_tile = yzone + 4 * ((((size_t)32 * n_sh * 4) + 1023) & ~(size_t)1023);
    if (d_lm && (n_sh & 1) == 0 && n_sh <= 4 * amx::kLutKhMax && 4 * kh2t - n_sh <= 4 && lds_tile <= 160 * 1024 && !ctx->opt_lut_regs) {
        auto launch = [&](auto kern) -> int {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
            long long blocks = (n_rows + 127) / 128;
            if (blocks > ctx->n_cu) blocks = ctx->n_cu;              // persistent, one workgroup per CU: the operator is staged once
            rec(ctx, 8, nullptr);
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_tile, nullptr, d_lm, d_ylm, d_idx, (long long)n_rows, n_sh,
                               n_out, nS, (float *)ctx->hextra.p);
            return AMX_OK;
        };
        if (kh2t == 23) rc = launch(amx::k_lut_resample_tile<23>);
        else if (kh2t == 46) rc = launch(amx::k_lut_resample_tile<46>);
        else rc = launch(amx::k_lut_resample_tile<64>);
        if (rc) return rc;
    } else
    if (n_sh > 4 * amx::kLutKhMax || lds > 80 * 1024) {              // generic shapes: operands from L2
        if (!d_lm) return amx_bad(ctx, "amx_lut_rotate_resample: shape beyond the fused kernel (K <= 256, operator <= 80 KB)");
        rec(ctx, 8, nullptr);
        hipLaunchKernelGGL(amx::k_lut_resample_generic, dim3((unsigned)((n_rows + 127) / 128)), dim3(256), 0, nullptr, d_lm, d_ylm,
                           d_idx, (long long)n_rows, n_sh, n_out, nS, (float *)ctx->hextra.p);
    } else {
        auto launch = [&](auto kern) -> int {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            long long blocks = (n_rows + 127) / 128;                  // persistent: the Ylm tile is staged once per workgroup
            const long long cap = 2LL * ctx->n_cu;
            if (blocks > cap) blocks = cap;
            rec(ctx, 8, nullptr);
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nullptr, d_lm, d_ylm, d_idx, (long long)n_rows, n_sh,
                               n_out, nS, (float *)ctx->hextra.p, d_z, d_r, ndirs, n_sh1);
            return AMX_OK;
        };
        if (kh2t == 23) rc = launch(amx::k_lut_resample<23>);
        else if (kh2t == 46) rc = launch(amx::k_lut_resample<46>);
        else rc = launch(amx::k_lut_resample<64>);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipGetLastError());
    rec(ctx, 9, nullptr);
    return AMX_OK;
}

extern "C" int amx_lut_resample(amx_ctx *ctx, const float *lm, int64_t n_rows, int n_sh, const float *ylm_out,
                                const int32_t *idx_out, int n_out, int nS, float *out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!lm || !ylm_out || !idx_out || !out || n_rows < 1 || n_sh < 1 || n_out < 1 || nS < n_out)
        return amx_bad(ctx, "amx_lut_resample: bad argument");
    for (int k = 0; k < n_out; k++) if (idx_out[k] < 0 || idx_out[k] >= nS) return amx_bad(ctx, "amx_lut_resample: idx_out out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t lb = (size_t)n_rows * n_sh * sizeof(float), yb = (size_t)n_out * n_sh * sizeof(float);
    const size_t ob = (size_t)n_rows * nS * sizeof(float);
    if ((rc = amx_ensure(ctx, ctx->hy, lb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hdirs, yb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hrmse, (size_t)n_out * sizeof(int)))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hextra, ob))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->hy.p, lm, lb, hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(ctx->hdirs.p, ylm_out, yb, hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(ctx->hrmse.p, idx_out, (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, nullptr));
    if ((rc = lut_gemm(ctx, (const float *)ctx->hy.p, nullptr, nullptr, 1, 1, n_rows, n_sh, (const float *)ctx->hdirs.p,
                       (const int *)ctx->hrmse.p, n_out, nS))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->hextra.p, ob, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

extern "C" int amx_lut_rotate_resample(amx_ctx *ctx, const float *zonal, int n_atoms, const float *ylm_rot, int ndirs,
                                       int n_sh_shell, int n_shells, const float *ylm_out, const int32_t *idx_out, int n_out,
                                       int nS, float *out)
{
    if (!ctx) return AMX_E_BADARG;
    if (!zonal || !ylm_rot || !ylm_out || !idx_out || !out || n_atoms < 1 || ndirs < 1 || n_sh_shell < 1 || n_shells < 1 ||
        n_out < 1 || nS < n_out)
        return amx_bad(ctx, "amx_lut_rotate_resample: bad argument");
    for (int k = 0; k < n_out; k++) if (idx_out[k] < 0 || idx_out[k] >= nS) return amx_bad(ctx, "amx_lut_rotate_resample: idx_out out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const int K = n_sh_shell * n_shells;
    const int64_t n_rows = (int64_t)n_atoms * ndirs;
    const size_t zb = (size_t)n_atoms * K * sizeof(float), rb = (size_t)ndirs * n_sh_shell * sizeof(float);
    const size_t yb = (size_t)n_out * K * sizeof(float), ob = (size_t)n_rows * nS * sizeof(float);
    if ((rc = amx_ensure(ctx, ctx->hy, zb + rb + 64))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hdirs, yb))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hrmse, (size_t)n_out * sizeof(int)))) return rc;
    if ((rc = amx_ensure(ctx, ctx->hextra, ob))) return rc;
    float *d_z = (float *)ctx->hy.p, *d_r = (float *)((char *)ctx->hy.p + ((zb + 15) & ~(size_t)15));
    HIPCHK(ctx, hipMemcpyAsync(d_z, zonal, zb, hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(d_r, ylm_rot, rb, hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(ctx->hdirs.p, ylm_out, yb, hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(ctx->hrmse.p, idx_out, (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, nullptr));
    if ((rc = lut_gemm(ctx, nullptr, d_z, d_r, ndirs, n_sh_shell, n_rows, K, (const float *)ctx->hdirs.p, (const int *)ctx->hrmse.p,
                       n_out, nS))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->hextra.p, ob, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(nullptr));
    return AMX_OK;
}

#ifdef AMX_LUT_PHASES
extern "C" int amx_debug_lut_phases(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::g_lut_ph), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(amx::g_lut_ph), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
