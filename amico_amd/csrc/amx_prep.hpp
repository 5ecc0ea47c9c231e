// amx_prep.hpp -- small bandwidth kernels around the solvers (included by amx_api.hip only):
// direction -> LUT index (lut.pyx:316-356), counting sort by orientation, dictionary tiles.
#pragma once
#include "amx_kernels.hpp"

namespace amx {

// ------------------------------------------------------------------ lut.pyx:316-356
__device__ __forceinline__ int dir_to_lut_idx_dev(double d0, double d1, double d2,
                                                  const short *__restrict__ ht, int &ii1, int &ii2)
{
    const double pi = 3.14159265358979323846;
    if (d1 < 0.0) { d0 = -d0; d1 = -d1; d2 = -d2; }          // on a copy: caller's DIRs stay const
    double i1, i2 = fmod(atan2(d1, d0), 2.0 * pi);
    if (i2 < 0.0) i2 = fmod(i2 + 2.0 * pi, 2.0 * pi);
    if (i2 > pi) {
        i2 = fmod(atan2(-d1, -d0), 2.0 * pi);
        i1 = atan2(sqrt(d0 * d0 + d1 * d1), -d2);
    } else {
        i1 = atan2(sqrt(d0 * d0 + d1 * d1), d2);
    }
    const double r1 = round(i1 / pi * 180.0), r2 = round(i2 / pi * 180.0);
    if (!(r1 >= -1.0 && r1 <= 181.0) || !(r2 >= -1.0 && r2 <= 181.0)) { ii1 = -1; ii2 = -1; return -1; }
    ii1 = (int)r1; ii2 = (int)r2;
    if (ii1 < 0 || ii1 > 180 || ii2 < 0 || ii2 > 180) return -1;
    return (int)ht[ii1 * 181 + ii2];
}

// LDS > 0: the block first counts its voxels per orientation in an LDS histogram and issues ONE global atomic per
// non-empty bin (1 M voxels on 500 orientations: 1 M contended global atomics otherwise).  kPrepSpan voxels per block.
constexpr int kPrepSpan = 8192;
// (round 6) ... of a LARGE call.  A thread's voxels are a chain of three atan2 each -- ~3 us a link --, and 8192 voxels per block are 13
// blocks for 100 000 voxels: the kernel lasted 24 us at every size up to 1 M.  prep_span(): a multiple of 1024 that gives every CU a block
// before any thread gets a second voxel (100 000 voxels: 24 -> 6 us; the per-bin global atomics grow with the blocks, hence the cap)
inline int prep_span(int64_t n) { const int64_t s = ((n + 255) / 256 + 1023) / 1024 * 1024; return (int)(s < 1024 ? 1024 : (s > kPrepSpan ? kPrepSpan : s)); }

// zrows (optional): the output rows [n][zcols] of the voxels that are SKIPPED (direction out of bounds: the call returns an error) are
// zeroed here, so that a fit need not clear its whole output first
__global__ __launch_bounds__(1024) void k_dir_to_lut(const double *__restrict__ dirs, int n, const short *__restrict__ ht,
                                                     int ndirs, int *__restrict__ lutidx, int *__restrict__ counts,
                                                     int *__restrict__ status, int use_lds, int vbase, int span,
                                                     double *__restrict__ zrows, int zcols)
{
    extern __shared__ int hist[];
    if (use_lds) {
        for (int i = threadIdx.x; i < ndirs; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    const int v0 = blockIdx.x * span;
    for (int v = v0 + threadIdx.x; v < v0 + span && v < n; v += blockDim.x) {
        int ii1, ii2;
        int idx = dir_to_lut_idx_dev(dirs[3 * (size_t)v], dirs[3 * (size_t)v + 1], dirs[3 * (size_t)v + 2], ht, ii1, ii2);
        if (idx < 0 || idx >= ndirs) {
            idx = -1;
            // first offending voxel AND its (i1, i2) in one 64-bit atomic -- two plain stores behind an atomicMin on the voxel alone could
            // pair the smallest voxel with another voxel's indices (vbase: offset of this batch in the caller's arrays; i1, i2 in -1 .. 181)
            atomicMin(reinterpret_cast<unsigned long long *>(status + ST_ERRPACK),
                      ((unsigned long long)(unsigned)(v + vbase) << 32) | (unsigned long long)(((unsigned)(ii1 + 1) << 16) | (unsigned)(ii2 + 1)));
            if (zrows) for (int j = 0; j < zcols; j++) zrows[(size_t)v * zcols + j] = 0.0;
        } else if (counts) {
            atomicAdd(use_lds ? &hist[idx] : &counts[idx], 1);
        }
        lutidx[v] = idx;
    }
    if (use_lds && counts) {
        __syncthreads();
        for (int i = threadIdx.x; i < ndirs; i += blockDim.x)
            if (hist[i]) atomicAdd(&counts[i], hist[i]);
    }
}

// The counters a fit starts from -- histogram, per-call words, chunk tickets and list counts -- cleared by one launch (three memset
// nodes cost a small call ~11 us each: the fill itself and the gap before the next packet of the stream)
__global__ void k_clear3(int *__restrict__ a, int na, int *__restrict__ b, int nb, int *__restrict__ c, int nc)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (int i = t; i < na; i += step) a[i] = 0;
    for (int i = t; i < nb; i += step) b[i] = 0;
    for (int i = t; i < nc; i += step) c[i] = 0;
}

// Longest chunks first for the lane kernels of the second plan: a kernel with one workgroup per chunk and one or two workgroups per
// CU runs two or three "rounds" of workgroups over the chip, and the populations of the orientations differ by +-30 %; started in
// order of decreasing size (LPT) the last round ends together instead of waiting for its largest member.  The kernels map
// blockIdx -> chunk through xcd_chunk(): the p-th largest chunk is stored where the p-th dispatched workgroup looks.
// (round 6: the tail of k_plan -- the same single block -- instead of a launch of its own)
constexpr int kOrderCap = 2048;
__device__ __forceinline__ void order_schunks(Chunk *__restrict__ chunks2, int n, unsigned long long *key, Chunk *tmp, int *map)
{
    constexpr int CAP = kOrderCap;
    if (n < 2 || n > CAP) return;
    int cap = 2;
    while (cap < n) cap <<= 1;                               // sort size: the next power of two
    for (int i = threadIdx.x; i < cap; i += blockDim.x) {
        key[i] = i < n ? (((unsigned long long)(unsigned)chunks2[i].count << 32) | (unsigned)(CAP - 1 - i)) : 0ull;   // ties: list order
        if (i < n) tmp[i] = chunks2[i];
    }
    {   // map[rank in dispatch order] = chunk index: block b = (row r = b >> 3, XCD x = b & 7) reads index x * per + r; the only
        // blocks without a chunk sit in the last column(s), rows r >= n - x * per
        const int per = (n + 7) >> 3, r0 = n - 7 * per;
        if (r0 >= 0) {
            for (int b = threadIdx.x; b < 8 * per; b += blockDim.x) {
                const int r = b >> 3, x = b & 7, cid = x * per + r;
                if (cid < n) map[b - (r > r0 ? r - r0 : 0)] = cid;
            }
        } else if (threadIdx.x == 0) {
            int r = 0;
            for (int b = 0; b < 8 * per; b++) {
                const int cid = (b & 7) * per + (b >> 3);
                if (cid < n) map[r++] = cid;
            }
        }
    }
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1) {                     // bitonic sort, descending
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = key[i], b = key[l];
                    if (up ? (a < b) : (a > b)) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int p = threadIdx.x; p < n; p += blockDim.x) chunks2[map[p]] = tmp[CAP - 1 - (int)(unsigned)(key[p] & 0xffffffffull)];
}


// single block: dir_start = exclusive scan(counts); ceil(count / ch) equal chunks of <= ch voxels per orientation
// (ch2 > 0, a multiple of 64: a second list of chunks of exactly ch2 voxels (the last one of an orientation shorter) for the
//  lane-per-voxel kernels, count at n_chunks[1]; Chunk::pad = index of the chunk's first 64-voxel block in the block-wise
//  tables of amx_seed.hpp (blocks never straddle orientations), total number of blocks at n_chunks[2])
__global__ void k_plan(const int *__restrict__ counts, int ndirs, int ch, int *__restrict__ dir_start,
                       int *__restrict__ cursor, Chunk *__restrict__ chunks, int *__restrict__ n_chunks,
                       int ch2 = 0, Chunk *__restrict__ chunks2 = nullptr, int order = 0)
{
    __shared__ int s_off, s_chk, s_chk2, s_blk;
    __shared__ unsigned long long okey[kOrderCap];       // order_schunks' arrays; the scans' four int rows alias otmp
    __shared__ Chunk otmp[kOrderCap];
    __shared__ int omap[kOrderCap];
    int *sa = reinterpret_cast<int *>(otmp), *sb = sa + 1024, *sc2 = sb + 1024, *sd = sc2 + 1024;
    if (threadIdx.x == 0) { s_off = 0; s_chk = 0; s_chk2 = 0; s_blk = 0; }
    __syncthreads();
    // ndirs is small (500..32761): a serial scan by one thread per 1024-wide tile is enough
    for (int base = 0; base < ndirs; base += blockDim.x) {
        const int dsel = base + threadIdx.x;
        const int c = (dsel < ndirs) ? counts[dsel] : 0;
        const int nc = (c + ch - 1) / ch;
        const int nc2 = ch2 > 0 ? (c + ch2 - 1) / ch2 : 0;
        // block-wide exclusive scans through shared memory (Hillis-Steele on 3 values)
        const int nblk = ch2 > 0 ? (c + 63) / 64 : 0;
        sa[threadIdx.x] = c; sb[threadIdx.x] = nc; sc2[threadIdx.x] = nc2; sd[threadIdx.x] = nblk;
        __syncthreads();
        for (int off = 1; off < (int)blockDim.x; off <<= 1) {
            int ta = 0, tb = 0, tc = 0, td = 0;
            if ((int)threadIdx.x >= off) { ta = sa[threadIdx.x - off]; tb = sb[threadIdx.x - off]; tc = sc2[threadIdx.x - off]; td = sd[threadIdx.x - off]; }
            __syncthreads();
            sa[threadIdx.x] += ta; sb[threadIdx.x] += tb; sc2[threadIdx.x] += tc; sd[threadIdx.x] += td;
            __syncthreads();
        }
        const int start = s_off + sa[threadIdx.x] - c;
        const int cstart = s_chk + sb[threadIdx.x] - nc;
        const int cstart2 = s_chk2 + sc2[threadIdx.x] - nc2;
        const int bstart = s_blk + sd[threadIdx.x] - nblk;
        if (dsel < ndirs) {
            dir_start[dsel] = start;
            cursor[dsel] = 0;
            // nc chunks of equal size (+-1) instead of full chunks plus a short remainder: no workgroup stages a
            // tile for a handful of voxels
            const int base = nc ? c / nc : 0, rem = nc ? c - base * nc : 0;
            for (int k = 0; k < nc; k++) {
                Chunk ck;
                ck.dir = dsel; ck.start = start + k * base + (k < rem ? k : rem);
                ck.count = base + (k < rem ? 1 : 0); ck.pad = 0;
                chunks[cstart + k] = ck;
            }
            for (int k = 0; k < nc2; k++) {
                Chunk ck;
                ck.dir = dsel; ck.start = start + k * ch2;
                ck.count = (c - k * ch2 < ch2) ? c - k * ch2 : ch2; ck.pad = bstart + k * (ch2 / 64);
                chunks2[cstart2 + k] = ck;
            }
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) { s_off += sa[threadIdx.x]; s_chk += sb[threadIdx.x]; s_chk2 += sc2[threadIdx.x]; s_blk += sd[threadIdx.x]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dir_start[ndirs] = s_off; n_chunks[0] = s_chk; if (ch2 > 0) { n_chunks[1] = s_chk2; n_chunks[2] = s_blk; } }
    if (order && ch2 > 0) {
        __threadfence_block();
        __syncthreads();                       // the list this block wrote, read back by all its threads
        order_schunks(chunks2, s_chk2, okey, otmp, omap);
    }
}

// scatter of the voxel ids into their orientation's range.  With LDS: the block reserves, per orientation, one range
// for all its voxels (one global atomic per non-empty bin) and hands out the slots with LDS atomics.
__global__ __launch_bounds__(1024) void k_bucket(const int *__restrict__ lutidx, int n, int ndirs,
                                                 const int *__restrict__ dir_start, int *__restrict__ cursor,
                                                 int *__restrict__ perm, int use_lds, int span)
{
    extern __shared__ int sh[];
    int *cnt = sh, *base = sh + ndirs;
    const int v0 = blockIdx.x * span;
    const int v1 = (v0 + span < n) ? v0 + span : n;
    if (!use_lds) {
        for (int v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
            const int d = lutidx[v];
            if (d >= 0) perm[dir_start[d] + atomicAdd(&cursor[d], 1)] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < ndirs; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (int v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        const int d = lutidx[v];
        if (d >= 0) atomicAdd(&cnt[d], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ndirs; i += blockDim.x) {
        const int c = cnt[i];
        base[i] = c ? dir_start[i] + atomicAdd(&cursor[i], c) : 0;
        cnt[i] = 0;
    }
    __syncthreads();
    for (int v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        const int d = lutidx[v];
        if (d >= 0) perm[base[d] + atomicAdd(&cnt[d], 1)] = v;
    }
}

// contiguous chunks for models without orientations (SANDI)
__global__ void k_plan_linear(int n, int ch, Chunk *__restrict__ chunks, int *__restrict__ n_chunks,
                              int *__restrict__ perm)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) perm[v] = v;
    const int nc = (n + ch - 1) / ch;
    if (v < nc) {
        Chunk ck; ck.dir = 0; ck.start = v * ch; ck.count = (v == nc - 1) ? n - v * ch : ch; ck.pad = 0;
        chunks[v] = ck;
    }
    if (v == 0) *n_chunks = nc;
}

// ------------------------------------------------------------------ dictionary tiles
// out[dir][i][j], j < n_lut from src[j][dir][i]; then n_fix shared columns from fix[c][i]
// (iso / CSF, or a column of ones when fix_ones[c] != 0); remaining columns zero.
__global__ void k_build_lut(const float *__restrict__ src, const float *__restrict__ fix,
                            const int *__restrict__ fix_ones, int n_lut, int n_fix, int ndirs, int nS,
                            int ldA, int tile_stride, float *__restrict__ out)
{
    const size_t per = (size_t)nS * ldA, tot = (size_t)ndirs * per;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < tot; o += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(o % ldA);
        const int i = (int)((o / ldA) % nS);
        const int dsel = (int)(o / per);
        float val = 0.f;
        if (j < n_lut) val = src[((size_t)j * ndirs + dsel) * nS + i];
        else if (j < n_lut + n_fix) val = fix_ones[j - n_lut] ? 1.0f : fix[(size_t)(j - n_lut) * nS + i];
        out[(size_t)dsel * tile_stride + (o - (size_t)dsel * per)] = val;
    }
}


// Gram matrices of the orientation tiles: G[dir][j][k] = sum_{i in rows} A[i][j] A[i][k] in fp64
// (products of fp32 values are exact in fp64); one workgroup per orientation, tile staged in LDS.
// rowsel == nullptr: all rows.  Row stride ldG (>= n_atoms, padding stays zero).
// (in_lds == 0: a tile larger than a CU's LDS is read where it lies -- a one-off per dictionary upload, the L2 serves it)
__global__ void k_build_gram(const float *__restrict__ tiles, int tile_stride, int nS, int ldA, int n_atoms,
                             const unsigned char *__restrict__ rowsel, int ldG, double *__restrict__ G, int in_lds = 1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    float *As = reinterpret_cast<float *>(smem_g);
    const float *g = tiles + (size_t)blockIdx.x * tile_stride;
    if (in_lds) {
        for (int k = threadIdx.x; k < nS * ldA; k += blockDim.x) {
            const int i = k / ldA;
            As[k] = (rowsel == nullptr || rowsel[i]) ? g[k] : 0.f;
        }
        __syncthreads();
    }
    double *out = G + (size_t)blockIdx.x * n_atoms * ldG;
    for (int e = threadIdx.x; e < n_atoms * ldG; e += blockDim.x) {
        const int j = e / ldG, c = e % ldG;
        double acc = 0.0;
        if (c < n_atoms) {
            if (in_lds) { for (int i = 0; i < nS; i++) acc += (double)As[i * ldA + j] * (double)As[i * ldA + c]; }
            else { for (int i = 0; i < nS; i++) if (rowsel == nullptr || rowsel[i]) acc += (double)g[i * ldA + j] * (double)g[i * ldA + c]; }
        }
        out[e] = acc;
    }
}

}  // namespace amx
