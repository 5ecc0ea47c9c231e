// Calibration of rocprofv3's HBM traffic counters on gfx950 against KNOWN byte counts, in the access patterns of this library's kernels
// (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/counter_probe tools/probes/counter_probe.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/p1 -- /tmp/counter_probe        (one counter set per run: tools/r06/calib.sh)
//
// Every kernel touches every one of its bytes / sectors ONCE, on a buffer far beyond the L2s (32 MiB) and the Infinity Cache (256 MiB), so
// each access is a miss all the way to HBM; the program prints the known counts, tools/r06/calib_summary.py puts the counters beside them.
//   k_read_wide16    16 B per lane, coalesced stream                               (the guide's calibrated pattern)
//   k_read_wide8     8 B per lane, coalesced (a wavefront reads 512 B in a row)    (table rows read by 64 consecutive voxels)
//   k_read_sector8   8 B per lane, ONE per 64-byte sector, adjacent sectors         (stride 64 B across the lanes)
//   k_read_row8      8 B per lane, lane l reads row r_l of a 176 x 512 B block     (the certificates' reads of the A'y table: k_nnls_gcert,
//                    at column l: sectors 512 B apart, rows scattered               k_lasso_gcert -- Crow[idx * 64], one row per lane)
//   k_read_line8     8 B per lane, ONE per 128-byte line (the other sector of the line is never read); k_read_row512: one per 512 B
//   k_read_planes    96 concurrent streams of 256-byte runs per wavefront, planes 22 MB apart (the mask gather's reads of a planar image)
//   k_read_lds4/16   global -> LDS direct loads (global_load_lds dword / dwordx4), coalesced                     (k_prep_gather, k_freewater_fused)
//   k_write_wide16   16 B per lane coalesced stores;  k_write_sector8: 8 B per lane, one per 64-byte sector      (WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_read_wide16(const uint4 *__restrict__ p, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345679u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_read_wide8(const double *__restrict__ p, size_t n8, unsigned *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e300) *sink = 1;
}
// one 8-byte load per 64-byte sector: lane -> sector (i), adjacent lanes adjacent sectors
__global__ void __launch_bounds__(256) k_read_sector8(const double *__restrict__ p, size_t n_sectors, unsigned *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sectors; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 8];
    if (acc == 1.2345e300) *sink = 1;
}
// one 8-byte load per 128-byte LINE (every other sector is never touched): does a lone access fetch 64 or 128 bytes?
__global__ void __launch_bounds__(256) k_read_line8(const double *__restrict__ p, size_t n_lines, unsigned *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 16];
    if (acc == 1.2345e300) *sink = 1;
}
// ... and one per 512 bytes, i.e. per row of a table block (a lane alone on its row)
__global__ void __launch_bounds__(256) k_read_row512(const double *__restrict__ p, size_t n_rows, unsigned *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 64 + (i & 63)];
    if (acc == 1.2345e300) *sink = 1;
}
// the mask gather's read pattern (k_prep_gather on a planar image): a wavefront reads ONE 256-byte run from each of NV planes that lie `plane`
// bytes apart (a tile of 64 voxels x NV volumes), all NV loads in flight together -- NV concurrent streams per wavefront, every request in another
// DRAM row.  Every byte of the buffer is read once, in full lines.
template <int NV>
__global__ void __launch_bounds__(256) k_read_planes(const float *__restrict__ p, size_t plane_el, size_t tiles, unsigned *sink)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += (size_t)gridDim.x * 4) {
        float v[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = p[(size_t)k * plane_el + t * 64 + lane];
#pragma unroll
        for (int k = 0; k < NV; k++) acc += v[k];
    }
    if (acc == 1.2345e30f) *sink = 1;
}
// the certificates' pattern: blocks of ROWS x 64 doubles (one block = 64 voxels of the A'y table, one 512-byte row per atom); a lane reads
// the entries of ITS voxel's atoms, i.e. 8 bytes of a row that the lanes beside it do not read -- sectors 512 B apart, 8 useful bytes of 64.
template <int ROWS>
__global__ void __launch_bounds__(256) k_read_row8(const double *__restrict__ p, size_t n_blocks, unsigned *sink)
{
    // every sector of a block exactly ONCE: pass t = 0 .. ROWS / 8 - 1; lane l = 8 k + j reads row 8 t + j at column 8 k + ((j + t) & 7):
    // the eight lanes 8 k .. 8 k + 7 read eight DIFFERENT rows (their 8-byte pieces lie 512 B apart), the eight groups k read eight different
    // sectors of each of those rows; over the passes every (row, k) sector is read once, 8 useful bytes of its 64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 3, j = lane & 7;
    double acc = 0.0;
    for (size_t b = (size_t)blockIdx.x * 4 + wave; b < n_blocks; b += (size_t)gridDim.x * 4) {
        const double *blk = p + b * (size_t)ROWS * 64;
#pragma unroll 4
        for (int t = 0; t < ROWS / 8; t++) acc += blk[(size_t)(8 * t + j) * 64 + 8 * k + ((j + t) & 7)];
    }
    if (acc == 1.2345e300) *sink = 1;
}
// global -> LDS direct loads (no VGPR in between): one dword / four dwords per lane, coalesced
template <int BYTES>
__global__ void __launch_bounds__(256) k_read_lds(const unsigned *__restrict__ p, size_t n4, unsigned *sink)
{
    __shared__ __attribute__((aligned(16))) unsigned tile[256 * (BYTES / 4)];
    const size_t per = (size_t)blockDim.x * (BYTES / 4);
    unsigned acc = 0;
    for (size_t base = (size_t)blockIdx.x * per; base + per <= n4; base += (size_t)gridDim.x * per) {
        const unsigned *src = p + base + (size_t)threadIdx.x * (BYTES / 4);
        // M0 base = the wavefront's slice of the tile; the instruction writes lane l's BYTES bytes at LDS[base + l * BYTES]
        unsigned *dst = tile + (threadIdx.x & ~63u) * (BYTES / 4);
        if constexpr (BYTES == 4) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
        else __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += tile[threadIdx.x * (BYTES / 4)];
    }
    if (acc == 0x12345679u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_write_wide16(uint4 *__restrict__ p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_write_sector8(double *__restrict__ p, size_t n_sectors)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sectors; i += (size_t)gridDim.x * blockDim.x) p[i * 8] = (double)i;
}

int main(int argc, char **argv)
{
    const size_t GiB = (size_t)1 << 30;
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2) * GiB;     // >> 256 MiB of Infinity Cache
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    char *buf; unsigned *sink;
    CK(hipMalloc((void **)&buf, bytes + 4096));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), blk(256);
    constexpr int ROWS = 176;
    const size_t n_blocks = bytes / ((size_t)ROWS * 512);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double useful, double sectors64, auto launch) {
        float best = 1e30f;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, nullptr)); launch(); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("PROBE %-18s useful_bytes %.0f  sector_bytes_64B %.0f  launches %d  best_ms %.3f  useful_GBps %.1f  sector_GBps %.1f\n", name, useful, sectors64, reps, best,
               useful / best / 1e6, sectors64 / best / 1e6);
    };
    timed("k_read_wide16", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(k_read_wide16, grid, blk, 0, nullptr, (const uint4 *)buf, bytes / 16, sink); });
    timed("k_read_wide8", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(k_read_wide8, grid, blk, 0, nullptr, (const double *)buf, bytes / 8, sink); });
    timed("k_read_sector8", (double)bytes / 8, (double)bytes, [&] { hipLaunchKernelGGL(k_read_sector8, grid, blk, 0, nullptr, (const double *)buf, bytes / 64, sink); });
    timed("k_read_row8", (double)n_blocks * ROWS * 64, (double)n_blocks * ROWS * 512, [&] { hipLaunchKernelGGL(k_read_row8<ROWS>, grid, blk, 0, nullptr, (const double *)buf, n_blocks, sink); });
    timed("k_read_line8", (double)bytes / 16, (double)bytes / 2, [&] { hipLaunchKernelGGL(k_read_line8, grid, blk, 0, nullptr, (const double *)buf, bytes / 128, sink); });
    timed("k_read_row512", (double)bytes / 64, (double)bytes / 8, [&] { hipLaunchKernelGGL(k_read_row512, grid, blk, 0, nullptr, (const double *)buf, bytes / 512, sink); });
    {
        constexpr int NV = 96;
        const size_t plane_el = bytes / 4 / NV, tiles = plane_el / 64;
        timed("k_read_planes<96>", (double)(tiles * 64 * NV * 4), (double)(tiles * 64 * NV * 4), [&] { hipLaunchKernelGGL(k_read_planes<NV>, grid, blk, 0, nullptr, (const float *)buf, plane_el, tiles, sink); });
    }
    timed("k_read_lds<4>", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(k_read_lds<4>, grid, blk, 0, nullptr, (const unsigned *)buf, bytes / 4, sink); });
    timed("k_read_lds<16>", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(k_read_lds<16>, grid, blk, 0, nullptr, (const unsigned *)buf, bytes / 4, sink); });
    timed("k_write_wide16", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(k_write_wide16, grid, blk, 0, nullptr, (uint4 *)buf, bytes / 16); });
    timed("k_write_sector8", (double)bytes / 8, (double)bytes, [&] { hipLaunchKernelGGL(k_write_sector8, grid, blk, 0, nullptr, (double *)buf, bytes / 64); });
    CK(hipDeviceSynchronize());
    return 0;
}
