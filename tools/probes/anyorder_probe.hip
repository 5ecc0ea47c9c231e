// Does hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM is still running on gfx950?
// k_wait (one workgroup) spins until k_flag's store arrives or 20 ms have passed; k_flag is launched after it, with and without
// the flag.  "saw the flag after N us" with the any-order launch and a time-out without it = the barrier bit is really dropped.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_wait(int *flag, long long *out)
{
    const long long t0 = wall_clock64();
    long long t = t0;
    int seen = 0;
    while (t - t0 < 2000000) {                     // 100 MHz constant clock: 20 ms
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(16);
        t = wall_clock64();
    }
    if (threadIdx.x == 0) { out[0] = seen; out[1] = t - t0; }
}
__global__ void k_flag(int *flag) { if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main()
{
    int *flag; long long *out;
    hipMalloc(&flag, sizeof(int)); hipMalloc(&out, 2 * sizeof(long long));
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; mode++) {
        hipMemsetAsync(flag, 0, sizeof(int), s);
        hipStreamSynchronize(s);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, flag, out);
        if (mode == 0) hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, flag);
        else hipExtLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
        hipStreamSynchronize(s);
        long long h[2]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        printf("%s launch of the second kernel: first kernel %s after %.1f us\n", mode ? "any-order" : "ordinary", h[0] ? "saw the flag" : "timed out", h[1] / 100.0);
    }
    return 0;
}
