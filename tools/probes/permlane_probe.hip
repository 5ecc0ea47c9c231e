// probe: lane mapping of v_permlane16_swap / v_permlane32_swap on gfx950 (diagnosis only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(v, v + 100, false, false);
    auto r2 = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = r2[0]; o[192 + threadIdx.x] = r2[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]"};
    for (int a = 0; a < 4; a++) { printf("%s:", names[a]); for (int i = 0; i < 64; i++) printf(" %u", h[a * 64 + i]); printf("\n"); }
    return 0;
}
