// layout probe of v_mfma_f32_16x16x4_f32: D[i][j] = i + 1 + 100 j  ->  which (i, j) does register r of lane l hold?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *out)
{
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    const float a = (kk == 0) ? (float)(i + 1) : ((kk == 1) ? 1.0f : 0.0f);       // A[i][0] = i + 1, A[i][1] = 1
    const float b = (kk == 0) ? 1.0f : ((kk == 1) ? 100.0f * (float)(l & 15) : 0.0f);   // B[0][j] = 1, B[1][j] = 100 j
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main()
{
    float *d; hipMalloc(&d, 256 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 16, 17, 32, 63}) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
