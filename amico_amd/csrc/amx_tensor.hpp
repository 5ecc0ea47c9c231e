// amx_tensor.hpp -- the arithmetic of the log-linear tensor fit (core.py:431-436, 456-458): logarithm, 3 x 3 Jacobi eigen-solver.
// Shared by k_dti_dirs (amx_signal.hip) and by the signal preparation kernel that takes the tensor fit along while the voxel's
// values are in LDS (k_prep_gather<.., DIRS>, amx_volume.hip).
#pragma once

namespace amx {

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

}  // namespace amx
