"""Rotation of the response functions in spherical-harmonic space and resampling to the subject's scheme
(SURVEY section 8 f, row 4) -- host mirror of the functions of amico/lut.pyx the kernel pipeline uses:

    precompute_rotation_matrices  (lut.pyx:94-141)   -> aux_matrices           host numpy, one-off per (lmax, ndirs)
    aux_structures_generate       (lut.pyx:171-193)  -> aux_structures_generate
    aux_structures_resample       (lut.pyx:196-224)  -> aux_structures_resample host numpy, one-off per scheme
    rotate_kernel                 (lut.pyx:227-271)  -> rotate_kernel           host numpy, one-off per protocol;
                                                         rotate_and_resample     GPU: fused with the resampling GEMM
    resample_kernel               (lut.pyx:274-311)  -> resample_kernels        GPU: one float32 GEMM for ALL atoms
                                                                                (amx_lut_resample, include/amico_amd.h)

The reference takes its real, even-order SH basis from dipy (`real_sh_descoteaux`, absent here) and its 500 high
resolution gradient directions from a table inside lut.pyx.  The pipeline's OUTPUT (the rotated, resampled LUT)
does not depend on either choice: any orthonormal real SH basis satisfies the addition theorem the rotation uses,
and any well-spread direction set supports the least-squares SH fit.  This module therefore uses its own basis
(scipy's complex harmonics, same (l, m) ordering: l = 0, 2, .. lmax, m = -l .. l) and a Fibonacci sphere.
"""
import numpy as np
from scipy import special

from . import _capi

HR_DIRS = 500          # samples per shell of the high-resolution response functions (lut.pyx:359-384)


def n_sh(lmax):
    return (lmax + 1) * (lmax + 2) // 2


def fibonacci_sphere(n):
    i = np.arange(n) + 0.5
    z = 1.0 - 2.0 * i / n
    phi = np.pi * (1.0 + 5.0 ** 0.5) * i
    s = np.sqrt(1.0 - z * z)
    return np.column_stack([s * np.cos(phi), s * np.sin(phi), z])


def real_sh_even(lmax, dirs):
    """[n_dirs, nSH] real orthonormal harmonics of even order: sqrt(2) Re Y_l^|m| (m < 0), Y_l^0, sqrt(2) Im Y_l^m (m > 0)"""
    d = np.asarray(dirs, dtype=np.float64)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    polar = np.arccos(np.clip(d[:, 2], -1.0, 1.0))
    azim = np.arctan2(d[:, 1], d[:, 0])
    out = np.zeros((d.shape[0], n_sh(lmax)))
    k = 0
    for l in range(0, lmax + 1, 2):
        for m in range(-l, l + 1):
            y = special.sph_harm_y(l, abs(m), polar, azim)
            out[:, k] = y.real if m == 0 else (np.sqrt(2.0) * y.real if m < 0 else np.sqrt(2.0) * y.imag)
            k += 1
    return out


def aux_matrices(lmax, lut_dirs, hr_dirs=None):
    """what precompute_rotation_matrices stores (lut.pyx:112-139): 'fit' (signal -> SH), 'Ylm_rot' (one row of basis
    values per LUT orientation), 'const' = sqrt(4 pi / (2l + 1)) and 'idx_m0' (position of m = 0) per coefficient"""
    hr = fibonacci_sphere(HR_DIRS) if hr_dirs is None else np.asarray(hr_dirs, dtype=np.float64)
    Y = real_sh_even(lmax, hr)
    aux = {'lmax': lmax, 'ndirs': len(lut_dirs), 'grad': hr}
    aux['fit'] = np.dot(np.linalg.pinv(np.dot(Y.T, Y)), Y.T)
    aux['Ylm_rot'] = real_sh_even(lmax, lut_dirs)
    const = np.zeros(n_sh(lmax))
    idx_m0 = np.zeros(n_sh(lmax), dtype=np.int32)
    i = 0
    for l in range(0, lmax + 1, 2):
        for _ in range(-l, l + 1):
            const[i] = np.sqrt(4.0 * np.pi / (2.0 * l + 1.0))
            idx_m0[i] = (l * l + l + 2) // 2 - 1
            i += 1
    aux['const'], aux['idx_m0'] = const, idx_m0
    return aux


def aux_structures_generate(scheme, lmax=12):
    nsh = n_sh(lmax)
    n = len(scheme.shells)
    return ([range(HR_DIRS * s, HR_DIRS * (s + 1)) for s in range(n)], [range(nsh * s, nsh * (s + 1)) for s in range(n)])


def aux_structures_resample(scheme, lmax=12):
    """(idx_OUT int32[dwi_count], Ylm_OUT f32[dwi_count, nSH * n_shells]): block-diagonal SH -> signal operator"""
    nsh = n_sh(lmax)
    shells = scheme.shells
    idx_out = np.zeros(scheme.dwi_count, dtype=np.int32)
    ylm_out = np.zeros((scheme.dwi_count, nsh * len(shells)), dtype=np.float32)
    pos = 0
    for s, sh in enumerate(shells):
        n = len(sh['idx'])
        idx_out[pos:pos + n] = sh['idx']
        ylm_out[pos:pos + n, nsh * s:nsh * (s + 1)] = real_sh_even(lmax, sh['grad'])
        pos += n
    return idx_out, ylm_out


def rotate_kernel(K, aux, idx_in, idx_out, is_isotropic, ndirs):
    """response function sampled on the high-resolution shells (symmetric about z) -> SH coefficients of its rotation
    to every LUT orientation, float32 [ndirs, nSH * n_shells] ([nSH * n_shells] if isotropic)"""
    klm = [np.dot(aux['fit'], K[list(idx_in[s])]) for s in range(len(idx_in))]
    n = len(idx_in) * aux['fit'].shape[0]
    if is_isotropic:
        out = np.zeros(n, dtype=np.float32)
        for s in range(len(idx_in)):
            out[list(idx_out[s])] = klm[s].astype(np.float32)
        return out
    out = np.zeros((ndirs, n), dtype=np.float32)
    for s in range(len(idx_in)):
        zonal = aux['const'] * klm[s][aux['idx_m0']]                    # addition theorem, one factor per (l, m)
        out[:, list(idx_out[s])] = zonal[None, :] * aux['Ylm_rot']
    return out


def zonal_factors(K, aux, idx_in, idx_out):
    """the part of rotate_kernel that depends on the response function only: per shell, const * Klm[idx_m0] with
    Klm = fit . K[shell] (lut.pyx:249-251, 262-264) -> float64 [nSH * n_shells]"""
    out = np.zeros(len(idx_in) * aux['fit'].shape[0])
    for s in range(len(idx_in)):
        klm = np.dot(aux['fit'], K[list(idx_in[s])])
        out[list(idx_out[s])] = aux['const'] * klm[aux['idx_m0']]
    return out


def rotate_and_resample(Ks, aux, idx_in, idx_sh, nS, idx_out, ylm_out, ctx=None):
    """rotate_kernel (lut.pyx:227-271) + resample_kernel (:274-311) for a list of anisotropic response functions Ks (each
    sampled on the high-resolution shells, symmetric about z) WITHOUT materialising the rotated SH coefficients: the GPU
    GEMM forms them in registers (amx_lut_rotate_resample).  -> float32 [n_atoms, ndirs, nS]"""
    from .models import get_context
    Z = np.stack([zonal_factors(np.asarray(K, dtype=np.float64), aux, idx_in, idx_sh) for K in Ks]).astype(np.float32)
    return _capi.lut_rotate_resample(ctx if ctx is not None else get_context(), Z, aux['Ylm_rot'], ylm_out, idx_out, nS)


def resample_kernels(lm, nS, idx_out, ylm_out, ctx=None):
    """lm f32[..., nSH * n_shells] (any leading shape: [n_atoms, ndirs, :], [ndirs, :] or [:]) -> f32[..., nS]; the
    entries idx_out of the last axis get dot(Ylm_out, lm[...]), the others (the b0 volumes) stay 1 -- on the GPU"""
    from .models import get_context
    return _capi.lut_resample(ctx if ctx is not None else get_context(), lm, ylm_out, idx_out, nS)


def resample_kernel(KRlm, nS, idx_out, Ylm_out, is_isotropic, ndirs, ctx=None):
    """signature of lut.pyx:274 for one atom"""
    lm = np.asarray(KRlm, dtype=np.float32)
    if (lm.ndim != 1) if is_isotropic else (lm.ndim != 2 or lm.shape[0] != ndirs):
        raise ValueError('Outdated LUT. Call "generate_kernels( regenerate=True )" to update the LUT')
    return resample_kernels(lm, nS, idx_out, Ylm_out, ctx)


def high_resolution_scheme(scheme, hr_dirs):
    """create_high_resolution_scheme (lut.pyx:359-384): HR_DIRS directions per shell, acquisition parameters unchanged"""
    from .synthetic import SimpleScheme
    shells = scheme.shells
    raw = np.zeros((len(hr_dirs) * len(shells), 4 if scheme.version == 0 else 7))
    for i, sh in enumerate(shells):
        blk = slice(len(hr_dirs) * i, len(hr_dirs) * (i + 1))
        raw[blk, 0:3] = hr_dirs
        if scheme.version == 0:
            raw[blk, 3] = sh['b']
        else:
            raw[blk, 3:7] = [sh['G'], sh['Delta'], sh['delta'], sh['TE']]
    return SimpleScheme(raw)
