"""LUT rotation / resampling (SURVEY section 8 f, row 4): host pipeline vs direct physics (CPU), GPU GEMM vs the numpy
statement of lut.pyx:274-311 (GPU)."""
import numpy as np
import pytest

from oracle import lut_np
from amico_amd import lut, synthetic as S


@pytest.fixture(scope='module')
def pipeline():
    scheme = S.make_scheme(n_b0=3, shells=((700.0, 20), (2000.0, 40)), seed=4)
    lut_dirs = S.fibonacci_hemisphere(60)
    aux = lut.aux_matrices(12, lut_dirs)
    idx_in, idx_o = lut.aux_structures_generate(scheme, 12)
    hr = lut.high_resolution_scheme(scheme, aux['grad'])
    vfs, ods = np.array([0.3, 0.7]), np.array([0.1, 0.5, 0.9])
    Kz = S.noddi_kernels(hr, np.array([[0.0, 0.0, 1.0]]), IC_VFs=vfs, IC_ODs=ods)          # fibre along z, 500 dirs / shell
    lm = np.stack([lut.rotate_kernel(Kz['wm'][a, 0].astype(np.float64), aux, idx_in, idx_o, False, len(lut_dirs))
                   for a in range(Kz['wm'].shape[0])])
    lm_iso = lut.rotate_kernel(Kz['iso'].astype(np.float64), aux, idx_in, idx_o, True, len(lut_dirs))
    idx_out, ylm_out = lut.aux_structures_resample(scheme, 12)
    direct = S.noddi_kernels(scheme, lut_dirs, IC_VFs=vfs, IC_ODs=ods)
    return dict(scheme=scheme, lut_dirs=lut_dirs, lm=lm, lm_iso=lm_iso, idx_out=idx_out, ylm_out=ylm_out, direct=direct,
                aux=aux, idx_in=idx_in, idx_o=idx_o, Kz=Kz)


def test_sh_basis_is_orthonormal_and_ordered():
    Y = lut.real_sh_even(12, lut.fibonacci_sphere(20000))
    assert Y.shape == (20000, 91)
    assert np.abs(Y.T @ Y * (4 * np.pi / 20000) - np.eye(91)).max() < 1e-4
    aux = lut.aux_matrices(4, S.fibonacci_hemisphere(10))
    assert aux['idx_m0'].tolist() == [0] + [3] * 5 + [10] * 9 and np.isclose(aux['const'][0], np.sqrt(4 * np.pi))
    pole = lut.real_sh_even(4, np.array([[0, 0, 1.0]]))[0]
    assert np.abs(np.delete(pole, [0, 3, 10])).max() < 1e-12            # only m = 0 survives on the axis


def test_rotate_and_resample_reproduce_the_physics(pipeline):
    """SH fit -> rotation by the addition theorem -> projection on the subject's gradients == the response function
    evaluated directly at those gradients for every LUT orientation (up to the lmax = 12 truncation)"""
    p = pipeline
    sc = p['scheme']
    n_dirs = len(p['lut_dirs'])
    for a in range(p['lm'].shape[0]):
        KR = lut_np.resample_kernel(p['lm'][a], sc.nS, p['idx_out'], p['ylm_out'], False, n_dirs)
        assert KR.dtype == np.float32 and np.all(KR[:, sc.b0_idx] == 1.0)
        assert np.abs(KR - p['direct']['wm'][a]).max() < 2e-3
    iso = lut_np.resample_kernel(p['lm_iso'], sc.nS, p['idx_out'], p['ylm_out'], True, n_dirs)
    assert np.abs(iso - p['direct']['iso']).max() < 1e-5


@pytest.mark.gpu
def test_resample_kernels_gemm_vs_numpy(pipeline):
    p = pipeline
    sc = p['scheme']
    n_dirs = len(p['lut_dirs'])
    got = lut.resample_kernels(p['lm'], sc.nS, p['idx_out'], p['ylm_out'])             # all atoms in one GEMM
    assert got.shape == p['lm'].shape[:2] + (sc.nS,) and got.dtype == np.float32
    for a in range(p['lm'].shape[0]):
        ref = lut_np.resample_kernel(p['lm'][a], sc.nS, p['idx_out'], p['ylm_out'], False, n_dirs)
        assert np.abs(got[a] - ref).max() < 2e-6
        assert np.array_equal(got[a][:, sc.b0_idx], ref[:, sc.b0_idx])
        one = lut.resample_kernel(p['lm'][a], sc.nS, p['idx_out'], p['ylm_out'], False, n_dirs)
        assert np.array_equal(one, got[a])
    iso = lut.resample_kernel(p['lm_iso'], sc.nS, p['idx_out'], p['ylm_out'], True, n_dirs)
    assert iso.shape == (sc.nS,) and np.abs(iso - lut_np.resample_kernel(p['lm_iso'], sc.nS, p['idx_out'], p['ylm_out'], True, n_dirs)).max() < 2e-6
    with pytest.raises(ValueError):
        lut.resample_kernel(p['lm'][0][:5], sc.nS, p['idx_out'], p['ylm_out'], False, n_dirs)
    with pytest.raises(ValueError):
        lut.resample_kernels(p['lm'][..., :50], sc.nS, p['idx_out'], p['ylm_out'])


@pytest.mark.gpu
def test_rotate_and_resample_fused(pipeline):
    """rotate_kernel + resample_kernel in ONE GEMM whose left operand (zonal factor x basis value) is formed in registers:
    equal to the two-step path up to float32 rounding of the rotated coefficients"""
    p = pipeline
    sc = p['scheme']
    Ks = [p['Kz']['wm'][a, 0].astype(np.float64) for a in range(p['Kz']['wm'].shape[0])]
    got = lut.rotate_and_resample(Ks, p['aux'], p['idx_in'], p['idx_o'], sc.nS, p['idx_out'], p['ylm_out'])
    lm = np.stack([lut.rotate_kernel(K, p['aux'], p['idx_in'], p['idx_o'], False, got.shape[1]) for K in Ks])
    ref = lut.resample_kernels(lm, sc.nS, p['idx_out'], p['ylm_out'])
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.abs(got - ref).max() < 2e-5
    assert np.array_equal(got[..., sc.b0_idx], np.ones_like(got[..., sc.b0_idx]))


@pytest.mark.gpu
def test_resample_random_shapes():
    rng = np.random.default_rng(0)
    for rows, k, n, nS in ((1, 7, 3, 5), (33, 91, 31, 40), (130, 455, 70, 77), (64, 182, 90, 99)):
        lm = rng.normal(size=(rows, k)).astype(np.float32)
        y = rng.normal(size=(n, k)).astype(np.float32)
        idx = np.sort(rng.choice(nS, n, replace=False)).astype(np.int32)
        got = lut.resample_kernels(lm, nS, idx, y)
        ref = np.ones((rows, nS), dtype=np.float32)
        ref[:, idx] = (lm.astype(np.float64) @ y.astype(np.float64).T).astype(np.float32)
        assert np.abs(got - ref).max() < 3e-6 * np.sqrt(k) * 4


@pytest.mark.gpu
def test_model_resample_builds_a_working_dictionary(tmp_path):
    """NODDI.resample / FreeWater.resample / SANDI.resample (GPU GEMM) from rotated SH coefficients: the dictionary
    matches the numpy statement atom by atom, files or arrays alike, and fits like the directly synthesised one"""
    import amico_amd
    from amico_amd import models
    from oracle import oracle
    scheme = S.make_scheme(n_b0=4, shells=((700.0, 24), (2000.0, 48)), seed=6)
    lut_dirs = S.fibonacci_hemisphere(500)
    ht = S.build_htable(lut_dirs)
    aux = lut.aux_matrices(12, lut_dirs)
    idx_in, idx_o = lut.aux_structures_generate(scheme, 12)
    hr = lut.high_resolution_scheme(scheme, aux['grad'])
    m = models.NODDI()
    m.set(IC_VFs=np.array([0.2, 0.5, 0.8]), IC_ODs=np.array([0.1, 0.4, 0.8]))
    m.scheme = scheme
    Kz = S.noddi_kernels(hr, np.array([[0.0, 0.0, 1.0]]), IC_VFs=m.IC_VFs, IC_ODs=m.IC_ODs)
    lms = [lut.rotate_kernel(Kz['wm'][a, 0].astype(np.float64), aux, idx_in, idx_o, False, 500) for a in range(9)]
    lms.append(lut.rotate_kernel(Kz['iso'].astype(np.float64), aux, idx_in, idx_o, True, 500))
    idx_out, ylm_out = lut.aux_structures_resample(scheme, 12)
    K = m.resample(lms, idx_out, ylm_out, False, 500)
    for i, a in enumerate(lms):
        np.save(tmp_path / f'A_{i + 1:03d}.npy', a)
    K2 = m.resample(str(tmp_path), idx_out, ylm_out, False, 500)
    assert all(np.array_equal(K[k], K2[k]) for k in ('wm', 'iso', 'norms', 'icvf', 'kappa'))
    for a in (0, 4, 8):
        ref = lut_np.resample_kernel(lms[a], scheme.nS, idx_out, ylm_out, False, 500)
        assert np.abs(K['wm'][a] - ref).max() < 2e-6
        assert np.allclose(K['norms'][:, a], 1 / np.linalg.norm(ref[0, scheme.dwi_idx]), rtol=1e-5)
    direct = S.noddi_kernels(scheme, lut_dirs, IC_VFs=m.IC_VFs, IC_ODs=m.IC_ODs)
    assert np.abs(K['wm'] - direct['wm']).max() < 2e-3 and np.allclose(K['icvf'], direct['icvf']) \
        and np.allclose(K['kappa'], direct['kappa'], rtol=1e-6)
    Km = m.resample(lms, idx_out, ylm_out, True, 500)
    assert Km['wm'].shape == (9, 500, 1 + scheme.dwi_count) and np.all(Km['wm'][:, :, 0] == 1.0)
    # the resampled dictionary drives the fit: same maps as the oracle on the same dictionary
    y, d = S.noddi_signals(3000, K, ht, scheme, seed=2)

    class Ev:
        pass
    ev = Ev()
    ev.y, ev.DIRs, ev.htable, ev.KERNELS, ev.nthreads = y, d, ht, K, 1
    ev.get_config = lambda k: False
    est = m.fit(ev)['estimates']
    ref = oracle.noddi_fit(y, d, K, ht, scheme.dwi_idx, nthreads=8)['estimates']
    assert np.abs(est - ref).max() < 1e-6
    # isotropic-only model (SANDI) and tensor model (FreeWater) through the same GEMM
    sm = models.SANDI()
    sm.scheme = scheme
    iso_lm = [lms[-1]] * (len(sm.Rs) + len(sm.d_in) + len(sm.d_isos))
    Ks = sm.resample(iso_lm, idx_out, ylm_out, False, 500)
    assert Ks['signal'].shape == (scheme.nS, 15) and np.allclose(np.linalg.norm(Ks['signal'], axis=0), 1.0)
    fm = models.FreeWater()
    fm.set(d_par=1.0e-3, d_perps=np.linspace(0.1e-3, 1.0e-3, 9), d_isos=[2.5e-3])
    fm.scheme = scheme
    Kf = fm.resample(lms[:9] + [lms[-1]], idx_out, ylm_out, False, 500)
    assert Kf['D'].shape == (9, 500, scheme.nS) and Kf['CSF'].shape == (1, scheme.nS)
    assert np.array_equal(Kf['D'], K['wm']) and np.array_equal(Kf['CSF'][0], K['iso'])
