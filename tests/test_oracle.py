"""CPU tests: the oracle (oracle/amico_oracle.c) against golden fixtures, third-party
solvers and KKT certificates.  No GPU needed."""
import numpy as np
import pytest
from scipy.optimize import nnls as sp_nnls

from oracle import oracle
from amico_amd import synthetic as S


# ------------------------------------------------------------------ dir_to_lut_idx (lut.pyx:316-356)
def test_htable_construction_rule(htable500):
    """the reference's own data file follows argmax |v.d| -> pins build_htable()"""
    assert np.array_equal(S.build_htable(htable500['dirs']), htable500['htable'])


def test_dir_to_lut_idx_hand_cases(htable500):
    ht = htable500['htable']
    cases = {
        (0, 0, 1): (0, 0), (0, 0, -1): (180, 0), (1, 0, 0): (90, 0), (-1, 0, 0): (90, 180),
        (0, 1, 0): (90, 90), (0, -1, 0): (90, 90), (0, 0, 0): (0, 0),
        (1, 1, 0): (90, 45), (-1, -1, 0): (90, 45), (1, -1, 0): (90, 135),
        (0, 1e-30, 1): (0, 90), (3, 0, 4): (37, 0),
    }
    for d, (e1, e2) in cases.items():
        idx, i1, i2 = oracle.dir_to_lut_idx(np.array(d, dtype=float), ht)
        assert (i1[0], i2[0]) == (e1, e2), (d, i1, i2)
        assert idx[0] == ht[e1 * 181 + e2]
    # rounding boundary: phi = 179.5 deg rounds half away from zero -> 180
    phi = np.deg2rad(179.5)
    idx, i1, i2 = oracle.dir_to_lut_idx(np.array([np.cos(phi), np.sin(phi), 0.0]), ht)
    assert i1[0] == 90 and i2[0] in (179, 180)
    # scale invariance and NaN
    a = oracle.dir_to_lut_idx(np.array([0.3, 0.5, 0.8]), ht)[0]
    b = oracle.dir_to_lut_idx(1e-9 * np.array([0.3, 0.5, 0.8]), ht)[0]
    assert a[0] == b[0]
    assert oracle.dir_to_lut_idx(np.array([np.nan, 0.0, 1.0]), ht)[0][0] == -1


def test_dir_to_lut_idx_is_nearest_direction(htable500):
    rng = np.random.default_rng(7)
    d = S.random_unit_vectors(2000, rng)
    idx, _, _ = oracle.dir_to_lut_idx(d, htable500['htable'])
    assert np.array_equal(idx, S.lut_indices(d, htable500['htable']))
    # the chosen LUT direction is within ~1 degree of the best one
    dots = np.abs(d @ htable500['dirs'].T)
    assert np.all(dots[np.arange(len(d)), idx] >= dots.max(axis=1) - 2e-3)
    # input must not be modified (the reference flips in place; we do not)
    d0 = d.copy()
    oracle.dir_to_lut_idx(d, htable500['htable'])
    assert np.array_equal(d, d0)


# ------------------------------------------------------------------ solver primitives
def kkt_nnls(A, y, x):
    w = A.T @ (y - A @ x)
    on = np.abs(w[x > 0]).max() if (x > 0).any() else 0.0
    off = max(0.0, w[x <= 0].max()) if (x <= 0).any() else 0.0
    return on, off


def kkt_enet(A, y, x, l1, l2):
    g = A.T @ (y - A @ x) - l2 * x - l1
    on = np.abs(g[x > 0]).max() if (x > 0).any() else 0.0
    off = max(0.0, g[x <= 0].max()) if (x <= 0).any() else 0.0
    return on, off


def test_nnls_random_vs_scipy():
    rng = np.random.default_rng(0)
    for (m, n) in [(20, 5), (6, 15), (99, 30), (50, 50), (10, 1)]:
        for _ in range(5):
            A = rng.standard_normal((m, n)); y = rng.standard_normal(m)
            x, rn, mode = oracle.nnls(A, y)
            xs, rs = sp_nnls(A, y)
            assert mode == 1
            assert np.allclose(x, xs, atol=1e-10)
            assert abs(rn - rs) < 1e-10
            assert (x >= 0).all()


def test_nnls_untouched_inputs_and_leading_columns():
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.random((12, 7))); y = rng.random(12)
    A0, y0 = A.copy(), y.copy()
    x, _, _ = oracle.nnls(A, y)
    assert np.array_equal(A, A0) and np.array_equal(y, y0)
    # n smaller than the allocated columns (models.pyx:940 passes positive_count)
    x4, _, _ = oracle.nnls(A[:, :4], y)
    assert np.allclose(x4, sp_nnls(A[:, :4], y)[0], atol=1e-12)


def test_nnls_noddi_dictionary(noddi_fix):
    f = noddi_fix
    for i in range(0, 160, 7):
        slot = np.searchsorted(f['lut_ids'], f['lut'][i])
        A = np.hstack([f['wm_slices'][:, slot, :].T.astype(float), f['iso'][:, None].astype(float)])
        x, _, mode = oracle.nnls(A, f['y'][i])
        assert mode == 1
        on, off = kkt_nnls(A, f['y'][i], x)
        assert on < 1e-11 and off < 1e-11
        assert np.allclose(x, f['x_stages'][i, 0], atol=1e-8)


def test_lasso_vs_golden_and_kkt(noddi_fix):
    f = noddi_fix
    l1, l2 = float(f['lambda1']), float(f['lambda2'])
    for i in range(0, 160, 5):
        slot = np.searchsorted(f['lut_ids'], f['lut'][i])
        A = f['wm_slices'][:, slot, :].T.astype(float)
        A2 = A[f['dwi_idx']] * f['norms']
        xiso = f['x_stages'][i, 0, -1]
        y2 = np.maximum(0.0, f['y'][i, f['dwi_idx']] - xiso * f['iso'][f['dwi_idx']])
        x, st = oracle.lasso(A2, y2, l1, l2)
        assert st == 0
        on, off = kkt_enet(A2, y2, x, l1, l2)
        assert on < 1e-10 and off < 1e-10
        assert np.allclose(x, f['x_stages'][i, 1, :144], atol=1e-8)


def test_lasso_small_random_vs_sklearn():
    from sklearn.linear_model import ElasticNet
    rng = np.random.default_rng(3)
    for (m, n, l1, l2) in [(30, 10, 0.3, 0.05), (8, 20, 0.1, 0.2), (40, 12, 0.0, 0.01)]:
        A = rng.random((m, n)); y = rng.random(m) * 3
        x, st = oracle.lasso(A, y, l1, l2)
        assert st == 0
        on, off = kkt_enet(A, y, x, l1, l2)
        assert on < 1e-10 and off < 1e-10
        if l1 > 0:
            en = ElasticNet(alpha=(l1 + l2) / m, l1_ratio=l1 / (l1 + l2), positive=True,
                            fit_intercept=False, tol=1e-13, max_iter=500000).fit(A, y)
            assert np.allclose(x, en.coef_, atol=1e-7)
    # all correlations below lambda1 -> zero solution
    x, st = oracle.lasso(rng.random((5, 4)), np.zeros(5), 0.5, 0.1)
    assert st == 0 and not x.any()


# ------------------------------------------------------------------ model glue vs golden fixtures
def test_noddi_fit_vs_golden(noddi_fix, htable500):
    f = noddi_fix
    out = oracle.noddi_fit(f['y'], f['dirs'], f['kernels'], htable500['htable'], f['dwi_idx'],
                           float(f['lambda1']), float(f['lambda2']), rmse=True, nrmse=True, mod=True,
                           return_x=True)
    assert out['err'] == 0
    assert np.allclose(out['x'], f['x_stages'], atol=1e-7)
    assert np.allclose(out['estimates'], f['estimates'], rtol=1e-7, atol=1e-9)
    assert np.allclose(out['rmse'], f['rmse'], atol=1e-10)
    assert np.allclose(out['nrmse'], f['nrmse'], atol=1e-10)
    tf = 1 - out['estimates'][:, 2]
    assert np.allclose(out['estimates_mod'], out['estimates'][:, :2] * tf[:, None], atol=1e-15)
    # hand-checkable voxels: all-zero signal -> NDI 0, ODI 1, FWF 0 ; pure iso -> FWF 1
    assert np.allclose(out['estimates'][1], [0.0, 1.0, 0.0], atol=1e-12)
    assert abs(out['estimates'][2, 2] - 1.0) < 1e-9
    assert out['nrmse'][1] == 0.0


def test_noddi_threads_and_dirs_const(noddi_fix, htable500):
    f = noddi_fix
    d0 = f['dirs'].copy()
    a = oracle.noddi_fit(f['y'], f['dirs'], f['kernels'], htable500['htable'], f['dwi_idx'], nthreads=1)
    b = oracle.noddi_fit(f['y'], f['dirs'], f['kernels'], htable500['htable'], f['dwi_idx'], nthreads=7)
    assert np.array_equal(a['estimates'], b['estimates'])
    assert np.array_equal(f['dirs'], d0)
    # flipped directions are the same axial orientation
    c = oracle.noddi_fit(f['y'], -f['dirs'], f['kernels'], htable500['htable'], f['dwi_idx'])
    assert np.array_equal(a['estimates'], c['estimates'])


def test_noddi_oob_direction_reports_voxel(noddi_fix, htable500):
    f = noddi_fix
    d = f['dirs'][:5].copy(); d[3] = np.nan
    out = oracle.noddi_fit(f['y'][:5], d, f['kernels'], htable500['htable'], f['dwi_idx'])
    assert out['err'] == -4


def test_freewater_fit_vs_golden(fw_fix, htable500):
    f = fw_fix
    out = oracle.freewater_fit(f['y'], f['dirs'], f['kernels'], htable500['htable'], 0.0, 1e-3,
                               corrected=True, return_x=True, rmse=True)
    assert np.allclose(out['x'], f['x'], atol=1e-8)
    assert np.allclose(out['estimates'], f['estimates'], atol=1e-8)
    assert np.allclose(out['y_corrected'], f['y_corrected'], atol=1e-8)
    assert np.allclose(out['estimates'][0], [0.0, 1.0])     # all-zero voxel


def test_sandi_fit_vs_golden(sandi_fix):
    f = sandi_fix
    out = oracle.sandi_fit(f['y'], f['kernels'], f['Rs'], f['d_in'], f['d_isos'], 0.0, 5e-3,
                           return_x=True, rmse=True, nrmse=True)
    assert np.allclose(out['x'], f['x'], atol=1e-8)
    assert np.allclose(out['estimates'], f['estimates'], rtol=1e-7, atol=1e-7)
    assert np.allclose(out['estimates'][0], 0.0)


def test_czb_oracle_matches_fixture(czb_fix, htable500):
    """CylinderZeppelinBall._fit restated (models.pyx:526-652) vs the fixture (scipy NNLS on the augmented system)"""
    from oracle import oracle
    f = czb_fix
    o = oracle.czb_fit(f['y'], f['dirs'], f['kernels'], f['Rs'], htable500['htable'], float(f['lambda1']), float(f['lambda2']),
                       rmse=True, nthreads=3, return_x=True)
    assert o['err'] == 0
    assert np.abs(o['x'] - f['x']).max() < 1e-10
    assert np.abs(o['estimates'] - f['estimates']).max() < 1e-9
    assert np.abs(o['rmse'] - f['rmse']).max() < 1e-12
    assert np.allclose(o['estimates'][0], 0.0)                      # all-zero voxel
    # KKT certificate of the non-negative ridge (lambda1 = 0): g = A'(y - A x) - lambda2 x
    K, lut = f['kernels'], f['lut']
    for v in range(0, len(lut), 9):
        A = np.concatenate([K['wmr'][:, lut[v]], K['wmh'][:, lut[v]], K['iso']], axis=0).astype(np.float64).T
        x = o['x'][v]
        g = A.T @ (f['y'][v] - A @ x) - float(f['lambda2']) * x
        assert np.abs(g[x > 0]).max(initial=0.0) < 1e-10 and g[x == 0].max(initial=0.0) < 1e-10
