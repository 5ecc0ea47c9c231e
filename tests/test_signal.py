"""Steps either side of model.fit (SURVEY section 8 f): oracle restatement vs golden vectors (CPU) and the HIP
kernels vs the oracle (GPU)."""
import numpy as np
import pytest

from conftest import load_npz
from oracle import signal_np
from amico_amd import synthetic as S


@pytest.fixture(scope='module')
def dti_fix():
    return load_npz('dti_fixture.npz')


def axis_error(a, b):
    """sin of the angle between two axes (sign-free: eigenvectors have no defined sign)"""
    a = a / np.linalg.norm(a, axis=1, keepdims=True)
    b = b / np.linalg.norm(b, axis=1, keepdims=True)
    return np.linalg.norm(np.cross(a, b), axis=1)


def well_separated(evals, rel=1e-6, absolute=1e-9):
    """voxels whose principal axis is defined: the two largest eigenvalues differ (isotropic / all-zero voxels have
    D = const * I or D ~ 0 and any axis is an eigenvector)"""
    gap = evals[:, 0] - evals[:, 1]
    return (gap > rel * np.abs(evals[:, 0])) & (gap > absolute)


# ----------------------------------------------------------------------------- CPU: oracle vs golden
def test_oracle_dti_vs_golden(dti_fix):
    sc = dti_fix['scheme']
    dirs, evals = signal_np.dti_directions(dti_fix['y'], sc[:, 3], sc[:, :3], return_evals=True)
    ok = well_separated(dti_fix['evals'])
    assert ok.sum() > 200
    assert axis_error(dirs[ok], dti_fix['dirs'][ok]).max() < 1e-8
    np.testing.assert_allclose(evals[ok], dti_fix['evals'][ok], rtol=0, atol=1e-12)
    # closed-form voxels: the axis that generated the signal
    assert axis_error(dirs[160:], dti_fix['closed_form_axis']).max() < 1e-10
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-12)


def test_oracle_gradient_table_rules():
    b = np.array([0.0, 5.0, 1000.0, 1000.0])
    g = np.array([[0, 0, 0], [1.0, 0, 0], [0, 1.0, 0], [0, 0.6, 0.8]])
    bb, gg = signal_np.gradient_table(b, g)
    assert bb.tolist() == [0.0, 5.0, 1000.0, 1000.0] and gg[0].tolist() == [0, 0, 0]
    with pytest.raises(ValueError):
        signal_np.gradient_table(np.array([1000.0]), np.array([[0.0, 0.5, 0.5]]))     # non-unit DWI vector
    B = signal_np.design_matrix(bb, gg)
    assert B.shape == (4, 7) and np.all(B[:, 6] == -1) and B[3, 4] == -2 * 1000 * 0.6 * 0.8
    # host mirror builds the same table / design matrix (no GPU needed for this part)
    from amico_amd import dti
    b2, g2 = dti.gradient_table(b, g)
    assert np.array_equal(bb, b2) and np.array_equal(gg, g2)
    assert np.array_equal(dti.design_matrix(b2, g2), B)


def test_min_signal_clip_matches_dipy_default():
    sc = S.make_scheme()
    y = np.zeros((3, sc.nS))
    y[1] = 1.0
    y[2] = -5.0
    d = signal_np.dti_directions(y, sc.b, sc.raw[:, :3])
    assert d.shape == (3, 3) and np.all(np.isfinite(d))          # log(max(y, 1e-4)): no -inf / nan


# ----------------------------------------------------------------------------- GPU: HIP kernel vs oracle
@pytest.mark.gpu
def test_dti_directions_golden_and_oracle(dti_fix):
    from amico_amd import dti
    sc = dti_fix['scheme']
    est = dti.TensorDirections(sc[:, 3], sc[:, :3])
    dirs = est.fit(dti_fix['y'])
    ok = well_separated(dti_fix['evals'])
    assert axis_error(dirs[ok], dti_fix['dirs'][ok]).max() < 1e-8
    assert axis_error(dirs[160:], dti_fix['closed_form_axis']).max() < 1e-10
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-12)
    assert est.fit(np.zeros((0, sc.shape[0]))).shape == (0, 3)
    with pytest.raises(ValueError):
        est.fit(np.zeros((4, 7)))


@pytest.mark.gpu
@pytest.mark.parametrize('n_vox', [1, 63, 64, 65, 20011])
def test_dti_directions_vs_oracle_synthetic(htable500, n_vox):
    """ragged tile sizes; realistic NODDI signals; the LUT index of the direction (what the fit consumes) is identical"""
    from amico_amd import dti
    from oracle import oracle
    sc = S.make_scheme()
    K = S.noddi_kernels(sc, htable500['dirs'])
    y, _ = S.noddi_signals(n_vox, K, htable500['htable'], sc, seed=5)
    ref, evals = signal_np.dti_directions(y, sc.b, sc.raw[:, :3], return_evals=True)
    dirs = dti.TensorDirections.from_scheme(sc).fit(y)
    ok = well_separated(evals)
    err = axis_error(dirs[ok], ref[ok])
    gap = ((evals[:, 0] - evals[:, 1]) / np.abs(evals[:, 0]))[ok]
    assert (err * gap).max() < 1e-12 and err.max() < 1e-8
    i_ref = np.atleast_1d(oracle.dir_to_lut_idx(ref[ok], htable500['htable']))
    i_gpu = np.atleast_1d(oracle.dir_to_lut_idx(dirs[ok], htable500['htable']))
    assert (i_ref != i_gpu).mean() <= 1e-4


@pytest.mark.gpu
def test_dti_other_schemes_and_merge_b0(htable500):
    from amico_amd import dti
    for sc, merge in ((S.make_scheme(n_b0=1, shells=((1000.0, 64),), seed=2), False),
                      (S.make_scheme(n_b0=6, shells=((1000.0, 20), (2000.0, 30), (3000.0, 94)), seed=3), True)):
        rng = np.random.default_rng(1)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        D = (q * np.array([1.7e-3, 0.4e-3, 0.3e-3])) @ q.T
        b, g = sc.b, sc.raw[:, :3]
        y = np.exp(-b * np.einsum('ij,jk,ik->i', g, D, g))[None, :].repeat(130, 0)
        if merge:
            y = np.hstack([y[:, sc.b0_idx].mean(1, keepdims=True), y[:, sc.dwi_idx]])
        d = dti.TensorDirections.from_scheme(sc, do_merge_b0=merge).fit(y)
        assert axis_error(d, q[:, 0][None, :].repeat(130, 0)).max() < 1e-10
