"""CylinderZeppelinBall on the GPU (SURVEY.md 8 row a-M, models.pyx:526-652): golden fixture, the oracle on a larger
seeded set, KKT certificate of the device coefficients, the model plug-in surface."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Holder:
    def __init__(self, y, dirs, htable, kernels, **cfg):
        self.y, self.DIRs, self.htable, self.KERNELS, self.nthreads = y, dirs, htable, kernels, 4
        self._cfg = cfg

    def get_config(self, k):
        return self._cfg.get(k, False)


def test_czb_golden_fixture_and_model_surface(czb_fix, htable500):
    from amico_amd import CylinderZeppelinBall
    f = czb_fix
    m = CylinderZeppelinBall()
    assert m.maps_name == ['v', 'a', 'd'] and m.solver_params == {'lambda1': 0.0, 'lambda2': 4.0}
    assert m.get_params()['isExvivo'] is False and len(m.Rs) == 21 and len(m.d_perps) == 4 and len(m.d_isos) == 1
    out = m.fit(Holder(f['y'], f['dirs'], htable500['htable'], f['kernels'], doComputeRMSE=True, doComputeNRMSE=True))
    assert out['estimates'].shape == (len(f['y']), 3) and out['estimates'].dtype == np.float64
    rel = np.abs(out['estimates'] - f['estimates']) / (np.abs(f['estimates']) + 1e-3)
    assert rel.max() < 1e-6, rel.max()
    assert np.abs(out['rmse'] - f['rmse']).max() < 1e-9
    assert np.allclose(out['estimates'][0], 0.0)
    # float32 signals through the host entry point: same maps
    out32 = m.fit(Holder(f['y'].astype(np.float32), f['dirs'], htable500['htable'], f['kernels']))
    ref32 = m.fit(Holder(f['y'].astype(np.float32).astype(np.float64), f['dirs'], htable500['htable'], f['kernels']))
    assert np.array_equal(out32['estimates'], ref32['estimates'])
    with pytest.raises(ValueError):
        m.set(Rs=np.linspace(1, 5, 4) * 1e-6)
        m.fit(Holder(f['y'], f['dirs'], htable500['htable'], f['kernels']))


def test_czb_vs_oracle_and_kkt(czb_fix, htable500):
    import os
    import torch
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    f = czb_fix
    ht, K, ids = htable500['htable'], f['kernels'], f['lut_ids']
    rng = np.random.default_rng(3)
    # 20 000 voxels whose directions fall into the LUT cells the fixture carries dictionaries for
    dirs = []
    while sum(len(d) for d in dirs) < 20000:
        d = S.random_unit_vectors(400000, rng)
        dirs.append(d[np.isin(S.lut_indices(d, ht), ids)])
    d = np.concatenate(dirs)[:20000]
    lut = S.lut_indices(d, ht)
    n = len(d)
    n_rs, n_p = K['wmr'].shape[0], K['wmh'].shape[0]
    w = rng.dirichlet([2.0, 2.0, 1.0], n)
    y0 = w[:, :1] * K['wmr'][rng.integers(n_rs, size=n), lut].astype(np.float64) + \
        w[:, 1:2] * K['wmh'][rng.integers(n_p, size=n), lut].astype(np.float64) + w[:, 2:] * K['iso'][0].astype(np.float64)
    y = np.abs(y0 + rng.normal(scale=1 / 20.0, size=y0.shape) + 1j * rng.normal(scale=1 / 20.0, size=y0.shape))
    y[5] = np.inf                                                     # non-finite signal -> NaN maps, no hang
    ctx = get_context()
    L = _capi.upload_czb(ctx, K, f['Rs'], ht)
    dev = torch.device('cuda', 0)
    est, r, _, xd = _capi.czb_fit_device(ctx, L, torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev), 0.0, 4.0,
                                         rmse=True, return_x=True)
    ctx.sync()
    est, x = est.cpu().numpy(), xd.cpu().numpy()
    assert np.isnan(est[5]).all()
    ok = np.ones(n, bool); ok[5] = False
    ref = oracle.czb_fit(y[ok], d[ok], K, f['Rs'], ht, 0.0, 4.0, rmse=True, nthreads=os.cpu_count() or 1, return_x=True)
    rel = np.abs(est[ok] - ref['estimates']) / (np.abs(ref['estimates']) + 1e-3)
    assert rel.max() < 1e-6, rel.max()
    assert np.abs(r.cpu().numpy()[ok] - ref['rmse']).max() < 1e-9
    assert ((x[ok] > 0) == (ref['x'] > 0)).all(axis=1).mean() >= 0.9999
    gp = gz = 0.0
    for lid in ids:
        rows = np.flatnonzero((lut == lid) & ok)
        A = np.concatenate([K['wmr'][:, lid], K['wmh'][:, lid], K['iso']], axis=0).astype(np.float64).T
        G = (y[rows] - x[rows] @ A.T) @ A - 4.0 * x[rows]
        P = x[rows] > 0
        gp, gz = max(gp, np.abs(G[P]).max(initial=0.0)), max(gz, G[~P].max(initial=0.0))
    assert x[ok].min() >= 0.0 and gp < 1e-9 and gz < 1e-9, (gp, gz)
    assert ctx.last_stats()['itercap_voxels'] == 0 and ctx.last_stats()['overflow_voxels'] == 0


@pytest.mark.parametrize('lam2', [0.0, 1e-8])
def test_czb_without_a_ridge(czb_fix, htable500, lam2):
    """the reference's lasso accepts any lambda2 >= 0 (models.pyx:439, 615): below 1e-6 the fit runs the thin-QR solver in A-space.
    Without a ridge `x` need not be unique (26 nearly collinear atoms), A x is: KKT certificate of the device x, and A x against
    the oracle's."""
    import os
    import torch
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    f = czb_fix
    ht, K, ids = htable500['htable'], f['kernels'], f['lut_ids']
    rng = np.random.default_rng(5)
    d = S.random_unit_vectors(200000, rng)
    d = d[np.isin(S.lut_indices(d, ht), ids)][:4000]
    lut = S.lut_indices(d, ht)
    n = len(d)
    n_rs, n_p = K['wmr'].shape[0], K['wmh'].shape[0]
    w = rng.dirichlet([2.0, 2.0, 1.0], n)
    y0 = w[:, :1] * K['wmr'][rng.integers(n_rs, size=n), lut].astype(np.float64) + \
        w[:, 1:2] * K['wmh'][rng.integers(n_p, size=n), lut].astype(np.float64) + w[:, 2:] * K['iso'][0].astype(np.float64)
    y = np.abs(y0 + rng.normal(scale=1 / 30.0, size=y0.shape) + 1j * rng.normal(scale=1 / 30.0, size=y0.shape))
    ctx = get_context()
    L = _capi.upload_czb(ctx, K, f['Rs'], ht)
    dev = torch.device('cuda', 0)
    est, _, _, xd = _capi.czb_fit_device(ctx, L, torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev), 0.0, lam2, return_x=True)
    ctx.sync()
    x = xd.cpu().numpy()
    st = ctx.last_stats()
    assert st['itercap_voxels'] == 0 and st['overflow_voxels'] == 0 and st['guard_trips'] == 0
    gp = gz = ax = 0.0
    for lid in ids:
        rows = np.flatnonzero(lut == lid)
        if len(rows) == 0:
            continue
        A = np.concatenate([K['wmr'][:, lid], K['wmh'][:, lid], K['iso']], axis=0).astype(np.float64).T
        G = (y[rows] - x[rows] @ A.T) @ A - lam2 * x[rows]
        P = x[rows] > 0
        gp, gz = max(gp, np.abs(G[P]).max(initial=0.0)), max(gz, G[~P].max(initial=0.0))
        for v in rows[::25]:                                  # lambda1 = 0 and (nearly) no ridge: plain NNLS, A x is unique
            xr, _, _ = oracle.nnls(A, y[v])
            ax = max(ax, np.abs(A @ x[v] - A @ xr).max())
    assert x.min() >= 0.0 and gp < 1e-9 and gz < 1e-9, (gp, gz)
    assert ax < (1e-8 if lam2 == 0.0 else 1e-4), ax            # (a ridge of 1e-8 against singular values of ~1e-4: A x moves by ~1e-6)
    assert np.isfinite(est.cpu().numpy()).all()


@pytest.mark.parametrize('lam2', [4.0, 1e-8])
def test_czb_fit_writes_every_voxel(czb_fix, htable500, lam2):
    """as test_noddi_fit_writes_every_voxel: the maps are not cleared before the fit (skipped voxels zeroed by k_dir_to_lut, every
    other voxel written by the solver) -- default ridge (lane solver) and the thin-QR route"""
    import torch
    from amico_amd import _capi, get_context, synthetic as S
    f = czb_fix
    ht, K, ids = htable500['htable'], f['kernels'], f['lut_ids']
    rng = np.random.default_rng(5)
    d = S.random_unit_vectors(600000, rng)
    d = d[np.isin(S.lut_indices(d, ht), ids)][:30000]
    lut = S.lut_indices(d, ht)
    n = len(d)
    w = rng.dirichlet([2.0, 2.0, 1.0], n)
    y = w[:, :1] * K['wmr'][rng.integers(K['wmr'].shape[0], size=n), lut].astype(np.float64) + \
        w[:, 1:2] * K['wmh'][rng.integers(K['wmh'].shape[0], size=n), lut].astype(np.float64) + w[:, 2:] * K['iso'][0].astype(np.float64)
    y = np.abs(y + rng.normal(scale=0.1, size=y.shape))
    y[:10] = 0.0; y[11] = np.nan; y[12, 3] = np.inf; y[n - 1] = np.nan
    y[20:60] = np.abs(rng.normal(size=(40, y.shape[1])))
    bad = [17, n // 2, n - 2]
    d[bad] = np.nan
    ctx = get_context()
    L = _capi.upload_czb(ctx, K, f['Rs'], ht)
    yt, dt = torch.from_numpy(y).cuda(), torch.from_numpy(d).cuda()
    lib = _capi.lib()
    outs = []
    for fill in (-7.25, 0.0):
        est = torch.full((n, 3), fill, dtype=torch.float64, device='cuda')
        rm = torch.full((n,), fill, dtype=torch.float64, device='cuda')
        assert lib.amx_czb_fit_device(ctx._h, L._h, yt.data_ptr(), dt.data_ptr(), n, 0.0, lam2, _capi.F_RMSE, est.data_ptr(), rm.data_ptr(), None, None) == 0
        with pytest.raises(RuntimeError, match=r'index out of bounds.*\[voxel 17\]'):
            ctx.sync()
        outs.append((est.cpu().numpy(), rm.cpu().numpy()))
    (e1, r1), (e0, r0) = outs
    ok = np.ones(n, bool); ok[bad] = False
    assert not (e1 == -7.25).any() and np.array_equal(e1, e0, equal_nan=True) and (e1[bad] == 0.0).all()
    assert np.array_equal(r1[ok], r0[ok], equal_nan=True) and not (r1[ok] == -7.25).any()
    L.close()
