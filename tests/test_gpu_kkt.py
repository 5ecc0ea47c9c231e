"""Certificates of the DEVICE solver output itself.

The oracle is parity-unpinned by the reference (no golden vectors, solver dependency absent), so the strongest
solver-independent evidence for the HIP path is the optimality (KKT) certificate of the coefficient vectors `x` the
kernels return through AMX_F_DEBUG_X (include/amico_amd.h) -- computed here in numpy from the device `x`, the
dictionary and the signals only:

  nnls  (models.pyx:911, 940):   x >= 0,  w = A'(y - A x):  w_j ~ 0 where x_j > 0,  w_j <= tol where x_j = 0
  lasso (models.pyx:926, 1238, 1569):  g = A'(y - A x) - lambda2 x - lambda1:  g_j ~ 0 on the support, g_j <= tol off it

on >= 100 000 voxels per model at SNR 30 and SNR 10; plus support equality with the oracle, the fixtures' per-stage
coefficients, and a hard cap on every voxel's map difference (BASELINE.json: 1e-4).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_VOX = 100_000
W_P_TOL = 1e-9       # |dual value| on the support (a least-squares solve on nearly collinear columns, cond up to 1e8)
W_Z_TOL = 1e-9       # largest admissible dual value off the support
CAP = 1e-4           # BASELINE.json: maps within 1e-4 of the reference -- on EVERY voxel


def _by_direction(lut_idx):
    order = np.argsort(lut_idx, kind='stable')
    bounds = np.flatnonzero(np.diff(lut_idx[order])) + 1
    return np.split(order, bounds)


def _noddi_certificates(K, sch, ht, y, d, x, lam1, lam2, exvivo=False):
    """max KKT violations of the three NODDI solves, from the device coefficients only (exvivo: the dot atom -- a column of
    ones, models.pyx:843-844 -- sits between the wm atoms and iso)"""
    from amico_amd import synthetic as S
    lut = S.lut_indices(d, ht)
    n_wm = K['wm'].shape[0]
    iso = K['iso'].astype(np.float64)
    fixed = ([np.ones_like(iso)] if exvivo else []) + [iso]
    dwi = np.asarray(sch.dwi_idx)
    norms = K['norms'][0]
    out = {k: 0.0 for k in ('s1_wP', 's1_wZ', 's2_gP', 's2_gZ', 's3_wP', 's3_wZ', 's3_off_support')}
    neg = 0.0
    for rows in _by_direction(lut):
        A = np.concatenate([K['wm'][:, lut[rows[0]], :].astype(np.float64)] + [f[None, :] for f in fixed], axis=0).T      # nS x n_atoms
        Y = y[rows]
        x1, x2, x3 = x[rows, 0], x[rows, 1], x[rows, 2]
        neg = min(neg, x1.min(), x2.min(), x3.min())
        # stage 1: NNLS over all atoms
        W = (Y - x1 @ A.T) @ A
        P = x1 > 0
        out['s1_wP'] = max(out['s1_wP'], np.abs(W[P]).max(initial=0.0))
        out['s1_wZ'] = max(out['s1_wZ'], W[~P].max(initial=0.0))
        # stage 2: non-negative elastic net on the column-normalised wm atoms, y2 clipped (models.pyx:914-926)
        A2 = A[dwi][:, :n_wm] * norms[None, :]
        Y2 = np.maximum(Y[:, dwi] - x1[:, n_wm:] @ A[dwi][:, n_wm:].T, 0.0)
        xl = x2[:, :n_wm]
        G = (Y2 - xl @ A2.T) @ A2 - lam2 * xl - lam1
        P = xl > 0
        out['s2_gP'] = max(out['s2_gP'], np.abs(G[P]).max(initial=0.0))
        out['s2_gZ'] = max(out['s2_gZ'], G[~P].max(initial=0.0))
        # stage 3: NNLS on the LASSO support + iso (models.pyx:929-942)
        allowed = np.concatenate([P, np.ones((len(rows), len(fixed)), dtype=bool)], axis=1)
        W = (Y - x3 @ A.T) @ A
        P3 = x3 > 0
        out['s3_off_support'] = max(out['s3_off_support'], np.abs(x3[~allowed]).max(initial=0.0))
        out['s3_wP'] = max(out['s3_wP'], np.abs(W[P3]).max(initial=0.0))
        out['s3_wZ'] = max(out['s3_wZ'], W[allowed & ~P3].max(initial=0.0))
    out['min_x'] = float(neg)
    return out


@pytest.mark.parametrize('snr,mapping', [(30.0, 'seeded'), (10.0, 'seeded'), (10.0, 'cold')])
def test_noddi_kkt_certificates_and_supports(htable500, snr, mapping, amx_env):
    """mapping: 'seeded' = seed solvers + Gram-space certificates + left-over kernels (default), 'cold' = the wavefront-per-voxel
    active-set solvers from the empty set for every voxel (AMX_NO_SEED=1)"""
    import torch
    if mapping == 'cold':
        amx_env(AMX_NO_SEED='1')
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    dirs, ht = htable500['dirs'], htable500['htable']
    sch = S.make_scheme(seed=0)
    K = S.noddi_kernels(sch, dirs)
    y, d = S.noddi_signals(N_VOX, K, ht, sch, seed=21 + int(snr), snr=snr)
    ctx = get_context()
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    dev = torch.device('cuda', 0)
    est, _, _, _, xd = _capi.noddi_fit_device(ctx, lut, torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev),
                                              0.5, 1e-3, 3, return_x=True)
    ctx.sync()
    x = xd.cpu().numpy()
    est = est.cpu().numpy()
    assert ctx.last_stats()['itercap_voxels'] == 0 and ctx.last_stats()['guard_trips'] == 0
    c = _noddi_certificates(K, sch, ht, y, d, x, 0.5, 1e-3)
    assert c['min_x'] >= 0.0, c
    assert c['s3_off_support'] == 0.0, c                       # exact zeros off the allowed set
    assert c['s1_wP'] < W_P_TOL and c['s3_wP'] < W_P_TOL and c['s2_gP'] < W_P_TOL, c
    assert c['s1_wZ'] < W_Z_TOL and c['s3_wZ'] < W_Z_TOL and c['s2_gZ'] < W_Z_TOL, c
    # against the oracle: stage-2 supports (they select the stage-3 atoms) and the maps on EVERY voxel
    ref = oracle.noddi_fit(y, d, K, ht, sch.dwi_idx, nthreads=os.cpu_count() or 1, return_x=True)
    n_wm = K['wm'].shape[0]
    same = ((x[:, 1, :n_wm] > 0) == (ref['x'][:, 1, :n_wm] > 0)).all(axis=1)
    assert same.mean() >= 0.9999, (same.size - same.sum())
    diff = np.abs(est - ref['estimates']).max(axis=1)
    assert diff.max() < CAP, (diff.max(), int((diff > 1e-6).sum()))
    assert (diff < 1e-6).mean() >= 0.9999
    assert np.abs(x[:, 0, -1] - ref['x'][:, 0, -1]).max() < 1e-7      # x_iso handed from stage 1 to stage 2


@pytest.mark.parametrize('rescue', [False, True, 'tight trip caps'])
@pytest.mark.parametrize('exvivo', [False, True])
def test_noddi_hard_mix_kkt_and_oracle(htable500, exvivo, rescue, amx_env):
    """Signals the dictionary does not explain (synthetic.noddi_hard_signals: crossings, wrong direction, CSF-dominated f_iso in
    [0.5, 1], pure noise, flat, half-zeroed, background, SNR 5 / 15 / 40) through the default path for this size -- the seed ->
    certificate chain, in vivo and ex vivo (dot atom): every voxel's coefficient vectors must satisfy the numpy
    KKT certificates, and the maps must equal the oracle's.  models.pyx:902-981 takes one path whatever the signal; the
    certificate thresholds of the fast path were tuned on clean single-atom voxels.
    rescue: with the second pass of the NNLS certificates that large calls run (k_nnls_gcert<., true>: ill-conditioned supports
    corrected with the signal itself instead of going to the wavefront-per-voxel kernel).
    tight trip caps: the seed solvers give a voxel up after 28 / 24 / 12 trips by default (it goes to the left-over kernels with no
    seed); with caps of 6 / 5 / 4 a large share of the voxels takes that road -- same certificates, same maps."""
    import torch
    if rescue == 'tight trip caps':
        amx_env(AMX_SEED_TRIPCAP='6,5,4')
    elif rescue:
        amx_env(AMX_RESCUE_FROM='0')
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    dirs, ht = htable500['dirs'], htable500['htable']
    sch = S.make_scheme(seed=9)
    K = S.noddi_kernels(sch, dirs)
    y, d, kind = S.noddi_hard_signals(N_VOX, K, ht, sch, seed=9)
    ctx = get_context()
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx, is_exvivo=exvivo)
    dev = torch.device('cuda', 0)
    est, _, _, _, xd = _capi.noddi_fit_device(ctx, lut, torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev),
                                              0.5, 1e-3, 4 if exvivo else 3, return_x=True)
    ctx.sync()
    st, ss = ctx.last_stats(), ctx.last_seed_stats()
    print('hard mix', 'ex vivo' if exvivo else 'in vivo', st, ss)
    assert st['itercap_voxels'] == 0 and st['guard_trips'] == 0 and st['overflow_voxels'] == 0
    assert ss['seeded_voxels'] == N_VOX                         # the fast path ran (ex vivo too); what it could not certify it handed on
    if rescue == 'tight trip caps':
        assert ss['leftover_stage1'] > N_VOX // 20 and ss['leftover_stage3'] > N_VOX // 50, ss
    x = xd.cpu().numpy()
    est = est.cpu().numpy()
    c = _noddi_certificates(K, sch, ht, y, d, x, 0.5, 1e-3, exvivo=exvivo)
    print(c)
    assert c['min_x'] >= 0.0 and c['s3_off_support'] == 0.0, c
    tol = 4e-9                                                  # (signals up to 2.0 and ||y|| up to 20: four times the clean-data bound)
    assert c['s1_wP'] < tol and c['s3_wP'] < tol and c['s2_gP'] < tol, c
    assert c['s1_wZ'] < tol and c['s3_wZ'] < tol and c['s2_gZ'] < tol, c
    ref = oracle.noddi_fit(y, d, K, ht, sch.dwi_idx, is_exvivo=exvivo, nthreads=os.cpu_count() or 1)['estimates']
    diff = np.abs(est - ref).max(axis=1)
    bad = np.flatnonzero(diff > 1e-6)
    print('max %.3e, > 1e-8: %d, > 1e-6: %d, kinds of those %s' % (diff.max(), int((diff > 1e-8).sum()), len(bad), np.bincount(kind[bad], minlength=8).tolist()))
    assert diff.max() < CAP, (diff.max(), kind[int(diff.argmax())])
    assert (diff < 1e-6).mean() >= 0.9999
    assert np.abs(est[:10] - ref[:10]).max() < 1e-12 and np.abs(est[:10, :3] - np.array([0.0, 1.0, 0.0])).max() < 1e-12   # all-zero voxels: NDI 0, ODI 1, FWF 0


def test_noddi_fixture_coefficients_per_stage(noddi_fix, htable500):
    """x_stages of the golden fixture (scipy NNLS / sklearn-checked elastic net per stage, make_fixtures.py)"""
    import torch
    from amico_amd import _capi, get_context
    f = noddi_fix
    ctx = get_context()
    lut = _capi.upload_noddi(ctx, f['kernels'], htable500['htable'], f['dwi_idx'])
    dev = torch.device('cuda', 0)
    est, _, _, _, xd = _capi.noddi_fit_device(ctx, lut, torch.from_numpy(f['y']).to(dev), torch.from_numpy(f['dirs']).to(dev),
                                              float(f['lambda1']), float(f['lambda2']), 3, return_x=True)
    ctx.sync()
    x, xs = xd.cpu().numpy(), f['x_stages']
    n_wm = f['kernels']['wm'].shape[0]
    assert np.abs(x[:, 0] - xs[:, 0]).max() < 1e-7
    assert np.abs(x[:, 1, :n_wm] - xs[:, 1, :n_wm]).max() < 1e-7
    assert np.abs(x[:, 2] - xs[:, 2]).max() < 1e-7
    # (scipy's NNLS leaves a few coefficients of 1e-17 .. 1e-19 where the device solver has exact zeros)
    assert ((x[:, 2] > 1e-12) == (xs[:, 2] > 1e-12)).all()
    assert np.abs(est.cpu().numpy() - f['estimates']).max() < 1e-6


def _lasso_certificate(A, Y, X, lam1, lam2):
    G = (Y - X @ A.T) @ A - lam2 * X - lam1
    P = X > 0
    return float(np.abs(G[P]).max(initial=0.0)), float(G[~P].max(initial=0.0)), float(X.min())


@pytest.mark.parametrize('snr', [30.0, 10.0])
@pytest.mark.parametrize('mapping', ['refill', 'lane', 'wave'])
def test_freewater_kkt_certificates(htable500, snr, mapping, amx_env):
    """mapping: 'refill' = lane per voxel, lanes refilled from a buffer (default), 'lane' = one solve per lane and pass,
    'wave' = one wavefront per voxel"""
    import torch
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    if mapping == 'wave':
        amx_env(AMX_WAVE_PER_VOXEL='1')
    if mapping == 'lane':
        amx_env(AMX_NO_REFILL='1')
    n = 20_000 if mapping == 'wave' else N_VOX
    dirs, ht = htable500['dirs'], htable500['htable']
    sch = S.make_scheme(1, ((1000.0, 64),), seed=3)
    K = S.freewater_kernels(sch, dirs)
    y, d = S.freewater_signals(n, K, ht, sch, seed=5 + int(snr), snr=snr)
    ctx = get_context()
    lut = _capi.upload_freewater(ctx, K, ht)
    dev = torch.device('cuda', 0)
    est, _, _, _, xd = _capi.freewater_fit_device(ctx, lut, torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev),
                                                  0.0, 1e-3, False, return_x=True)
    ctx.sync()
    x = xd.cpu().numpy()
    idx = S.lut_indices(d, ht)
    gp = gz = 0.0
    for rows in _by_direction(idx):
        A = np.concatenate([K['D'][:, idx[rows[0]], :], K['CSF']], axis=0).astype(np.float64).T
        a, b, mn = _lasso_certificate(A, y[rows], x[rows], 0.0, 1e-3)
        gp, gz = max(gp, a), max(gz, b)
        assert mn >= 0.0
    assert gp < W_P_TOL and gz < W_Z_TOL, (gp, gz)
    ref = oracle.freewater_fit(y, d, K, ht, nthreads=os.cpu_count() or 1, return_x=True)
    assert np.abs(est.cpu().numpy() - ref['estimates']).max() < 1e-6
    assert ((x > 0) == (ref['x'] > 0)).all(axis=1).mean() >= 0.9999
    assert ctx.last_stats()['itercap_voxels'] == 0


@pytest.mark.parametrize('snr', [30.0, 10.0])
@pytest.mark.parametrize('mapping', ['rows', 'lane', 'wave'])
def test_sandi_kkt_certificates(snr, mapping, amx_env):
    import torch
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    if mapping == 'wave':
        amx_env(AMX_WAVE_PER_VOXEL='1')
    if mapping == 'lane':
        amx_env(AMX_SANDI_ATOM_SPACE='1')
    n = N_VOX if mapping == 'rows' else 20_000
    avg = S.directional_average_scheme(S.make_sandi_scheme())
    K, Rs, d_in, d_isos = S.sandi_kernels(avg)
    y = S.sandi_signals(n, K, avg, seed=9 + int(snr), snr=snr)
    ctx = get_context()
    lut = _capi.upload_sandi(ctx, K, Rs, d_in, d_isos)
    dev = torch.device('cuda', 0)
    est, _, _, xd = _capi.sandi_fit_device(ctx, lut, torch.from_numpy(y).to(dev), 0.0, 5e-3, return_x=True)
    ctx.sync()
    x = xd.cpu().numpy() / K['norms'][None, :]                 # undo models.pyx:1570-1571 for the certificate
    A = np.asarray(K['signal'], dtype=np.float64)
    gp, gz, mn = _lasso_certificate(A, y, x, 0.0, 5e-3)
    assert mn >= 0.0 and gp < W_P_TOL and gz < W_Z_TOL, (gp, gz, mn)
    ref = oracle.sandi_fit(y, K, Rs, d_in, d_isos, nthreads=os.cpu_count() or 1, return_x=True)
    assert ((xd.cpu().numpy() > 0) == (ref['x'] > 0)).all(axis=1).mean() >= 0.9999
    e, r = est.cpu().numpy(), ref['estimates']
    assert np.abs(e[:, :3] - r[:, :3]).max() < 1e-6            # volume fractions
    assert (np.abs(e[:, 3:] - r[:, 3:]) / (np.abs(r[:, 3:]) + 1e-3)).max() < 1e-6   # Rsoma / Din / De (um, um^2/ms)
    assert ctx.last_stats()['itercap_voxels'] == 0
