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


# ============================================================================= signal preparation / scatter
def _volume(shape, scheme, order, seed=0, with_bad=True):
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.0, 900.0, shape + (scheme.nS,)).astype(np.float32)
    img[..., scheme.b0_idx] += 600.0
    if with_bad:
        img[0, 0, 0, :] = 0.0                      # b0 mean 0 -> norm factor 0
        img[1, 2, 1, 3] = -7.5                     # negative sample -> clipped
        img[2, 1, 0, scheme.b0_idx] = -1.0         # negative b0 mean -> norm factor 0 (mean <= 0)
    mask = (rng.uniform(size=shape) < 0.6).astype(np.uint8)
    mask[0, 0, 0] = mask[1, 2, 1] = mask[2, 1, 0] = 1
    mask[3, 3, 2] = 2                              # only == 1 counts (core.py:451)
    return np.asarray(img, order=order), mask


def test_oracle_prepare_signal_hand_cases():
    sc = S.make_scheme(n_b0=3, shells=((1000.0, 4),), seed=1)
    img = np.ones((2, 2, 2, sc.nS), dtype=np.float32)
    img[..., sc.b0_idx] = [2.0, 4.0, 6.0]          # mean b0 = 4
    img[1, 1, 1] = 0.0
    mask = np.ones((2, 2, 2), dtype=np.uint8)
    mask[0, 0, 1] = 0
    y, mb0 = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx)
    assert y.shape == (7, sc.nS) and mb0[0, 0, 0] == 4.0
    assert np.array_equal(y[0, sc.b0_idx], [0.5, 1.0, 1.5]) and np.all(y[0, sc.dwi_idx] == 0.25)
    assert np.all(y[-1] == 0.0)                                          # b0 = 0 voxel: factor 0
    ym, _ = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, do_merge_b0=True)
    assert ym.shape == (7, 5) and ym[0, 0] == 1.0 and np.all(ym[0, 1:] == 0.25)
    vol = signal_np.scatter_results(np.arange(14.0).reshape(7, 2), mask)
    assert vol.dtype == np.float32 and vol.shape == (2, 2, 2, 2) and np.all(vol[0, 0, 1] == 0) and vol[0, 1, 0, 1] == 3.0


def _prep_scheme(b):
    g = np.zeros((len(b), 3))
    g[b > 0, 0] = 1.0
    return S.SimpleScheme(np.hstack([g, b[:, None]]))


def test_oracle_prepare_signal_vs_golden():
    f = load_npz('prep_fixture.npz')
    sc = _prep_scheme(f['b'])
    img, mask = f['img'], f['mask']
    y, mb0 = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx)
    assert np.array_equal(y, f['y_plain']) and np.array_equal(mb0, f['mean_b0s'])
    assert np.array_equal(signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, do_normalize=False)[0], f['y_raw'])
    assert np.array_equal(signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, b0_min_signal=0.9)[0], f['y_b0min'])
    assert np.array_equal(signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, do_merge_b0=True)[0], f['y_merge'])
    assert np.array_equal(signal_np.prepare_signal(np.asfortranarray(img), mask, sc.b0_idx, sc.dwi_idx, shells=sc.shells,
                                                   do_directional_average=True)[0], f['y_diravg'])
    assert np.array_equal(signal_np.scatter_results(f['values'], mask), f['volume'])


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
def test_prepare_signal_golden(order):
    from amico_amd import prep
    f = load_npz('prep_fixture.npz')
    sc = _prep_scheme(f['b'])
    img, mask = np.asarray(f['img'], order=order), f['mask']
    for key, opts in (('y_plain', {}), ('y_raw', dict(do_normalize=False)), ('y_b0min', dict(b0_min_signal=0.9)),
                      ('y_merge', dict(do_merge_b0=True)), ('y_diravg', dict(do_directional_average=True))):
        sp = prep.SignalPreparation(sc, img, mask, **opts)
        y, _ = sp.gather(img)
        assert np.array_equal(y, f[key]), key
    assert np.array_equal(sp.scatter(f['values']), f['volume'])
    assert np.array_equal(prep.SignalPreparation(sc, img, mask)._plan.mean_b0(img), f['mean_b0s'])


def test_volume_groups_follow_the_reference_rules():
    from amico_amd import prep
    sc = S.make_sandi_scheme(bvals=(4000.0, 1000.0, 2500.0), ndir_per_shell=5, n_b0=2)
    g = prep.volume_groups(sc, do_directional_average=True)
    assert g[0] == [0, 1] and g[1] == list(range(7, 12)) and g[2] == list(range(12, 17)) and g[3] == list(range(2, 7))
    t = prep.directional_average_table(sc)
    assert t.shape == (4, 7) and t[0].tolist() == [1, 0, 0, 0, 0, 0, 0] and np.all(np.diff(t[1:, 3]) > 0)
    sc2 = S.make_scheme(n_b0=2, shells=((700.0, 3),), seed=0)
    assert prep.volume_groups(sc2, do_merge_b0=True) == [[0, 1], [2], [3], [4]]
    assert prep.volume_groups(sc2) == [[0], [1], [2], [3], [4]]


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
@pytest.mark.parametrize('opts', [dict(), dict(do_normalize=False), dict(do_merge_b0=True), dict(b0_min_signal=0.9)])
def test_prepare_signal_bit_exact(order, opts):
    from amico_amd import prep
    sc = S.make_scheme()
    img, mask = _volume((70, 9, 5), sc, order, seed=3)
    ref, mb0 = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, **opts)
    sp = prep.SignalPreparation(sc, img, mask, **opts)
    y, m = sp.gather(img)
    assert y.shape == ref.shape and np.array_equal(y, ref)               # bit-exact, incl. -0.0 / clipping
    if opts.get('do_normalize', True):
        assert np.array_equal(m, mb0[mask == 1])
        assert np.array_equal(sp._plan.mean_b0(img), mb0)
    vals = np.random.default_rng(0).normal(size=(sp.n_vox, 3))
    assert np.array_equal(sp.scatter(vals), signal_np.scatter_results(vals, mask))
    assert np.array_equal(sp.scatter(vals[:, 0]), signal_np.scatter_results(vals[:, 0], mask))


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
def test_directional_average_bit_exact_including_the_view_aliasing(order):
    from amico_amd import prep
    for bvals in ((1000.0, 2500.0, 4000.0, 6000.0, 8000.0), (4000.0, 1000.0, 2500.0)):   # 2nd: shells not in b order
        sc = S.make_sandi_scheme(bvals=bvals, ndir_per_shell=12 if len(bvals) == 3 else 60, n_b0=2 if len(bvals) == 3 else 6)
        img, mask = _volume((33, 6, 4), sc, order, seed=5, with_bad=False)
        ref, _ = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx, shells=sc.shells, do_directional_average=True)
        sp = prep.SignalPreparation(sc, img, mask, do_directional_average=True)
        y, _ = sp.gather(img)
        assert y.shape == (int((mask == 1).sum()), len(bvals) + 1) and np.array_equal(y, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
@pytest.mark.parametrize('density', [0.004, 0.5, 1.0])
def test_prepare_signal_sparse_and_full_masks(order, density):
    """the gather kernel walks a plan-time list of the tiles that hold masked voxels (amx_prep::live64): a mask with a handful of
    voxels (most tiles dead, the live ones with a single voxel), a ragged one, and a full one -- extents that are no multiples of the
    64-voxel tile, both memory orders, float64 and float32 rows -- all bit-exact with the numpy statements of core.py:209-223, 451-452"""
    from amico_amd import prep
    sc = S.make_scheme()
    shape = (131, 7, 9) if order == 'F' else (5, 6, 131)
    rng = np.random.default_rng(11)
    img = rng.uniform(0.0, 900.0, shape + (sc.nS,)).astype(np.float32)
    img[..., sc.b0_idx] += 600.0
    mask = (rng.uniform(size=shape) < density).astype(np.uint8)
    mask[-1, -1, -1] = 1                                                 # the very last voxel of the image
    mask[0, 0, 0] = 1
    img = np.asarray(img, order=order)
    ref, mb0 = signal_np.prepare_signal(img, mask, sc.b0_idx, sc.dwi_idx)
    sp = prep.SignalPreparation(sc, img, mask)
    for _ in range(2):                                                   # the plan's tile counter is reset by every launch
        y, m = sp.gather(img)
        assert y.shape == ref.shape and np.array_equal(y, ref)
        assert np.array_equal(m, mb0[mask == 1])


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
@pytest.mark.parametrize('opts', [dict(), dict(do_normalize=False), dict(do_merge_b0=True)])
@pytest.mark.parametrize('f32', [True, False])
def test_gather_with_directions_equals_gather_then_tensor_fit(order, opts, f32):
    """amx_prep_gather_directions_device[_f32] (the tensor fit rides on the gather's LDS tile: one pass over the image) against
    amx_prep_gather_device + amx_dti_directions_device on the same image: y bit for bit, the directions to rounding (the fused
    kernel sums the volumes in index order), on both memory orders, with and without normalisation / b0 merge, ragged mask"""
    import torch
    from amico_amd import prep, dti, _capi
    sc = S.make_scheme()
    img, mask = _volume((70, 9, 5), sc, order, seed=3)
    sp = prep.SignalPreparation(sc, img, mask, **opts)
    td = dti.TensorDirections.from_scheme(sc, do_merge_b0=opts.get('do_merge_b0', False), ctx=sp.ctx)
    L, c = _capi.lib(), sp.ctx
    dev = torch.device('cuda', 0)
    flat = np.lib.stride_tricks.as_strided(img, shape=(img.size,), strides=(4,))
    d_img = torch.from_numpy(flat.copy()).to(dev)
    n, m = sp.n_vox, sp.n_out
    ydt = torch.float32 if f32 else torch.float64
    y_a, y_b = torch.zeros((n, m), dtype=ydt, device=dev), torch.zeros((n, m), dtype=ydt, device=dev)
    mb_a, mb_b = torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)
    d_a, d_b = torch.zeros((n, 3), dtype=torch.float64, device=dev), torch.zeros((n, 3), dtype=torch.float64, device=dev)
    norm = int(sp.do_normalize)
    gather = L.amx_prep_gather_device_f32 if f32 else L.amx_prep_gather_device
    fused = L.amx_prep_gather_directions_device_f32 if f32 else L.amx_prep_gather_directions_device
    c.check(gather(c._h, sp._plan._h, d_img.data_ptr(), norm, 0.0, y_a.data_ptr(), mb_a.data_ptr(), None))
    td.fit_device(y_a.data_ptr(), n, d_a.data_ptr(), None, f32=f32)
    c.check(fused(c._h, sp._plan._h, td._dti._h, d_img.data_ptr(), norm, 0.0, y_b.data_ptr(), mb_b.data_ptr(), d_b.data_ptr(), None))
    c.sync(None)
    assert torch.equal(y_a, y_b) and torch.equal(mb_a, mb_b)
    a, b = d_a.cpu().numpy(), d_b.cpu().numpy()
    assert np.isfinite(b).all() == np.isfinite(a).all()
    yv = y_a.cpu().numpy().astype(np.float64)
    flat_row = np.ptp(np.log(np.maximum(yv, 1e-4)), axis=1) == 0.0          # (norm factor 0 -> a constant row: a zero tensor, any direction)
    ok = np.isfinite(a).all(axis=1) & ~flat_row
    assert ok.sum() > 0.9 * n and np.abs(a[ok] - b[ok]).max() < 1e-12
    # a helper for another scheme is refused
    other = dti.TensorDirections.from_scheme(S.make_scheme(shells=((1000.0, 8),), seed=1), ctx=sp.ctx)
    with pytest.raises(Exception):
        c.check(fused(c._h, sp._plan._h, other._dti._h, d_img.data_ptr(), norm, 0.0, y_b.data_ptr(), mb_b.data_ptr(), d_b.data_ptr(), None))


@pytest.mark.gpu
def test_prepare_signal_errors_and_edges():
    from amico_amd import prep
    sc = S.make_scheme(n_b0=0, shells=((1000.0, 8),), seed=1)
    img = np.ones((4, 4, 4, 8), dtype=np.float32)
    with pytest.raises(RuntimeError):
        prep.SignalPreparation(sc, img, np.ones((4, 4, 4)))              # no b0 to normalise with
    sp = prep.SignalPreparation(sc, img, np.zeros((4, 4, 4)), do_normalize=False)   # empty mask
    y, _ = sp.gather(img)
    assert y.shape == (0, 8) and not sp.scatter(np.zeros((0, 2))).any()
    with pytest.raises(ValueError):
        prep.SignalPreparation(sc, img[..., :5], np.ones((4, 4, 4)), do_normalize=False)
    with pytest.raises(ValueError):
        sp.gather(np.ones((4, 4, 5, 8), dtype=np.float32))


@pytest.mark.gpu
def test_c_abi_argument_checks_of_the_widened_rows():
    """bad arguments come back as AMX_E_BADARG (ValueError) with a message, never as a crash"""
    from amico_amd import _capi, get_context
    ctx = get_context()
    with pytest.raises(ValueError):
        _capi.Dti(ctx, np.zeros((7, 3)))                                 # fewer volumes than parameters
    with pytest.raises(ValueError):
        _capi.Dti(ctx, np.zeros((6, 30)))                                # not pinv of a 7-column design matrix
    with pytest.raises(ValueError):
        _capi.Dti(ctx, np.zeros((7, 30)), min_signal=0.0)
    rank = np.full((3, 3, 3), -1, dtype=np.int32)
    rank[0, 0, 0] = 0
    rank[1, 1, 1] = 0                                                    # row 0 used twice
    with pytest.raises(ValueError):
        _capi.Prep(ctx, (3, 3, 3, 4), (36, 12, 4, 1), rank, [[0], [1], [2], [3]], [0])
    rank[1, 1, 1] = 1
    with pytest.raises(ValueError):
        _capi.Prep(ctx, (3, 3, 3, 4), (36, 12, 4, 1), rank, [[0], [7]], [0])          # group index out of range
    with pytest.raises(ValueError):
        _capi.Prep(ctx, (3, 3, 3, 4), (36, 12, 4, 0), rank, [[0]], [0])               # zero stride
    p = _capi.Prep(ctx, (3, 3, 3, 4), (36, 12, 4, 1), rank, [[0], [1], [2], [3]], [])
    with pytest.raises(ValueError):
        p.gather(np.zeros((3, 3, 3, 4), dtype=np.float32), normalize=True)            # no b0 to normalise with
    y, _ = p.gather(np.arange(108, dtype=np.float32).reshape(3, 3, 3, 4), normalize=False)
    assert y.tolist() == [[0.0, 1.0, 2.0, 3.0], [52.0, 53.0, 54.0, 55.0]]
    assert ctx.selftest().shape == (12, 64)                              # the context is still usable
