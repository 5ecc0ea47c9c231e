#!/usr/bin/env python3
"""Golden fixture of the CylinderZeppelinBall fit (models.pyx:375-652); run ONLY in the build container.

Like make_fixtures.py: the dictionary atoms come from the reference's importable physics (amico.synthesis CylinderGPD /
Zeppelin / Ball on a STEJSKALTANNER scheme, gradients rotated so that the LUT direction maps to z), the expected
coefficients from an independent third-party solver (scipy NNLS on the augmented system = the non-negative ridge of
cyspams lasso with lambda1 = 0, lambda2 = 4), the maps from a numpy restatement of models.pyx:616-633."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as MF            # noqa: E402  (registers the bare `amico` namespace, helpers)
import amico.scheme                   # noqa: E402
import amico.synthesis as syn         # noqa: E402
from amico_amd import synthetic as S  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    REF = '/root/reference/amico'
    htable = np.fromfile(os.path.join(REF, 'directions', 'htable_ndirs=500.bin'), dtype=np.int16)
    ref_dirs = np.fromfile(os.path.join(REF, 'directions', 'ndirs=500.bin'), dtype=np.float64).reshape(500, 3)
    sch = S.make_sandi_scheme(bvals=(1000., 2000., 3000.), ndir_per_shell=30, n_b0=6, seed=2)     # 96 volumes, 7 columns
    table = sch.raw.copy()
    Rs = np.concatenate(([0.01], np.linspace(0.5, 8.0, 20))) * 1e-6          # models.pyx:407 defaults
    d_perps = np.array([1.19e-3, 0.85e-3, 0.51e-3, 0.17e-3])
    d_isos = np.array([2.0e-3])
    d_par = 0.6e-3
    ids = np.array([5, 140, 260, 391, 470])
    nS = sch.nS
    wmr = np.zeros((len(Rs), len(ids), nS), dtype=np.float32)
    wmh = np.zeros((len(d_perps), len(ids), nS), dtype=np.float32)
    b0 = np.asarray(sch.b0_idx)
    for li, lid in enumerate(ids):
        rs = MF.rotated_scheme(table, ref_dirs[lid])
        rs_table = rs.raw.copy() if hasattr(rs, 'raw') else None
        cyl = syn.CylinderGPD(rs); zep = syn.Zeppelin(rs)
        for k, R in enumerate(Rs):
            s = np.asarray(cyl.get_signal(d_par, R), dtype=np.float64)
            s[b0] = 1.0                                                # resample_kernel leaves b0 entries at 1 (lut.pyx:298)
            wmr[k, li] = s.astype(np.float32)
        for k, dp in enumerate(d_perps):
            s = np.asarray(zep.get_signal(d_par, dp), dtype=np.float64)
            s[b0] = 1.0
            wmh[k, li] = s.astype(np.float32)
    ball = syn.Ball(amico.scheme.Scheme(table.copy(), 0))
    iso = np.zeros((len(d_isos), nS), dtype=np.float32)
    for k, di in enumerate(d_isos):
        s = np.asarray(ball.get_signal(di), dtype=np.float64)
        s[b0] = 1.0
        iso[k] = s.astype(np.float32)
    assert np.isfinite(wmr).all() and np.isfinite(wmh).all() and np.isfinite(iso).all()
    n = 150
    dirs = MF.pick_voxel_dirs(ids, ref_dirs, htable, n, rng)
    lut = S.lut_indices(dirs, htable)
    slot = np.searchsorted(ids, lut)
    n_atoms = len(Rs) + len(d_perps) + len(d_isos)
    y = np.zeros((n, nS))
    for i in range(n):
        k = rng.integers(len(Rs)); h = rng.integers(len(d_perps))
        f = rng.dirichlet([2.0, 2.0, 1.0])
        y0 = f[0] * wmr[k, slot[i]].astype(np.float64) + f[1] * wmh[h, slot[i]].astype(np.float64) + f[2] * iso[0]
        y[i] = S._finish(S._rician(y0[None, :], 30.0, rng), sch)[0]
    y[0] = 0.0
    y[1] = wmr[7, slot[1]].astype(np.float64)
    lam1, lam2 = 0.0, 4.0
    xs = np.zeros((n, n_atoms)); est = np.zeros((n, 3)); rmse = np.zeros(n)
    for i in range(n):
        A = np.hstack([wmr[:, slot[i], :].T.astype(np.float64), wmh[:, slot[i], :].T.astype(np.float64), iso.T.astype(np.float64)])
        x = MF.enet_pos(A, y[i], lam1, lam2)
        xs[i] = x
        f1 = x[:len(Rs)].sum(); f2 = x[len(Rs):len(Rs) + len(d_perps)].sum() + 1e-16
        v = f1 / (f1 + f2 + 1e-16)
        f1 += 1e-16
        a = 1e6 * 2.0 * (Rs * x[:len(Rs)]).sum() / f1
        d = (4.0 * v) / (np.pi * a ** 2.0 + 1e-16)
        est[i] = [v, a, d]
        r = y[i] - A @ x
        rmse[i] = np.sqrt((r ** 2).sum() / nS)
    np.savez_compressed(os.path.join(HERE, 'czb_fixture.npz'), scheme=table, lut_ids=ids, wmr_slices=wmr, wmh_slices=wmh,
                        iso=iso, Rs=Rs, d_perps=d_perps, d_isos=d_isos, y=y, dirs=dirs, lut=lut, lambda1=lam1, lambda2=lam2,
                        x=xs, estimates=est, rmse=rmse)
    print('czb_fixture.npz', os.path.getsize(os.path.join(HERE, 'czb_fixture.npz')), 'support sizes', (xs > 0).sum(1).mean(), (xs > 0).sum(1).max())


if __name__ == '__main__':
    main()
