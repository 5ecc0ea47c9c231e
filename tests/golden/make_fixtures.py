#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONLY in the build container).

What comes from /root/reference (read-only, never shipped):
  * data files amico/directions/htable_ndirs=500.bin and ndirs=500.bin -> copied as
    fixture DATA (htable500.npz);
  * the importable pure-Python physics amico.synthesis / amico.scheme (numpy+scipy only),
    used to synthesise realistic dictionary atoms.  The LUT slice of orientation d_k is
    obtained by rotating the gradient table so that d_k maps to z (the reference's
    generators assume a z-aligned fibre, synthesis.py:500-509, 766-776) instead of the
    SH rotation of lut.pyx (needs dipy, absent here).
The hot path itself (models.pyx -> cyspams) can NOT be imported or compiled here, so the
expected outputs are produced by INDEPENDENT THIRD-PARTY solvers following models.pyx:
  stage NNLS      scipy.optimize.nnls            (models.pyx:911, 940)
  stage LASSO     scipy.optimize.nnls on the augmented system of SURVEY appendix A.3,
                  asserted equal to sklearn ElasticNet(positive=True) at generation time
  glue / maps     numpy restatement of models.pyx:905-967, 1231-1276, 1567-1619
Usage:  python tests/golden/make_fixtures.py
"""
import os
import sys
import types
import numpy as np
from scipy.optimize import nnls as sp_nnls

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
REF = '/root/reference/amico'
m = types.ModuleType('amico')
m.__path__ = [REF]
sys.modules['amico'] = m
import amico.scheme            # noqa: E402
import amico.synthesis as syn  # noqa: E402
from amico_amd import synthetic as S  # noqa: E402


def rot_to_z(d):
    d = d / np.linalg.norm(d)
    z = np.array([0, 0, 1.0])
    v = np.cross(d, z)
    s = np.linalg.norm(v)
    c = d @ z
    if s < 1e-12:
        return np.eye(3) if c > 0 else np.diag([1, -1, -1.0])
    vx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + vx + vx @ vx * ((1 - c) / s ** 2)


def rotated_scheme(table, d):
    t = table.copy()
    t[:, :3] = t[:, :3] @ rot_to_z(d).T
    return amico.scheme.Scheme(t, 0)


def enet_pos(A, y, lam1, lam2):
    """argmin_{x>=0} 1/2|y-Ax|^2 + lam1 sum x + lam2/2 |x|^2 via scipy NNLS (appendix A.3)."""
    n = A.shape[1]
    Ap = np.vstack([A, np.sqrt(lam2) * np.eye(n)])
    yp = np.hstack([y, np.zeros(n)])
    if lam1 != 0.0:
        c = Ap @ np.linalg.solve(Ap.T @ Ap, lam1 * np.ones(n))
        yp = yp - c
    x, _ = sp_nnls(Ap, yp, maxiter=50 * n)
    return x


def enet_sklearn(A, y, lam1, lam2):
    from sklearn.linear_model import ElasticNet
    mrows = A.shape[0]
    en = ElasticNet(alpha=(lam1 + lam2) / mrows, l1_ratio=lam1 / (lam1 + lam2), positive=True,
                    fit_intercept=False, tol=1e-14, max_iter=2000000)
    en.fit(A, y)
    return en.coef_


def pick_voxel_dirs(lut_ids, ref_dirs, htable, n_vox, rng):
    """random directions whose LUT index is one of lut_ids"""
    out = []
    while len(out) < n_vox:
        d = S.random_unit_vectors(20000, rng)
        idx = S.lut_indices(d, htable)
        for k in np.where(np.isin(idx, lut_ids))[0]:
            out.append(d[k])
            if len(out) == n_vox:
                break
    return np.array(out)


def main():
    rng = np.random.default_rng(0)
    htable = np.fromfile(os.path.join(REF, 'directions', 'htable_ndirs=500.bin'), dtype=np.int16)
    ref_dirs = np.fromfile(os.path.join(REF, 'directions', 'ndirs=500.bin'), dtype=np.float64).reshape(500, 3)
    np.savez_compressed(os.path.join(HERE, 'htable500.npz'), htable=htable, dirs=ref_dirs)

    # ------------------------------------------------------------------ NODDI
    sch = S.make_scheme(seed=0)                      # 9 b0 + 30@700 + 60@2000
    table = sch.raw.copy()
    IC_VFs = np.linspace(0.1, 0.99, 12)
    IC_ODs = np.hstack((np.array([0.03, 0.06]), np.linspace(0.09, 0.99, 10)))
    lut_ids = np.array([3, 77, 151, 260, 388, 499])
    n_wm = 144
    wm_s = np.zeros((n_wm, len(lut_ids), sch.nS), dtype=np.float32)
    for li, lid in enumerate(lut_ids):
        rs = rotated_scheme(table, ref_dirs[lid])
        ic = syn.NODDIIntraCellular(rs); ec = syn.NODDIExtraCellular(rs); iso_g = syn.NODDIIsotropic(rs)
        idx = 0
        for od in IC_ODs:
            kappa = 1.0 / np.tan(od * np.pi / 2.0)
            s_ic = ic.get_signal(1.7e-3, kappa)
            for v in IC_VFs:
                s_ec = ec.get_signal(1.7e-3, kappa, v)
                wm_s[idx, li] = (v * s_ic + (1 - v) * s_ec).astype(np.float32)
                idx += 1
    iso = np.squeeze(iso_g.get_signal(3.0e-3)).astype(np.float32)
    kappa = np.array([1.0 / np.tan(od * np.pi / 2.0) for od in IC_ODs for v in IC_VFs], dtype=np.float32)
    icvf = np.array([v for od in IC_ODs for v in IC_VFs], dtype=np.float32)
    dwi_idx = sch.dwi_idx
    # norms: taken at LUT direction 0 in the reference (models.pyx:784); direction 0 is not among
    # the stored slices, so its atoms are synthesised as well (only the norms are kept)
    rs0 = rotated_scheme(table, ref_dirs[0])
    ic0 = syn.NODDIIntraCellular(rs0); ec0 = syn.NODDIExtraCellular(rs0)
    norms = np.zeros((len(dwi_idx), n_wm))
    idx = 0
    for od in IC_ODs:
        kap = 1.0 / np.tan(od * np.pi / 2.0)
        s_ic = ic0.get_signal(1.7e-3, kap)
        for v in IC_VFs:
            a0 = (v * s_ic + (1 - v) * ec0.get_signal(1.7e-3, kap, v)).astype(np.float32)
            norms[:, idx] = 1.0 / np.linalg.norm(a0[dwi_idx])
            idx += 1

    n_vox = 160
    dirs = pick_voxel_dirs(lut_ids, ref_dirs, htable, n_vox, rng)
    lut = S.lut_indices(dirs, htable)
    slot = np.searchsorted(lut_ids, lut)
    y = np.zeros((n_vox, sch.nS))
    for i in range(n_vox):
        k = rng.integers(n_wm); f = rng.uniform(0, 0.5)
        y0 = (1 - f) * wm_s[k, slot[i]].astype(np.float64) + f * iso
        y[i] = S._finish(S._rician(y0[None, :], 30.0, rng), sch)[0]
    # hand-checkable cases appended: exact atom, all-zero, pure iso
    y[0] = wm_s[37, slot[0]].astype(np.float64)
    y[1] = 0.0
    y[2] = iso.astype(np.float64)

    lam1, lam2 = 0.5, 1e-3
    n_atoms = n_wm + 1
    xs = np.zeros((n_vox, 3, n_atoms))
    est = np.zeros((n_vox, 3)); rmse = np.zeros(n_vox); nrmse = np.zeros(n_vox)
    max_sk = 0.0
    for i in range(n_vox):
        A = np.hstack([wm_s[:, slot[i], :].T.astype(np.float64), iso[:, None].astype(np.float64)])
        x1, _ = sp_nnls(A, y[i], maxiter=50 * n_atoms)
        xs[i, 0] = x1
        A2 = A[dwi_idx, :n_wm] * norms
        y2 = np.maximum(0.0, y[i, dwi_idx] - x1[-1] * iso[dwi_idx])
        x2 = enet_pos(A2, y2, lam1, lam2)
        if i < 40:
            max_sk = max(max_sk, np.abs(enet_sklearn(A2, y2, lam1, lam2) - x2).max())
        x = np.zeros(n_atoms); x[:n_wm] = x2; x[-1] = x1[-1]
        xs[i, 1] = x
        x[-1] = 1.0
        pos = np.where(x > 0)[0]
        x3, _ = sp_nnls(A[:, pos], y[i], maxiter=50 * n_atoms)
        x[:] = 0; x[pos] = x3
        xs[i, 2] = x
        s = x.sum() + 1e-16
        swm = (x[:n_wm] / s).sum() + 1e-16
        f1 = (icvf * x[:n_wm] / s / swm).sum()
        f2 = ((1.0 - icvf.astype(np.float64)).astype(np.float32) * x[:n_wm] / s / swm).sum()
        k1 = (kappa * x[:n_wm] / s / swm).sum()
        est[i] = [f1 / (f1 + f2 + 1e-16), 2.0 / np.pi * np.arctan2(1.0, k1), x[-1] / s]
        r = y[i] - A @ x
        rmse[i] = np.sqrt((r ** 2).sum() / sch.nS)
        den = (y[i] ** 2).sum()
        nrmse[i] = np.sqrt((r ** 2).sum() / den) if den > 1e-16 else 0.0
    print('NODDI: max |sklearn - augmented nnls| on stage 2 =', max_sk)
    assert max_sk < 1e-7
    np.savez_compressed(os.path.join(HERE, 'noddi_fixture.npz'),
                        scheme=table, lut_ids=lut_ids, wm_slices=wm_s, iso=iso, norms=norms, icvf=icvf,
                        kappa=kappa, dwi_idx=dwi_idx, y=y, dirs=dirs, lut=lut, lambda1=lam1, lambda2=lam2,
                        x_stages=xs, estimates=est, rmse=rmse, nrmse=nrmse)

    # ------------------------------------------------------------------ FreeWater
    fsch = S.make_scheme(1, ((1000.0, 64),), seed=3)
    ftable = fsch.raw.copy()
    d_perps = np.linspace(0.1, 1.0, 10) * 1e-3
    f_ids = np.array([10, 123, 250, 377, 480])
    D_s = np.zeros((10, len(f_ids), fsch.nS), dtype=np.float32)
    for li, lid in enumerate(f_ids):
        rs = rotated_scheme(ftable, ref_dirs[lid])
        zep = syn.Zeppelin(rs)
        for k, dp in enumerate(d_perps):
            D_s[k, li] = zep.get_signal(1.0e-3, dp).astype(np.float32)
    CSF = syn.Ball(amico.scheme.Scheme(ftable.copy(), 0)).get_signal(2.5e-3).astype(np.float32)[None, :]
    n_fw = 120
    fdirs = pick_voxel_dirs(f_ids, ref_dirs, htable, n_fw, rng)
    flut = S.lut_indices(fdirs, htable)
    fslot = np.searchsorted(f_ids, flut)
    fy = np.zeros((n_fw, fsch.nS))
    for i in range(n_fw):
        k = rng.integers(10); f = rng.uniform(0, 0.5)
        y0 = (1 - f) * D_s[k, fslot[i]].astype(np.float64) + f * CSF[0]
        fy[i] = S._finish(S._rician(y0[None, :], 30.0, rng), fsch)[0]
    fy[0] = 0.0
    fx = np.zeros((n_fw, 11)); fest = np.zeros((n_fw, 2)); fcorr = np.zeros_like(fy)
    for i in range(n_fw):
        A = np.hstack([D_s[:, fslot[i], :].T.astype(np.float64), CSF.T.astype(np.float64)])
        x = enet_pos(A, fy[i], 0.0, 1e-3)
        fx[i] = x
        s = x.sum() + 1e-16
        v = x[:10].sum() / s
        fest[i] = [v, 1.0 - v]
        xi = x.copy(); xi[:10] = 0
        fcorr[i] = np.maximum(0.0, fy[i] - A @ xi)
    np.savez_compressed(os.path.join(HERE, 'freewater_fixture.npz'),
                        scheme=ftable, lut_ids=f_ids, D_slices=D_s, CSF=CSF, y=fy, dirs=fdirs, lut=flut,
                        lambda1=0.0, lambda2=1e-3, x=fx, estimates=fest, y_corrected=fcorr)

    # ------------------------------------------------------------------ SANDI
    full = S.make_sandi_scheme()
    avg = S.directional_average_scheme(full)
    rsch = amico.scheme.Scheme(avg.raw.copy(), 0)
    Rs = np.linspace(1.0, 12.0, 5) * 1e-6
    d_in = np.linspace(0.25, 3.0, 5) * 1e-3
    d_isos = np.linspace(0.25, 3.0, 5) * 1e-3
    sph = syn.SphereGPD(rsch); ast = syn.Astrosticks(rsch); ball = syn.Ball(rsch)
    cols = [sph.get_signal(3.0e-3, R) for R in Rs] + [ast.get_signal(d) for d in d_in] + \
           [ball.get_signal(d) for d in d_isos]
    signal = np.zeros((avg.nS, 15), order='F'); snorms = np.zeros(15)
    for k, c in enumerate(cols):
        c = np.asarray(c, dtype=np.float64).astype(np.float32).astype(np.float64)
        c[avg.b0_idx] = 1.0
        snorms[k] = 1.0 / np.linalg.norm(c)
        signal[:, k] = c * snorms[k]
    KS = {'signal': signal, 'norms': snorms}
    n_sa = 200
    sy = S.sandi_signals(n_sa, KS, avg, seed=5)
    sy[0] = 0.0
    sx = np.zeros((n_sa, 15)); sest = np.zeros((n_sa, 6))
    for i in range(n_sa):
        x = enet_pos(signal, sy[i], 0.0, 5e-3) * snorms
        sx[i] = x
        s = x.sum() + 1e-16
        xs_, xk, xi = x[:5].sum(), x[5:10].sum(), x[10:].sum()
        sest[i] = [xs_ / s, xk / s, xi / s,
                   1e6 * (Rs * x[:5]).sum() / (xs_ + 1e-16),
                   1e3 * (d_in * x[5:10]).sum() / (xk + 1e-16),
                   1e3 * (d_isos * x[10:]).sum() / (xi + 1e-16)]
    np.savez_compressed(os.path.join(HERE, 'sandi_fixture.npz'),
                        scheme=avg.raw, signal=signal, norms=snorms, Rs=Rs, d_in=d_in, d_isos=d_isos,
                        y=sy, lambda1=0.0, lambda2=5e-3, x=sx, estimates=sest)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
