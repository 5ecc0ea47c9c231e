#!/usr/bin/env python3
"""Can the Gram-space certificate keep the supports it refuses for conditioning (Cholesky pivot ratio < 1e-3, 4 % of the voxels)?
Corrected semi-normal equations: x <- x + G_PP^-1 A_P'(y - A_P x) with the factor already in the lane's registers; the residual of the
correction uses the signal and the atoms themselves, not the squared-condition products.  Reports, for the refused supports of the
bench mix, the error against the 80-bit solution after 0 .. 3 corrections."""
import os, sys
import numpy as np
from scipy.optimize import nnls
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from amico_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=17)
lut = S.lut_indices(d, ht)
wm, iso = K['wm'], K['iso']
rows = []
for v in range(n):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso.astype(np.float64)[:, None]], axis=1)    # [nS][n_atoms]
    x, _ = nnls(A, y[v], maxiter=3000)
    P = np.nonzero(x > 0)[0]
    if len(P) == 0 or len(P) > 8:
        continue
    AP = A[:, P]
    G = AP.T @ AP; c = AP.T @ y[v]
    L = np.linalg.cholesky(G)
    dg = np.diag(L); ratio = dg.min() / dg.max()
    xq = np.linalg.lstsq(AP, y[v], rcond=None)[0]
    # the truth: the same corrected iteration in 80-bit arithmetic, to convergence
    APl = AP.astype(np.longdouble); yl = y[v].astype(np.longdouble); Gl = APl.T @ APl
    k = len(P); Ll = np.zeros((k, k), np.longdouble)
    for j in range(k):
        Ll[j, j] = np.sqrt(Gl[j, j] - (Ll[j, :j] ** 2).sum())
        for i in range(j + 1, k):
            Ll[i, j] = (Gl[i, j] - (Ll[i, :j] * Ll[j, :j]).sum()) / Ll[j, j]
    def solvel(b):
        z = np.zeros(k, np.longdouble)
        for j in range(k):
            z[j] = (b[j] - (Ll[j, :j] * z[:j]).sum()) / Ll[j, j]
        for j in range(k - 1, -1, -1):
            z[j] = (z[j] - (Ll[j + 1:, j] * z[j + 1:]).sum()) / Ll[j, j]
        return z
    xt = solvel(APl.T @ yl)
    for it in range(6):
        xt = xt + solvel(APl.T @ (yl - APl @ xt))
    qr_err = float(np.abs(xq - xt).max() / np.abs(xt).max())
    xq = xt.astype(np.float64)
    solve = lambda b: np.linalg.solve(L.T, np.linalg.solve(L, b))
    xs = [solve(c)]
    for it in range(3):
        r = y[v] - AP @ xs[-1]
        xs.append(xs[-1] + solve(AP.T @ r))
    err = [np.abs(xx - xq).max() / np.abs(xq).max() for xx in xs]
    step = [np.abs(xs[k + 1] - xs[k]).max() / np.abs(xq).max() for k in range(3)]
    rows.append((ratio, len(P), *err, *step, qr_err))
R = np.array(rows)
for lo, hi in ((1e-3, 1.0), (1e-4, 1e-3), (1e-5, 1e-4), (1e-6, 1e-5), (0.0, 1e-6)):
    m = (R[:, 0] >= lo) & (R[:, 0] < hi)
    if m.sum() == 0:
        continue
    print('pivot ratio [%g, %g): %5d voxels (%.1f %%)  max rel err vs QR after 0/1/2/3 corrections: %.1e %.1e %.1e %.1e   max step sizes %.1e %.1e %.1e   QR (lstsq) itself %.1e'
          % (lo, hi, m.sum(), 100.0 * m.sum() / len(R), *R[m, 2:6].max(axis=0), *R[m, 6:9].max(axis=0), R[m, 9].max()))
