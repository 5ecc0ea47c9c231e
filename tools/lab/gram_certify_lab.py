#!/usr/bin/env python3
"""Gram-space certificate (x = G_PP^-1 c_P, duals c_t - G_tP x, no signal pass): accuracy against the exact solution as a
function of the pivot-ratio threshold.  CPU lab."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs); y, d = S.noddi_signals(n, K, ht, sch, seed=5)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64)
rows = []
cache = {}
for v in range(n):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        cache[lut[v]] = (A, A.T @ A)
    A, G = cache[lut[v]]
    x, _ = nnls(A, y[v], maxiter=5000); P = np.nonzero(x > 0)[0]
    if not len(P): continue
    c = A.T @ y[v]
    L = np.linalg.cholesky(G[np.ix_(P, P)])
    xg = np.linalg.solve(L.T, np.linalg.solve(L, c[P]))
    xq = np.linalg.lstsq(A[:, P], y[v], rcond=None)[0]
    piv = np.diag(L); ratio = piv.min() / piv.max()
    u = c - G[:, P] @ xg; u[P] = -1
    ue = A.T @ (y[v] - A[:, P] @ xq); ue[P] = -1
    rows.append((ratio, np.abs(xg - xq).max() / np.abs(xq).max(), np.abs(u - ue).max(), (xg > 0).all(), u.max(), ue.max()))
r = np.array(rows, float)
print('voxels', len(r))
for thr in (1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 3e-4, 1e-4, 0):
    sel = r[:, 0] >= thr
    print('pivot ratio >= %-6g: accepted %.1f%%  max rel |dx| %.2e  median %.2e  p99 %.2e  max |du| %.2e' % (thr, 100 * sel.mean(), r[sel, 1].max(), np.median(r[sel, 1]), np.percentile(r[sel, 1], 99), r[sel, 2].max()))
print('largest non-passive dual: exact median %.2e, closest to zero %.2e' % (np.median(r[:, 5]), r[:, 5].max()))
