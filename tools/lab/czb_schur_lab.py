#!/usr/bin/env python3
"""CylinderZeppelinBall's ridge problem (models.pyx:615: lasso(lambda1 = 0, lambda2 = 4)) in COMPLEMENTARY form: with H = A'A + l2 I,
M = H^-1 (one 26 x 26 matrix per orientation) and z0 = M A'y (a GEMM over the voxels) the optimum with the atoms Z clamped to zero is
    nu = -M_ZZ^-1 z0_Z,   x = z0 + M[:, Z] nu   (x_Z = 0),   gradient on Z = nu
so block principal pivoting needs a |Z| x |Z| Cholesky per step (|Z| ~ 4 of 26) instead of a |P| x |P| one.  numpy emulation against the
oracle; prints steps / |Z| statistics.   usage: czb_schur_lab.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle
f = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'czb_fixture.npz'), allow_pickle=False))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lam2 = 4.0
rng = np.random.default_rng(1)
wmr, wmh, iso = f['wmr_slices'], f['wmh_slices'], f['iso']
lid = 0
A = np.concatenate([wmr[:, lid], wmh[:, lid], iso], axis=0).astype(np.float64).T        # nS x 26
N = A.shape[1]
w = rng.dirichlet([2.0, 2.0, 1.0], n)
y0 = w[:, :1] * wmr[rng.integers(wmr.shape[0], size=n), lid] + w[:, 1:2] * wmh[rng.integers(wmh.shape[0], size=n), lid] + w[:, 2:] * iso[0]
kinds = rng.integers(0, 3, n)
y = np.abs(y0 + rng.normal(scale=1 / 20.0, size=y0.shape))
y[kinds == 1] = np.abs(rng.normal(size=((kinds == 1).sum(), A.shape[0])))            # pure noise
H = A.T @ A + lam2 * np.eye(N)
M = np.linalg.inv(H)
Z0 = (y @ A) @ M                                                                       # z0 = M A'y
kBackup = 3
steps, zmax, worst = [], [], 0.0
for v in range(n):
    z0 = Z0[v]
    Z = z0 <= 0
    ninf, backup, it = N + 1, 0, 0
    while True:
        idx = np.flatnonzero(Z)
        nu = -np.linalg.solve(M[np.ix_(idx, idx)], z0[idx]) if len(idx) else np.zeros(0)
        x = z0 + M[:, idx] @ nu
        x[idx] = 0.0
        g = np.zeros(N); g[idx] = nu                                                   # gradient on Z (KKT: >= 0)
        bad = (~Z & ~(x > 0)) | (Z & (g < -1e-13))
        nbad = int(bad.sum())
        zmax.append(len(idx))
        if nbad == 0 or it > 4 * N + 16:
            break
        if nbad < ninf:
            ninf, backup, block = nbad, kBackup, True
        elif backup > 0:
            backup -= 1; block = True
        else:
            block = False
        if block:
            Z ^= bad
        else:
            j = np.flatnonzero(bad).max(); Z[j] = ~Z[j]
        it += 1
    steps.append(it + 1)
    xo, _ = oracle.lasso(A, y[v], 0.0, lam2)
    worst = max(worst, np.abs(x - xo).max())
print('voxels %d  max |x - oracle| %.2e  solves per voxel mean %.2f max %d  |Z| mean %.1f max %d' % (n, worst, np.mean(steps), max(steps), np.mean(zmax), max(zmax)))
for k in range(3):
    print(' kind', k, 'solves', np.mean(np.array(steps)[kinds == k]))
