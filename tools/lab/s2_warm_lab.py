#!/usr/bin/env python3
"""k_lasso_seed's trip rule (the most negative passive atom leaves + the best atom enters, else the two best enter) in numpy, started from
the empty set and from the stage-1 NNLS support: trips per voxel.   usage: s2_warm_lab.py n_vox k snr"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
src = open(os.path.join(os.path.dirname(__file__), 's2_lab.py')).read()
exec(src.split("def seed(")[0])


def trips(Sk, yt, P0):
    kk, n = Sk.shape
    P = P0.copy(); t_ = 0
    for it in range(200):
        t_ += 1
        SP = Sk[:, P]
        M = lam2 * np.eye(kk) + SP @ SP.T
        w = np.linalg.solve(M, SP @ (SP.T @ yt - lam1))
        t = Sk.T @ (yt - w) - lam1
        neg = np.where(P & (t <= 0))[0]
        cand = np.where(~P, t, -np.inf)
        order = np.argsort(-cand)[:2]
        if len(neg):
            P[neg[np.argmin(t[neg])]] = False
            if cand[order[0]] > 1e-9: P[order[0]] = True
        elif cand[order[0]] > 1e-9:
            P[order[0]] = True
            if cand[order[1]] > 1e-9: P[order[1]] = True
        else:
            break
    return P, t_


cache = {}; res = {'empty': [], 'stage1': [], 'stage1+nb': []}
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        A2 = A[dwi][:, :144] * norms[None, :]
        U2 = rrqr(A2, k); cache[lut[v]] = (A, A2, U2, U2.T @ A2)
    A, A2, U2, S2 = cache[lut[v]]
    x1, _ = sp_nnls(A, y[v], maxiter=5000)
    y2 = np.maximum(y[v][dwi] - x1[144] * iso[dwi], 0.0)
    x2 = lasso_exact(A2, y2); P2 = x2 > 0
    yt = U2.T @ y2
    P1 = x1[:144] > 0
    Pn = P1.copy()
    for j in np.where(P1)[0]:                  # + the grid neighbours of the stage-1 atoms (12 x 12 grid)
        for dj in (-12, 12, -1, 1):
            q = j + dj
            if 0 <= q < 144: Pn[q] = True
    for name, P0 in (('empty', np.zeros(144, bool)), ('stage1', P1), ('stage1+nb', Pn)):
        P, t_ = trips(S2, yt, P0)
        res[name].append(((P == P2).all(), t_, P.sum(), P0.sum(), (P0 & ~P2).sum()))
print('n', n_vox, 'k', k, 'snr', snr)
for name, r in res.items():
    a = np.array(r, float)
    print('%-10s exact %.1f%%  trips mean %.2f p95 %.0f max %.0f  |P| %.2f  |P0| %.2f  P0 not in final %.2f' % (
        name, 100 * a[:, 0].mean(), a[:, 1].mean(), np.percentile(a[:, 1], 95), a[:, 1].max(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean()))
