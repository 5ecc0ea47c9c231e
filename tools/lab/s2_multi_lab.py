#!/usr/bin/env python3
"""k_lasso_seed's trip rule with NA atoms entering per trip / all non-positive passive atoms leaving at once: trips per voxel and rank-one
changes per voxel (numpy).   usage: s2_multi_lab.py n_vox k snr"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
src = open(os.path.join(os.path.dirname(__file__), 's2_lab.py')).read()
exec(src.split("def seed(")[0])


def trips(Sk, yt, na, allneg, addwithrem):
    kk, n = Sk.shape
    P = np.zeros(n, bool); t_ = 0; ch = 0
    for it in range(300):
        t_ += 1
        SP = Sk[:, P]
        M = lam2 * np.eye(kk) + SP @ SP.T
        w = np.linalg.solve(M, SP @ (SP.T @ yt - lam1))
        t = Sk.T @ (yt - w) - lam1
        neg = np.where(P & (t <= 0))[0]
        cand = np.where(~P, t, -np.inf)
        order = np.argsort(-cand)
        if len(neg):
            if allneg: P[neg] = False; ch += len(neg)
            else: P[neg[np.argmin(t[neg])]] = False; ch += 1
            for q in order[:addwithrem]:
                if cand[q] > 1e-9: P[q] = True; ch += 1
        elif cand[order[0]] > 1e-9:
            for q in order[:na]:
                if cand[q] > 1e-9: P[q] = True; ch += 1
        else:
            break
    return P, t_, ch


rules = {'kernel (2 enter | 1 leaves + 1 enters)': (2, False, 1), '3 enter': (3, False, 1), '4 enter': (4, False, 1),
         '2 enter, all neg leave + 1': (2, True, 1), '3 enter, all neg leave + 1': (3, True, 1), '3 enter | 1 leaves + 2 enter': (3, False, 2),
         '2 enter | 1 leaves + 2 enter': (2, False, 2), '1 enter (classic)': (1, False, 1)}
cache = {}; res = {r: [] for r in rules}
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
    for name, (na, allneg, awr) in rules.items():
        P, t_, ch = trips(S2, yt, na, allneg, awr)
        res[name].append(((P == P2).all(), t_, P.sum(), ch))
print('n', n_vox, 'k', k, 'snr', snr)
for name, r in res.items():
    a = np.array(r, float)
    print('%-40s exact %.1f%%  trips mean %.2f p95 %.0f max %.0f  |P| %.2f  changes %.2f' % (
        name, 100 * a[:, 0].mean(), a[:, 1].mean(), np.percentile(a[:, 1], 95), a[:, 1].max(), a[:, 2].mean(), a[:, 3].mean()))
