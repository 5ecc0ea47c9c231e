#!/usr/bin/env python3
"""Stage-2 (LASSO) seeds: compressed rank-k problem, heuristic 'greedy add + block removal' in Woodbury form (state: P, M).
usage: s2_lab.py n_vox k"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
snr = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64)
dwi = np.asarray(sch.dwi_idx); norms = K['norms'][0]; lam1, lam2 = 0.5, 1e-3

def rrqr(A, k):
    R = A.copy(); Q = []
    for _ in range(k):
        nr = (R * R).sum(0); j = int(np.argmax(nr)); q = R[:, j] / np.sqrt(nr[j])
        for _ in range(2):
            for p in Q: q = q - p * (p @ q)
            q /= np.linalg.norm(q)
        Q.append(q); R = R - np.outer(q, q @ R)
    return np.array(Q).T

def lasso_exact(A2, y2):
    n = A2.shape[1]
    Aa = np.vstack([A2, np.sqrt(lam2) * np.eye(n)]); ya = np.concatenate([y2, np.zeros(n)])
    cvec = Aa @ np.linalg.solve(Aa.T @ Aa, lam1 * np.ones(n))
    return sp_nnls(Aa, ya - cvec, maxiter=20000)[0]

def seed(Sk, yt, mode):
    """Woodbury form.  mode 'greedy': add the best atom, drop all passive atoms with t <= 0; 'bpp': block exchange"""
    kk, n = Sk.shape
    P = np.zeros(n, bool); steps = 0; ninf = n + 1; backup = 3
    for it in range(200):
        steps += 1
        SP = Sk[:, P]
        M = lam2 * np.eye(kk) + SP @ SP.T
        rhs = SP @ (SP.T @ yt - lam1)
        w = np.linalg.solve(M, rhs)
        t = Sk.T @ (yt - w) - lam1            # passive: lam2 * x_j ; others: dual
        neg = P & (t <= 0); pos = (~P) & (t > 1e-9)
        if mode == 'greedy':
            if neg.any(): P &= ~neg; continue
            if not pos.any(): break
            P[int(np.argmax(np.where(pos, t, -np.inf)))] = True
        else:
            bad = neg | pos; nb = bad.sum()
            if nb == 0: break
            if nb < ninf: ninf = nb; backup = 3; P ^= bad
            elif backup > 0: backup -= 1; P ^= bad
            else: j = int(np.nonzero(bad)[0].max()); P[j] = ~P[j]
    return P, steps

cache = {}; res = {'greedy': [], 'bpp': []}
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        A2 = A[dwi][:, :144] * norms[None, :]
        U2 = rrqr(A2, k); cache[lut[v]] = (A, A2, U2, U2.T @ A2)
    A, A2, U2, S2 = cache[lut[v]]
    x1, _ = sp_nnls(A, y[v], maxiter=5000)
    y2 = np.maximum(y[v][dwi] - x1[144] * iso[dwi], 0.0)
    x2 = lasso_exact(A2, y2); P2 = x2 > 0
    for mode in res:
        P, steps = seed(S2, U2.T @ y2, mode)
        res[mode].append(((P == P2).all(), steps, P.sum()))
print('n', n_vox, 'k', k, 'snr', snr)
for mode, r in res.items():
    a = np.array(r, float)
    print('%-7s exact %.1f%%  steps mean %.1f p95 %.0f max %.0f  |P| %.2f' % (mode, 100 * a[:, 0].mean(), a[:, 1].mean(), np.percentile(a[:, 1], 95), a[:, 1].max(), a[:, 2].mean()))
