#!/usr/bin/env python3
"""Warm starts for the NNLS stage from approximate (low-rank / low-precision) solves: polish cost (CPU lab)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
exec(open(os.path.join(os.path.dirname(__file__), 'path_lab2.py')).read().split("rules = {")[0].split("n_vox = int")[0])
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 200
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64); n_wm = 144
def lsq(A, P, y):
    z = np.zeros(A.shape[1])
    if P: z[P] = np.linalg.lstsq(A[:, P], y, rcond=None)[0]
    return z
def solve(A, y, P0=()):
    n = A.shape[1]
    P = list(P0); x = np.zeros(n); adds = len(P); rems = 0
    while P:
        z = lsq(A, P, y); bad = [j for j in P if not z[j] > 0]
        if not bad: x = z; break
        for j in bad: P.remove(j); rems += 1
    banned = set(); steps = 0
    for it in range(500):
        w = A.T @ (y - A @ x)
        ok = np.ones(n, bool); ok[P] = False; ok[list(banned)] = False; ok &= w > 0
        if not ok.any(): break
        t = int(np.argmax(np.where(ok, w, -np.inf)))
        P.append(t); adds += 1; steps += 1
        z = lsq(A, P, y)
        if not z[t] > 0: P.remove(t); adds -= 1; banned.add(t); continue
        banned.clear()
        while True:
            neg = [j for j in P if not z[j] > 0]
            if not neg: x = z; break
            ratios = {j: x[j] / (x[j] - z[j]) for j in neg}; al = min(ratios.values())
            x = x + al * (z - x)
            for j in [j for j in neg if ratios[j] <= al]: P.remove(j); x[j] = 0.0; rems += 1
            z = lsq(A, P, y)
    return x, adds, rems, steps
res = {}
cache = {}
for v in range(n_vox):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
    if lut[v] not in cache: cache[lut[v]] = np.linalg.svd(A, full_matrices=False)
    U, s, Vt = cache[lut[v]]
    x0, a, r, st = solve(A, y[v]); res.setdefault('cold', []).append((a, r, st, 0, 1))
    P1 = set(np.nonzero(x0 > 0)[0])
    for k in (4, 6, 8, 10, 13, 20):
        Ak = (U[:, :k] * s[:k]) @ Vt[:k]
        xk, _ = sp_nnls(Ak, y[v], maxiter=5000)
        P0 = [int(j) for j in np.argsort(-xk)[:8] if xk[j] > 0]
        x, a, r, st = solve(A, y[v], P0)
        res.setdefault('rank %d' % k, []).append((a, r, st, len(P0), float(set(P0) == P1)))
    # float32-rounded exact problem
    xk, _ = sp_nnls(A.astype(np.float32).astype(np.float64), (y[v] * (1 + 1e-7 * np.random.default_rng(v).standard_normal(99))), maxiter=5000)
    P0 = [int(j) for j in np.nonzero(xk > 0)[0]]
    x, a, r, st = solve(A, y[v], P0); res.setdefault('y perturbed 1e-7', []).append((a, r, st, len(P0), float(set(P0) == P1)))
for k, v in res.items():
    a = np.array(v, float).mean(axis=0)
    print('%-18s |P0| %.2f exact-support %.0f%%  polish: adds(incl P0) %.2f removals %.2f outer steps %.2f' % (k, a[3], 100 * a[4], a[0], a[1], a[2]))
