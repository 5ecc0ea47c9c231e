#!/usr/bin/env python3
"""Entering-rule experiments for the NNLS stage (CPU lab).  usage: python tools/lab/path_lab2.py [n_vox] [snr]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls

n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 300
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64); n_wm = 144


def lsq(A, P, y):
    z = np.zeros(A.shape[1])
    if P: z[P] = np.linalg.lstsq(A[:, P], y, rcond=None)[0]
    return z


def solve(A, y, rule, P0=()):
    n = A.shape[1]
    P = list(P0); x = np.zeros(n); adds = len(P); rems = 0
    while P:                                  # warm start: block removals
        z = lsq(A, P, y); bad = [j for j in P if not z[j] > 0]
        if not bad: x = z; break
        for j in bad: P.remove(j); rems += 1
    banned = set()
    for it in range(500):
        r = y - A @ x; w = A.T @ r
        ok = np.ones(n, bool); ok[P] = False; ok[list(banned)] = False; ok &= w > 0
        if not ok.any(): break
        t = rule(A, P, w, ok, x, y)
        P.append(t); adds += 1
        z = lsq(A, P, y)
        if not z[t] > 0:
            P.remove(t); adds -= 1; banned.add(t); continue
        banned.clear()
        while True:
            neg = [j for j in P if not z[j] > 0]
            if not neg: x = z; break
            ratios = {j: x[j] / (x[j] - z[j]) for j in neg}
            al = min(ratios.values())
            x = x + al * (z - x)
            out = [j for j in neg if ratios[j] <= al]
            for j in out: P.remove(j); x[j] = 0.0; rems += 1
            z = lsq(A, P, y)
    return x, adds, rems, len(P)


def r_max(A, P, w, ok, x, y): return int(np.argmax(np.where(ok, w, -np.inf)))
NRM = None
def r_norm(A, P, w, ok, x, y): return int(np.argmax(np.where(ok, w / NRM, -np.inf)))
def r_ols(A, P, w, ok, x, y):
    # exact best improvement: w_j^2 / |a_j perp|^2
    if P:
        Q, _ = np.linalg.qr(A[:, P]); perp = (A * A).sum(0) - ((Q.T @ A) ** 2).sum(0)
    else:
        perp = (A * A).sum(0)
    perp = np.maximum(perp, 1e-30)
    return int(np.argmax(np.where(ok & (perp > 1e-12), w * w / perp, -np.inf)))


def pair_start(G, c, iso_j):
    gjj = np.diag(G); gji = G[:, iso_j]; gii = G[iso_j, iso_j]
    det = gjj * gii - gji ** 2
    with np.errstate(all='ignore'):
        xj = (gii * c - gji * c[iso_j]) / det; xi = (gjj * c[iso_j] - gji * c) / det
    ok2 = (xj > 0) & (xi > 0) & (det > 1e-12 * gjj * gii); ok2[iso_j] = False
    g2 = np.where(ok2, c * xj + c[iso_j] * xi, -1); g1 = np.where(c > 0, c ** 2 / gjj, -1)
    j2 = int(np.argmax(g2)); j1 = int(np.argmax(g1))
    return [j2, iso_j] if g2[j2] >= g1[j1] else [j1]


rules = {'max': r_max, 'norm': r_norm, 'ols': r_ols}
res = {}
for v in range(n_vox):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
    NRM = np.linalg.norm(A, axis=0)
    xr, _ = sp_nnls(A, y[v], maxiter=2000)
    G = A.T @ A; c = A.T @ y[v]
    for name, rule in rules.items():
        for st in ('empty', 'pair'):
            x, a, r, p = solve(A, y[v], rule, pair_start(G, c, n_wm) if st == 'pair' else ())
            res.setdefault((name, st), []).append((a, r, p, np.abs(A @ (x - xr)).max()))
for k, v in res.items():
    a = np.array(v)
    print('%-5s %-6s adds %.2f rems %.2f |P| %.2f  max|A dx| %.1e' % (k[0], k[1], a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].max()))
