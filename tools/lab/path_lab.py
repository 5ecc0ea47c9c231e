#!/usr/bin/env python3
"""CPU laboratory for the active-set PATH of the NODDI stages (not product code).
Counts column additions / removals of Lawson-Hanson variants that all end at the same KKT point, on the
synthetic voxels of the bench.  usage: python tools/lab/path_lab.py [n_vox]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls

n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht)
wm = K['wm']; iso = K['iso'].astype(np.float64)
n_wm = wm.shape[0]; nS = sch.nS


def ls(A, P, y):
    z = np.zeros(A.shape[1])
    if len(P):
        z[P] = np.linalg.lstsq(A[:, P], y, rcond=None)[0]
    return z


def active_set(A, y, P0=(), rule='max', allowed=None, stats=None, lam1=0.0):
    """Lawson-Hanson from a warm start P0 (block removals until feasible), entering rule 'max' | 'norm'."""
    m, n = A.shape
    allowed = np.ones(n, bool) if allowed is None else allowed
    nrm = np.linalg.norm(A, axis=0)
    P = list(P0)
    x = np.zeros(n)
    adds = len(P); rems = 0; solves = 0
    # warm start: block removal until feasible
    while len(P):
        z = ls(A, P, y); solves += 1
        bad = [j for j in P if not z[j] > 0]
        if not bad:
            x = z; break
        for j in bad: P.remove(j); rems += 1
    banned = set()
    for it in range(3 * n + 10):
        r = y - A @ x
        w = A.T @ r
        cand = [j for j in range(n) if allowed[j] and j not in P and j not in banned]
        if not cand: break
        score = w[cand] if rule == 'max' else w[cand] / nrm[cand]
        k = int(np.argmax(score)); t = cand[k]
        if not w[t] > 0: break
        P.append(t); adds += 1
        z = ls(A, P, y); solves += 1
        if not z[t] > 0:
            P.remove(t); adds -= 1; banned.add(t); continue
        banned.clear()
        while True:
            neg = [j for j in P if not z[j] > 0]
            if not neg:
                x = z; break
            al = min(x[j] / (x[j] - z[j]) for j in neg)
            x = x + al * (z - x)
            out = [j for j in P if not x[j] > 1e-300 or (j in neg and x[j] / max(x[j] - z[j], 1e-300) <= al * (1 + 1e-12))]
            if not out: out = [min(neg, key=lambda j: x[j] / (x[j] - z[j]))]
            for j in out: P.remove(j); x[j] = 0.0; rems += 1
            z = ls(A, P, y); solves += 1
    if stats is not None:
        stats.append((adds, rems, solves, len(P)))
    return x


def pair_start(A, y, G, c, iso_j):
    """best {j, iso} pair with both coefficients positive (closed form from Gram entries)"""
    gjj = np.diag(G); gji = G[:, iso_j]; gii = G[iso_j, iso_j]
    det = gjj * gii - gji ** 2
    xj = (gii * c - gji * c[iso_j]) / det
    xi = (gjj * c[iso_j] - gji * c) / det
    gain2 = c * xj + c[iso_j] * xi
    gain1 = c ** 2 / gjj
    ok2 = (xj > 0) & (xi > 0) & (det > 1e-12 * gjj * gii)
    ok2[iso_j] = False
    g2 = np.where(ok2, gain2, -1)
    g1 = np.where(c > 0, gain1, -1)
    j2 = int(np.argmax(g2)); j1 = int(np.argmax(g1))
    if g2[j2] >= g1[j1]: return [j2, iso_j]
    return [j1]


variants = {
    'LH max (shipped rule)': dict(rule='max', start=None),
    'LH normalised entering': dict(rule='norm', start=None),
    'pair start + max': dict(rule='max', start='pair'),
    'pair start + norm': dict(rule='norm', start='pair'),
}
res = {k: [] for k in variants}
err = {k: 0.0 for k in variants}
sub = 0; sizes = []
t0 = time.time()
for v in range(n_vox):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
    xr, _ = sp_nnls(A, y[v], maxiter=2000)
    G = A.T @ A; c = A.T @ y[v]
    for name, cfg in variants.items():
        P0 = pair_start(A, y[v], G, c, n_wm) if cfg['start'] == 'pair' else ()
        x = active_set(A, y[v], P0, cfg['rule'], stats=res[name])
        err[name] = max(err[name], np.abs(A @ x - A @ xr).max())
print('voxels', n_vox, 'snr', snr, 'time %.1fs' % (time.time() - t0))
for name in variants:
    a = np.array(res[name], float)
    print('%-26s adds %.2f rems %.2f solves %.2f |P| %.2f   max|A dx| %.1e' % (name, *a.mean(axis=0), err[name]))
