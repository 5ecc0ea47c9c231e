#!/usr/bin/env python3
"""Emulation of a lane-per-voxel approximate NNLS in a rank-k compressed space (Gram-space Cholesky solves, fp64 or fp32
dual scan), and the cost of polishing its support with the exact A-space Lawson-Hanson.  CPU lab."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 300
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
KS = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 10, 12, 16]
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64); n_wm = 144
MAXP = 8

def lsq(A, P, y):
    z = np.zeros(A.shape[1])
    if P: z[P] = np.linalg.lstsq(A[:, P], y, rcond=None)[0]
    return z

def exact(A, y, P0=()):
    n = A.shape[1]; P = list(P0); x = np.zeros(n); adds = len(P); rems = 0
    while P:
        z = lsq(A, P, y); bad = [j for j in P if not z[j] > 0]
        if not bad: x = z; break
        for j in bad: P.remove(j); rems += 1
    banned = set(); steps = 0
    for it in range(500):
        w = A.T @ (y - A @ x)
        ok = np.ones(n, bool); ok[P] = False; ok[list(banned)] = False; ok &= w > 0
        if not ok.any(): break
        t = int(np.argmax(np.where(ok, w, -np.inf))); P.append(t); adds += 1; steps += 1
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

def chol_solve(H, c):
    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return None
    return np.linalg.solve(L.T, np.linalg.solve(L, c))

def lane_solver(Sk, yt, f32scan=False, tol=1e-10, allowed=None):
    """LH in compressed space; passive systems on H_PP = S_P'S_P by Cholesky (fp64); returns (P, steps, maxnp, status)"""
    k, n = Sk.shape
    P = []; x = np.zeros(n); steps = 0; maxnp = 0
    S32 = Sk.astype(np.float32)
    banned = set()
    for it in range(200):
        r = yt - Sk[:, P] @ x[P] if P else yt.copy()
        w = (S32.T @ r.astype(np.float32)).astype(np.float64) if f32scan else Sk.T @ r
        ok = np.ones(n, bool); ok[P] = False; ok[list(banned)] = False
        if allowed is not None: ok &= allowed
        ok &= w > tol
        if not ok.any(): return P, steps, maxnp, 0
        t = int(np.argmax(np.where(ok, w, -np.inf)))
        if len(P) >= MAXP: return P, steps, maxnp, 1
        P.append(t); steps += 1; maxnp = max(maxnp, len(P))
        z = np.zeros(n); s = chol_solve(Sk[:, P].T @ Sk[:, P], Sk[:, P].T @ yt)
        if s is None or not s[-1] > 0:
            P.pop(); banned.add(t); continue
        z[P] = s; banned.clear()
        while True:
            neg = [j for j in P if not z[j] > 0]
            if not neg: x = z; break
            ratios = {j: x[j] / (x[j] - z[j]) for j in neg}; al = min(ratios.values())
            x = x + al * (z - x)
            for j in [j for j in neg if ratios[j] <= al]: P.remove(j); x[j] = 0.0
            steps += 1
            z = np.zeros(n)
            if P:
                s = chol_solve(Sk[:, P].T @ Sk[:, P], Sk[:, P].T @ yt)
                if s is None: return P, steps, maxnp, 2
                z[P] = s
    return P, steps, maxnp, 3

cache = {}; res = {}
for v in range(n_vox):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
    if lut[v] not in cache: cache[lut[v]] = np.linalg.svd(A, full_matrices=False)
    U, s, Vt = cache[lut[v]]
    x0, a, r, st = exact(A, y[v]); P1 = set(np.nonzero(x0 > 0)[0])
    res.setdefault('cold', []).append((a, r, st, 0, 1, 0, 0, 0))
    for k in KS:
        for f32 in (False, True):
            Sk = s[:k, None] * Vt[:k]; yt = U[:, :k].T @ y[v]
            P0, steps, maxnp, status = lane_solver(Sk, yt, f32)
            x, a, r, st = exact(A, y[v], P0)
            res.setdefault('k=%d %s' % (k, 'f32scan' if f32 else 'f64'), []).append((a, r, st, len(P0), float(set(P0) == P1), steps, maxnp, float(status != 0)))
print('n', n_vox, 'snr', snr)
for k, v in res.items():
    a = np.array(v, float); m = a.mean(axis=0)
    print('%-14s |P0| %.2f exact %.0f%% | polish adds %.2f rems %.2f outer %.2f | lane steps mean %.1f p95 %.0f max %.0f, maxnp max %d, fail %.1f%%' %
          (k, m[3], 100 * m[4], m[0], m[1], m[2], m[5], np.percentile(a[:, 5], 95), a[:, 5].max(), a[:, 6].max(), 100 * m[7]))
