#!/usr/bin/env python3
"""All three NODDI stages: lane-style approximate active set in a rank-k compressed space (SVD or pivoted-QR basis) vs the
exact supports.  CPU lab.  usage: three_lab.py n_vox snr k basis(svd|rrqr)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 300
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
basis = sys.argv[4] if len(sys.argv) > 4 else 'svd'
k2 = int(sys.argv[5]) if len(sys.argv) > 5 else k
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=3, snr=snr)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64); n_wm = 144
dwi = np.asarray(sch.dwi_idx); norms = K['norms'][0]
lam1, lam2 = 0.5, 1e-3

def rrqr_basis(A, k):
    R = A.copy(); Q = []
    for _ in range(k):
        nr = (R * R).sum(0); j = int(np.argmax(nr))
        q = R[:, j] / np.sqrt(nr[j])
        for _ in range(2):
            for p in Q: q = q - p * (p @ q)
            q /= np.linalg.norm(q)
        Q.append(q); R = R - np.outer(q, q @ R)
    return np.array(Q).T

def get_basis(A, k):
    if basis == 'svd': return np.linalg.svd(A, full_matrices=False)[0][:, :k]
    return rrqr_basis(A, k)

def gram_as(H, c, tol, maxp, allowed=None, f32=True, Sk=None, yt=None, lam1_=0.0, lam2_=0.0):
    """Lawson-Hanson on (H, c): min 1/2 x'Hx - c'x, x >= 0; passive systems by Cholesky; returns P, steps, maxnp, status"""
    n = len(c); P = []; x = np.zeros(n); steps = 0; maxnp = 0; banned = set()
    H32 = H.astype(np.float32)
    for it in range(400):
        if Sk is not None:
            r = yt - Sk[:, P] @ x[P]
            w = (Sk.astype(np.float32).T @ r.astype(np.float32)).astype(np.float64) - lam1_ - lam2_ * x
        else:
            w = (c.astype(np.float32) - H32[:, P] @ x[P].astype(np.float32)).astype(np.float64) if (f32 and P) else c - H[:, P] @ x[P]
        ok = np.ones(n, bool); ok[P] = False; ok[list(banned)] = False
        if allowed is not None: ok &= allowed
        ok &= w > tol
        if not ok.any(): return P, x, steps, maxnp, 0
        t = int(np.argmax(np.where(ok, w, -np.inf)))
        if len(P) >= maxp: return P, x, steps, maxnp, 1
        P.append(t); steps += 1; maxnp = max(maxnp, len(P))
        try: s = np.linalg.solve(H[np.ix_(P, P)], c[P]); np.linalg.cholesky(H[np.ix_(P, P)])
        except np.linalg.LinAlgError: s = None
        if s is None or not s[-1] > 0: P.pop(); banned.add(t); continue
        z = np.zeros(n); z[P] = s; banned.clear()
        while True:
            neg = [j for j in P if not z[j] > 0]
            if not neg: x = z; break
            ratios = {j: x[j] / (x[j] - z[j]) for j in neg}; al = min(ratios.values())
            x = x + al * (z - x)
            for j in [j for j in neg if ratios[j] <= al]: P.remove(j); x[j] = 0.0
            steps += 1; z = np.zeros(n)
            if P: z[P] = np.linalg.solve(H[np.ix_(P, P)], c[P])
    return P, x, steps, maxnp, 3

def lasso_exact(A2, y2):
    n = A2.shape[1]
    Aa = np.vstack([A2, np.sqrt(lam2) * np.eye(n)]); ya = np.concatenate([y2, np.zeros(n)])
    cvec = Aa @ np.linalg.solve(Aa.T @ Aa, lam1 * np.ones(n))
    return sp_nnls(Aa, ya - cvec, maxiter=20000)[0]

cache = {}; st = {1: [], 2: [], 3: []}
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        A2 = A[dwi][:, :144] * norms[None, :]
        U = get_basis(A, k); U2 = get_basis(A2, k2)
        cache[lut[v]] = (A, A2, U, U.T @ A, U2, U2.T @ A2)
    A, A2, U, Sk, U2, S2k = cache[lut[v]]
    # ---- stage 1
    x1, _ = sp_nnls(A, y[v], maxiter=5000); P1 = set(np.nonzero(x1 > 0)[0])
    yt = U.T @ y[v]
    P0, xa, steps, maxnp, status = gram_as(Sk.T @ Sk, Sk.T @ yt, 1e-7, 8, None, True, Sk, yt)
    st[1].append((set(P0) == P1, steps, maxnp, status != 0, abs(xa[144] - x1[144])))
    # ---- stage 2 (exact x_iso from stage 1)
    y2 = np.maximum(y[v][dwi] - x1[144] * iso[dwi], 0.0)
    x2 = lasso_exact(A2, y2); P2 = set(np.nonzero(x2 > 0)[0])
    y2t = U2.T @ y2
    P0, xa, steps, maxnp, status = gram_as(S2k.T @ S2k + lam2 * np.eye(144), S2k.T @ y2t - lam1, 1e-7, 20, None, True, S2k, y2t, lam1, lam2)
    st[2].append((set(P0) == P2, steps, maxnp, status != 0, len(P2)))
    # ---- stage 3
    allowed = np.zeros(145, bool); allowed[list(P2)] = True; allowed[144] = True
    idx = np.nonzero(allowed)[0]
    x3s, _ = sp_nnls(A[:, idx], y[v], maxiter=5000); P3 = set(idx[x3s > 0])
    P0, xa, steps, maxnp, status = gram_as(Sk.T @ Sk, Sk.T @ yt, 1e-7, 8, allowed, True, Sk, yt)
    st[3].append((set(P0) == P3, steps, maxnp, status != 0, len(P3)))
print('n', n_vox, 'snr', snr, 'k', k, k2, basis)
for s in (1, 2, 3):
    a = np.array(st[s], float)
    print('stage %d: exact support %.1f%%  lane steps mean %.1f p95 %.0f max %.0f  maxnp p95 %.0f p99 %.0f max %d  fail %.1f%%  last col mean %.3g' %
          (s, 100 * a[:, 0].mean(), a[:, 1].mean(), np.percentile(a[:, 1], 95), a[:, 1].max(), np.percentile(a[:, 2], 95), np.percentile(a[:, 2], 99), a[:, 2].max(), 100 * a[:, 3].mean(), a[:, 4].mean()))
