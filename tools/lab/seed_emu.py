#!/usr/bin/env python3
"""numpy emulation of k_nnls_seed's per-lane state machine (same decisions): trips per voxel, outcomes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
KD, MS, tol, cap = 12, int(sys.argv[2]) if len(sys.argv) > 2 else 8, 1e-10, 96
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n_vox, K, ht, sch, seed=5)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64)

def rrqr(A, k):
    R = A.copy(); Q = []
    for _ in range(k):
        nr = (R * R).sum(0); j = int(np.argmax(nr)); q = R[:, j] / np.sqrt(nr[j])
        for _ in range(2):
            for p in Q: q = q - p * (p @ q)
            q /= np.linalg.norm(q)
        Q.append(q); R = R - np.outer(q, q @ R)
    return np.array(Q).T

def seed(Sk, yt):
    idx = []; x = []; trips = 0; last = -1; ban = [-1, -1]; nrej = 0
    while True:
        trips += 1
        npn = len(idx)
        kmin = -1
        if npn:
            SP = Sk[:, idx]; H = SP.T @ SP; c = SP.T @ yt
            try:
                L = np.linalg.cholesky(H); z = np.linalg.solve(L.T, np.linalg.solve(L, c)); piv = True
            except np.linalg.LinAlgError:
                z = np.zeros(npn); piv = False
            alpha = np.inf
            for s in range(npn):
                if not z[s] > 0:
                    den = x[s] - z[s]; ratio = x[s] / den if den > 0 else 0.0
                    if ratio < alpha: alpha = ratio; kmin = s
            if not piv and kmin < 0: kmin = npn - 1; alpha = 0.0
            if kmin >= 0:
                x = [xs + alpha * (zs - xs) for xs, zs in zip(x, z)]
                gone = idx[kmin]
                if gone == last: ban = [gone, ban[0]]; nrej += 1
                del idx[kmin]; del x[kmin]
                continue
            x = list(z); ban = [-1, -1]
        r = yt - (Sk[:, idx] @ np.array(x) if idx else 0)
        w = Sk.T @ r
        w[[b for b in ban if b >= 0]] = -np.inf
        bj = int(np.argmax(w))
        if not w[bj] > tol or bj in idx: return idx, trips, 'kkt', nrej
        if len(idx) >= MS or trips > cap: return idx, trips, 'overflow' if len(idx) >= MS else 'cap', nrej
        idx.append(bj); x.append(0.0); last = bj
        if trips > 2 * cap: return idx, trips, 'cap2', nrej

cache = {}; out = []
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        U = rrqr(A, KD); cache[lut[v]] = (A, U, U.T @ A)
    A, U, Sk = cache[lut[v]]
    idx, trips, why, nrej = seed(Sk, U.T @ y[v])
    x1, _ = sp_nnls(A, y[v], maxiter=5000)
    out.append((trips, why, nrej, set(idx) == set(np.nonzero(x1 > 0)[0])))
tr = np.array([o[0] for o in out])
print('trips mean %.1f p50 %d p95 %d p99 %d max %d' % (tr.mean(), np.median(tr), np.percentile(tr, 95), np.percentile(tr, 99), tr.max()))
for why in ('kkt', 'overflow', 'cap', 'cap2'):
    sel = [o for o in out if o[1] == why]
    if sel: print(why, len(sel), 'mean trips %.1f' % np.mean([o[0] for o in sel]), 'exact %.1f%%' % (100 * np.mean([o[3] for o in sel])), 'rejections/voxel %.2f' % np.mean([o[2] for o in sel]))
