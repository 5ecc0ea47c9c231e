#!/usr/bin/env python3
"""Stage-1 seeds: Lawson-Hanson (step back) vs greedy (drop the most negative atom, no coefficient vector) on the rank-12
compressed problem: exact-support rate and trips.  CPU lab."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from scipy.optimize import nnls as sp_nnls
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 600
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
KD, MS, tol = 12, 8, 1e-10
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs); y, d = S.noddi_signals(n_vox, K, ht, sch, seed=5, snr=snr)
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
def greedy(Sk, yt, drop='most'):
    idx = []; trips = 0; ban = []
    while True:
        trips += 1
        if trips > 80: return idx, trips, 'cap'
        if idx:
            SP = Sk[:, idx]
            try: z = np.linalg.solve(SP.T @ SP, SP.T @ yt); np.linalg.cholesky(SP.T @ SP)
            except np.linalg.LinAlgError: z = None
            if z is None: ban.append(idx.pop()); continue
            if (z <= 0).any():
                k = int(np.argmin(z)) if drop == 'most' else int(np.nonzero(z <= 0)[0][-1])
                gone = idx.pop(k)
                if k == len(idx): ban.append(gone)       # the newest atom was refused
                continue
            r = yt - SP @ z
        else:
            r = yt
        w = Sk.T @ r; w[idx] = -np.inf; w[ban] = -np.inf
        bj = int(np.argmax(w))
        if not w[bj] > tol: return idx, trips, 'kkt'
        if len(idx) >= MS: return idx, trips, 'overflow'
        idx.append(bj)
        if len(ban) and bj not in ban: pass
        # bans are forgotten once an addition stands (checked at the next feasible solve)
        if len(idx) and False: ban = []
cache = {}; out = []
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        U = rrqr(A, KD); cache[lut[v]] = (A, U, U.T @ A)
    A, U, Sk = cache[lut[v]]
    x1, _ = sp_nnls(A, y[v], maxiter=5000); P1 = set(np.nonzero(x1 > 0)[0])
    idx, trips, why = greedy(Sk, U.T @ y[v])
    out.append((set(idx) == P1, trips, why))
tr = np.array([o[1] for o in out])
print('greedy: exact %.1f%%  trips mean %.1f p95 %d max %d  outcomes' % (100 * np.mean([o[0] for o in out]), tr.mean(), np.percentile(tr, 95), tr.max()),
      {w: sum(o[2] == w for o in out) for w in ('kkt', 'overflow', 'cap')})
