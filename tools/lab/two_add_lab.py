#!/usr/bin/env python3
"""k_nnls_seed<1,8> trip structure in numpy (solve -> at most one removal + solve again -> scan -> append), and what appending
TWO atoms per trip (the best and the runner-up of the dual scan) would do to the number of trips and to the proposed support.
usage: two_add_lab.py [n_vox] [rule]   rule: 0 = one atom, 1 = two whenever the runner-up is positive, 2 = two only while np < 2,
3 = two unless the runner-up is a grid neighbour of the best (same kappa or same v_ic index +-1)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
KD, MS, tol, cap = 12, 8, 1e-10, 28
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
if len(sys.argv) > 2 and sys.argv[2] == 'hard': y, d, _ = S.noddi_hard_signals(n_vox, K, ht, sch, seed=9)
else: y, d = S.noddi_signals(n_vox, K, ht, sch, seed=5)
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

def solve(Sk, idx, yt):
    SP = Sk[:, idx]; H = SP.T @ SP; c = SP.T @ yt
    return np.linalg.lstsq(H, c, rcond=1e-14)[0]

def step(x, z):
    kmin, alpha = -1, np.inf
    for s in range(len(z)):
        if not z[s] > 0:
            den = x[s] - z[s]; ratio = x[s] / den if den > 0 else 0.0
            if ratio < alpha: alpha, kmin = ratio, s
    if kmin < 0: return list(z), -1
    return [xs + alpha * (zs - xs) for xs, zs in zip(x, z)], kmin

def seed(Sk, yt, rule, scale=None):
    idx, x, trips, last, ban = [], [], 0, [], [-1, -1]
    if rule in (7, 8): idx, x = [Sk.shape[1] - 1], [0.0]                      # start with the iso atom passive
    if rule == 8:
        w0 = (Sk.T @ yt) * scale[4]; w0[-1] = -np.inf; b0 = int(np.argmax(w0))
        if w0[b0] > tol: idx.append(b0); x.append(0.0)
    while True:
        trips += 1
        scan = True
        if idx:
            z = solve(Sk, idx, yt)
            x, kmin = step(x, z)
            if kmin < 0: ban = [-1, -1]
            else:
                gone = idx[kmin]
                if gone in last: ban = [gone, ban[0]]
                del idx[kmin]; del x[kmin]
                if idx:
                    z = solve(Sk, idx, yt)
                    x, k2 = step(x, z)
                    if k2 >= 0: scan = False
        if scan:
            r = yt - (Sk[:, idx] @ np.array(x) if idx else 0)
            w = Sk.T @ r
            if rule in (9, 10) and idx:
                Q, _ = np.linalg.qr(Sk[:, idx]); out = Sk - Q @ (Q.T @ Sk); nrm = np.sqrt((out * out).sum(0)); nrm[nrm < 1e-9] = np.inf
                w = np.where(w > tol, w / nrm, w)
            elif rule >= 4: w = np.where(w > tol, w * scale[min(rule, 4) if rule >= 7 else rule], w)
            for b in ban:
                if b >= 0: w[b] = -np.inf
            order = np.argsort(-w)
            bj = int(order[0])
            if not w[bj] > tol or bj in idx: return idx, trips, 'kkt'
            if len(idx) >= MS or trips > cap: return idx, trips, 'noseed'
            if rule in (10, 11) and not idx and bj != Sk.shape[1] - 1 and (Sk[:, -1] @ yt) > tol: idx.append(Sk.shape[1] - 1); x.append(0.0)
            idx.append(bj); x.append(0.0); last = [bj]
            b2 = int(order[1])
            two = rule == 1 or (rule == 2 and len(idx) <= 2) or rule == 3
            if rule == 3 and abs(b2 - bj) in (1, 12): two = False           # (144 wm atoms = 12 x 12 grid)
            if two and w[b2] > tol and b2 not in idx and len(idx) < MS:
                idx.append(b2); x.append(0.0); last = [bj, b2]
        if trips > 2 * cap: return idx, trips, 'noseed'

cache = {}
res = {r: [] for r in (0, 4, 11, 9, 10)}
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        U = rrqr(A, KD); Sk0 = U.T @ A; nr = np.sqrt((Sk0 * Sk0).sum(0)); cache[lut[v]] = Sk0, U, {4: 1.0 / nr, 5: 1.0 / nr ** 2, 6: 1.0 / np.sqrt(nr)}
    Sk, U, scale = cache[lut[v]]
    base = None
    for r in res:
        idx, trips, why = seed(Sk, U.T @ y[v], r, scale)
        if r == 0: base = set(idx)
        res[r].append((trips, why, set(idx) == base))
for r, out in res.items():
    tr = np.array([o[0] for o in out])
    print('rule %d: trips mean %.2f p50 %d p95 %d max %d | no seed %d | same support as rule 0: %.2f%%' % (r, tr.mean(), np.median(tr), np.percentile(tr, 95), tr.max(), sum(o[1] == 'noseed' for o in out), 100 * np.mean([o[2] for o in out])))
