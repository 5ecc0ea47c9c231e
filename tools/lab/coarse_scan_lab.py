#!/usr/bin/env python3
"""k_nnls_seed<1,8> with a COARSE dual scan first ("a coarse-to-fine dual scan"): the scan is 39 % of a trip; if the entering
atom is taken from a sub-grid of the 12 x 12 (kappa, v_ic) atoms whenever one of THEM has a positive (normalised) dual value, and the full scan
runs only when none has (to find the rest, and to certify the stop), how many trips does the path take, and how many of them need the full scan?
Same trip structure as two_add_lab.py (rule 11 = the shipped entering rule: normalised dual values, iso first).
usage: coarse_scan_lab.py [n_vox] [hard]"""
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
n_atoms = wm.shape[0] + 1
grid = np.arange(144).reshape(12, 12)
SUBSETS = {
    'full': None,
    'every 2nd x 2nd (36 + iso)': np.concatenate([grid[::2, ::2].ravel(), [144]]),
    'every 2nd row (72 + iso)': np.concatenate([grid[::2, :].ravel(), [144]]),
    'every 3rd x 3rd (16 + iso)': np.concatenate([grid[::3, ::3].ravel(), [144]]),
    'first 4 tiles (64 atoms)': np.arange(64),
}

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

def seed(Sk, yt, scale, sub):
    idx, x, trips, last, ban = [], [], 0, [], [-1, -1]
    n_coarse = n_full = 0
    mask = None
    if sub is not None:
        mask = np.zeros(Sk.shape[1], bool); mask[sub] = True
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
            w = np.where(w > tol, w * scale, w)
            for b in ban:
                if b >= 0: w[b] = -np.inf
            bj = -1
            if mask is not None:
                n_coarse += 1
                wc = np.where(mask, w, -np.inf)
                cj = int(np.argmax(wc))
                if wc[cj] > tol and cj not in idx: bj = cj
            if bj < 0:
                n_full += 1
                bj = int(np.argmax(w))
                if not w[bj] > tol or bj in idx: return idx, trips, 'kkt', n_coarse, n_full
            if len(idx) >= MS or trips > cap: return idx, trips, 'noseed', n_coarse, n_full
            if not idx and bj != Sk.shape[1] - 1 and (Sk[:, -1] @ yt) > tol: idx.append(Sk.shape[1] - 1); x.append(0.0)
            idx.append(bj); x.append(0.0); last = [bj]
        if trips > 2 * cap: return idx, trips, 'noseed', n_coarse, n_full

cache = {}
res = {k: [] for k in SUBSETS}
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        U = rrqr(A, KD); Sk0 = U.T @ A; nr = np.sqrt((Sk0 * Sk0).sum(0)); cache[lut[v]] = Sk0, U, 1.0 / nr
    Sk, U, scale = cache[lut[v]]
    base = None
    for k, sub in SUBSETS.items():
        idx, trips, why, nc, nf = seed(Sk, U.T @ y[v], scale, sub)
        if base is None: base = set(idx)
        res[k].append((trips, why, set(idx) == base, nc, nf))
for k, out in res.items():
    tr = np.array([o[0] for o in out]); nc = np.array([o[3] for o in out]); nf = np.array([o[4] for o in out])
    sub = SUBSETS[k]
    frac = 1.0 if sub is None else (np.ceil(len(sub) / 16.0) / 10.0)          # MFMA tiles of the coarse scan against the 10 of the full one
    cost = nf.mean() + frac * nc.mean() if sub is not None else nf.mean()
    print('%-28s trips mean %.2f p95 %d max %d | no seed %d | same support %.2f%% | scans per voxel: coarse %.2f full %.2f -> scan work %.2f full-scan units' % (
        k, tr.mean(), np.percentile(tr, 95), tr.max(), sum(o[1] == 'noseed' for o in out), 100 * np.mean([o[2] for o in out]), nc.mean(), nf.mean(), cost))
