#!/usr/bin/env python3
"""LASSO seed solver: check the device's final passive sets against the compressed problem's own KKT conditions (numpy)."""
import os, sys
os.environ.setdefault('AMX_SEED_MIN_VOXELS', '0')      # (small inputs would take the unseeded kernels: read at context creation)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
res = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, return_x=True); ctx.sync()
X = res[-1].cpu().numpy()
perm = _capi.debug_fetch(ctx, None, 0, (n,), np.int32)
y2t = _capi.debug_fetch(ctx, None, 3, (n, 12), np.float64)[:, :8]
s2 = _capi.debug_fetch(ctx, None, 4, (n, 4), np.uint64)
U2 = _capi.debug_fetch(ctx, lut, 12, (500, 99, 12), np.float64)[:, :, :8]
S2 = _capi.debug_fetch(ctx, lut, 13, (500, 144, 12), np.float64)[:, :, :8]
li = S.lut_indices(d, ht); dwi = np.asarray(sch.dwi_idx); norms = K['norms'][0]; iso = K['iso'].astype(np.float64)
lam1, lam2 = 0.5, 1e-3
# basis / projection checks on voxel perm[0]
for p in (0, 10, 20):
    v = perm[p]; A = K['wm'][:, li[v], :].astype(np.float64).T
    A2 = np.zeros((99, 144)); A2[dwi] = A[dwi] * norms[None, :]
    print('pos', p, ': |U2\'U2 - I|', np.abs(U2[li[v]].T @ U2[li[v]] - np.eye(8)).max(), ' |S2 - U2\'A2|', np.abs(U2[li[v]].T @ A2 - S2[li[v]].T).max(),
          ' captured', np.linalg.norm(A2 - U2[li[v]] @ (U2[li[v]].T @ A2)) / np.linalg.norm(A2))
    y2 = np.zeros(99); y2[dwi] = np.maximum(y[v][dwi] - X[v, 0, 144] * iso[dwi], 0)
    print('    projection err', np.abs(U2[li[v]].T @ y2 - y2t[p]).max())
    Sk = S2[li[v]].T; yt_ = y2t[p]
    P = np.array([(int(s2[p, j >> 6]) >> (j & 63)) & 1 for j in range(144)], bool)
    SP = Sk[:, P]; M = lam2 * np.eye(8) + SP @ SP.T
    w = np.linalg.solve(M, SP @ (SP.T @ yt_ - lam1))
    t = Sk.T @ (yt_ - w) - lam1
    print('    seed', np.nonzero(P)[0].tolist(), ' x_P', np.round(t[P] / lam2, 4).tolist())
    print('    largest duals outside P:', [(int(j), float('%.3g' % t[j])) for j in np.argsort(-np.where(P, -9, t))[:4]], ' exact support', np.nonzero(X[v, 1, :144] > 0)[0].tolist())

def dev(Sk, yt):
    k = 8; L = np.sqrt(lam2) * np.eye(k); g = np.zeros(k); P = np.zeros(144, bool); trips = 0; log = []
    while True:
        trips += 1
        w = np.linalg.solve(L.T, np.linalg.solve(L, g)); r = yt - w
        t = Sk.T @ r - lam1
        dj = -1; worst = 0.0
        for j in np.nonzero(P)[0]:
            if t[j] <= worst: worst = t[j]; dj = j
        tt = np.where(P, -np.inf, t); bj = int(np.argmax(tt)); best = tt[bj]
        if dj >= 0: jj, sg = dj, -1.0
        elif best > 1e-9: jj, sg = bj, 1.0
        else: return P, trips, log
        if trips > 64: return P, trips, log
        log.append(('+' if sg > 0 else '-') + str(jj) + '(best %.4g@%d r0 %.6g g0 %.6g t75 %.4g t51 %.4g)' % (best, bj, r[0], g[0], t[75], t[51]))
        v = Sk[:, jj].copy(); cj = v @ yt - lam1; g += sg * cj * v; P[jj] = ~P[jj]
        for j in range(k):
            al, bl = L[j, j], v[j]
            n2 = al * al + sg * bl * bl; rr = np.sqrt(n2); c = rr / al; s = bl / al; L[j, j] = rr
            for i in range(j + 1, k):
                tnew = (L[i, j] + sg * s * v[i]) / c; v[i] = c * v[i] - s * tnew; L[i, j] = tnew
for p in (0, 10, 20):
    v = perm[p]
    P, trips, log = dev(S2[li[v]].T, y2t[p])
    print('host emulation on the device tables, pos', p, ':', np.nonzero(P)[0].tolist(), trips, ' '.join(log))

if os.environ.get('SEED2_TRACE'):
    tr = _capi.debug_fetch(ctx, None, 2, (80, 8), np.float64)
    for row in tr[:24]:
        if row[0] > 0: print('trip %d jj %d sigma %+d best %.4g bj %d dj %d r0 %.6g g0 %.6g' % (row[0], row[1], row[2], row[3], row[4], row[5], row[6], row[7]))
