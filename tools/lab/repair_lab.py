#!/usr/bin/env python3
"""Would a lane-per-voxel REPAIR in the NNLS Gram certificate pay?  For the voxels whose compressed-space seed is not the full
problem's support: add the most violating atom (exact dual values), solve, drop non-positive coefficients, test again -- how many
rounds until the Kuhn-Tucker conditions hold (numpy; seed solver = tools/lab/seed_emu.py's emulation)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
src = open(os.path.join(os.path.dirname(__file__), 'seed_emu.py')).read().split("cache = {}; out = []")[0]
src = src.replace("sch = S.make_scheme(seed=0)", "sch = S.make_scheme(18, ((1000.0, 90), (2000.0, 90), (3000.0, 90)), seed=4) if os.environ.get('PROTO') == 'hcp' else S.make_scheme(seed=0)")
exec(src)
cache = {}
hist = {}
n_bad = 0
for v in range(n_vox):
    if lut[v] not in cache:
        A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
        U = rrqr(A, KD); cache[lut[v]] = (A, U, U.T @ A)
    A, U, Sk = cache[lut[v]]
    idx, trips, why, nrej = seed(Sk, U.T @ y[v])
    if why != 'kkt': continue
    P = list(idx)
    rounds = 0; outcome = None
    for rounds in range(0, 6):
        x = np.linalg.lstsq(A[:, P], y[v], rcond=None)[0] if P else np.zeros(0)
        if P and not (x > 0).all():
            if rounds == 0: outcome = 'seed x<=0'; 
            # drop the non-positive ones (all at once) and go on
            P = [p for p, xv in zip(P, x) if xv > 0]
            continue
        r = y[v] - (A[:, P] @ x if P else 0)
        w = A.T @ r
        w[P] = -np.inf
        j = int(np.argmax(w))
        if not w[j] > 1e-10:
            outcome = 'ok after %d' % rounds; break
        P.append(j)
    else:
        outcome = 'not within 6'
    hist[outcome] = hist.get(outcome, 0) + 1
for k in sorted(hist): print('%-16s %6d  %.2f %%' % (k, hist[k], 100.0 * hist[k] / n_vox))
