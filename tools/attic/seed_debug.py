#!/usr/bin/env python3
"""Check the pieces of the seed pipeline against numpy: basis, compressed dictionary, projection, seeds."""
import os, sys
os.environ.setdefault('AMX_SEED_MIN_VOXELS', '0')      # (small inputs would take the unseeded kernels: read at context creation)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
from scipy.optimize import nnls
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
U = _capi.debug_fetch(ctx, lut, 10, (500, 99, 12), np.float64)
Sb = _capi.debug_fetch(ctx, lut, 11, (500, 145, 12), np.float64)
A0 = np.concatenate([K['wm'][:, 0, :].astype(np.float64).T, K['iso'].astype(np.float64)[:, None]], axis=1)
print('U orthonormal: max |U\'U - I| =', np.abs(U[0].T @ U[0] - np.eye(12)).max(), ' S = U\'A: max err', np.abs(U[0].T @ A0 - Sb[0].T).max())
print('captured energy: |A - U U\'A| / |A| =', np.linalg.norm(A0 - U[0] @ (U[0].T @ A0)) / np.linalg.norm(A0))
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
res = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, return_x=True)
ctx.sync()
X = res[-1].cpu().numpy()          # [n][3][145]
perm = _capi.debug_fetch(ctx, None, 0, (n,), np.int32)
ytil = _capi.debug_fetch(ctx, None, 1, (n, 12), np.float64)
seeds = _capi.debug_fetch(ctx, None, 2, (n,), np.uint64)     # NOTE: stage-3 seeds (last written)
li = S.lut_indices(d, ht)
err = 0
for p in range(0, n, max(1, n // 50)):
    v = perm[p]; err = max(err, np.abs(U[li[v]].T @ y[v] - ytil[p]).max())
print('projection max err', err)
stage = 0 if os.environ.get('AMX_SEED_STAGES', '3') == '1' else 2
print('seeds of stage', 1 if stage == 0 else 3, ': no-seed fraction', (seeds == np.uint64(0xffffffffffffffff)).mean())
same = 0; tot = 0; shown = 0
for p in range(0, n, max(1, n // 2000)):
    v = perm[p]; sd = int(seeds[p])
    ids = sorted(b for b in ((sd >> (8 * k)) & 0xff for k in range(8)) if b < 0xf0)
    ref = sorted(np.nonzero(X[v, stage] > 0)[0].tolist())
    tot += 1; same += ids == ref
    if ids != ref and shown < 8:
        shown += 1; print('  pos', p, 'vox', v, 'seed', ids, 'exact', ref, hex(sd))
print('seed == exact support: %d / %d' % (same, tot))
# ---- LASSO stage seeds vs the exact stage-2 supports
if os.environ.get('AMX_SEED_STAGES', '7') in ('7', '4', '5', '6'):
    s2 = _capi.debug_fetch(ctx, None, 4, (n, 4), np.uint64)
    same = 0; tot = 0; shown = 0; nos = 0
    for p in range(0, n, max(1, n // 2000)):
        v = perm[p]
        if int(s2[p, 3]) != 0: nos += 1; tot += 1; continue
        ids = [j for j in range(144) if (int(s2[p, j >> 6]) >> (j & 63)) & 1]
        ref = np.nonzero(X[v, 1, :144] > 0)[0].tolist()
        tot += 1; same += ids == ref
        if ids != ref and shown < 6:
            shown += 1; print('  pos', p, 'vox', v, 'seed', ids, 'exact', ref)
    print('LASSO seed == exact support: %d / %d (no seed: %d)' % (same, tot, nos))
