#!/usr/bin/env python3
"""NODDI fit rate on a SKEWED orientation distribution (half the voxels in six fibre bundles of 15 degrees spread, half uniform):
orientation populations then differ by an order of magnitude, as in a brain.  usage: skew_ab.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
rng = np.random.default_rng(3)
d = S.random_unit_vectors(n, rng)
centres = S.random_unit_vectors(6, rng)
pick = rng.integers(0, 6, n)
bund = centres[pick] + 0.27 * rng.normal(size=(n, 3))
bund /= np.linalg.norm(bund, axis=1, keepdims=True)
sel = rng.uniform(size=n) < 0.5
d[sel] = bund[sel]
lut = S.lut_indices(d, ht)
c = np.bincount(lut, minlength=500)
print('orientation populations: mean %.0f  max %d  min %d  max/mean %.1f  empty %d' % (c.mean(), c.max(), c.min(), c.max() / c.mean(), (c == 0).sum()))
wm, iso = K['wm'], K['iso'].astype(np.float64)
y = np.empty((n, sch.nS))
for s in range(0, n, 65536):
    e = min(n, s + 65536)
    k = rng.integers(0, wm.shape[0], e - s); f = rng.uniform(0.0, 0.5, e - s)[:, None]
    y0 = (1.0 - f) * wm[k, lut[s:e], :].astype(np.float64) + f * iso[None, :]
    y[s:e] = S._finish(S._rician(y0, 30.0, rng), sch)
ctx = _capi.Context(0)
L = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
for _ in range(3):
    out = _capi.noddi_fit_device(ctx, L, yt, dt, 0.5, 1e-3, 3); ctx.sync()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8):
    out = _capi.noddi_fit_device(ctx, L, yt, dt, 0.5, 1e-3, 3); ctx.sync()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
print('%s: %d voxels %.3f ms  %.1f M voxels/s' % (os.environ.get('AMICO_AMD_LIB', 'default').split('/')[-2] if os.environ.get('AMICO_AMD_LIB') else 'default', n, 1e3 * el, n / el / 1e6), ctx.last_seed_stats())
