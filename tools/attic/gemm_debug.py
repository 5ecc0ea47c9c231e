#!/usr/bin/env python3
"""k_noddi_gemm against numpy: C = [A | U]'y per voxel, block-wise layout, ||y||^2"""
import os, sys
os.environ.setdefault('AMX_SEED_MIN_VOXELS', '0')      # (small inputs would take the unseeded kernels: read at context creation)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
_capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3); ctx.sync()
perm = _capi.debug_fetch(ctx, None, 0, (n,), np.int32)
misc = _capi.debug_fetch(ctx, None, 7, (16,), np.int32)
nsc, nblk = int(misc[1]), int(misc[2])
sc = _capi.debug_fetch(ctx, None, 6, (nsc, 4), np.int32)
Cb = _capi.debug_fetch(ctx, None, 5, (nblk, 160, 64), np.float64)
U = _capi.debug_fetch(ctx, lut, 10, (500, 99, 12), np.float64)
li = S.lut_indices(d, ht)
print('seed chunks', nsc, 'blocks', nblk, 'expected blocks', sum((c[2] + 63) // 64 for c in sc))
err = 0.0; erry = 0.0; erru = 0.0
rng = np.random.default_rng(0)
for c in sc[rng.integers(0, nsc, 40)]:
    dsel, start, count, pad = [int(v) for v in c]
    for k in rng.integers(0, count, 6):
        v = perm[start + k]
        A = np.concatenate([K['wm'][:, dsel, :].astype(np.float64).T, K['iso'].astype(np.float64)[:, None]], axis=1)
        row = Cb[pad + (k >> 6), :, k & 63]
        assert li[v] == dsel
        err = max(err, np.abs(A.T @ y[v] - row[:145]).max()); erru = max(erru, np.abs(U[dsel].T @ y[v] - row[146:158]).max()); erry = max(erry, abs(y[v] @ y[v] - row[158]))
print('max |A\'y - C| %.3g   max |U\'y - C| %.3g   max | |y|^2 - C | %.3g' % (err, erru, erry))
