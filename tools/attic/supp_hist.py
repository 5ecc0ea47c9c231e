#!/usr/bin/env python3
"""histogram of the LASSO seed support sizes and of the stage-1 / stage-3 seed sizes (device, synthetic NODDI voxels)"""
import os, sys
os.environ.setdefault('AMX_SEED_MIN_VOXELS', '0')      # (small inputs would take the unseeded kernels: read at context creation)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
res = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, return_x=True); ctx.sync()
X = res[-1].cpu().numpy()
for st, name in ((0, 'stage-1 NNLS'), (1, 'LASSO'), (2, 'stage-3 NNLS')):
    c = (X[:, st, :] > 0).sum(1)
    h = np.bincount(c, minlength=30)
    print(name, 'support sizes: mean %.2f' % c.mean(), ' cumulative %:', ' '.join('%d:%.1f' % (k, 100 * h[:k + 1].sum() / n) for k in range(0, 30) if h[k]))
