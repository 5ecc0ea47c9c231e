#!/usr/bin/env python3
"""Same input twice through the host-buffer path and the device path: results must be bit-identical."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=3)
y = y.astype(np.float32).astype(np.float64)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
a = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
b = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
c = _capi.noddi_fit(ctx, lut, y.astype(np.float32), d, 0.5, 1e-3, 3, rmse=True)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
e = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, rmse=True); ctx.sync()
f = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, rmse=True); ctx.sync()
def cmp(name, p, q):
    dd = np.abs(p - q).max(axis=1) if p.ndim > 1 else np.abs(p - q)
    bad = np.nonzero(dd > 0)[0]
    print('%-28s differing voxels %d  max %.3g  first %s' % (name, len(bad), dd.max(), bad[:8].tolist()))
cmp('host f64 vs host f64', a[0], b[0]); cmp('host f64 vs host f32', a[0], c[0])
cmp('device vs device', e[0].cpu().numpy(), f[0].cpu().numpy()); cmp('host vs device', a[0], e[0].cpu().numpy())
cmp('rmse host vs host', a[1], b[1]); cmp('rmse host vs device', a[1], e[1].cpu().numpy())
