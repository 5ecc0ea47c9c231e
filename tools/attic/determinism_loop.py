#!/usr/bin/env python3
"""bitwise repeatability of the NODDI fit over many calls (device path and host-buffer path)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=3)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
ref = None
for r in range(reps):
    e = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, rmse=True, return_x=True); ctx.sync()
    cur = [t.cpu().numpy() for t in (e[0], e[1], e[-1])]
    if ref is None: ref = cur; continue
    for k, nm in enumerate(('estimates', 'rmse', 'x')):
        if not np.array_equal(ref[k], cur[k]):
            dd = np.abs(ref[k] - cur[k]).reshape(n, -1).max(axis=1); bad = np.nonzero(dd > 0)[0]
            print('device rep', r, nm, 'differs on', len(bad), 'voxels, max', dd.max(), bad[:5].tolist())
ref = None
for r in range(reps):
    cur = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)[:2]
    if ref is None: ref = cur; continue
    for k, nm in enumerate(('estimates', 'rmse')):
        if not np.array_equal(ref[k], cur[k]):
            dd = np.abs(ref[k] - cur[k]).reshape(n, -1).max(axis=1); bad = np.nonzero(dd > 0)[0]
            print('host rep', r, nm, 'differs on', len(bad), 'voxels, max', dd.max(), bad[:5].tolist())
print('done', reps)
