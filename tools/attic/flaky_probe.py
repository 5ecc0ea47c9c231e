#!/usr/bin/env python3
"""fresh context, first call = float32 host path, second = float64 host path (the order of tests/test_gpu_boundary.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amico_amd import _capi, get_context, synthetic as S
n = 900000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=3)
ctx = get_context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
y32 = y.astype(np.float32); y64 = y32.astype(np.float64)
seen = []
ctx.set_progress(lambda done, total: seen.append((done, total)))
e32, r32, _, _ = _capi.noddi_fit(ctx, lut, y32, d, 0.5, 1e-3, 3, rmse=True)
ctx.set_progress(None)
e64, r64, _, _ = _capi.noddi_fit(ctx, lut, y64, d, 0.5, 1e-3, 3, rmse=True)
dd = np.abs(e32 - e64).max(axis=1); bad = np.nonzero(dd > 0)[0]
print('differing voxels', len(bad), 'max', dd.max(), 'first', bad[:10].tolist(), 'rmse differ', int((r32 != r64).sum()), 'stats', ctx.last_stats(), 'progress', seen[:3], len(seen))
if len(bad):
    e3 = _capi.noddi_fit(ctx, lut, y64, d, 0.5, 1e-3, 3, rmse=True)[0]
    print('  third call vs second: differing', int((np.abs(e3 - e64).max(axis=1) > 0).sum()), ' vs first:', int((np.abs(e3 - e32).max(axis=1) > 0).sum()))
    print('  e32', e32[bad[0]], 'e64', e64[bad[0]])
