#!/usr/bin/env python3
"""NODDI parity on one synthetic batch: stress_one.py N seed snr  (prints the voxels above 1e-6)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402
from oracle import oracle                                  # noqa: E402
n, seed, snr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ctx = get_context()
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=seed); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=seed + 1, snr=snr)
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
est, _, _, _ = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)
sel = np.arange(n) if len(sys.argv) < 5 else np.array([int(v) for v in sys.argv[4].split(',')])
ref = oracle.noddi_fit(y[sel], d[sel], K, ht, sch.dwi_idx, nthreads=os.cpu_count())['estimates']
dd = np.abs(est[sel] - ref).max(axis=1)
print(os.environ.get('AMICO_AMD_LIB', 'default'), 'max', dd.max(), 'bad', [(int(sel[i]), float(dd[i])) for i in np.where(dd > 1e-6)[0]], ctx.last_stats())
