#!/usr/bin/env python3
"""One-million-voxel NODDI fit from host buffers, a few calls (for a rocprofv3 timeline: tools/host_timeline.sh)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
f32 = len(sys.argv) > 2 and sys.argv[2] == 'f32'
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=1)
if f32:
    y = y.astype(np.float32)
ctx = get_context()
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
for i in range(4):
    t = time.perf_counter()
    _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)
    print('call %d: %.2f ms' % (i, 1e3 * (time.perf_counter() - t)), flush=True)
