#!/usr/bin/env python3
"""Timeline of the host-buffer NODDI fit (AMX_HOST_TRACE=1): where the 18 ms (float64) / 12 ms (float32) of a 1 M-voxel call go.
usage: AMX_HOST_TRACE=1 python tools/r05/host_trace.py [n_voxels]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=17)
ctx = _capi.Context(0)
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
# (the synthetic signals are float32 values in float64 buffers, like evaluation.y of the reference: core.py:136, 451)
for name, yy in (('float64 holding float32 values', y), ('genuine float64 values', y * (1.0 + 2.0 ** -30)), ('float32', y.astype(np.float32))):
    ts = []
    for rep in range(calls):
        sys.stderr.write('--- %s call %d\n' % (name, rep)); sys.stderr.flush()
        t0 = time.perf_counter(); e = _capi.noddi_fit(ctx, lut, yy, d, 0.5, 1e-3, 3)[0]; ts.append(time.perf_counter() - t0)
    print('%s: %d voxels, calls %s ms -> median %.2f ms = %.1f M voxels/s (%d batches as float32)' % (name, n, ' '.join('%.2f' % (1e3 * t) for t in ts), 1e3 * np.median(ts[1:]), n / np.median(ts[1:]) / 1e6, ctx.last_host_narrowed()), flush=True)
