#!/usr/bin/env python3
"""host numpy in -> host numpy out (amx_noddi_fit / _f32): ms per 1 M voxels over the batch plan (AMX_HOST_BATCH, AMX_HOST_RAMP)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=5)
y32 = y.astype(np.float32)
for batch in (131072, 196608, 262144, 393216, 524288):
    for ramp in (0, 65536, 131072):
        os.environ['AMX_HOST_BATCH'] = str(batch); os.environ['AMX_HOST_RAMP'] = str(ramp)
        ctx = _capi.Context(0)
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
        out = []
        for yy in (y, y32):
            _capi.noddi_fit(ctx, lut, yy, d, 0.5, 1e-3, 3)
            ts = []
            for _ in range(4):
                t = time.perf_counter(); _capi.noddi_fit(ctx, lut, yy, d, 0.5, 1e-3, 3); ts.append(time.perf_counter() - t)
            out.append(1e3 * float(np.median(ts)))
        print('batch %7d ramp %6d: f64 %.2f ms (%.1f M voxels/s)  f32 %.2f ms (%.1f M voxels/s)' % (batch, ramp, out[0], n / out[0] / 1e3, out[1], n / out[1] / 1e3), flush=True)
        del lut, ctx
