#!/usr/bin/env python3
"""Parity of the NODDI fit on the protocols off the tuned point, at the call size that selects the round-6 paths (1 M voxels: third
certificate pass per chunk, 8-wavefront left-over builds, rescue pass for > 128 volumes), on seeds no bench or test uses: every
SAMPLE-th voxel against the CPU oracle.  usage: stress_protocols.py [n] [sample_every] [seed]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402
from oracle import oracle                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
ctx = get_context()
print('#', _capi.build_id())
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)
cases = [
    ('105 volumes', 5, ((700.0, 50), (2000.0, 50)), False, None),
    ('150 volumes', 10, ((700.0, 40), (2000.0, 60), (3000.0, 40)), False, None),
    ('181 volumes', 1, ((1000.0, 90), (2500.0, 90)), False, None),
    ('288 volumes', 18, ((1000.0, 90), (2000.0, 90), (3000.0, 90)), False, None),
    ('99 volumes, SNR 50', 9, ((700.0, 30), (2000.0, 60)), False, 50.0),
    ('99 volumes, SNR 10, ex vivo', 9, ((700.0, 30), (2000.0, 60)), True, 10.0),
]
for name, n_b0, shells, exvivo, snr in cases:
    sch = S.make_scheme(n_b0, shells, seed=seed)
    K = S.noddi_kernels(sch, dirs)
    y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=seed + 1, **({} if snr is None else {'snr': snr}))
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx, exvivo)
    t0 = time.perf_counter()
    est = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 4 if exvivo else 3)[0]
    t1 = time.perf_counter()
    pick = np.arange(0, n, every)
    ref = oracle.noddi_fit(np.ascontiguousarray(y[pick]), np.ascontiguousarray(d[pick]), K, ht, sch.dwi_idx, is_exvivo=exvivo, nthreads=os.cpu_count())
    df = np.abs(est[pick] - ref['estimates']).max(axis=1)
    print('%-28s n=%d checked=%d max=%.3e median=%.2e >1e-8: %d >1e-6: %d >1e-4: %d | path %s | %s %s' % (
        name, n, len(pick), df.max(), np.median(df), (df > 1e-8).sum(), (df > 1e-6).sum(), (df > 1e-4).sum(),
        ctx.last_path(), ctx.last_seed_stats(), ctx.last_stats()), flush=True)
    for b in np.where(df > 1e-6)[0][:5]:
        print('   voxel', pick[b], 'gpu', est[pick[b]], 'oracle', ref['estimates'][b])
    lut.close()
