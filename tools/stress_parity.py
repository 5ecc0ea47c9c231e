#!/usr/bin/env python3
"""Large-sample parity: HIP path vs the CPU oracle on N synthetic voxels (NODDI / FreeWater / SANDI).
Prints the distribution of max|dmap| and the solver statistics.  usage: stress_parity.py [N] [seed]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402
from oracle import oracle                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 123
cores = os.cpu_count()
ctx = get_context()
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)


def report(name, got, ref):
    d = np.abs(got - ref).max(axis=1)
    print(f'{name}: n={len(d)} max={d.max():.3e} median={np.median(d):.2e} '
          f'>1e-8: {(d > 1e-8).sum()} >1e-6: {(d > 1e-6).sum()} >1e-4: {(d > 1e-4).sum()}  stats={ctx.last_stats()}', flush=True)
    return d


sch = S.make_scheme(seed=seed)
K = S.noddi_kernels(sch, dirs)
for snr in (30.0, 10.0):
    y, d = S.noddi_signals(n, K, ht, sch, seed=seed + 1, snr=snr)
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    est, _, _, _ = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)
    ref = oracle.noddi_fit(y, d, K, ht, sch.dwi_idx, nthreads=cores)
    dd = report(f'NODDI snr={snr:g}', est, ref['estimates'])
    bad = np.where(dd > 1e-6)[0][:5]
    for b in bad:
        print('   voxel', b, 'gpu', est[b], 'oracle', ref['estimates'][b])
fs = S.make_scheme(1, ((1000.0, 64),), seed=seed)
KF = S.freewater_kernels(fs, dirs)
yf, df = S.freewater_signals(n, KF, ht, fs, seed=seed + 2)
lf = _capi.upload_freewater(ctx, KF, ht)
ef, _, _, _ = _capi.freewater_fit(ctx, lf, yf, df, 0.0, 1e-3, False)
report('FreeWater', ef, oracle.freewater_fit(yf, df, KF, ht, nthreads=cores)['estimates'])
avg = S.directional_average_scheme(S.make_sandi_scheme())
KS, Rs, din, diso = S.sandi_kernels(avg)
ys = S.sandi_signals(n, KS, avg, seed=seed + 3)
ls = _capi.upload_sandi(ctx, KS, Rs, din, diso)
es, _, _ = _capi.sandi_fit(ctx, ls, ys, 0.0, 5e-3)
rs = oracle.sandi_fit(ys, KS, Rs, din, diso, nthreads=cores)['estimates']
report('SANDI fractions', es[:, :3], rs[:, :3])
report('SANDI Rsoma/Din/De', es[:, 3:], rs[:, 3:])
