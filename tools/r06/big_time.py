#!/usr/bin/env python3
"""NODDI with lambda1 = 0 (dense LASSO optimum: k_noddi_lasso_big for every voxel): fit time, support sizes, parity sample.  usage: big_time.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=5)
ctx = _capi.Context(0); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt, dt = torch.from_numpy(y).cuda(), torch.from_numpy(d).cuda()
for lam1 in (0.0, 0.05, 0.5):
    for _ in range(2):
        t0 = time.perf_counter()
        est = _capi.noddi_fit_device(ctx, lut, yt, dt, lam1, 1e-3, 3)[0]
        ctx.sync(); el = time.perf_counter() - t0
    ref = oracle.noddi_fit(y[:4000], d[:4000], K, ht, sch.dwi_idx, lambda1=lam1, lambda2=1e-3, nthreads=os.cpu_count() or 1)
    diff = np.abs(est[:4000].cpu().numpy() - ref['estimates']).max(axis=1)
    print('lambda1 %.3f: %d voxels %.2f ms  %.2f M voxels/s | max |dmap| on 4000 voxels %.1e | %s | %s' % (lam1, n, 1e3 * el, n / el / 1e6, diff.max(), ctx.last_stats(), ctx.last_path()[-120:]), flush=True)
