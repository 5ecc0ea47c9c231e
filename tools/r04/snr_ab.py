#!/usr/bin/env python3
"""NODDI fit rate on the bench mix at another noise level (the bench's is SNR 30): do the trip caps, tuned at SNR 30, cost anything on
noisier data?  usage: AMX_SEED_TRIPCAP=... snr_ab.py [n] [snr]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=5, snr=snr)
ctx = _capi.Context(0)
L = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
for _ in range(3):
    out = _capi.noddi_fit_device(ctx, L, yt, dt, 0.5, 1e-3, 3); ctx.sync()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8):
    out = _capi.noddi_fit_device(ctx, L, yt, dt, 0.5, 1e-3, 3); ctx.sync()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
print('caps %s SNR %g: %d voxels %.3f ms  %.1f M voxels/s' % (os.environ.get('AMX_SEED_TRIPCAP', 'default'), snr, n, 1e3 * el, n / el / 1e6), ctx.last_seed_stats())
