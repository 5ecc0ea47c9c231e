#!/usr/bin/env python3
"""where the first host-buffer fit of a process spends its time: digest of KERNELS, dictionary upload, first fit (with AMX_HOST_TRACE=1: its timeline)"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
from amico_amd import _capi, synthetic as S
from amico_amd.models import _fingerprint
n = 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=5)
y = y.astype(np.float32).astype(np.float64)
import torch
torch.cuda.init(); torch.zeros(1, device='cuda')
T = lambda: time.perf_counter()
t0 = T(); _fingerprint(K); t1 = T()
ctx = _capi.Context(-1); t2 = T()
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx); t3 = T()
out = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3); t4 = T()
out = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3); t5 = T()
print('PARTS digest %.1f ms | context %.1f | dictionary upload + tables %.1f | first fit %.1f | second fit %.1f' % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
