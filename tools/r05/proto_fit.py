#!/usr/bin/env python3
"""NODDI fit of one acquisition protocol, inputs resident in HBM: voxels/s, HIP-event groups, seed-chain rates.
usage: proto_fit.py <hcp|bench|105|150> [n_voxels] [steps]   (run under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
which = sys.argv[1] if len(sys.argv) > 1 else 'hcp'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
proto = {'hcp': (18, ((1000.0, 90), (2000.0, 90), (3000.0, 90))), 'bench': (9, ((700.0, 30), (2000.0, 60))),
         '105': (5, ((700.0, 50), (2000.0, 50))), '150': (10, ((700.0, 40), (2000.0, 60), (3000.0, 40)))}[which]
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(proto[0], proto[1], seed=4) if which != 'bench' else S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=17)
ctx = _capi.Context(0)
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
est = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
L = _capi.lib(); ctx.set_profiling(True)
stream = torch.cuda.current_stream().cuda_stream
def fit():
    ctx.check(L.amx_noddi_fit_device(ctx._h, lut._h, yt.data_ptr(), dt.data_ptr(), n, 0.5, 1e-3, 0, est.data_ptr(), None, None, None, stream))
for _ in range(2):
    fit(); ctx.sync(stream)
torch.cuda.synchronize(); t0 = time.perf_counter()
kms = np.zeros(10)
for _ in range(steps):
    fit(); ctx.sync(stream)
    for w in (0, 1, 2, 3, 5, 6, 7, 8, 9):
        try: kms[w] += ctx.last_kernel_ms(w)
        except Exception: pass
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
kms /= steps
print('%s: %d volumes, %d voxels: %.3f ms  %.1f M voxels/s | kernels %.3f | groups s1 %.3f s2 %.3f s3 %.3f | left-overs %.3f %.3f %.3f | seed solvers %.3f %.3f'
      % (which, sch.nS, n, 1e3 * el, n / el / 1e6, kms[0], kms[5], kms[6], kms[7], kms[1], kms[2], kms[3], kms[8], kms[9]))
print(ctx.last_seed_stats(), ctx.last_stats())
