#!/usr/bin/env python3
"""Does splitting one NODDI call over k contexts / streams (each its own workspace, the parts' kernels free to fill each other's tails)
beat the single chain?  usage: split_ab.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y_h, d_h = S.noddi_signals_parallel(n, K, ht, sch, seed=17)
dev = torch.device('cuda', 0)
y = torch.from_numpy(y_h).to(dev); d = torch.from_numpy(d_h).to(dev)
L = _capi.lib()
ref = None
for parts in (1, 2, 3, 4):
    ctxs = [_capi.Context(0) for _ in range(parts)]
    luts = [_capi.upload_noddi(c, K, ht, sch.dwi_idx, False) for c in ctxs]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    est = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    cut = [n * i // parts for i in range(parts + 1)]
    def fit():
        for i in range(parts):
            s, e = cut[i], cut[i + 1]
            ctxs[i].check(L.amx_noddi_fit_device(ctxs[i]._h, luts[i]._h, y[s:e].data_ptr(), d[s:e].data_ptr(), e - s, 0.5, 1e-3, 0,
                                                 est[s:e].data_ptr(), None, None, None, streams[i].cuda_stream))
        for i in range(parts):
            ctxs[i].sync(streams[i].cuda_stream)
    for _ in range(3):
        fit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        fit()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 8
    e = est.cpu().numpy()
    if ref is None:
        ref = e
    print('%d part(s): %d voxels %.3f ms  %.1f M voxels/s  max |dmap| vs one part %.1e' % (parts, n, el * 1e3, n / el / 1e6, np.abs(e - ref).max()))
    del luts, ctxs
