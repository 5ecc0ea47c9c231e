#!/usr/bin/env python3
"""Do two half-size FreeWater fits on two HIP streams (two contexts) overlap?  (projection = HBM-bound, solver = VALU-bound)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(1, ((1000.0, 64),), seed=3); K = S.freewater_kernels(sch, dirs)
y, d = S.freewater_signals(n, K, ht, sch, seed=1)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
ctxs = [_capi.Context() for _ in range(parts)]
luts = [_capi.upload_freewater(c, K, ht) for c in ctxs]
streams = [torch.cuda.Stream() for _ in range(parts)]
def whole():
    r = _capi.freewater_fit_device(ctxs[0], luts[0], yt, dt, 0.0, 1e-3, False); torch.cuda.synchronize(); return r[0]
def split():
    out = []; h = n // parts
    for p in range(parts):
        i, j = p * h, (n if p == parts - 1 else (p + 1) * h)
        out.append(_capi.freewater_fit_device(ctxs[p], luts[p], yt[i:j], dt[i:j], 0.0, 1e-3, False, stream=streams[p].cuda_stream)[0])
    torch.cuda.synchronize()
    return torch.cat(out)
for f, name in ((whole, 'one fit'), (split, '%d concurrent fits' % parts)):
    for _ in range(3): r = f()
    t0 = time.perf_counter()
    for _ in range(10): r = f()
    dt_ = (time.perf_counter() - t0) / 10
    print('%-20s %.3f ms  %.2f G voxels/s' % (name, dt_ * 1e3, n / dt_ / 1e9))
    if name == 'one fit': ref = r
print('max |difference|', float((r - ref).abs().max()))
