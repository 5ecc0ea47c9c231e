#!/usr/bin/env python3
"""Is k_fw_project limited by the random gather of signal rows?  Same fit with (a) random directions (rows gathered in
bucket order), (b) directions sorted by LUT index beforehand (rows of a bucket contiguous in memory)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402

n = 2_000_000
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)
sch = S.make_scheme(1, ((1000.0, 64),), seed=3)
K = S.freewater_kernels(sch, dirs)
y, d = S.freewater_signals(n, K, ht, sch, seed=1)
ctx = get_context()
lut = _capi.upload_freewater(ctx, K, ht)
dev = torch.device('cuda', 0)
idx = S.lut_indices(d, ht)
order = np.argsort(idx, kind='stable')
for name, yy, dd in (('random order', y, d), ('sorted by orientation', y[order], d[order])):
    yt, dt = torch.from_numpy(np.ascontiguousarray(yy)).to(dev), torch.from_numpy(np.ascontiguousarray(dd)).to(dev)
    for it in range(3):
        _capi.freewater_fit_device(ctx, lut, yt, dt, 0.0, 1e-3, False)
    ctx.sync()
    t = time.perf_counter()
    for it in range(10):
        _capi.freewater_fit_device(ctx, lut, yt, dt, 0.0, 1e-3, False)
    ctx.sync()
    print('%-24s %.3f ms per fit' % (name, 1e2 * (time.perf_counter() - t)))
