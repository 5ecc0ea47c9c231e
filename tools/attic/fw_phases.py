#!/usr/bin/env python3
"""Cycle split of k_freewater_refill (diagnosis build: tools/build_variant.sh fwph -DAMX_FW_PHASES, then
AMICO_AMD_LIB=variants/fwph/libamico_amd.so python tools/fw_phases.py)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(dirs)
sch = S.make_scheme(1, ((1000.0, 64),), seed=3)
K = S.freewater_kernels(sch, dirs)
y, d = S.freewater_signals(n, K, ht, sch, seed=1)
ctx = get_context()
lut = _capi.upload_freewater(ctx, K, ht)
dev = torch.device('cuda', 0)
yt, dt = torch.from_numpy(y).to(dev), torch.from_numpy(d).to(dev)
lib = _capi.lib() if hasattr(_capi, 'lib') else _capi._lib
out = (ctypes.c_ulonglong * 16)()
for it in range(2):
    _capi.freewater_fit_device(ctx, lut, yt, dt, 0.0, 1e-3, False)
    ctx.sync()
    lib.amx_debug_fw_phases(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ['refill (phase 1)', 'take', 'solve+feasibility', 'gradient+pick', 'maps', 'trips', 'active lane-trips', 'total']
tot = v[7]
for k in (0, 1, 2, 3, 4):
    print('%-20s %6.1f%%' % (names[k], 100 * v[k] / tot))
print('trips per wavefront-voxel-batch: %.2f trips per 64 voxels, lane utilisation %.1f%%, trips per voxel %.2f' % (
    v[5] / (n / 64), 100 * v[6] / (64 * v[5]), v[6] / n))
pv = v[8:]
print('project kernel: setup+first loads %.1f%%, tile writes (wait for loads) %.1f%%, issue+contract %.1f%%, epilogue %.1f%%; '
      '%.0f cycles (100 MHz counter) per batch' % (100 * pv[0] / pv[7], 100 * pv[1] / pv[7], 100 * pv[2] / pv[7], 100 * pv[3] / pv[7],
                                                   pv[7] / max(pv[5], 1)))
