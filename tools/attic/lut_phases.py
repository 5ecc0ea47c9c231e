#!/usr/bin/env python3
"""Cycle split of k_lut_resample (diagnosis build: tools/build_variant.sh lutph -DAMX_LUT_PHASES, then
AMICO_AMD_LIB=variants/lutph/libamico_amd.so python tools/lut_phases.py)."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, '.')
from amico_amd import _capi, get_context, lut, synthetic as S   # noqa: E402

scheme = S.make_scheme(seed=0)
idx_out, ylm_out = lut.aux_structures_resample(scheme, 12)
rng = np.random.default_rng(0)
lm = rng.normal(size=(144, 500, ylm_out.shape[1])).astype(np.float32)
ctx = get_context()
L = _capi.lib()
out = (ctypes.c_ulonglong * 8)()
for it in range(3):
    _capi.lut_resample(ctx, lm, ylm_out, idx_out, scheme.nS)
    L.amx_debug_lut_phases(out, 1)
v = np.array(list(out), dtype=np.float64)
print('waves x total cycles %.3g; staging %.1f%%, A loads %.1f%%, MFMA blocks %.1f%%, stores %.1f%%; tiles %d' % (
    v[7], 100 * v[0] / v[7], 100 * v[1] / v[7], 100 * v[2] / v[7], 100 * v[3] / v[7], v[5]))
