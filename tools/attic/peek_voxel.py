#!/usr/bin/env python3
"""diagnosis: stage intermediates (x_iso, LASSO support) of given voxels on the GPU vs the oracle.
usage: AMICO_AMD_LIB=<lib built with -DAMX_PEEK> peek_voxel.py N seed snr v1,v2"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402
from oracle import oracle                                  # noqa: E402
n, seed, snr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
vs = [int(v) for v in sys.argv[4].split(',')]
ctx = get_context()
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=seed); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=seed + 1, snr=snr)
lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
est, _, _, _ = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)
L = _capi.lib()
for v in vs:
    xi = (C.c_double * 2)(); sp = (C.c_uint64 * 4)()
    L.amx_peek(ctx._h, C.c_int64(v), xi, sp)
    supp = [64 * q + b for q in range(4) for b in range(64) if (sp[q] >> b) & 1]
    r = oracle.noddi_fit(y[v:v + 1], d[v:v + 1], K, ht, sch.dwi_idx, nthreads=1, return_x=True)
    x = r['x'][0]
    print('voxel', v, 'gpu xiso', xi[0], 'oracle xiso', x[0][-1], 'diff', xi[0] - x[0][-1])
    print('   gpu supp   ', supp)
    print('   oracle supp', list(np.nonzero(x[1])[0]))
    print('   gpu maps', est[v], 'oracle', r['estimates'][0])
