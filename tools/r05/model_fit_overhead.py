#!/usr/bin/env python3
"""NODDI().fit(evaluation) against the bare host-buffer call on the same context, alternating: what the plug-in surface adds per call"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from amico_amd import NODDI, _capi, get_context, synthetic as S, models as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=1)


class Ev:
    def __init__(self):
        self.y, self.DIRs, self.htable, self.KERNELS, self.nthreads = y, d, ht, K, 1

    def get_config(self, k):
        return False


m = NODDI(); m.scheme = sch; ev = Ev()
m.fit(ev)
ctx = get_context(); lut = next(iter(m._lut_cache.values()))[1]
_capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)
a, b, c = [], [], []
for _ in range(9):
    t = time.perf_counter(); m.fit(ev); a.append(time.perf_counter() - t)
    t = time.perf_counter(); _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3); b.append(time.perf_counter() - t)
    t = time.perf_counter(); M._fingerprint(K); c.append(time.perf_counter() - t)
f = lambda v: '%.2f' % (1e3 * float(np.median(v)))
print('%d voxels: model.fit %s ms, bare host-buffer call %s ms, digest of KERNELS alone %s ms' % (n, f(a), f(b), f(c)))
print('  model.fit calls:', ' '.join('%.2f' % (1e3 * v) for v in a))
print('  bare calls:     ', ' '.join('%.2f' % (1e3 * v) for v in b))
