#!/usr/bin/env python3
"""first call of a process: NODDI().fit(evaluation) on 1 M voxels, host numpy in / out -- the one fit a subject gets.  usage: first_call.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
t00 = time.perf_counter()
from amico_amd import NODDI, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals_parallel(n, K, ht, sch, seed=5)
y = y.astype(np.float32).astype(np.float64)


class Ev:
    def __init__(self):
        self.y, self.DIRs, self.htable, self.KERNELS, self.nthreads = y, d, ht, K, 1

    def get_config(self, k):
        return False


import torch
torch.cuda.init(); torch.zeros(1, device='cuda')          # (the HIP runtime itself: a process pays it once whatever it runs)
m = NODDI(); m.scheme = sch
ts = []
for k in range(4):
    t0 = time.perf_counter(); out = m.fit(Ev()); ts.append(1e3 * (time.perf_counter() - t0))
print('AMX_HOST_PREFETCH=%s: first call %.1f ms, then %s ms' % (os.environ.get('AMX_HOST_PREFETCH', '1'), ts[0], ' '.join('%.1f' % t for t in ts[1:])))
