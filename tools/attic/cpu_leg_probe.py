"""How fast is the CPU oracle on this box? (thread counts x builds, FreeWater and NODDI)"""
import numpy as np, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amico_amd import synthetic as S
from oracle import oracle
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(1, ((1000.0, 64),), seed=3); K = S.freewater_kernels(sch, dirs)
y, d = S.freewater_signals(500000, K, ht, sch, seed=1)
for fast in (False, True):
    oracle.use_fast_build(fast)
    for nt in (64, 128, 256):
        ts = []
        for _ in range(4):
            t = time.perf_counter(); oracle.freewater_fit(y, d, K, ht, nthreads=nt); ts.append(time.perf_counter() - t)
        print('FW', 'fast' if fast else 'O2', nt, 'threads: best %.2f M/s median %.2f M/s' % (5e5 / min(ts) / 1e6, 5e5 / np.median(ts) / 1e6), flush=True)
sn = S.make_scheme(seed=0); Kn = S.noddi_kernels(sn, dirs)
yn, dn = S.noddi_signals(100000, Kn, ht, sn, seed=1)
for fast in (False, True):
    oracle.use_fast_build(fast)
    for nt in (128, 256):
        ts = []
        for _ in range(3):
            t = time.perf_counter(); oracle.noddi_fit(yn, dn, Kn, ht, sn.dwi_idx, nthreads=nt); ts.append(time.perf_counter() - t)
        print('NODDI', 'fast' if fast else 'O2', nt, 'threads: best %.1f k/s median %.1f k/s' % (1e5 / min(ts) / 1e3, 1e5 / np.median(ts) / 1e3), flush=True)
