#!/usr/bin/env python3
"""Can the stage-3 support be read off the LASSO coefficients?  (oracle x of stages 2 and 3 on synthetic voxels)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5)
r = oracle.noddi_fit(y, d, K, ht, sch.dwi_idx, nthreads=os.cpu_count() or 1, return_x=True)
x = r['x'] if 'x' in r else r['x_stages']
x2, x3 = x[:, 1, :144], x[:, 2, :144]
s2, s3 = x2 > 0, x3 > 0
print('mean |S2| %.2f  |S3 wm| %.2f  S3 subset of S2: %.4f' % (s2.sum(1).mean(), s3.sum(1).mean(), (s3 <= s2).all(1).mean()))
for k in (2, 3, 4, 5):
    top = np.argsort(-x2, axis=1)[:, :k]
    m = np.zeros_like(s2); np.put_along_axis(m, top, True, axis=1); m &= s2
    print('top-%d LASSO atoms == S3: %.3f   S3 subset of top-%d: %.3f' % (k, (m == s3).all(1).mean(), k, (s3 <= m).all(1).mean()))
for thr in (0.02, 0.05, 0.1, 0.2):
    m = x2 > thr * x2.max(1, keepdims=True)
    print('x2 > %.2f max == S3: %.3f   subset: %.3f  size %.2f' % (thr, (m == s3).all(1).mean(), (s3 <= m).all(1).mean(), m.sum(1).mean()))
