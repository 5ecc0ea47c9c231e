import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=3, snr=snr)
r = oracle.noddi_fit(y, d, K, ht, sch.dwi_idx, nthreads=os.cpu_count(), return_x=True)
x = r['x']
s1 = x[:, 0, :] > 0; s2 = x[:, 1, :144] > 0; s3 = x[:, 2, :] > 0
s3allowed = np.concatenate([s2, np.ones((n, 1), bool)], axis=1)
sub = (s1 & ~s3allowed).sum(axis=1) == 0
print('n', n, 'snr', snr)
print('|S1| %.2f  |S2| %.2f  |S3| %.2f' % (s1.sum(1).mean(), s2.sum(1).mean(), s3.sum(1).mean()))
print('S1 subset of S2+iso: %.1f %%' % (100 * sub.mean()))
print('|S1 minus allowed| mean %.2f ; |S1 & allowed| mean %.2f' % ((s1 & ~s3allowed).sum(1).mean(), (s1 & s3allowed).sum(1).mean()))
print('S3 == S1&allowed: %.1f %%' % (100 * ((s3 == (s1 & s3allowed)).all(axis=1)).mean()))
print('S3 superset of S1&allowed: %.1f %%' % (100 * (((s1 & s3allowed) & ~s3).sum(1) == 0).mean()))
print('|S3 minus S1| mean %.2f' % ((s3 & ~s1).sum(1).mean()))
# stage 2: S1(wm) subset of S2 ?
s1w = s1[:, :144]
print('S1wm subset S2 %.1f %%, |S2 minus S1wm| %.2f, |S1wm minus S2| %.2f' % (100 * ((s1w & ~s2).sum(1) == 0).mean(), (s2 & ~s1w).sum(1).mean(), (s1w & ~s2).sum(1).mean()))
