import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from amico_amd import synthetic as S
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0)
K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(40, K, ht, sch, seed=3)
lut = S.lut_indices(d, ht); wm = K['wm']; iso = K['iso'].astype(np.float64); n_wm = 144
def nm(j): return 'iso' if j == 144 else '(%d,%d)' % (j // 12, j % 12)   # (od index, vf index)
for v in range(6):
    A = np.concatenate([wm[:, lut[v], :].astype(np.float64).T, iso[:, None]], axis=1)
    P = []; x = np.zeros(145); log = []
    for it in range(60):
        w = A.T @ (y[v] - A @ x)
        w[P] = -1
        t = int(np.argmax(w))
        if w[t] <= 0: break
        P.append(t); log.append('+' + nm(t))
        while True:
            z = np.zeros(145); z[P] = np.linalg.lstsq(A[:, P], y[v], rcond=None)[0]
            neg = [j for j in P if z[j] <= 0]
            if not neg: x = z; break
            al = min(x[j] / (x[j] - z[j]) for j in neg)
            x = x + al * (z - x)
            out = min(neg, key=lambda j: x[j] / (x[j] - z[j]))
            P.remove(out); x[out] = 0; log.append('-' + nm(out))
    print(v, ' '.join(log), '| final', [nm(j) + ':%.3f' % x[j] for j in sorted(P)])
