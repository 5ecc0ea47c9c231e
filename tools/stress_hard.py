#!/usr/bin/env python3
"""NODDI parity on signals the dictionary does not explain: two crossing compartments, wrong direction, pure noise,
flat and zero signals; in-vivo and ex-vivo dictionaries.  usage: stress_hard.py [N] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amico_amd import _capi, get_context, synthetic as S   # noqa: E402
from oracle import oracle                                  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9
rng = np.random.default_rng(seed)
ctx = get_context()
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=seed); K = S.noddi_kernels(sch, dirs)
wm = K['wm']; iso = K['iso'].astype(np.float64)
d1 = S.random_unit_vectors(n, rng); d2 = S.random_unit_vectors(n, rng)
l1 = S.lut_indices(d1, ht); l2 = S.lut_indices(d2, ht)
k1 = rng.integers(0, wm.shape[0], n); k2 = rng.integers(0, wm.shape[0], n)
f = rng.dirichlet([1, 1, 1], n)
y0 = f[:, :1] * wm[k1, l1].astype(np.float64) + f[:, 1:2] * wm[k2, l2].astype(np.float64) + f[:, 2:3] * iso[None, :]
kind = rng.integers(0, 6, n)
sig = 1.0 / rng.choice([5.0, 15.0, 40.0], n)
y = np.sqrt((y0 + sig[:, None] * rng.normal(size=y0.shape)) ** 2 + (sig[:, None] * rng.normal(size=y0.shape)) ** 2)
y[kind == 3] = np.abs(rng.normal(size=(int((kind == 3).sum()), y.shape[1])))          # pure noise
y[kind == 4] = rng.uniform(0.0, 2.0, (int((kind == 4).sum()), 1))                     # flat signal
y[kind == 5] *= (rng.uniform(size=(int((kind == 5).sum()), y.shape[1])) < 0.5)         # half the volumes zeroed
y[:10] = 0.0
y = y.astype(np.float32).astype(np.float64)
dgiven = np.where((kind == 2)[:, None], d2, d1)                                       # kind 2: direction of the minor fibre
cores = os.cpu_count()
for exvivo in (False, True):
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx, is_exvivo=exvivo)
    est, _, _, _ = _capi.noddi_fit(ctx, lut, y, dgiven, 0.5, 1e-3, 4 if exvivo else 3)
    ref = oracle.noddi_fit(y, dgiven, K, ht, sch.dwi_idx, is_exvivo=exvivo, nthreads=cores)['estimates']
    dd = np.abs(est - ref).max(axis=1)
    bad = np.where(dd > 1e-6)[0]
    print('exvivo' if exvivo else 'invivo', 'n', n, 'max %.3e' % dd.max(), '>1e-8:', int((dd > 1e-8).sum()), '>1e-6:', len(bad),
          'kinds of bad', np.bincount(kind[bad], minlength=6).tolist(), [(int(b), float(dd[b])) for b in bad[:6]], ctx.last_stats(), flush=True)
