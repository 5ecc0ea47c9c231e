#!/usr/bin/env python3
"""wall time of amico_amd.Evaluation.fit() (numpy in / numpy out) on a 128x128x80x99 volume, by phase"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amico_amd
from amico_amd import synthetic as S
sch = S.make_scheme(); dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); K = S.noddi_kernels(sch, dirs)
shape = (128, 128, 80)
y, _ = S.noddi_signals(int(np.prod(shape)), K, ht, sch, seed=1)
img = np.asfortranarray((y.reshape(shape + (-1,)) * 900.0).astype(np.float32))
xx, yy, zz = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij')
mask = ((xx * xx + yy * yy + zz * zz) < 0.92).astype(np.uint8)
ae = amico_amd.Evaluation()
ae.set_data(img, sch, mask)
ae.set_model('NODDI'); ae.set_kernels(K, ht)
for rep in range(3):
    t = time.perf_counter(); ae.fit(); el = time.perf_counter() - t
    print('Evaluation.fit: %.1f ms total (directions+prepare %.1f ms, model.fit %.1f ms) for %d voxels -> %.2f M voxels/s'
          % (1e3 * el, 1e3 * ae.get_config('dirs_precomputing_time'), 1e3 * ae.get_config('fit_time'), ae.y.shape[0], ae.y.shape[0] / el / 1e6))
