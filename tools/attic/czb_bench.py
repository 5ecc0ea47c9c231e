#!/usr/bin/env python3
"""CylinderZeppelinBall throughput (not a BASELINE config): dictionary from model.generate + load_kernels on a 3-shell
STEJSKALTANNER scheme, voxels = one cylinder + one zeppelin + ball with Rician noise, device-resident fit."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import amico_amd                                             # noqa: E402
from amico_amd import _capi, get_context, synthetic as S     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
sch = S.make_sandi_scheme(bvals=(1000., 2500., 4000.), ndir_per_shell=30, n_b0=3)
lut_dirs = S.fibonacci_hemisphere(500)
ht = S.build_htable(lut_dirs)
ae = amico_amd.Evaluation()
ae.set_data(np.ones((2, 2, 2, sch.nS), dtype=np.float32), sch, np.ones((2, 2, 2), dtype=np.uint8))
ae.set_model('CylinderZeppelinBall')
t = time.time()
lms = ae.generate_kernels(lut_dirs)
ae.load_kernels(lms, lut_dirs)
print('generate + load kernels: %.2f s' % (time.time() - t))
K = ae.KERNELS
rng = np.random.default_rng(1)
ori = rng.integers(0, 500, n)
a1, a2 = rng.integers(0, K['wmr'].shape[0], n), rng.integers(0, K['wmh'].shape[0], n)
f = rng.dirichlet([2, 2, 1], n)
y = (f[:, :1] * K['wmr'][a1, ori] + f[:, 1:2] * K['wmh'][a2, ori] + f[:, 2:] * K['iso'][0][None, :]).astype(np.float64)
y = np.sqrt((y + rng.normal(0, 1 / 30, y.shape)) ** 2 + rng.normal(0, 1 / 30, y.shape) ** 2)
ctx = get_context()
m = ae.model
lut = _capi.upload_czb(ctx, K, m.Rs, ht)
dev = torch.device('cuda', 0)
yt, dt = torch.from_numpy(y).to(dev), torch.from_numpy(lut_dirs[ori]).to(dev)
for _ in range(2):
    _capi.czb_fit_device(ctx, lut, yt, dt, 0.0, 4.0)
ctx.sync()
t = time.perf_counter()
for _ in range(5):
    _capi.czb_fit_device(ctx, lut, yt, dt, 0.0, 4.0)
ctx.sync()
dt_s = (time.perf_counter() - t) / 5
print('CylinderZeppelinBall: %d voxels, %d volumes, %d atoms: %.2f ms per fit, %.1f M voxels/s; stats %s' % (
    n, sch.nS, lut.n_atoms, 1e3 * dt_s, n / dt_s / 1e6, ctx.last_stats()))
