#!/usr/bin/env python3
"""Seeded NNLS stages vs the Lawson-Hanson path from the empty set (AMX_NO_SEED=1) vs the CPU oracle, with kernel times.
usage: python tools/seed_check.py [n_vox] [snr]   (run with an -DAMX_STATS library + AMX_DEBUG=1 for the seed statistics)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                # noqa: E402
from amico_amd import _capi, synthetic as S                 # noqa: E402
from oracle import oracle                                   # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=5, snr=snr)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
out = {}
for tag, env in (('seeded', '0'), ('cold', '1')):
    os.environ['AMX_NO_SEED'] = env
    ctx = _capi.Context()
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    ctx.set_profiling(True)
    for rep in range(3):
        t0 = time.time()
        res = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3)
        ctx.sync()
        dtm = time.time() - t0
    est = res[0].cpu().numpy()
    ms = [ctx.last_kernel_ms(k) for k in (0, 1, 2, 3)]
    print('%-7s wall %.2f ms  total/stage kernel ms %s  stats %s' % (tag, 1e3 * dtm, ['%.2f' % m for m in ms], ctx.last_stats()), flush=True)
    out[tag] = est
dd = np.abs(out['seeded'] - out['cold']).max(axis=1)
print('seeded vs cold: max %.3e, voxels > 1e-9: %d, > 1e-6: %d' % (dd.max(), (dd > 1e-9).sum(), (dd > 1e-6).sum()))
m = min(n, 100000)
ref = oracle.noddi_fit(y[:m], d[:m], K, ht, sch.dwi_idx, nthreads=os.cpu_count())['estimates']
for tag in out:
    dd = np.abs(out[tag][:m] - ref).max(axis=1)
    print('%-7s vs oracle (%d voxels): max %.3e, > 1e-8: %d, > 1e-6: %d' % (tag, m, dd.max(), (dd > 1e-8).sum(), (dd > 1e-6).sum()))
