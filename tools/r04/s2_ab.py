#!/usr/bin/env python3
"""Stage-2 products derived from the stage-1 table (default) against the exact pass for every voxel (AMX_S2_EXACT=1): maps and the
stage coefficient vectors of the same voxels, bench mix / SNR 10 / hard mix.  usage: s2_ab.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
cases = {'bench mix': S.noddi_signals(n, K, ht, sch, seed=5), 'snr 10': S.noddi_signals(n, K, ht, sch, seed=6, snr=10.0),
         'hard mix': S.noddi_hard_signals(n, K, ht, sch, seed=7)[:2]}
res = {}
for mode in ('derived', 'exact'):
    if mode == 'exact':
        os.environ['AMX_S2_EXACT'] = '1'
    ctx = _capi.Context(0)
    os.environ.pop('AMX_S2_EXACT', None)
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    for name, (y, d) in cases.items():
        yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
        out = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, return_x=True)
        ctx.sync()
        res[(mode, name)] = (out[0].cpu().numpy(), out[-1].cpu().numpy(), ctx.last_seed_stats(), ctx.last_stats())
for name in cases:
    e0, x0, s0, t0 = res[('derived', name)]; e1, x1, s1, t1 = res[('exact', name)]
    print('%-10s max |dmap| %.3e  max |dx| stage 1/2/3 %.2e %.2e %.2e  supports differ in %d voxels' % (
        name, np.abs(e0 - e1).max(), np.abs(x0[:, 0] - x1[:, 0]).max(), np.abs(x0[:, 1] - x1[:, 1]).max(), np.abs(x0[:, 2] - x1[:, 2]).max(),
        int(((x0[:, 1] > 0) != (x1[:, 1] > 0)).any(axis=1).sum())))
    print('   derived:', s0, t0)
    print('   exact  :', s1, t1)
