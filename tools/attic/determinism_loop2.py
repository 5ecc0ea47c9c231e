#!/usr/bin/env python3
"""which stage / which seeds change between repeated device fits"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amico_amd import _capi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs); sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
y, d = S.noddi_signals(n, K, ht, sch, seed=3)
ctx = _capi.Context(); lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
yt = torch.from_numpy(y).cuda(); dt = torch.from_numpy(d).cuda()
ref = None
for r in range(reps):
    e = _capi.noddi_fit_device(ctx, lut, yt, dt, 0.5, 1e-3, 3, return_x=True); ctx.sync()
    X = e[-1].cpu().numpy()
    perm = _capi.debug_fetch(ctx, None, 0, (n,), np.int32)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    sd = _capi.debug_fetch(ctx, None, 2, (n,), np.uint64)[inv]           # last written seeds (stage 3 unless AMX_SEED_STAGES says otherwise), voxel order
    s2 = _capi.debug_fetch(ctx, None, 4, (n, 4), np.uint64)[inv]
    yt12 = _capi.debug_fetch(ctx, None, 1, (n, 12), np.float64)[inv]
    yt2 = _capi.debug_fetch(ctx, None, 3, (n, 12), np.float64)[inv]
    cur = (X, sd, s2, yt12, yt2)
    if ref is None: ref = cur; continue
    msg = []
    for st in range(3):
        bad = np.nonzero(np.abs(ref[0][:, st] - X[:, st]).max(axis=1) > 0)[0]
        if len(bad): msg.append('x stage %d: %d voxels %s' % (st + 1, len(bad), bad[:4].tolist()))
    for k, nm in ((1, 'seeds'), (2, 'lasso seeds'), (3, 'ytil'), (4, 'ytil2')):
        a = ref[k].reshape(n, -1); b = cur[k].reshape(n, -1)
        bad = np.nonzero((a != b).any(axis=1))[0]
        if len(bad): msg.append('%s: %d voxels %s' % (nm, len(bad), bad[:4].tolist()))
    print('rep', r, '; '.join(msg) if msg else 'identical')
