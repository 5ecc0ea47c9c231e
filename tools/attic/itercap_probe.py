#!/usr/bin/env python3
"""Which voxel of the NODDI golden fixture trips the iteration cap?  (one-voxel fits)"""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from amico_amd import _capi, get_context   # noqa: E402
from conftest import expand_lut   # noqa: E402
f = dict(np.load('tests/golden/noddi_fixture.npz', allow_pickle=True))
ht = dict(np.load('tests/golden/htable500.npz', allow_pickle=True))
K = {'model': 'NODDI', 'wm': expand_lut(f['wm_slices'], f['lut_ids']), 'iso': f['iso'], 'norms': f['norms'], 'icvf': f['icvf'],
     'kappa': f['kappa']}
ctx = get_context()
lut = _capi.upload_noddi(ctx, K, ht['htable'], f['dwi_idx'])
y, d = f['y'], f['dirs']
order = [5, 0, 0, 1] + list(range(y.shape[0]))
for v in order:
    _capi.noddi_fit(ctx, lut, y[v:v + 1].copy(), d[v:v + 1].copy(), 0.5, 1e-3, 3)
    st = ctx.last_stats()
    if st['itercap_voxels'] or st['guard_trips'] or st['overflow_voxels']:
        print(v, st, 'y range', y[v].min(), y[v].max(), 'dir', d[v])
print('done')
