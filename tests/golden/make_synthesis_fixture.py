#!/usr/bin/env python3
"""Golden vectors for amico_amd/synthesis.py: the reference's response functions (amico/synthesis.py, imported from
/root/reference in the build container -- pure Python, numpy + scipy) evaluated on two small high-resolution-like schemes.
Only inputs and outputs are stored (tests/golden/synthesis_fixture.npz); nothing of the reference travels.

    python tests/golden/make_synthesis_fixture.py
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
m = types.ModuleType('amico')
m.__path__ = ['/root/reference/amico']
sys.modules['amico'] = m
from amico import synthesis as ref            # noqa: E402
from amico.scheme import Scheme               # noqa: E402


def fib(n):
    i = np.arange(n) + 0.5
    z = 1.0 - 2.0 * i / n
    phi = np.pi * (1.0 + 5.0 ** 0.5) * i
    s = np.sqrt(1.0 - z * z)
    return np.column_stack([s * np.cos(phi), s * np.sin(phi), z])


rng = np.random.default_rng(7)
d = fib(40)
d[0] = [0.0, 0.0, 1.0]                                      # exactly parallel / perpendicular cases
d[1] = [1.0, 0.0, 0.0]
# b-value scheme (VERSION 0): one b0 row, three shells
raw0 = np.vstack([[[0, 0, 0, 0.0]]] + [np.column_stack([d, np.full(len(d), b)]) for b in (700.0, 2000.0, 3000.0)])
# STEJSKALTANNER scheme (VERSION 1): G, Delta, delta, TE
raw1 = np.vstack([[[0, 0, 0, 0.0, 0.04, 0.02, 0.08]]] +
                 [np.column_stack([d, np.full(len(d), G), np.full(len(d), 0.040), np.full(len(d), 0.020), np.full(len(d), 0.080)])
                  for G in (0.020, 0.045, 0.070)] +
                 [np.column_stack([d[:10], np.full(10, 0.060), np.full(10, 0.030), np.full(10, 0.012), np.full(10, 0.070)])])
out = {'raw0': raw0, 'raw1': raw1}
for tag, raw in (('v0', raw0), ('v1', raw1)):
    sch = Scheme(raw.copy(), 0)
    out[tag + '_b'] = np.asarray(sch.b, dtype=np.float64)
    out[tag + '_stick'] = ref.Stick(sch).get_signal(1.7e-3)
    out[tag + '_zeppelin'] = ref.Zeppelin(sch).get_signal(1.7e-3, 0.4e-3)
    out[tag + '_ball'] = ref.Ball(sch).get_signal(3.0e-3)
    out[tag + '_tensor'] = ref.Tensor(sch).get_signal(1.5e-3, 0.5e-3, 0.2e-3)
    kappas = np.array([0.0, 1e-6, 0.05, 0.5, 2.0, 8.0, 21.2, 29.0, 35.0, 64.0])
    out[tag + '_kappas'] = kappas
    out[tag + '_noddi_ic'] = np.stack([ref.NODDIIntraCellular(sch).get_signal(1.7e-3, k) for k in kappas])
    out[tag + '_noddi_ec'] = np.stack([ref.NODDIExtraCellular(sch).get_signal(1.7e-3, k, 0.6) for k in kappas])
    out[tag + '_noddi_iso'] = ref.NODDIIsotropic(sch).get_signal(3.0e-3)
    out[tag + '_watson_coeff'] = np.stack([ref.NODDIIntraCellular(sch)._watson_SH_coeff(k) for k in kappas[1:]])
    out[tag + '_lgi_x'] = np.array([0.0, 1e-3, 0.05, 0.0500001, 0.3, 1.0, 3.4, 9.0])
    out[tag + '_lgi'] = ref.NODDIIntraCellular(sch)._legendre_gaussian_integral(out[tag + '_lgi_x'].copy(), 6)
sch1 = Scheme(raw1.copy(), 0)
out['radii'] = np.array([1.0e-6, 4.0e-6, 12.0e-6])
out['v1_sphere'] = np.stack([ref.SphereGPD(sch1).get_signal(3.0e-3, R) for R in out['radii']])
out['v1_cylinder'] = np.stack([ref.CylinderGPD(sch1).get_signal(0.6e-3, R) for R in np.array([0.01e-6, 2.0e-6, 8.0e-6])])
out['cyl_radii'] = np.array([0.01e-6, 2.0e-6, 8.0e-6])
out['v1_cylinder_tilted'] = ref.CylinderGPD(sch1).get_signal(0.6e-3, 3.0e-6, 0.7, 1.1)
out['v1_astrosticks'] = ref.Astrosticks(sch1).get_signal(1.2e-3)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'synthesis_fixture.npz'), **out)
print('wrote synthesis_fixture.npz:', {k: np.shape(v) for k, v in out.items()})
