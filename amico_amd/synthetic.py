"""Synthetic acquisition schemes, dictionaries (LUTs) and noisy signals.

Everything the fit path consumes (``KERNELS`` dict, ``htable``, ``y``, ``DIRs``) is produced
here in the reference's layouts so that tests and ``bench.py`` can run without NIfTI files,
dipy or the reference package:

* ``KERNELS`` layouts follow ``NODDI.resample`` (amico/models.pyx:754-792),
  ``FreeWater.resample`` (:1113-1144) and ``SANDI.resample`` (:1446-1486);
* ``htable`` follows the construction rule of ``amico/directions/htable_ndirs=*.bin``
  (``htable[theta*181+phi] = argmax_k |v(theta,phi) . d_k|``);
* the signal model is SURVEY.md section 8(d): one random atom + isotropic fraction,
  Rician noise, b0 normalisation, float32 round trip (amico/core.py:136,222,451-452).

The compartment physics is written independently (quadrature over the Watson distribution,
closed forms for zeppelin / ball / astro-sticks, Gaussian-phase sphere); it is only meant to
give *realistic* dictionaries (coherent atoms, rank-deficient A) for tests and benchmarks.
"""
import numpy as np
from scipy import special, optimize

GAMMA = 2.675987e8  # rad s^-1 T^-1


# ----------------------------------------------------------------------------- directions
def fibonacci_hemisphere(n):
    """n unit vectors spread over the half sphere y >= 0 (like amico/directions/ndirs=*.bin)."""
    k = np.arange(n) + 0.5
    z = 1.0 - k / n                      # upper hemisphere of a Fibonacci lattice ...
    phi = np.pi * (1.0 + 5 ** 0.5) * k
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    v = np.stack([r * np.cos(phi), z, r * np.sin(phi)], axis=1)   # ... rotated so that y >= 0
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def build_htable(dirs):
    """int16[181*181]: for integer (theta, phi) degrees the index of the closest (axial) direction."""
    th = np.deg2rad(np.arange(181.0))[:, None]
    ph = np.deg2rad(np.arange(181.0))[None, :]
    v = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th) * np.ones_like(ph)], axis=-1)
    v = v.reshape(-1, 3)
    dt = np.asarray(dirs, dtype=np.float64).T
    out = np.empty(v.shape[0], dtype=np.int16)
    step = max(1, (1 << 24) // max(1, dt.shape[1]))          # (32 761 grid points x 32 761 directions would not fit at once)
    for s in range(0, v.shape[0], step):
        out[s:s + step] = np.argmax(np.abs(v[s:s + step] @ dt), axis=1)
    return out


def random_unit_vectors(n, rng):
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# ----------------------------------------------------------------------------- schemes
class SimpleScheme:
    """The few fields of amico.scheme.Scheme the fit path reads (scheme.py:76-84)."""

    def __init__(self, table, b0_thr=0):
        table = np.asarray(table, dtype=np.float64)
        self.raw = table
        if table.shape[1] == 4:
            self.version = 0
            self.b = table[:, 3].copy()
        elif table.shape[1] == 7:
            self.version = 1
            self.b = (GAMMA * table[:, 3] * table[:, 5]) ** 2 * (table[:, 4] - table[:, 5] / 3.0) * 1e-6
        else:
            raise ValueError('Unrecognized scheme format')
        self.b0_thr = b0_thr
        self.b0_idx = np.where(self.b <= b0_thr)[0]
        self.b0_count = len(self.b0_idx)
        self.dwi_idx = np.where(self.b > b0_thr)[0]
        self.dwi_count = len(self.dwi_idx)

    @property
    def nS(self):
        return self.b0_count + self.dwi_count

    @property
    def shells(self):
        """one dict per distinct acquisition row raw[:, 3:] above the b0 threshold, in order of first appearance,
        with the indices of its volumes (scheme.py:88-120)"""
        out, seen = [], []
        tmp = self.raw[:, 3:]
        for i in range(tmp.shape[0]):
            if any(np.array_equal(tmp[i], k) for k in seen):
                continue
            seen.append(tmp[i].copy())
            if self.b[i] <= self.b0_thr:
                continue
            sh = {'b': self.b[i], 'G': None, 'Delta': None, 'delta': None, 'TE': None}
            if self.version == 1:
                sh['G'], sh['Delta'], sh['delta'], sh['TE'] = (float(v) for v in tmp[i])
            sh['idx'] = np.where((tmp == tmp[i]).all(axis=1))[0]
            sh['grad'] = self.raw[sh['idx'], 0:3]
            out.append(sh)
        return out


def make_scheme(n_b0=9, shells=((700.0, 30), (2000.0, 60)), seed=0):
    """Nx4 b-value scheme, b0 volumes first (the 99-volume 2-shell protocol by default)."""
    rng = np.random.default_rng(seed)
    g = [np.zeros((n_b0, 3))]
    b = [np.zeros(n_b0)]
    for bval, n in shells:
        g.append(random_unit_vectors(n, rng))
        b.append(np.full(n, float(bval)))
    return SimpleScheme(np.hstack([np.vstack(g), np.hstack(b)[:, None]]))


# ----------------------------------------------------------------------------- NODDI physics
def _watson_stick_table(bd, kappa, ncos=513, nquad=48):
    """E(c) = int exp(-bd (g.n)^2) W(n; z, kappa) dn for g at cos-angle c from the mean axis."""
    x, wx = np.polynomial.legendre.leggauss(nquad)          # cos(polar) of n
    az = (np.arange(2 * nquad) + 0.5) * np.pi / nquad       # azimuth of n
    dens = np.exp(kappa * (x * x - 1.0))
    dens = dens * wx
    dens /= dens.sum() * len(az)
    c = np.linspace(0.0, 1.0, ncos)
    s = np.sqrt(1.0 - c * c)
    sx = np.sqrt(1.0 - x * x)
    # g = (s, 0, c); n = (sx cos az, sx sin az, x)  ->  g.n = s*sx*cos(az) + c*x
    gn = s[:, None, None] * sx[None, :, None] * np.cos(az)[None, None, :] + c[:, None, None] * x[None, :, None]
    E = (np.exp(-bd * gn * gn) * dens[None, :, None]).sum(axis=(1, 2))
    return c, E


def _watson_tau1(kappa):
    """<cos^2> of a Watson distribution."""
    if kappa < 1e-5:
        return 1.0 / 3.0 + 4.0 * kappa / 45.0
    sk = np.sqrt(kappa)
    return -1.0 / (2.0 * kappa) + 1.0 / (2.0 * sk * special.dawsn(sk))


def noddi_kernels(scheme, dirs, IC_VFs=None, IC_ODs=None, dPar=1.7e-3, dIso=3.0e-3):
    """KERNELS dict in the layout of NODDI.resample (models.pyx:763-789)."""
    if IC_VFs is None:
        IC_VFs = np.linspace(0.1, 0.99, 12)
    if IC_ODs is None:
        IC_ODs = np.hstack((np.array([0.03, 0.06]), np.linspace(0.09, 0.99, 10)))
    dirs = np.asarray(dirs, dtype=np.float64)
    ndirs, nS = dirs.shape[0], scheme.nS
    n_wm = len(IC_ODs) * len(IC_VFs)
    g = scheme.raw[:, :3]
    b = scheme.b
    cosang = np.abs(dirs @ g.T)                     # ndirs x nS
    cosang = np.minimum(cosang, 1.0)
    K = {'model': 'NODDI',
         'wm': np.ones((n_wm, ndirs, nS), dtype=np.float32),
         'iso': np.ones(nS, dtype=np.float32),
         'kappa': np.zeros(n_wm, dtype=np.float32),
         'icvf': np.zeros(n_wm, dtype=np.float32),
         'norms': np.zeros((scheme.dwi_count, n_wm))}
    bvals = np.unique(b[scheme.dwi_idx])
    idx = 0
    for od in IC_ODs:
        kappa = 1.0 / np.tan(od * np.pi / 2.0)
        tau1 = _watson_tau1(kappa)
        ic = np.ones((ndirs, nS))
        for bv in bvals:
            c, E = _watson_stick_table(bv * dPar, kappa)
            sel = np.where(b == bv)[0]
            ic[:, sel] = np.interp(cosang[:, sel], c, E)
        for v in IC_VFs:
            dperp = dPar * (1.0 - v)
            dw_par = dperp + (dPar - dperp) * tau1
            dw_perp = dperp + (dPar - dperp) * (1.0 - tau1) / 2.0
            ec = np.exp(-b[None, :] * ((dw_par - dw_perp) * cosang ** 2 + dw_perp))
            sig = v * ic + (1.0 - v) * ec
            sig[:, scheme.b0_idx] = 1.0
            K['wm'][idx] = sig.astype(np.float32)
            K['kappa'][idx] = kappa
            K['icvf'][idx] = v
            K['norms'][:, idx] = 1.0 / np.linalg.norm(K['wm'][idx, 0, scheme.dwi_idx])
            idx += 1
    iso = np.exp(-b * dIso)
    iso[scheme.b0_idx] = 1.0
    K['iso'] = iso.astype(np.float32)
    return K


# ----------------------------------------------------------------------------- FreeWater physics
def freewater_kernels(scheme, dirs, d_par=1.0e-3, d_perps=None, d_isos=(2.5e-3,)):
    """KERNELS dict in the layout of FreeWater.resample (models.pyx:1120-1142)."""
    if d_perps is None:
        d_perps = np.linspace(0.1, 1.0, 10) * 1e-3
    dirs = np.asarray(dirs, dtype=np.float64)
    ndirs, nS = dirs.shape[0], scheme.nS
    cos2 = np.minimum(np.abs(dirs @ scheme.raw[:, :3].T), 1.0) ** 2
    b = scheme.b
    K = {'model': 'FreeWater',
         'D': np.zeros((len(d_perps), ndirs, nS), dtype=np.float32),
         'CSF': np.zeros((len(d_isos), nS), dtype=np.float32)}
    for i, dp in enumerate(d_perps):
        sig = np.exp(-b[None, :] * ((d_par - dp) * cos2 + dp))
        sig[:, scheme.b0_idx] = 1.0
        K['D'][i] = sig.astype(np.float32)
    for i, d in enumerate(d_isos):
        sig = np.exp(-b * d)
        sig[scheme.b0_idx] = 1.0
        K['CSF'][i] = sig.astype(np.float32)
    return K


# ----------------------------------------------------------------------------- SANDI physics
def make_sandi_scheme(bvals=(1000., 2500., 4000., 6000., 8000.), ndir_per_shell=60, n_b0=6,
                      Delta=0.040, delta=0.020, TE=0.080, seed=0):
    """Nx7 STEJSKALTANNER table: 5 shells x 60 dirs + 6 b0 = 306 volumes."""
    rng = np.random.default_rng(seed)
    rows = [np.hstack([np.zeros((n_b0, 3)), np.zeros((n_b0, 1)), np.full((n_b0, 1), Delta),
                       np.full((n_b0, 1), delta), np.full((n_b0, 1), TE)])]
    for bv in bvals:
        G = np.sqrt(bv * 1e6 / ((GAMMA * delta) ** 2 * (Delta - delta / 3.0)))
        g = random_unit_vectors(ndir_per_shell, rng)
        rows.append(np.hstack([g, np.full((ndir_per_shell, 1), G), np.full((ndir_per_shell, 1), Delta),
                               np.full((ndir_per_shell, 1), delta), np.full((ndir_per_shell, 1), TE)]))
    return SimpleScheme(np.vstack(rows))


def directional_average_scheme(scheme):
    """One row per shell (+ one b0 row first), as after doDirectionalAverage (core.py:232-268)."""
    shells = []
    seen = []
    for i in scheme.dwi_idx:
        key = tuple(np.round(scheme.raw[i, 3:], 12))
        if key not in seen:
            seen.append(key)
            shells.append(np.hstack([[1.0, 0.0, 0.0], scheme.raw[i, 3:]]))
    b0 = np.hstack([[0.0, 0.0, 0.0], scheme.raw[scheme.b0_idx[0], 3:]])
    if scheme.version == 1:
        b0[3] = 0.0
    return SimpleScheme(np.vstack([b0] + shells))


_SPHERE_ROOTS = None


def _sphere_roots(n=20):
    global _SPHERE_ROOTS
    if _SPHERE_ROOTS is None or len(_SPHERE_ROOTS) < n:
        # roots of d/dx j1(x) = 0
        def f(x):
            return special.spherical_jn(1, x, derivative=True)
        roots, x0 = [], 1.0
        while len(roots) < n:
            x1 = x0 + 0.1
            if f(x0) * f(x1) < 0:
                roots.append(optimize.brentq(f, x0, x1))
            x0 = x1
        _SPHERE_ROOTS = np.array(roots)
    return _SPHERE_ROOTS[:n]


def _sphere_gpd(G, Delta, delta, D, R):
    """Gaussian-phase signal of restricted diffusion in a sphere (SI units in, unitless out)."""
    am = _sphere_roots() / R
    am2 = am * am
    num = (2.0 * delta - (2.0 + np.exp(-am2 * D * (Delta - delta)) - 2.0 * np.exp(-am2 * D * delta)
                          - 2.0 * np.exp(-am2 * D * Delta) + np.exp(-am2 * D * (Delta + delta))) / (am2 * D))
    s = np.sum(num / (am2 * am2 * (am2 * R * R - 2.0)))
    return np.exp(-2.0 * (GAMMA * G) ** 2 / D * s)


def sandi_kernels(avg_scheme, d_is=3.0e-3, Rs=None, d_in=None, d_isos=None):
    """KERNELS dict in the layout of SANDI.resample (models.pyx:1455-1484) for a
    direction-averaged scheme (nS = n_shells + 1)."""
    if Rs is None:
        Rs = np.linspace(1.0, 12.0, 5) * 1e-6
    if d_in is None:
        d_in = np.linspace(0.25, 3.0, 5) * 1e-3
    if d_isos is None:
        d_isos = np.linspace(0.25, 3.0, 5) * 1e-3
    nS = avg_scheme.nS
    b = avg_scheme.b
    n_atoms = len(Rs) + len(d_in) + len(d_isos)
    K = {'model': 'SANDI',
         'signal': np.zeros((nS, n_atoms), dtype=np.float64, order='F'),
         'norms': np.zeros(n_atoms, dtype=np.float64)}
    cols = []
    for R in Rs:
        sig = np.ones(nS)
        for i in avg_scheme.dwi_idx:
            G, Delta, delta = avg_scheme.raw[i, 3], avg_scheme.raw[i, 4], avg_scheme.raw[i, 5]
            sig[i] = _sphere_gpd(G, Delta, delta, d_is * 1e-6, R)
        cols.append(sig)
    for d in d_in:
        sig = np.ones(nS)
        bd = b[avg_scheme.dwi_idx] * d
        sig[avg_scheme.dwi_idx] = np.sqrt(np.pi / (4.0 * bd)) * special.erf(np.sqrt(bd))
        cols.append(sig)
    for d in d_isos:
        sig = np.exp(-b * d)
        sig[avg_scheme.b0_idx] = 1.0
        cols.append(sig)
    for idx, sig in enumerate(cols):
        sig = sig.astype(np.float32).astype(np.float64)   # resample_kernel returns float32
        K['norms'][idx] = 1.0 / np.linalg.norm(sig)
        K['signal'][:, idx] = sig * K['norms'][idx]
    return K, np.asarray(Rs), np.asarray(d_in), np.asarray(d_isos)


# ----------------------------------------------------------------------------- signals
def _rician(y0, snr, rng):
    s = 1.0 / snr
    return np.sqrt((y0 + s * rng.standard_normal(y0.shape)) ** 2 + (s * rng.standard_normal(y0.shape)) ** 2)


def _finish(y, scheme):
    if scheme.b0_count > 0:
        y = y / y[:, scheme.b0_idx].mean(axis=1, keepdims=True)
    y = y.astype(np.float32).astype(np.float64)     # core.py:136 float32 volume, :451 -> double
    y[y < 0] = 0
    return np.ascontiguousarray(y)


def lut_indices(dirs, htable):
    """numpy restatement of the effective rule of lut.pyx:316-356 (used only to synthesise data)."""
    d = np.array(dirs, dtype=np.float64, copy=True)
    flip = d[:, 1] < 0
    d[flip] = -d[flip]
    i2 = np.arctan2(d[:, 1], d[:, 0])
    i2 = np.where(i2 < 0, i2 + 2 * np.pi, i2)
    i1 = np.arctan2(np.hypot(d[:, 0], d[:, 1]), d[:, 2])
    ii1 = np.floor(i1 / np.pi * 180.0 + 0.5).astype(np.int64)
    ii2 = np.floor(i2 / np.pi * 180.0 + 0.5).astype(np.int64)
    return np.asarray(htable)[ii1 * 181 + ii2].astype(np.int64)


def noddi_signals(n_vox, kernels, htable, scheme, seed=1, snr=30.0, chunk=65536):
    """(y f64[n_vox,nS], DIRs f64[n_vox,3]) following the signal model of SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    wm, iso = kernels['wm'], kernels['iso'].astype(np.float64)
    n_wm = wm.shape[0]
    dirs = random_unit_vectors(n_vox, rng)
    y = np.empty((n_vox, scheme.nS))
    lut = lut_indices(dirs, htable)
    for s in range(0, n_vox, chunk):
        e = min(n_vox, s + chunk)
        k = rng.integers(0, n_wm, e - s)
        f = rng.uniform(0.0, 0.5, e - s)[:, None]
        y0 = (1.0 - f) * wm[k, lut[s:e], :].astype(np.float64) + f * iso[None, :]
        y[s:e] = _finish(_rician(y0, snr, rng), scheme)
    return y, dirs


def noddi_signals_parallel(n_vox, kernels, htable, scheme, seed=1, snr=30.0, block=250_000, threads=None):
    """noddi_signals for multi-million-voxel batches: blocks of `block` voxels with their own seeds, generated by a thread
    pool (numpy releases the GIL in the heavy statements), written in place -- 8 M voxels in a few seconds"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    y = np.empty((n_vox, scheme.nS))
    dirs = np.empty((n_vox, 3))
    starts = list(range(0, n_vox, block))

    def one(i):
        s = starts[i]
        e = min(n_vox, s + block)
        y[s:e], dirs[s:e] = noddi_signals(e - s, kernels, htable, scheme, seed=seed * 1000 + i, snr=snr)
    with ThreadPoolExecutor(threads or min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(len(starts))))
    return y, dirs


HARD_KINDS = ('crossing', 'single+iso', 'wrong direction', 'pure noise', 'flat', 'half zeroed', 'CSF dominated', 'background')


def noddi_hard_signals(n_vox, kernels, htable, scheme, seed=9):
    """Signals the dictionary does NOT explain -- what the wiki data sets the reference is verified on contain besides clean white
    matter (README.md:18-19): two crossing compartments, the fit given the minor fibre's direction, pure noise, flat signals,
    half of the volumes zeroed, CSF-dominated voxels (f_iso in [0.5, 1]), near-zero background; SNR 5 / 15 / 40 mixed; the first
    ten voxels all zero.  Returns (y, DIRs, kind) with kind indexing HARD_KINDS.  (models.pyx:902-981 takes the same path whatever
    the signal; the fast path of the MI355X build must too.)"""
    rng = np.random.default_rng(seed)
    wm, iso = kernels['wm'], kernels['iso'].astype(np.float64)
    n = n_vox
    d1 = random_unit_vectors(n, rng); d2 = random_unit_vectors(n, rng)
    l1 = lut_indices(d1, htable); l2 = lut_indices(d2, htable)
    k1 = rng.integers(0, wm.shape[0], n); k2 = rng.integers(0, wm.shape[0], n)
    kind = rng.integers(0, len(HARD_KINDS), n)
    f = rng.dirichlet([1, 1, 1], n)
    one = kind == 1
    f[one, 1] = 0.0; f[one, 2] = rng.uniform(0.0, 0.5, int(one.sum())); f[one, 0] = 1.0 - f[one, 2]
    csf = kind == 6
    fi = rng.uniform(0.5, 1.0, int(csf.sum()))
    f[csf, 2] = fi; f[csf, 0] = 1.0 - fi; f[csf, 1] = 0.0
    y0 = f[:, :1] * wm[k1, l1].astype(np.float64) + f[:, 1:2] * wm[k2, l2].astype(np.float64) + f[:, 2:3] * iso[None, :]
    sig = 1.0 / rng.choice([5.0, 15.0, 40.0], n)
    y = np.sqrt((y0 + sig[:, None] * rng.normal(size=y0.shape)) ** 2 + (sig[:, None] * rng.normal(size=y0.shape)) ** 2)
    y[kind == 3] = np.abs(rng.normal(size=(int((kind == 3).sum()), y.shape[1])))
    y[kind == 4] = rng.uniform(0.0, 2.0, (int((kind == 4).sum()), 1))
    y[kind == 5] *= (rng.uniform(size=(int((kind == 5).sum()), y.shape[1])) < 0.5)
    bg = kind == 7
    y[bg] = np.abs(rng.normal(scale=0.02, size=(int(bg.sum()), y.shape[1])))
    y[:10] = 0.0
    y = np.ascontiguousarray(y.astype(np.float32).astype(np.float64))
    dirs = np.ascontiguousarray(np.where((kind == 2)[:, None], d2, d1))
    return y, dirs, kind


def freewater_signals(n_vox, kernels, htable, scheme, seed=1, snr=30.0, chunk=65536):
    rng = np.random.default_rng(seed)
    D, CSF = kernels['D'], kernels['CSF'].astype(np.float64)
    dirs = random_unit_vectors(n_vox, rng)
    y = np.empty((n_vox, scheme.nS))
    lut = lut_indices(dirs, htable)
    for s in range(0, n_vox, chunk):
        e = min(n_vox, s + chunk)
        k = rng.integers(0, D.shape[0], e - s)
        f = rng.uniform(0.0, 0.5, e - s)[:, None]
        y0 = (1.0 - f) * D[k, lut[s:e], :].astype(np.float64) + f * CSF[0][None, :]
        y[s:e] = _finish(_rician(y0, snr, rng), scheme)
    return y, dirs


def sandi_signals(n_vox, kernels, avg_scheme, seed=1, snr=30.0, navg=60):
    """Direction-averaged signals (noise std reduced by sqrt(navg) as after averaging a shell)."""
    rng = np.random.default_rng(seed)
    A = kernels['signal'] / kernels['norms'][None, :]
    n_atoms = A.shape[1]
    w = rng.dirichlet(np.ones(3), n_vox)
    ks = np.stack([rng.integers(0, 5, n_vox), 5 + rng.integers(0, 5, n_vox), 10 + rng.integers(0, 5, n_vox)], 1)
    ks = np.minimum(ks, n_atoms - 1)
    y0 = np.zeros((n_vox, A.shape[0]))
    for c in range(3):
        y0 += w[:, c:c + 1] * A[:, ks[:, c]].T
    y = np.abs(y0 + rng.standard_normal(y0.shape) / (snr * np.sqrt(navg)))
    return _finish(y, avg_scheme)
