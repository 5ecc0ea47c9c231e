"""Response functions of the compartments AMICO's dictionaries are built from (amico/synthesis.py), for a fibre along z.

This is the one-off, per-protocol step in front of the LUT rotation (`lut.rotate_kernel`, `lut.rotate_and_resample`):
`model.generate()` evaluates these functions on the high-resolution scheme (500 directions per shell).  Host numpy, like the
reference -- a few thousand values per atom, nothing for a GPU to do.  Everything is vectorised over the measurements.

Each class has the reference's name, constructor (`scheme`) and `get_signal` signature, and follows the same physical
definitions (cited per class); the numerical evaluation is this module's own:

* tensor family (Stick, Zeppelin, Ball, Tensor): exp(-b g'Dg), D diagonal in the scanner frame (synthesis.py:122-143);
* SphereGPD / CylinderGPD: Gaussian-phase-distribution sums over the roots of the Bessel-derivative equations
  (synthesis.py:12-28, 309-346, 445-493); the roots are computed here (scipy) instead of tabulated -- the reference's sphere
  table repeats one root and skips two beyond the 37th (no effect above its 1e-7 stopping precision), its cylinder table is
  accurate to 7 digits only (responses differ by up to 6e-7 of the unit signal);
* Astrosticks: powder average of sticks, sqrt(pi)/(2 sqrt(b d)) erf(sqrt(b d)) (synthesis.py:368-392);
* NODDI (synthesis.py:495-845, a port of the NODDI MATLAB toolbox): intra-cellular = Watson-distributed sticks in the
  spherical-harmonic form of the toolbox, E = 1/2 sum_{n<=6} L_n(b d) C_n(kappa) N_2n P_2n(cos theta), with L_n the
  Legendre-Gaussian integrals and C_n the SH coefficients of the Watson distribution -- both evaluated by Gauss-Legendre
  quadrature of their defining integrals (the reference uses closed forms and recurrences that agree to ~1e-9 where they are
  stable; its recurrence for L_n loses digits just above its switch point x = 0.05, i.e. for b < 90 s/mm^2), except where the
  toolbox deliberately substitutes approximations: polynomial fits in log(kappa/30) for kappa > 30 and the leading Taylor
  terms for kappa <= 0.1 -- those are restated so that the result is the reference's there too; extra-cellular = the Watson-averaged zeppelin with tortuosity; isotropic = exp(-b d).
  Only the b-value of a measurement enters (the toolbox's PGSE protocol conversion, synthesis.py:30-92, reproduces b).
"""
import numpy as np
from scipy import special

GAMMA = 2.675987e8                 # proton gyromagnetic ratio used by the reference (synthesis.py:10), rad / (s T)
_GPD_PRECISION = 1e-7              # synthesis.py:7: a GPD sum stops when a term falls below this fraction of the sum
_N_ROOTS = 60


def _b(scheme):
    return np.asarray(scheme.b, dtype=np.float64)


def _dirs(scheme):
    return np.asarray(scheme.raw, dtype=np.float64)[:, :3]


# ------------------------------------------------------------------------------------------------ tensor family
class BaseTensor:
    def __init__(self, scheme):
        self.scheme = scheme

    def _get_signal(self, evals):
        g = _dirs(self.scheme)
        return np.exp(-_b(self.scheme) * ((g * g) @ np.asarray(evals, dtype=np.float64)))


class Tensor(BaseTensor):
    def get_signal(self, diff_par, diff_perp1, diff_perp2):
        return self._get_signal([diff_perp1, diff_perp2, diff_par])


class Stick(BaseTensor):
    def get_signal(self, diff):
        return self._get_signal([0.0, 0.0, diff])


class Zeppelin(BaseTensor):
    def get_signal(self, diff_par, diff_perp):
        return self._get_signal([diff_perp, diff_perp, diff_par])


class Ball(BaseTensor):
    def get_signal(self, diff):
        return self._get_signal([diff, diff, diff])


# ------------------------------------------------------------------------------------------------ restricted compartments
def _sphere_roots(n=_N_ROOTS):
    """positive roots of d/dx j_1(x) = 0 (Neumann condition on a sphere): x j_1'(x) = 0  <=>  tan x = 2x / (2 - x^2)"""
    f = lambda x: special.spherical_jn(1, x, derivative=True)
    roots, x = [], 1.0
    from scipy.optimize import brentq
    while len(roots) < n:
        if f(x) * f(x + 0.5) < 0:
            roots.append(brentq(f, x, x + 0.5, xtol=1e-14))
        x += 0.5
    return np.array(roots)


_SPHERE_AM = None
_CYL_AM = None


def _gpd_sum(am, big_delta, small_delta, diff, radius, n):
    """sum over the roots a_m of  [2 D a^2 delta - 2 + 2 e^{-D a^2 delta} + 2 e^{-D a^2 Delta} - e^{-D a^2 (Delta - delta)}
    - e^{-D a^2 (Delta + delta)}] / [D^2 a^6 (R^2 a^2 - n)]   (n = 2 sphere, 1 cylinder), stopped like synthesis.py:12-28"""
    total = 0.0
    for a in am:
        dam = diff * a * a
        nom = (2 * dam * small_delta - 2 + 2 * np.exp(-dam * small_delta) + 2 * np.exp(-dam * big_delta)
               - np.exp(-dam * (big_delta - small_delta)) - np.exp(-dam * (big_delta + small_delta)))
        term = nom / (dam * dam * a * a * (radius * radius * a * a - n))
        total += term
        if term < _GPD_PRECISION * total:
            break
    return total


def _per_sequence(scheme, fn):
    """fn(G, Delta, delta) evaluated once per distinct (Delta, delta) pair of a STEJSKALTANNER scheme"""
    raw = np.asarray(scheme.raw, dtype=np.float64)
    out = np.zeros(raw.shape[0])
    keys = {}
    for i in range(raw.shape[0]):
        keys.setdefault((raw[i, 4], raw[i, 5]), []).append(i)
    return raw, keys, out


class SphereGPD:
    """impermeable sphere, GPD approximation (Murday-Cotts / Balinov); diff in mm^2/s, radius in m (synthesis.py:309-346)"""
    def __init__(self, scheme):
        self.scheme = scheme

    def get_signal(self, diff, radius):
        global _SPHERE_AM
        if _SPHERE_AM is None:
            _SPHERE_AM = _sphere_roots()
        d = diff * 1e-6
        raw, keys, sig = _per_sequence(self.scheme, None)
        gmod = np.linalg.norm(raw[:, :3], axis=1) * raw[:, 3]
        for (big, small), idx in keys.items():
            s = _gpd_sum(_SPHERE_AM / radius, big, small, d, radius, 2)
            sig[idx] = np.exp(-2.0 * GAMMA * GAMMA * gmod[idx] ** 2 * s)
        sig[np.all(raw[:, :3] == 0, axis=1)] = 1.0
        return sig


class CylinderGPD:
    """impermeable cylinder along (theta, phi), GPD approximation perpendicular (Van Gelderen), free diffusion along the axis
    (synthesis.py:445-493)"""
    def __init__(self, scheme):
        self.scheme = scheme

    def get_signal(self, diff, radius, theta=0, phi=0):
        global _CYL_AM
        if _CYL_AM is None:
            _CYL_AM = special.jnp_zeros(1, _N_ROOTS)
        d = diff * 1e-6
        n = np.array([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)])
        raw, keys, sig = _per_sequence(self.scheme, None)
        gvec = raw[:, :3] * raw[:, 3:4]
        gmod = np.linalg.norm(gvec, axis=1)
        with np.errstate(invalid='ignore', divide='ignore'):
            cosw = np.where(gmod > 0, (gvec @ n) / (gmod * np.linalg.norm(n)), 0.0)
        cosw = np.clip(cosw, -1.0, 1.0)
        sin2 = 1.0 - cosw * cosw
        for (big, small), idx in keys.items():
            s = _gpd_sum(_CYL_AM / radius, big, small, d, radius, 1)
            perp = np.exp(-2.0 * GAMMA * GAMMA * gmod[idx] ** 2 * sin2[idx] * s)
            q_par = GAMMA * small * gmod[idx] * cosw[idx]
            sig[idx] = perp * np.exp(-(big - small / 3.0) * q_par * q_par * d)
        sig[np.all(raw[:, :3] == 0, axis=1)] = 1.0
        return sig


class Astrosticks:
    """sticks with uniformly distributed orientations (synthesis.py:368-392)"""
    def __init__(self, scheme):
        self.scheme = scheme

    def get_signal(self, diff):
        raw = np.asarray(self.scheme.raw, dtype=np.float64)
        bd = _b(self.scheme) * diff
        with np.errstate(invalid='ignore', divide='ignore'):
            sig = np.where(bd > 0, np.sqrt(np.pi) / (2.0 * np.sqrt(bd)) * special.erf(np.sqrt(bd)), 1.0)
        sig[np.all(raw[:, :3] == 0, axis=1)] = 1.0
        return sig


# ------------------------------------------------------------------------------------------------ NODDI
_GL_T, _GL_W = np.polynomial.legendre.leggauss(96)

# SH coefficients of the Watson distribution for kappa > 30: the toolbox replaces the closed forms (which lose all digits there)
# by sixth-order polynomials in ln(kappa / 30) (synthesis.py:732-745); rows = orders 2, 4, .. 12, columns = powers 0 .. 6
_WATSON_FIT_LARGE = np.array([
    [7.52308, 0.411538, -0.214588, 0.0784091, -0.023981, 0.00731537, -0.0026467],
    [8.93718, 1.62147, -0.733421, 0.191568, -0.0202906, -0.00779095, 0.00574847],
    [8.87905, 3.35689, -1.15935, 0.0673053, 0.121857, -0.066642, 0.0180215],
    [7.84352, 5.03178, -1.0193, -0.426362, 0.328816, -0.0688176, -0.0229398],
    [6.30113, 6.09914, -0.16088, -1.05578, 0.338069, 0.0937157, -0.106935],
    [4.65678, 6.30069, 1.13754, -1.38393, -0.0134758, 0.331686, -0.105954]])


def legendre_gaussian_integral(x, n=6):
    """L_k(x) = int_{-1}^{1} exp(-x t^2) P_2k(t) dt, k = 0 .. n, for an array x >= 0 -> [len(x), n + 1]
    (synthesis.py:600-656 evaluates the same integrals by recurrence for x > 0.05 and by Taylor series below)"""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    e = np.exp(-x[:, None] * (_GL_T * _GL_T)[None, :]) * _GL_W[None, :]
    P = np.stack([special.eval_legendre(2 * k, _GL_T) for k in range(n + 1)], axis=1)
    return e @ P


def watson_sh_coeff(kappa, n=6):
    """C_k = int f(x) Y_{2k,0}(x) dOmega for the Watson density f = exp(kappa (x.mu)^2) normalised to int f dOmega = 4 pi
    (so C_0 = 2 sqrt(pi)), k = 0 .. n (synthesis.py:658-759)"""
    C = np.zeros(n + 1)
    C[0] = 2.0 * np.sqrt(np.pi)
    if kappa > 30.0:
        ln = np.log(kappa) - np.log(30.0)
        C[1:] = _WATSON_FIT_LARGE[:n] @ ln ** np.arange(7)
        return C
    if kappa <= 0.1:
        # the toolbox switches to the leading terms of the Taylor series here (synthesis.py:747-758); kept, so that the
        # default dictionary's most dispersed atoms (IC_OD = 0.99: kappa = 0.016) are the reference's to rounding
        k = float(kappa)
        lead = [np.sqrt(np.pi / 5.0) * (4.0 * k / 3.0 + 8.0 * k ** 2 / 63.0),
                np.sqrt(np.pi) * 0.2 * (8.0 * k ** 2 / 21.0 + 32.0 * k ** 3 / 693.0),
                np.sqrt(np.pi / 13.0) * (16.0 * k ** 3 / 693.0 + 32.0 * k ** 4 / 10395.0),
                np.sqrt(np.pi / 17.0) * 32.0 * k ** 4 / 19305.0,
                np.sqrt(np.pi / 21.0) * 64.0 * k ** 5 / 692835.0,
                np.sqrt(np.pi) * 128.0 * k ** 6 / 152108775.0]
        C[1:] = lead[:n]
        return C
    # exp(kappa (t^2 - 1)) keeps the integrand in range for every kappa <= 30
    w = np.exp(kappa * (_GL_T * _GL_T - 1.0)) * _GL_W
    norm = w.sum()
    for k in range(1, n + 1):
        C[k] = 4.0 * np.pi * np.sqrt((4 * k + 1) / (4.0 * np.pi)) * (w @ special.eval_legendre(2 * k, _GL_T)) / norm
    return C


def _cos_to_z(scheme):
    g = _dirs(scheme).copy()
    b0 = _b(scheme) == 0
    g[b0] = [1.0, 0.0, 0.0]                                   # synthesis.py:84-86
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return np.clip(g[:, 2], -1.0, 1.0)


class NODDIIntraCellular:
    """Watson-distributed sticks (synthesis.py:500-569)"""
    def __init__(self, scheme):
        self.scheme = scheme

    def get_signal(self, diff_par, kappa):
        bd = _b(self.scheme) * diff_par                        # = -LePar of the toolbox: gamma^2 delta^2 G^2 (Delta - delta/3) d
        L = legendre_gaussian_integral(bd, 6)
        C = watson_sh_coeff(float(kappa), 6)
        ct = _cos_to_z(self.scheme)
        sh = np.stack([np.sqrt((k + 0.25) / np.pi) * special.eval_legendre(2 * k, ct) for k in range(7)], axis=1)
        E = (L * C[None, :] * sh).sum(axis=1)
        if np.any(E <= 0):
            E[E <= 0] = E[E > 0].min() * 0.1                   # the truncated series may dip below zero (synthesis.py:565-567)
        return 0.5 * E


class NODDIExtraCellular:
    """hindered compartment: zeppelin with tortuosity d_perp = d_par (1 - v_ic), averaged over the Watson distribution in
    closed form (synthesis.py:766-826)"""
    def __init__(self, scheme):
        self.scheme = scheme

    @staticmethod
    def _watson_hindered(d_par, d_perp, kappa):
        dm = d_par - d_perp
        if kappa < 1e-5:
            dp2 = d_par + 2.0 * d_perp
            k2 = kappa * kappa
            return (dp2 / 3.0 + 4.0 * dm * kappa / 45.0 + 8.0 * dm * k2 / 945.0,
                    dp2 / 3.0 - 2.0 * dm * kappa / 45.0 - 4.0 * dm * k2 / 945.0)
        sk = np.sqrt(kappa)
        factor = sk / special.dawsn(sk)                        # dawsn(x) = sqrt(pi)/2 exp(-x^2) erfi(x)
        return ((-dm + 2.0 * d_perp * kappa + dm * factor) / (2.0 * kappa),
                (dm + 2.0 * (d_par + d_perp) * kappa - dm * factor) / (4.0 * kappa))

    def get_signal(self, diff_par, kappa, vol_ic):
        d_par_w, d_perp_w = self._watson_hindered(diff_par, diff_par * (1.0 - vol_ic), float(kappa))
        c2 = _cos_to_z(self.scheme) ** 2
        return np.exp(-_b(self.scheme) * ((d_par_w - d_perp_w) * c2 + d_perp_w))


class NODDIIsotropic:
    def __init__(self, scheme):
        self.scheme = scheme

    def get_signal(self, diff_iso):
        return np.exp(-_b(self.scheme) * diff_iso)
