"""ctypes binding of the CPU oracle (oracle/libamico_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the amico_amd product path.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)


class NoddiArgs(C.Structure):
    _fields_ = [("n_vox", C.c_int), ("nS", C.c_int), ("ndirs", C.c_int), ("n_wm", C.c_int),
                ("is_exvivo", C.c_int), ("dwi_count", C.c_int),
                ("dwi_idx", C.POINTER(C.c_int64)), ("wm", c_fp), ("iso", c_fp),
                ("norms", c_dp), ("icvf", c_fp), ("kappa", c_fp),
                ("htable", C.POINTER(C.c_int16)), ("lambda1", C.c_double), ("lambda2", C.c_double),
                ("compute_rmse", C.c_int), ("compute_nrmse", C.c_int), ("compute_mod", C.c_int),
                ("nthreads", C.c_int)]


class FwArgs(C.Structure):
    _fields_ = [("n_vox", C.c_int), ("nS", C.c_int), ("ndirs", C.c_int), ("n_perp", C.c_int),
                ("n_iso", C.c_int), ("is_mouse", C.c_int), ("D", c_fp), ("CSF", c_fp),
                ("htable", C.POINTER(C.c_int16)), ("lambda1", C.c_double), ("lambda2", C.c_double),
                ("compute_rmse", C.c_int), ("compute_nrmse", C.c_int), ("save_corrected", C.c_int),
                ("nthreads", C.c_int)]


class SandiArgs(C.Structure):
    _fields_ = [("n_vox", C.c_int), ("nS", C.c_int), ("n_rs", C.c_int), ("n_in", C.c_int),
                ("n_iso", C.c_int), ("signal", c_dp), ("norms", c_dp), ("Rs", c_dp),
                ("d_in", c_dp), ("d_isos", c_dp), ("lambda1", C.c_double), ("lambda2", C.c_double),
                ("compute_rmse", C.c_int), ("compute_nrmse", C.c_int), ("nthreads", C.c_int)]


class CzbArgs(C.Structure):
    _fields_ = [("n_vox", C.c_int), ("nS", C.c_int), ("ndirs", C.c_int), ("n_rs", C.c_int), ("n_perp", C.c_int),
                ("n_iso", C.c_int), ("wmr", c_fp), ("wmh", c_fp), ("iso", c_fp), ("Rs", c_dp),
                ("htable", C.POINTER(C.c_int16)), ("lambda1", C.c_double), ("lambda2", C.c_double),
                ("compute_rmse", C.c_int), ("compute_nrmse", C.c_int), ("nthreads", C.c_int)]


def build(force=False, fast=False):
    so = os.path.join(_HERE, "libamico_oracle_fast.so" if fast else "libamico_oracle.so")
    src = os.path.join(_HERE, "amico_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def use_fast_build(on=True):
    """switch to the -O3 -march=native build of the same source (bench.py's cpu_baseline leg only; the checker build
    stays the default)"""
    global _LIB, _FAST
    _FAST, _LIB = bool(on), None
    if on:      # -march=native: always compile on the machine that is going to run it
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libamico_oracle_fast.so"])


_FAST = False


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build(fast=_FAST))
        L.amo_dir_to_lut_idx.restype = C.c_int
        L.amo_dir_to_lut_idx.argtypes = [c_dp, C.POINTER(C.c_int16), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.amo_nnls.restype = C.c_int
        L.amo_nnls.argtypes = [c_dp, c_dp, C.c_int, C.c_int, c_dp, c_dp]
        L.amo_lasso.restype = C.c_int
        L.amo_lasso.argtypes = [c_dp, c_dp, C.c_int, C.c_int, c_dp, C.c_double, C.c_double]
        L.amo_noddi_fit.restype = C.c_int64
        L.amo_noddi_fit.argtypes = [C.POINTER(NoddiArgs), c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.amo_freewater_fit.restype = C.c_int64
        L.amo_freewater_fit.argtypes = [C.POINTER(FwArgs), c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.amo_sandi_fit.restype = C.c_int64
        L.amo_sandi_fit.argtypes = [C.POINTER(SandiArgs), c_dp, c_dp, c_dp, c_dp, c_dp]
        L.amo_czb_fit.restype = C.c_int64
        L.amo_czb_fit.argtypes = [C.POINTER(CzbArgs), c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _fp(a):
    return a.ctypes.data_as(c_fp)


def dir_to_lut_idx(dirs, htable):
    """Vectorised wrapper: dirs f64[n,3] -> (idx int32[n], ii1, ii2); idx=-1 if out of bounds."""
    dirs = np.ascontiguousarray(np.atleast_2d(dirs), dtype=np.float64)
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    n = dirs.shape[0]
    out = np.empty(n, np.int32); i1 = np.empty(n, np.int32); i2 = np.empty(n, np.int32)
    a, b = C.c_int(), C.c_int()
    L = lib()
    for k in range(n):
        out[k] = L.amo_dir_to_lut_idx(dirs[k].ctypes.data_as(c_dp), ht.ctypes.data_as(C.POINTER(C.c_int16)),
                                      C.byref(a), C.byref(b))
        i1[k], i2[k] = a.value, b.value
    return out, i1, i2


def nnls(A, y):
    A = np.asfortranarray(A, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    m, n = A.shape
    x = np.zeros(n); rn = C.c_double()
    mode = lib().amo_nnls(_dp(A), _dp(y), m, n, _dp(x), C.byref(rn))
    return x, rn.value, mode


def lasso(A, y, lambda1, lambda2):
    A = np.asfortranarray(A, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    m, n = A.shape
    x = np.zeros(n)
    st = lib().amo_lasso(_dp(A), _dp(y), m, n, _dp(x), float(lambda1), float(lambda2))
    return x, st


def noddi_fit(y, dirs, kernels, htable, dwi_idx, lambda1=0.5, lambda2=1e-3, is_exvivo=False,
              rmse=False, nrmse=False, mod=False, nthreads=1, return_x=False):
    y = np.ascontiguousarray(y, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
    wm = np.ascontiguousarray(kernels['wm'], dtype=np.float32)
    iso = np.ascontiguousarray(kernels['iso'], dtype=np.float32)
    norms = np.ascontiguousarray(kernels['norms'], dtype=np.float64)
    icvf = np.ascontiguousarray(kernels['icvf'], dtype=np.float32)
    kappa = np.ascontiguousarray(kernels['kappa'], dtype=np.float32)
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    dwi_idx = np.ascontiguousarray(dwi_idx, dtype=np.int64)
    n_vox, nS = y.shape
    n_wm, ndirs, _ = wm.shape
    n_atoms = n_wm + 1 + (1 if is_exvivo else 0)
    nmaps = 3 + (1 if is_exvivo else 0)
    a = NoddiArgs(n_vox, nS, ndirs, n_wm, int(is_exvivo), len(dwi_idx),
                  dwi_idx.ctypes.data_as(C.POINTER(C.c_int64)), _fp(wm), _fp(iso), _dp(norms),
                  _fp(icvf), _fp(kappa), ht.ctypes.data_as(C.POINTER(C.c_int16)),
                  float(lambda1), float(lambda2), int(rmse), int(nrmse), int(mod), int(nthreads))
    est = np.zeros((n_vox, nmaps))
    r = np.zeros(n_vox) if rmse else None
    nr = np.zeros(n_vox) if nrmse else None
    md = np.zeros((n_vox, 2)) if mod else None
    xd = np.zeros((n_vox, 3, n_atoms)) if return_x else None
    err = lib().amo_noddi_fit(C.byref(a), _dp(y), _dp(dirs), _dp(est), _dp(r), _dp(nr), _dp(md), _dp(xd))
    out = {'estimates': est, 'err': err}
    if rmse: out['rmse'] = r
    if nrmse: out['nrmse'] = nr
    if mod: out['estimates_mod'] = md
    if return_x: out['x'] = xd
    return out


def freewater_fit(y, dirs, kernels, htable, lambda1=0.0, lambda2=1e-3, is_mouse=False,
                  rmse=False, nrmse=False, corrected=False, nthreads=1, return_x=False):
    y = np.ascontiguousarray(y, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
    D = np.ascontiguousarray(kernels['D'], dtype=np.float32)
    CSF = np.ascontiguousarray(kernels['CSF'], dtype=np.float32)
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    n_vox, nS = y.shape
    n_perp, ndirs, _ = D.shape
    n_iso = CSF.shape[0]
    nmaps = 4 if is_mouse else 2
    a = FwArgs(n_vox, nS, ndirs, n_perp, n_iso, int(is_mouse), _fp(D), _fp(CSF),
               ht.ctypes.data_as(C.POINTER(C.c_int16)), float(lambda1), float(lambda2),
               int(rmse), int(nrmse), int(corrected), int(nthreads))
    est = np.zeros((n_vox, nmaps))
    r = np.zeros(n_vox) if rmse else None
    nr = np.zeros(n_vox) if nrmse else None
    yc = np.zeros((n_vox, nS)) if corrected else None
    xd = np.zeros((n_vox, n_perp + n_iso)) if return_x else None
    err = lib().amo_freewater_fit(C.byref(a), _dp(y), _dp(dirs), _dp(est), _dp(r), _dp(nr), _dp(yc), _dp(xd))
    out = {'estimates': est, 'err': err}
    if rmse: out['rmse'] = r
    if nrmse: out['nrmse'] = nr
    if corrected: out['y_corrected'] = yc
    if return_x: out['x'] = xd
    return out


def sandi_fit(y, kernels, Rs, d_in, d_isos, lambda1=0.0, lambda2=5e-3, rmse=False, nrmse=False,
              nthreads=1, return_x=False):
    y = np.ascontiguousarray(y, dtype=np.float64)
    sig = np.asfortranarray(kernels['signal'], dtype=np.float64)
    norms = np.ascontiguousarray(kernels['norms'], dtype=np.float64)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64); d_in = np.ascontiguousarray(d_in, dtype=np.float64)
    d_isos = np.ascontiguousarray(d_isos, dtype=np.float64)
    n_vox, nS = y.shape
    a = SandiArgs(n_vox, nS, len(Rs), len(d_in), len(d_isos), _dp(sig), _dp(norms), _dp(Rs), _dp(d_in),
                  _dp(d_isos), float(lambda1), float(lambda2), int(rmse), int(nrmse), int(nthreads))
    est = np.zeros((n_vox, 6))
    r = np.zeros(n_vox) if rmse else None
    nr = np.zeros(n_vox) if nrmse else None
    xd = np.zeros((n_vox, sig.shape[1])) if return_x else None
    lib().amo_sandi_fit(C.byref(a), _dp(y), _dp(est), _dp(r), _dp(nr), _dp(xd))
    out = {'estimates': est, 'err': 0}
    if rmse: out['rmse'] = r
    if nrmse: out['nrmse'] = nr
    if return_x: out['x'] = xd
    return out


def czb_fit(y, dirs, kernels, Rs, htable, lambda1=0.0, lambda2=4.0, rmse=False, nrmse=False, nthreads=1, return_x=False):
    """CylinderZeppelinBall._fit (models.pyx:526-652): KERNELS['wmr'] / ['wmh'] / ['iso'], model.Rs -> v, a, d"""
    y = np.ascontiguousarray(y, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
    wmr = np.ascontiguousarray(kernels['wmr'], dtype=np.float32)
    wmh = np.ascontiguousarray(kernels['wmh'], dtype=np.float32)
    iso = np.ascontiguousarray(kernels['iso'], dtype=np.float32)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64)
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    n_vox, nS = y.shape
    n_rs, ndirs, _ = wmr.shape
    a = CzbArgs(n_vox, nS, ndirs, n_rs, wmh.shape[0], iso.shape[0], _fp(wmr), _fp(wmh), _fp(iso), _dp(Rs),
                ht.ctypes.data_as(C.POINTER(C.c_int16)), float(lambda1), float(lambda2), int(rmse), int(nrmse), int(nthreads))
    est = np.zeros((n_vox, 3))
    r = np.zeros(n_vox) if rmse else None
    nr = np.zeros(n_vox) if nrmse else None
    xd = np.zeros((n_vox, n_rs + wmh.shape[0] + iso.shape[0])) if return_x else None
    err = lib().amo_czb_fit(C.byref(a), _dp(y), _dp(dirs), _dp(est), _dp(r), _dp(nr), _dp(xd))
    out = {'estimates': est, 'err': err}
    if rmse: out['rmse'] = r
    if nrmse: out['nrmse'] = nr
    if return_x: out['x'] = xd
    return out
