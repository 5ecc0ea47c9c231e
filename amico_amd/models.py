"""Model plug-ins with the surface of ``amico.models`` (amico/models.pyx:75-217, 655-991,
995-1286, 1344-1627): same names, constructor defaults, ``set`` / ``get_params`` /
``set_solver`` and ``fit(evaluation) -> dict``; the per-voxel work is done by the HIP library
through the C ABI (``amico_amd._capi``).  ``generate`` (response functions, ``amico_amd.synthesis``) and ``resample`` (the
GPU GEMM of ``amico_amd.lut``) produce the reference's ``KERNELS`` dict; ``amico_amd.synthetic`` builds realistic
dictionaries directly in signal space for tests and benchmarks.
"""
from abc import ABC, abstractmethod
import numpy as np
from . import _capi

_CTX = None
_CTXS = None          # one context per device of the fit's device set (set_devices / AMX_DEVICES); _CTXS[0] is _CTX
_DEVICES = None


def set_devices(ids=None):
    """The devices `model.fit(evaluation)` spreads a host-buffer fit over (round 6).  The reference's caller is ONE process whose fit
    cuts the voxels into contiguous chunks, one per host thread (core.py:465-466, models.pyx:204-211); here the chunks are one per GPU:
    a context and a host thread per device, each shard travelling over its own PCIe link, the results written straight into the
    caller's arrays (no collective: they go home anyway).  ids: a list of HIP device numbers (a device may be named twice: two
    contexts on it -- what the one-GPU test box exercises), 'all', or None = the current device only (the default; the environment
    variable AMX_DEVICES=all | 0,1,... sets the same without touching the caller's script)."""
    global _DEVICES
    if isinstance(ids, str):
        ids = None if not ids.strip() else ('all' if ids.strip().lower() == 'all' else [int(t) for t in ids.replace(',', ' ').split()])
    _DEVICES = ids if ids is None or ids == 'all' else [int(i) for i in ids]
    reset_context()


def _device_ids():
    import os
    ids = _DEVICES
    if ids is None:
        e = os.environ.get('AMX_DEVICES', '').strip()
        if not e:
            return None
        ids = 'all' if e.lower() == 'all' else [int(t) for t in e.replace(',', ' ').split()]
    if ids == 'all':
        ids = list(range(_capi.device_count()))
    return ids or None


def get_contexts():
    """the contexts of the fit's device set, in shard order (one, on the current device, unless set_devices / AMX_DEVICES says otherwise)"""
    global _CTX, _CTXS
    if _CTXS is None:
        ids = _device_ids()
        _CTXS = [_capi.Context(-1)] if ids is None else [_capi.Context(i) for i in ids]
        _CTX = _CTXS[0]
    return _CTXS


def get_context():
    """process-wide amx_ctx (created on first use): on the current HIP device, or the first device of the fit's device set"""
    global _CTX
    if _CTX is None:
        get_contexts()
    return _CTX


def reset_context():
    """forget the process-wide context(s): the next get_context() creates new ones (the library reads its AMX_* environment
    switches once, when a context is created -- tools and tests that flip a switch call this afterwards)"""
    global _CTX, _CTXS
    _CTX = None
    _CTXS = None


def _digest(a):
    """content digest of a whole array (xxh3-128 when xxhash is importable, sha1 otherwise)"""
    b = np.ascontiguousarray(a)
    try:
        import xxhash
        return xxhash.xxh3_128(b.data).hexdigest()
    except ImportError:
        import hashlib
        return hashlib.sha1(b.data).hexdigest()


def _fingerprint(K):
    return tuple((k, v.shape, str(v.dtype), _digest(v)) for k, v in sorted(K.items()) if isinstance(v, np.ndarray))


class _PendingFingerprint:
    """the digest of KERNELS on a helper thread (xxhash / hashlib release the GIL), started when a fit finds a cached dictionary whose
    cheap key matches: the fit runs with the cached dictionary meanwhile and `matches()` is asked before its result is returned"""

    def __init__(self, K, expected):
        import threading
        self.expected, self.got, self.err = expected, None, None
        self._t = threading.Thread(target=self._run, args=(K,), daemon=True)
        self._t.start()

    def _run(self, K):
        try:
            self.got = _fingerprint(K)
        except BaseException as e:          # (the caller's thread decides what to do with it)
            self.err = e

    def matches(self):
        self._t.join()
        if self.err is not None:
            raise self.err
        return self.got == self.expected


def _verified_fit(fit):
    """``fit`` with the dictionary cache checked BEHIND the solver: the reference re-reads KERNELS on every fit (models.pyx:840-847), here
    the device dictionary is reused when every byte of KERNELS is what was uploaded -- a 1.2 ms digest of 28 MB per call that need not
    sit in front of a 12.6 ms fit.  A fit that ran on a stale dictionary (KERNELS edited in place since the upload) is discarded and
    run again on the rebuilt one, so the caller never sees its result -- but a progress callback registered on the context does see
    the discarded fit's ticks too (the fit runs twice in that case), and an exception of the stale fit is only believed once the
    digest has confirmed the dictionary (otherwise: upload again, fit again)."""
    import functools

    @functools.wraps(fit)
    def wrapper(self, evaluation):
        self._lut_pending = None            # None: inside a verified fit, nothing started yet (False / absent: outside)
        try:
            try:
                out = fit(self, evaluation)
            except Exception:
                # a fit on a STALE dictionary may also fail (shapes still match, values do not): ask the digest before believing the error
                pend, self._lut_pending = self._lut_pending, False
                if isinstance(pend, _PendingFingerprint) and not pend.matches():
                    self._lut_cache = {}
                    return fit(self, evaluation)
                raise
            pend, self._lut_pending = self._lut_pending, False
            if pend is not None and not pend.matches():
                self._lut_cache = {}        # stale: upload again (every context), fit again
                out = fit(self, evaluation)
            return out
        finally:
            self._lut_pending = False
    return wrapper


class BaseModel(ABC):
    """models.pyx:75-217"""

    @abstractmethod
    def __init__(self):
        self.id = 'BaseModel'
        self.name = 'Base Model'
        self.maps_name = []
        self.maps_descr = []
        self.scheme = None

    @abstractmethod
    def set(self, *args, **kwargs):
        return

    @abstractmethod
    def get_params(self):
        return

    @abstractmethod
    def set_solver(self):
        self.solver_params = {}

    # ---- response functions on the high-resolution scheme -> rotated SH coefficients (models.pyx:444-477, 727-751, 1088-1110,
    #      1414-1444).  `_atoms()` yields (signal on the high-resolution scheme, is_isotropic) in the order of the A_###.npy files.
    def _atoms(self, scheme_high):
        raise NotImplementedError

    def generate(self, out_path, aux, idx_in, idx_out, ndirs):
        """response functions (amico_amd.synthesis) -> lut.rotate_kernel -> `A_%03d.npy` under `out_path` (skipped when
        out_path is None); returns the list of arrays, which `resample` / `Evaluation.load_kernels` accept directly"""
        from . import lut as _lut
        scheme_high = _lut.high_resolution_scheme(self.scheme, aux['grad'])
        lms = []
        for i, (signal, isotropic) in enumerate(self._atoms(scheme_high)):
            lm = _lut.rotate_kernel(signal, aux, idx_in, idx_out, isotropic, ndirs)
            if out_path is not None:
                from os.path import join as pjoin
                np.save(pjoin(out_path, f'A_{i + 1:03d}.npy'), lm)
            lms.append(lm)
        return lms

    # ---- resampling of the rotated SH coefficients to the subject's scheme (models.pyx:754-792, 1113-1144,
    #      1446-1486): `in_path` is the reference's folder of A_###.npy files or a list of the arrays themselves.
    @staticmethod
    def _load_lm(in_path, n_atoms):
        if isinstance(in_path, (str, bytes)):
            from os.path import join as pjoin
            return [np.load(pjoin(in_path, f'A_{i + 1:03d}.npy')) for i in range(n_atoms)]
        if len(in_path) != n_atoms:
            raise ValueError('Outdated LUT. Call "generate_kernels( regenerate=True )" to update the LUT')
        return list(in_path)

    def _merge(self, doMergeB0):
        """(nS of the resampled kernels, columns kept): with doMergeB0 the first b0 + the DWI volumes"""
        sc = self.scheme
        if doMergeB0:
            return 1 + sc.dwi_count, np.hstack((sc.b0_idx[0], sc.dwi_idx))
        return sc.nS, np.arange(sc.nS)

    def _resample_rotated(self, lms, idx_out, Ylm_out, ndirs):
        """all anisotropic atoms in ONE GEMM on the GPU (amx_lut_resample) -> f32[n, ndirs, scheme.nS]"""
        from . import lut as _lut
        for lm in lms:
            if np.ndim(lm) != 2 or lm.shape[0] != ndirs:
                raise ValueError('Outdated LUT. Call "generate_kernels( regenerate=True )" to update the LUT')
        return _lut.resample_kernels(np.stack(lms).astype(np.float32, copy=False), self.scheme.nS, idx_out, Ylm_out)

    def _resample_isotropic(self, lms, idx_out, Ylm_out):
        from . import lut as _lut
        return _lut.resample_kernels(np.stack(lms).astype(np.float32, copy=False), self.scheme.nS, idx_out, Ylm_out)

    def resample(self, in_path, idx_out, Ylm_out, doMergeB0, ndirs):
        raise NotImplementedError

    @abstractmethod
    def fit(self, evaluation):
        # models.pyx:204-217: chunks are only kept for introspection -- the GPU path shards
        # nothing across host threads, results are voxel-order identical by construction
        dev = getattr(evaluation, '_dev', None)
        n = dev['y'].shape[0] if dev is not None else evaluation.y.shape[0]     # (no download of a device-resident y)
        nthreads = max(1, int(evaluation.nthreads or 1))
        c = max(1, n // nthreads)
        self.chunks = [(i, j) for i, j in zip(range(0, n, c), range(c, n + 1, c))]
        if self.chunks and self.chunks[-1][1] != n:
            self.chunks[-1] = (self.chunks[-1][0], n)
        self.configs = {
            'compute_rmse': evaluation.get_config('doComputeRMSE'),
            'compute_nrmse': evaluation.get_config('doComputeNRMSE'),
        }

    # ---- device-resident inputs: Evaluation.fit leaves `y` (and `DIRs`) in HBM (evaluation._dev) when it produced
    #      them on the GPU; the fit then reads them in place and keeps its outputs there for the scatter
    @staticmethod
    def _warn_if_capped(ctx):
        """voxels whose active-set iteration hit its cap keep the last iterate: say so (the reference's solvers are silent)"""
        st = ctx.last_stats()
        if st['itercap_voxels'] > 0:
            import warnings
            warnings.warn('amico_amd: %d voxel(s) stopped at the iteration cap of the active-set solver' % st['itercap_voxels'],
                          RuntimeWarning)

    @classmethod
    def _finish_device(cls, ctx, dev, named):
        ctx.sync()
        cls._warn_if_capped(ctx)
        dev['out'] = {k: v for k, v in named.items() if v is not None}
        return {k: v.cpu().numpy() for k, v in dev['out'].items()}

    @staticmethod
    def _dev_dirs(evaluation, dev):
        """directions of the device-resident path: the tensor Evaluation.fit left in HBM, or -- when the caller has
        assigned `evaluation.DIRs` since -- that array, uploaded"""
        d = dev.get('dirs')
        if d is None and getattr(evaluation, '_DIRs', None) is not None:
            import torch
            d = torch.from_numpy(np.ascontiguousarray(evaluation._DIRs, dtype=np.float64)).to(dev['y'].device)
            dev['dirs'] = d
        return d

    # ---- dictionary cache: one upload per (KERNELS, htable) object pair AND model / scheme state that shapes the
    #      device dictionary.  The keyed objects are held (an id() can be recycled once its object is collected), and
    #      a digest of EVERY byte of the arrays catches in-place edits of KERNELS (the reference re-reads KERNELS on every
    #      fit, models.pyx:840-847; xxh3 of the default NODDI dictionary, 28 MB: 4 ms).
    def _lut_extra_key(self):
        return ()

    def _lut(self, evaluation, builder, ctx=None):
        """the device dictionary of `ctx` (default: the process-wide context) for evaluation.KERNELS: cached per context"""
        K, ht = evaluation.KERNELS, getattr(evaluation, 'htable', None)
        sc = self.scheme
        skey = None if sc is None else (int(getattr(sc, 'nS', 0)), tuple(np.asarray(getattr(sc, 'dwi_idx', ())).tolist()))
        if ctx is None:
            ctx = get_context()             # (a dictionary lives in ONE context: reset_context() must not leave a stale upload behind)
        # everything but the content: objects, shapes, types, the model / scheme state that shapes the device dictionary
        cheap = (id(K), id(ht), tuple((k, v.shape, str(v.dtype)) for k, v in sorted(K.items()) if isinstance(v, np.ndarray)),
                 skey, self._lut_extra_key(), id(ctx))
        caches = getattr(self, '_lut_cache', None)
        if not isinstance(caches, dict):
            caches = self._lut_cache = {}
        for key in [k for k, c in caches.items() if c[4] is not ctx and c[4] not in (_CTXS or [])]:
            del caches[key]                 # (uploads of contexts that reset_context() has dropped)
        cache = caches.get(id(ctx))
        if cache is not None and cache[0][0] == cheap:
            # the content is checked while the fit runs (_verified_fit); outside a fit (no wrapper to ask the question) right here
            pend = getattr(self, '_lut_pending', False)
            if pend is None:
                self._lut_pending = _PendingFingerprint(K, cache[0][1])
                return cache[1]
            if isinstance(pend, _PendingFingerprint) and pend.expected == cache[0][1]:
                return cache[1]             # (another context of the same fit: one digest of KERNELS answers for all of them)
            if _fingerprint(K) == cache[0][1]:
                return cache[1]
        if K.get('model') != self.id:
            raise ValueError('Response functions were not created with the same model')
        caches[id(ctx)] = ((cheap, _fingerprint(K)), builder(), K, ht, ctx)     # K, ht, ctx: strong references to the keyed objects
        return caches[id(ctx)][1]

    # ---- host-buffer fit on several devices: contiguous shards (models.pyx:204-211), a context + host thread per device
    def _fit_on_devices(self, evaluation, n, upload, fit_shard, outs):
        """upload(ctx) -> Lut; fit_shard(ctx, lut, lo, hi, out_views) fits voxels [lo, hi) into row slices of `outs` (a tuple of
        arrays / None like the return value of the _capi fit).  Returns outs.  One device: a plain call."""
        from .parallel import shard_range
        ctxs = get_contexts()
        luts = [self._lut(evaluation, (lambda c=c: upload(c)), c) for c in ctxs]        # (uploads one after the other, first fit only)
        world = len(ctxs)
        bounds = [shard_range(n, r, world) for r in range(world)]

        def work(r):
            lo, hi = bounds[r]
            if hi <= lo:
                return
            ctxs[r].set_call_voxels(n if world > 1 else 0)      # every shard takes the paths the whole call's size asks for
            try:
                fit_shard(ctxs[r], luts[r], lo, hi, tuple(None if o is None else o[lo:hi] for o in outs))
            except _capi.AmxError as e:
                import re
                m = re.search(r'\[voxel (\d+)\]', str(e))       # (a shard counts its voxels from 0: name the caller's voxel)
                if m and lo:
                    raise _capi.AmxError(e.code, str(e).replace(m.group(0), '[voxel %d]' % (int(m.group(1)) + lo))) from None
                raise
            finally:
                ctxs[r].set_call_voxels(0)
        if world == 1:
            work(0)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(world) as ex:        # (ctypes releases the GIL for the length of the C call)
                futs = [ex.submit(work, r) for r in range(world)]
            errs = [f.exception() for f in futs]
            for e in errs:
                if e is not None:
                    raise e                              # the lowest shard's error first, like the reference's serial loop would hit it
        for c in ctxs:
            self._warn_if_capped(c)
        return outs


class CylinderZeppelinBall(BaseModel):
    """models.pyx:375-652.  A = [cylinders (Rs) | zeppelins (d_perps) | balls (d_isos)] of the voxel's orientation, one
    lasso(lambda1=0, lambda2=4) per voxel, maps v (intra-cellular volume fraction), a (mean axonal diameter, um), d (axonal
    density).  Reference quirk (models.pyx:435, 549): ``get_params`` and ``_fit`` read ``self.isExvivo``, which the class
    never sets -- there the fit raises AttributeError until the user assigns it; here it is False (the "ex vivo" branch of
    the reference only appends an all-zero atom, :552-554, so it changes nothing)."""

    def __init__(self):
        self.id = 'CylinderZeppelinBall'
        self.name = 'Cylinder-Zeppelin-Ball'
        self.maps_name = ['v', 'a', 'd']
        self.maps_descr = ['Intra-cellular volume fraction', 'Mean axonal diameter', 'Axonal density']
        self.scheme = None
        self.isExvivo = False
        self.set()
        self.set_solver()

    def set(self, d_par=0.6E-3, Rs=np.concatenate(([0.01], np.linspace(0.5, 8.0, 20))) * 1E-6,
            d_perps=np.array([1.19E-3, 0.85E-3, 0.51E-3, 0.17E-3]), d_isos=np.array([2.0E-3])):
        self.d_par = d_par
        self.Rs = np.array(Rs)
        self.d_perps = np.array(d_perps)
        self.d_isos = np.array(d_isos)

    def get_params(self):
        return {'id': self.id, 'name': self.name, 'd_par': self.d_par, 'Rs': self.Rs, 'd_perps': self.d_perps,
                'd_isos': self.d_isos, 'isExvivo': self.isExvivo}

    def _lut_extra_key(self):
        return (np.asarray(self.Rs, dtype=np.float64).tobytes(),)

    def set_solver(self, lambda1=0.0, lambda2=4.0):
        super().set_solver()
        self.solver_params['lambda1'] = lambda1
        self.solver_params['lambda2'] = lambda2

    def _atoms(self, scheme_high):
        """models.pyx:444-477"""
        from . import synthesis as syn
        if self.scheme.version != 1:
            raise RuntimeError('This model requires a "VERSION: STEJSKALTANNER" scheme')
        cylinder, zeppelin, ball = syn.CylinderGPD(scheme_high), syn.Zeppelin(scheme_high), syn.Ball(scheme_high)
        for R in self.Rs:
            yield cylinder.get_signal(self.d_par, R), False
        for d in self.d_perps:
            yield zeppelin.get_signal(self.d_par, d), False
        for d in self.d_isos:
            yield ball.get_signal(d), True

    def resample(self, in_path, idx_out, Ylm_out, doMergeB0, ndirs):
        """models.pyx:480-522"""
        n_r, n_p, n_i = len(self.Rs), len(self.d_perps), len(self.d_isos)
        lms = self._load_lm(in_path, n_r + n_p + n_i)
        _, merge_idx = self._merge(doMergeB0)
        K = {'model': self.id}
        rot = self._resample_rotated(lms[:n_r + n_p], idx_out, Ylm_out, ndirs)[:, :, merge_idx]
        K['wmr'] = np.ascontiguousarray(rot[:n_r])
        K['wmh'] = np.ascontiguousarray(rot[n_r:])
        K['iso'] = np.ascontiguousarray(self._resample_isotropic(lms[n_r + n_p:], idx_out, Ylm_out)[:, merge_idx])
        return K

    @_verified_fit
    def fit(self, evaluation):
        super().fit(evaluation)
        ctx = get_context()
        K = evaluation.KERNELS
        if K['wmr'].shape[0] != len(self.Rs) or K['wmh'].shape[0] != len(self.d_perps) or K['iso'].shape[0] != len(self.d_isos):
            raise ValueError('KERNELS do not match Rs / d_perps / d_isos of the model')
        lut = self._lut(evaluation, lambda: _capi.upload_czb(ctx, K, self.Rs, evaluation.htable))
        kw = dict(rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']))
        dev = getattr(evaluation, '_dev', None)
        if dev is not None:
            est, rmse, nrmse = _capi.czb_fit_device(ctx, lut, dev['y'], self._dev_dirs(evaluation, dev),
                                                    self.solver_params['lambda1'], self.solver_params['lambda2'], **kw)
            return self._finish_device(ctx, dev, {'estimates': est, 'rmse': rmse, 'nrmse': nrmse})
        if len(get_contexts()) > 1:
            n, y, d = evaluation.y.shape[0], evaluation.y, evaluation.DIRs
            est, rmse, nrmse = self._fit_on_devices(
                evaluation, n, lambda c: _capi.upload_czb(c, K, self.Rs, evaluation.htable),
                lambda c, l, lo, hi, o: _capi.czb_fit(c, l, y[lo:hi], d[lo:hi], self.solver_params['lambda1'], self.solver_params['lambda2'], out=o, **kw),
                (np.zeros((n, 3)), np.zeros(n) if kw['rmse'] else None, np.zeros(n) if kw['nrmse'] else None))
        else:
            est, rmse, nrmse = _capi.czb_fit(ctx, lut, evaluation.y, evaluation.DIRs, self.solver_params['lambda1'],
                                             self.solver_params['lambda2'], **kw)
            self._warn_if_capped(ctx)
        results = {'estimates': est}
        if self.configs['compute_rmse']:
            results['rmse'] = rmse
        if self.configs['compute_nrmse']:
            results['nrmse'] = nrmse
        return results


class NODDI(BaseModel):
    """models.pyx:655-991"""

    def __init__(self):
        self.id = 'NODDI'
        self.name = 'NODDI'
        self.maps_name = ['NDI', 'ODI', 'FWF']
        self.maps_descr = ['Neurite Density Index', 'Orientation Dispersion Index', 'Free Water Fraction']
        self.scheme = None
        self.set()
        self.set_solver()

    def set(self, dPar=1.7E-3, dIso=3.0E-3, IC_VFs=np.linspace(0.1, 0.99, 12),
            IC_ODs=np.hstack((np.array([0.03, 0.06]), np.linspace(0.09, 0.99, 10))), isExvivo=False):
        self.dPar = dPar
        self.dIso = dIso
        self.IC_VFs = np.array(IC_VFs) if isinstance(IC_VFs, list) else IC_VFs
        self.IC_ODs = np.array(IC_ODs) if isinstance(IC_ODs, list) else IC_ODs
        self.isExvivo = isExvivo
        if isExvivo:
            self.maps_name.append('dot')
            self.maps_descr.append('Dot volume fraction')

    def get_params(self):
        return {'id': self.id, 'name': self.name, 'dPar': self.dPar, 'dIso': self.dIso, 'IC_VFs': self.IC_VFs,
                'IC_ODs': self.IC_ODs, 'isExvivo': self.isExvivo}

    def _lut_extra_key(self):
        return (bool(self.isExvivo),)

    def set_solver(self, lambda1=5e-1, lambda2=1e-3):
        super().set_solver()
        self.solver_params['lambda1'] = lambda1
        self.solver_params['lambda2'] = lambda2

    def _atoms(self, scheme_high):
        """models.pyx:727-751: one atom per (kappa, v_ic) -- v_ic * intra-cellular + (1 - v_ic) * extra-cellular, the orientation
        dispersion index mapped to the Watson concentration kappa = 1 / tan(OD pi / 2) -- then the isotropic atom"""
        from . import synthesis as syn
        ic, ec, iso = syn.NODDIIntraCellular(scheme_high), syn.NODDIExtraCellular(scheme_high), syn.NODDIIsotropic(scheme_high)
        for kappa in 1.0 / np.tan(np.asarray(self.IC_ODs) * np.pi / 2.0):
            signal_ic = ic.get_signal(self.dPar, kappa)
            for v_ic in self.IC_VFs:
                yield v_ic * signal_ic + (1.0 - v_ic) * ec.get_signal(self.dPar, kappa, v_ic), False
        yield iso.get_signal(self.dIso), True

    def resample(self, in_path, idx_out, Ylm_out, doMergeB0, ndirs):
        """models.pyx:754-792"""
        n_wm = len(self.IC_ODs) * len(self.IC_VFs)
        lms = self._load_lm(in_path, n_wm + 1)
        nS, merge_idx = self._merge(doMergeB0)
        K = {'model': self.id}
        K['wm'] = np.ascontiguousarray(self._resample_rotated(lms[:n_wm], idx_out, Ylm_out, ndirs)[:, :, merge_idx])
        K['iso'] = self._resample_isotropic(lms[n_wm:], idx_out, Ylm_out)[0][merge_idx]
        K['kappa'] = np.repeat(1.0 / np.tan(np.asarray(self.IC_ODs) * np.pi / 2.0), len(self.IC_VFs)).astype(np.float32)
        K['icvf'] = np.tile(np.asarray(self.IC_VFs), len(self.IC_ODs)).astype(np.float32)
        K['norms'] = np.zeros((self.scheme.dwi_count, n_wm))
        cols = slice(1, None) if doMergeB0 else self.scheme.dwi_idx
        for a in range(n_wm):
            K['norms'][:, a] = 1 / np.linalg.norm(K['wm'][a, 0, cols])     # norm of coupled atoms (for l1 minimization)
        return K

    @_verified_fit
    def fit(self, evaluation):
        super().fit(evaluation)
        self.configs['compute_modulated_maps'] = evaluation.get_config('doSaveModulatedMaps')
        ctx = get_context()
        n_wm = len(self.IC_ODs) * len(self.IC_VFs)
        if evaluation.KERNELS['wm'].shape[0] != n_wm:
            raise ValueError('KERNELS do not match IC_VFs / IC_ODs of the model')
        lut = self._lut(evaluation, lambda: _capi.upload_noddi(ctx, evaluation.KERNELS, evaluation.htable,
                                                               self.scheme.dwi_idx, self.isExvivo))
        dev = getattr(evaluation, '_dev', None)
        if dev is not None:
            est, rmse, nrmse, mod = _capi.noddi_fit_device(
                ctx, lut, dev['y'], self._dev_dirs(evaluation, dev), self.solver_params['lambda1'], self.solver_params['lambda2'],
                len(self.maps_name), rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']),
                mod=bool(self.configs['compute_modulated_maps']))
            return self._finish_device(ctx, dev, {'estimates': est, 'rmse': rmse, 'nrmse': nrmse, 'estimates_mod': mod})
        kw = dict(rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']), mod=bool(self.configs['compute_modulated_maps']))
        if len(get_contexts()) > 1:
            n, y, d, nm = evaluation.y.shape[0], evaluation.y, evaluation.DIRs, len(self.maps_name)
            est, rmse, nrmse, mod = self._fit_on_devices(
                evaluation, n, lambda c: _capi.upload_noddi(c, evaluation.KERNELS, evaluation.htable, self.scheme.dwi_idx, self.isExvivo),
                lambda c, l, lo, hi, o: _capi.noddi_fit(c, l, y[lo:hi], d[lo:hi], self.solver_params['lambda1'], self.solver_params['lambda2'], nm, out=o, **kw),
                (np.zeros((n, nm)), np.zeros(n) if kw['rmse'] else None, np.zeros(n) if kw['nrmse'] else None, np.zeros((n, 2)) if kw['mod'] else None))
        else:
            est, rmse, nrmse, mod = _capi.noddi_fit(ctx, lut, evaluation.y, evaluation.DIRs, self.solver_params['lambda1'],
                                                    self.solver_params['lambda2'], len(self.maps_name), **kw)
            self._warn_if_capped(ctx)
        results = {'estimates': est}
        if self.configs['compute_rmse']:
            results['rmse'] = rmse
        if self.configs['compute_nrmse']:
            results['nrmse'] = nrmse
        if self.configs['compute_modulated_maps']:
            results['estimates_mod'] = mod
        return results


class FreeWater(BaseModel):
    """models.pyx:995-1286"""

    def __init__(self):
        self.id = 'FreeWater'
        self.name = 'Free-Water'
        self.scheme = None
        self.set()
        self.set_solver()

    def set(self, d_par=None, d_perps=None, d_isos=None, type='Human'):
        self.type = type
        if self.type == 'Mouse':
            self.maps_name = ['FiberVolume', 'FW', 'FW_blood', 'FW_csf']
            self.maps_descr = ['fiber volume fraction', 'Isotropic free-water volume fraction', 'FW blood', 'FW csf']
            self.d_par = 1.0E-3 if d_par is None else d_par
            self.d_perps = np.linspace(0.15, 0.55, 10) * 1E-3 if d_perps is None else d_perps
            self.d_isos = [1.5E-3, 3E-3] if d_isos is None else d_isos
        else:
            self.maps_name = ['FiberVolume', 'FW']
            self.maps_descr = ['fiber volume fraction', 'Isotropic free-water volume fraction']
            self.d_par = 1.0E-3 if d_par is None else d_par
            self.d_perps = np.linspace(0.1, 1.0, 10) * 1E-3 if d_perps is None else d_perps
            self.d_isos = [2.5E-3] if d_isos is None else d_isos

    def get_params(self):
        return {'id': self.id, 'name': self.name, 'd_par': self.d_par, 'd_perps': self.d_perps,
                'd_isos': self.d_isos, 'type': self.type}

    def set_solver(self, lambda1=0.0, lambda2=1e-3):
        super().set_solver()
        self.solver_params['lambda1'] = lambda1
        self.solver_params['lambda2'] = lambda2
        # NB: the reference assigns lambda2 = 0.25 for Mouse to a dead local (models.pyx:1082-1085):
        # it has no effect there and therefore none here.

    def _atoms(self, scheme_high):
        """models.pyx:1088-1110"""
        from . import synthesis as syn
        zeppelin, ball = syn.Zeppelin(scheme_high), syn.Ball(scheme_high)
        for d in self.d_perps:
            yield zeppelin.get_signal(self.d_par, d), False
        for d in self.d_isos:
            yield ball.get_signal(d), True

    def resample(self, in_path, idx_out, Ylm_out, doMergeB0, ndirs):
        """models.pyx:1113-1144"""
        n_t, n_i = len(self.d_perps), len(self.d_isos)
        lms = self._load_lm(in_path, n_t + n_i)
        _, merge_idx = self._merge(doMergeB0)
        K = {'model': self.id}
        K['D'] = np.ascontiguousarray(self._resample_rotated(lms[:n_t], idx_out, Ylm_out, ndirs)[:, :, merge_idx])
        K['CSF'] = np.ascontiguousarray(self._resample_isotropic(lms[n_t:], idx_out, Ylm_out)[:, merge_idx])
        return K

    @_verified_fit
    def fit(self, evaluation):
        super().fit(evaluation)
        self.configs['save_corrected_DWI'] = evaluation.get_config('doSaveCorrectedDWI')
        ctx = get_context()
        lut = self._lut(evaluation, lambda: _capi.upload_freewater(ctx, evaluation.KERNELS, evaluation.htable))
        dev = getattr(evaluation, '_dev', None)
        if dev is not None:
            est, rmse, nrmse, yc = _capi.freewater_fit_device(
                ctx, lut, dev['y'], self._dev_dirs(evaluation, dev), self.solver_params['lambda1'], self.solver_params['lambda2'],
                self.type == 'Mouse', rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']),
                corrected=bool(self.configs['save_corrected_DWI']))
            return self._finish_device(ctx, dev, {'estimates': est, 'rmse': rmse, 'nrmse': nrmse, 'y_corrected': yc})
        kw = dict(rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']), corrected=bool(self.configs['save_corrected_DWI']))
        mouse = self.type == 'Mouse'
        if len(get_contexts()) > 1:
            n, y, d = evaluation.y.shape[0], evaluation.y, evaluation.DIRs
            est, rmse, nrmse, yc = self._fit_on_devices(
                evaluation, n, lambda c: _capi.upload_freewater(c, evaluation.KERNELS, evaluation.htable),
                lambda c, l, lo, hi, o: _capi.freewater_fit(c, l, y[lo:hi], d[lo:hi], self.solver_params['lambda1'], self.solver_params['lambda2'], mouse, out=o, **kw),
                (np.zeros((n, 4 if mouse else 2)), np.zeros(n) if kw['rmse'] else None, np.zeros(n) if kw['nrmse'] else None,
                 np.zeros((n, y.shape[1])) if kw['corrected'] else None))
        else:
            est, rmse, nrmse, yc = _capi.freewater_fit(ctx, lut, evaluation.y, evaluation.DIRs, self.solver_params['lambda1'],
                                                       self.solver_params['lambda2'], mouse, **kw)
            self._warn_if_capped(ctx)
        results = {'estimates': est}
        if self.configs['compute_rmse']:
            results['rmse'] = rmse
        if self.configs['compute_nrmse']:
            results['nrmse'] = nrmse
        if self.configs['save_corrected_DWI']:
            results['y_corrected'] = yc
        return results


class SANDI(BaseModel):
    """models.pyx:1344-1627"""

    def __init__(self):
        self.id = 'SANDI'
        self.name = 'SANDI'
        self.maps_name = ['fsoma', 'fneurite', 'fextra', 'Rsoma', 'Din', 'De']
        self.maps_descr = ['Intra-soma volume fraction', 'Intra-neurite volume fraction',
                           'Extra-cellular volume fraction', 'Apparent soma radius', 'Neurite axial diffusivity',
                           'Extra-cellular mean diffusivity']
        self.scheme = None
        self.set()
        self.set_solver()

    def set(self, d_is=3.0E-3, Rs=np.linspace(1.0, 12.0, 5) * 1E-6, d_in=np.linspace(0.25, 3.0, 5) * 1E-3,
            d_isos=np.linspace(0.25, 3.0, 5) * 1E-3):
        self.d_is = d_is
        self.Rs = np.array(Rs)
        self.d_in = np.array(d_in)
        self.d_isos = np.array(d_isos)

    def get_params(self):
        return {'id': self.id, 'name': self.name, 'd_is': self.d_is, 'Rs': self.Rs, 'd_in': self.d_in,
                'd_isos': self.d_isos}

    def _lut_extra_key(self):
        return tuple(np.asarray(v, dtype=np.float64).tobytes() for v in (self.Rs, self.d_in, self.d_isos))

    def set_solver(self, lambda1=0.0, lambda2=5.0E-3):
        super().set_solver()
        self.solver_params['lambda1'] = lambda1
        self.solver_params['lambda2'] = lambda2

    def _atoms(self, scheme_high):
        """models.pyx:1407-1444: all three compartments are isotropic (soma = sphere, neurites = astrosticks, extra-cellular = ball)"""
        from . import synthesis as syn
        if self.scheme.version != 1:
            raise RuntimeError('This model requires a "VERSION: STEJSKALTANNER" scheme')
        sphere, sticks, ball = syn.SphereGPD(scheme_high), syn.Astrosticks(scheme_high), syn.Ball(scheme_high)
        for R in self.Rs:
            yield sphere.get_signal(self.d_is, R), True
        for d in self.d_in:
            yield sticks.get_signal(d), True
        for d in self.d_isos:
            yield ball.get_signal(d), True

    def resample(self, in_path, idx_out, Ylm_out, doMergeB0, ndirs):
        """models.pyx:1446-1486: isotropic atoms, each scaled to unit norm"""
        n_atoms = len(self.Rs) + len(self.d_in) + len(self.d_isos)
        lms = self._load_lm(in_path, n_atoms)
        nS, merge_idx = self._merge(doMergeB0)
        sig = self._resample_isotropic(lms, idx_out, Ylm_out)[:, merge_idx]
        K = {'model': self.id, 'signal': np.zeros((nS, n_atoms), dtype=np.float64, order='F'),
             'norms': np.zeros(n_atoms, dtype=np.float64)}
        for a in range(n_atoms):
            K['norms'][a] = 1.0 / np.linalg.norm(sig[a])
            K['signal'][:, a] = sig[a] * K['norms'][a]
        return K

    @_verified_fit
    def fit(self, evaluation):
        super().fit(evaluation)
        ctx = get_context()
        lut = self._lut(evaluation, lambda: _capi.upload_sandi(ctx, evaluation.KERNELS, self.Rs, self.d_in, self.d_isos))
        dev = getattr(evaluation, '_dev', None)
        if dev is not None:
            est, rmse, nrmse = _capi.sandi_fit_device(ctx, lut, dev['y'], self.solver_params['lambda1'],
                                                      self.solver_params['lambda2'], rmse=bool(self.configs['compute_rmse']),
                                                      nrmse=bool(self.configs['compute_nrmse']))
            return self._finish_device(ctx, dev, {'estimates': est, 'rmse': rmse, 'nrmse': nrmse})
        kw = dict(rmse=bool(self.configs['compute_rmse']), nrmse=bool(self.configs['compute_nrmse']))
        if len(get_contexts()) > 1:
            n, y = evaluation.y.shape[0], evaluation.y
            est, rmse, nrmse = self._fit_on_devices(
                evaluation, n, lambda c: _capi.upload_sandi(c, evaluation.KERNELS, self.Rs, self.d_in, self.d_isos),
                lambda c, l, lo, hi, o: _capi.sandi_fit(c, l, y[lo:hi], self.solver_params['lambda1'], self.solver_params['lambda2'], out=o, **kw),
                (np.zeros((n, 6)), np.zeros(n) if kw['rmse'] else None, np.zeros(n) if kw['nrmse'] else None))
        else:
            est, rmse, nrmse = _capi.sandi_fit(ctx, lut, evaluation.y, self.solver_params['lambda1'], self.solver_params['lambda2'], **kw)
            self._warn_if_capped(ctx)
        results = {'estimates': est}
        if self.configs['compute_rmse']:
            results['rmse'] = rmse
        if self.configs['compute_nrmse']:
            results['nrmse'] = nrmse
        return results
