"""``Evaluation`` with the fields ``model.fit(evaluation)`` reads (amico/core.py:42-104, 407-498): ``y``,
``DIRs``, ``htable``, ``KERNELS``, ``nthreads``, ``get_config`` -- and the caller contract around the hot path for
in-memory volumes, every per-voxel step on the GPU:

  set_data   ~ load_data's preprocessing inputs (core.py:209-268): raw float32 image, scheme, mask
  fit        b0 normalisation / b0 merge / shell average + mask gather + clip  -> ``y``      (amx_prep_gather)
             principal directions from the log-linear tensor fit                -> ``DIRs``   (amx_dti_directions)
             model.fit(self)                                                    -> maps       (amx_*_fit)
             scatter into float32 volumes (core.py:472-498)                     -> ``RESULTS`` (amx_prep_scatter)

NIfTI / scheme-file I/O, Rician debiasing and LUT generation stay out of scope.
"""
import inspect
import time
from os import cpu_count
import numpy as np
from . import models as _models
from . import _capi
from . import prep as _prep
from . import dti as _dti
from .synthetic import SimpleScheme


class Evaluation:
    def __init__(self, study_path='.', subject='.', output_path=None):
        self.niiDWI_img = None
        self.scheme = None
        self.niiMASK_img = None
        self.model = None
        self.KERNELS = None
        self._y = None
        self._DIRs = None
        self._dev = None
        self.nthreads = None
        self.RESULTS = None
        self.mean_b0s = None
        self.htable = None
        self._dirs_img = None
        self.CONFIG = {}
        self.set_config('study_path', study_path)
        self.set_config('subject', subject)
        self.set_config('OUTPUT_path', output_path)
        # defaults of core.py:82-96
        self.set_config('peaks_filename', None)
        self.set_config('doNormalizeSignal', True)
        self.set_config('doKeepb0Intact', False)
        self.set_config('doComputeRMSE', False)
        self.set_config('doComputeNRMSE', False)
        self.set_config('doSaveModulatedMaps', False)
        self.set_config('doSaveCorrectedDWI', False)
        self.set_config('doMergeB0', False)
        self.set_config('doDebiasSignal', False)
        self.set_config('DWI-SNR', None)
        self.set_config('doDirectionalAverage', False)
        self.set_config('nthreads', -1)
        self.set_config('DTI_fit_method', 'OLS')
        self.set_config('BLAS_nthreads', 1)

    # `y` / `DIRs` (core.py:451-458) are produced on the GPU by fit() and stay there for model.fit (self._dev); the
    # numpy arrays the reference exposes are fetched on first access
    @property
    def y(self):
        if self._y is None and self._dev is not None:
            self._y = self._dev['y'].cpu().numpy().astype(np.float64, copy=False)      # (held as float32 in HBM: core.py:451-452 widen)
        return self._y

    @y.setter
    def y(self, value):
        self._y = value
        self._dev = None

    @property
    def DIRs(self):
        if self._DIRs is None and self._dev is not None and self._dev.get('dirs') is not None:
            self._DIRs = self._dev['dirs'].cpu().numpy()
        return self._DIRs

    @DIRs.setter
    def DIRs(self, value):
        self._DIRs = value
        if self._dev is not None:
            self._dev.pop('dirs', None)       # directions assigned by the caller replace the ones held in HBM

    def set_config(self, key, value):
        self.CONFIG[key] = value

    def get_config(self, key):
        return self.CONFIG.get(key)

    # ---- in-memory replacement of load_data (core.py:107-278): volumes are given directly
    def set_data(self, dwi, scheme, mask=None, directions=None, b0_min_signal=0):
        """dwi [X,Y,Z,nS] raw signal (C or Fortran order), mask [X,Y,Z], directions [X,Y,Z,3] (optional peaks,
        core.py:438-447: when absent they come from the tensor fit).  The options doNormalizeSignal / doMergeB0 /
        doDirectionalAverage are read here, like load_data reads them."""
        img = np.asarray(dwi)
        if img.ndim != 4:
            raise ValueError('DWI file is not a 4D image')                       # core.py:138-139
        self.niiDWI_img = img.astype(np.float32, copy=False)                     # core.py:136
        if any(st % 4 or st <= 0 for st in self.niiDWI_img.strides):
            self.niiDWI_img = np.ascontiguousarray(self.niiDWI_img)
        self.set_config('dim', self.niiDWI_img.shape[:3])
        self.set_config('b0_min_signal', b0_min_signal)
        if scheme.nS != self.niiDWI_img.shape[3]:
            raise ValueError('Scheme does not match with DWI data')
        self.niiMASK_img = np.ones(self.niiDWI_img.shape[:3], dtype=np.uint8) if mask is None \
            else np.asarray(mask, dtype=np.uint8)
        if self.niiMASK_img.shape != self.niiDWI_img.shape[:3]:
            raise ValueError('MASK geometry does not match with DWI data')
        if directions is not None and (np.ndim(directions) != 4 or np.shape(directions)[:3] != self.niiMASK_img.shape
                                       or np.shape(directions)[3] < 3):
            raise ValueError('PEAKS geometry does not match with DWI data')      # core.py:444-445
        # a peaks file holds 3*npeaks values per voxel: the fit uses the first peak
        self._dirs_img = None if directions is None else np.ascontiguousarray(np.asarray(directions, dtype=np.float32)[..., :3])   # core.py:442
        self._raw_scheme = scheme
        self._prep = _prep.SignalPreparation(
            scheme, self.niiDWI_img, self.niiMASK_img, do_normalize=self.get_config('doNormalizeSignal'),
            do_merge_b0=self.get_config('doMergeB0'), do_directional_average=self.get_config('doDirectionalAverage'),
            b0_min_signal=b0_min_signal)
        # the scheme the model sees: one row per shell after the directional average (core.py:254-255)
        self.scheme = SimpleScheme(_prep.directional_average_table(scheme), scheme.b0_thr) \
            if self.get_config('doDirectionalAverage') else scheme

    def set_model(self, model_name):
        if not hasattr(_models, model_name):
            raise ValueError(f'Model "{model_name}" not recognized')
        self.model = getattr(_models, model_name)()
        self.set_solver()

    def set_solver(self, **params):
        if self.model is None:
            raise RuntimeError('Model not set; call "set_model()" method first')
        allowed = list(inspect.signature(self.model.set_solver).parameters)
        params_new = {k: v for k, v in params.items() if k in allowed}     # core.py:314-322
        self.model.set_solver(**params_new)
        self.set_config('solver_params', params_new)

    def set_kernels(self, kernels, htable=None):
        """stands in for generate_kernels()/load_kernels() (core.py:328-404)"""
        self.KERNELS = kernels
        self.htable = None if htable is None else np.ascontiguousarray(htable, dtype=np.int16)
        if self.model is not None:
            self.model.scheme = self.scheme

    def generate_kernels(self, lut_dirs, out_path=None, lmax=12):
        """core.py:328-372 `generate_kernels`: response functions of the model on the high-resolution scheme (500 directions
        per shell) -> SH coefficients rotated to every LUT orientation.  Writes `A_###.npy` under `out_path` when given and
        returns the list of arrays (what `load_kernels` takes as `in_path`).  `lut_dirs` [ndirs, 3]: the LUT orientations."""
        from . import lut as _lut
        if self.model is None:
            raise RuntimeError('Model not set; call "set_model()" method first')
        if self.scheme is None:
            raise RuntimeError('Scheme not loaded; call "set_data()" first')
        t = time.time()
        self.model.scheme = self.scheme
        lut_dirs = np.asarray(lut_dirs, dtype=np.float64)
        aux = _lut.aux_matrices(lmax, lut_dirs)
        idx_in, idx_out = _lut.aux_structures_generate(self.scheme, lmax)
        if out_path is not None:
            import os
            os.makedirs(out_path, exist_ok=True)
        lms = self.model.generate(out_path, aux, idx_in, idx_out, len(lut_dirs))
        self.set_config('generate_kernels_time', time.time() - t)
        return lms

    def load_kernels(self, in_path, lut_dirs, lmax=12):
        """core.py:374-404 `load_kernels`: resample the rotated SH coefficients (folder of A_###.npy files written by
        generate_kernels, or the list of arrays) to this subject's scheme -- one GEMM on the GPU -- and set KERNELS /
        htable.  `lut_dirs` [ndirs, 3]: the LUT orientations (amico/directions/ndirs=*.bin in the reference)."""
        from . import lut as _lut
        from .synthetic import build_htable
        if self.model is None:
            raise RuntimeError('Model not set; call "set_model()" method first')
        if self.scheme is None:
            raise RuntimeError('Scheme not loaded; call "set_data()" first')
        t = time.time()
        self.model.scheme = self.scheme
        idx_out, ylm_out = _lut.aux_structures_resample(self.scheme, lmax)
        self.KERNELS = self.model.resample(in_path, idx_out, ylm_out, self.get_config('doMergeB0'), len(lut_dirs))
        self.htable = build_htable(np.asarray(lut_dirs, dtype=np.float64))
        self.set_config('ndirs', len(lut_dirs))
        self.set_config('lmax', lmax)
        self.set_config('load_kernels_time', time.time() - t)

    def fit(self):
        if self.niiDWI_img is None:
            raise RuntimeError('Data not loaded; call "set_data()" first')
        if self.model is None:
            raise RuntimeError('Model not set; call "set_model()" first')
        if self.KERNELS is None:
            raise RuntimeError('Response functions not set; call "set_kernels()" first')
        if self.KERNELS['model'] != self.model.id:
            raise RuntimeError('Response functions were not created with the same model')
        nt = self.get_config('nthreads')
        self.nthreads = nt if nt > 0 else cpu_count()
        self.model.scheme = self.scheme
        sel = self.niiMASK_img == 1                                   # core.py:451 (== 1, not nonzero)
        import torch                                                  # device buffers only
        dev = torch.device('cuda', torch.cuda.current_device())
        L, ctx, plan = _capi.lib(), self._prep.ctx, self._prep._plan
        t = time.time()
        # ---- raw image -> HBM once; everything up to the map volumes stays there (one stream, default)
        img = self.niiDWI_img
        d_img = torch.from_numpy(np.lib.stride_tricks.as_strided(img, shape=(plan.extent,), strides=(4,))).to(dev)
        n = self._prep.n_vox
        # the prepared signals stay float32 in HBM (every value of core.py:209-268 is a float32; core.py:451-452 only widen them):
        # the tensor fit and the model fit read them in place, half the bytes of the float64 rows
        d_y = torch.empty((n, self._prep.n_out), dtype=torch.float32, device=dev)
        d_mb0 = torch.empty(n, dtype=torch.float32, device=dev)
        thr = 0.0
        if self._prep.do_normalize and self._prep.b0_min_signal != 0.0:              # core.py:217
            d_vol = torch.empty(img.shape[:3], dtype=torch.float32, device=dev)
            ctx.check(L.amx_prep_mean_b0_device(ctx._h, plan._h, d_img.data_ptr(), d_vol.data_ptr(), None))
            mean_b0s = d_vol.cpu().numpy()
            thr = float(self._prep.b0_min_signal * mean_b0s[mean_b0s > 0].mean())
        ctx.check(L.amx_prep_gather_device_f32(ctx._h, plan._h, d_img.data_ptr(), int(self._prep.do_normalize), thr,
                                           d_y.data_ptr(), d_mb0.data_ptr(), None))  # core.py:209-268 + 451-452
        # precompute directions (core.py:428-458)
        d_dirs = None
        if self.get_config('doDirectionalAverage'):
            pass
        elif self._dirs_img is not None:
            d_dirs = torch.from_numpy(np.ascontiguousarray(self._dirs_img[sel, :], dtype=np.float64)).to(dev)
        else:
            if self.get_config('DTI_fit_method') not in ('OLS', 'LS'):
                raise NotImplementedError('only the default DTI_fit_method (OLS) runs on the GPU')
            est = _dti.TensorDirections.from_scheme(self._raw_scheme, do_merge_b0=self.get_config('doMergeB0'), ctx=ctx)
            d_dirs = torch.empty((n, 3), dtype=torch.float64, device=dev)
            est.fit_device(d_y.data_ptr(), n, d_dirs.data_ptr(), f32=True)
        ctx.sync()
        del d_img
        self._y, self._DIRs = None, None
        self._dev = {'y': d_y, 'dirs': d_dirs}
        self.mean_b0s = d_mb0.cpu().numpy() if self._prep.do_normalize else None
        self.set_config('dirs_precomputing_time', time.time() - t)
        t = time.time()
        # models.pyx:28-43, 981 feed a ProgressBar while the chunks are fitted; here the whole fit is enqueued at once and
        # the library reports from the stream (amx_set_progress: after each NODDI stage / at the end of the other models)
        cb = self.get_config('progress_callback')
        if callable(cb):
            ctx.set_progress(cb)
        try:
            results = self.model.fit(self)                            # reads self._dev in place
        finally:
            if callable(cb):
                ctx.sync()
                ctx.set_progress(None)
        self.set_config('fit_time', time.time() - t)
        out = self._dev.get('out', {})

        def sc(key, host_values):
            """per-voxel values -> float32 volume (core.py:472-498); from the device copy when the fit left one"""
            t_ = out.get(key)
            if t_ is None:
                return self._prep.scatter(host_values)
            k = 1 if t_.dim() == 1 else t_.shape[1]
            vol = torch.empty(img.shape[:3] + (k,), dtype=torch.float32, device=dev)
            ctx.check(L.amx_prep_scatter_device(ctx._h, plan._h, t_.data_ptr(), k, vol.data_ptr(), None))
            ctx.sync()
            v = vol.cpu().numpy()
            return v[..., 0] if t_.dim() == 1 else v

        self.RESULTS = {}
        self.RESULTS['MAPs'] = sc('estimates', results['estimates'])
        if d_dirs is not None:
            out['DIRs'] = d_dirs
            self.RESULTS['DIRs'] = sc('DIRs', None)
        if self.get_config('doComputeRMSE'):
            self.RESULTS['RMSE'] = sc('rmse', results['rmse'])
        if self.get_config('doComputeNRMSE'):
            self.RESULTS['NRMSE'] = sc('nrmse', results['nrmse'])
        if self.model.name == 'NODDI' and self.get_config('doSaveModulatedMaps'):
            self.RESULTS['MAPs_mod'] = sc('estimates_mod', results['estimates_mod'])
        if self.model.name == 'Free-Water' and self.get_config('doSaveCorrectedDWI'):
            y_corrected = results['y_corrected']                      # core.py:488-498
            b0_idx = self.scheme.b0_idx
            if self.get_config('doNormalizeSignal') and self.scheme.b0_count > 0:
                y_corrected = y_corrected * np.reshape(self.mean_b0s, (-1, 1))
            if self.get_config('doKeepb0Intact') and self.scheme.b0_count > 0:
                y_corrected[:, b0_idx] = self.y[:, b0_idx] * np.reshape(self.mean_b0s, (-1, 1))
            self.RESULTS['DWI_corrected'] = self._prep.scatter(y_corrected)
        self._dev.pop('out', None)
        return results
