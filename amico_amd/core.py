"""A thin ``Evaluation`` holder with the fields ``model.fit(evaluation)`` reads
(amico/core.py:42-104, 407-498): ``y``, ``DIRs``, ``htable``, ``KERNELS``, ``nthreads``,
``get_config``.  It reproduces the caller contract around the hot path -- mask gather + clip
(core.py:451-452) and the scatter of the results into float32 volumes (core.py:472-498) --
for in-memory (synthetic) volumes; NIfTI I/O, DTI and LUT generation stay out of scope.
"""
import inspect
import time
from os import cpu_count
import numpy as np
from . import models as _models


class Evaluation:
    def __init__(self, study_path='.', subject='.', output_path=None):
        self.niiDWI_img = None
        self.scheme = None
        self.niiMASK_img = None
        self.model = None
        self.KERNELS = None
        self.y = None
        self.DIRs = None
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

    def set_config(self, key, value):
        self.CONFIG[key] = value

    def get_config(self, key):
        return self.CONFIG.get(key)

    # ---- in-memory replacement of load_data (core.py:107-278): volumes are given directly
    def set_data(self, dwi, scheme, mask=None, directions=None):
        """dwi [X,Y,Z,nS] (already b0-normalised), mask [X,Y,Z], directions [X,Y,Z,3]"""
        self.niiDWI_img = np.asarray(dwi, dtype=np.float32)          # core.py:136
        self.scheme = scheme
        self.set_config('dim', self.niiDWI_img.shape[:3])
        self.niiMASK_img = np.ones(self.niiDWI_img.shape[:3], dtype=np.uint8) if mask is None \
            else np.asarray(mask, dtype=np.uint8)
        self._dirs_img = None if directions is None else np.asarray(directions, dtype=np.float64)

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
        self.y = self.niiDWI_img[sel, :].astype(np.double)
        self.y[self.y < 0] = 0
        if self.model.id != 'SANDI':
            if self._dirs_img is None:
                raise RuntimeError('principal directions not set (DTI estimation is outside this path)')
            self.DIRs = np.ascontiguousarray(self._dirs_img[sel, :], dtype=np.float64)
        t = time.time()
        results = self.model.fit(self)
        self.set_config('fit_time', time.time() - t)
        dim = self.get_config('dim')
        self.RESULTS = {}
        self.RESULTS['MAPs'] = np.zeros([dim[0], dim[1], dim[2], len(self.model.maps_name)], dtype=np.float32)
        self.RESULTS['MAPs'][sel, :] = results['estimates']
        if self.DIRs is not None:
            self.RESULTS['DIRs'] = np.zeros([dim[0], dim[1], dim[2], 3], dtype=np.float32)
            self.RESULTS['DIRs'][sel, :] = self.DIRs
        if self.get_config('doComputeRMSE'):
            self.RESULTS['RMSE'] = np.zeros(dim, dtype=np.float32)
            self.RESULTS['RMSE'][sel] = results['rmse']
        if self.get_config('doComputeNRMSE'):
            self.RESULTS['NRMSE'] = np.zeros(dim, dtype=np.float32)
            self.RESULTS['NRMSE'][sel] = results['nrmse']
        if self.model.name == 'NODDI' and self.get_config('doSaveModulatedMaps'):
            self.RESULTS['MAPs_mod'] = np.zeros([dim[0], dim[1], dim[2], 2], dtype=np.float32)
            self.RESULTS['MAPs_mod'][sel, :] = results['estimates_mod']
        if self.model.name == 'Free-Water' and self.get_config('doSaveCorrectedDWI'):
            self.RESULTS['DWI_corrected'] = np.zeros(self.niiDWI_img.shape, dtype=np.float32)
            self.RESULTS['DWI_corrected'][sel, :] = results['y_corrected']
        return results
