"""ctypes binding of the C ABI in include/amico_amd.h (amico_amd/csrc/libamico_amd.so).

There is NO CPU fallback: if the HIP library is missing or no gfx950 GPU is visible every
entry point raises.  Device memory for the ``*_device`` calls is plain pointers (e.g.
``torch.Tensor.data_ptr()``) -- torch is plumbing, never part of the signatures.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AMICO_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libamico_amd.so')   # override: A/B builds

AMX_OK, AMX_E_BADARG, AMX_E_HIP, AMX_E_DIR_OOB, AMX_E_OVERFLOW, AMX_E_NODEVICE = 0, -1, -2, -3, -4, -5
F_RMSE, F_NRMSE, F_MODULATED, F_CORRECTED, F_DEBUG_X = 1, 2, 4, 8, 16

# every symbol include/amico_amd.h declares (tests check that the library exports them all)
SYMBOLS = ['amx_version', 'amx_build_id', 'amx_device_count', 'amx_set_call_voxels', 'amx_ctx_create', 'amx_ctx_destroy', 'amx_last_error',
           'amx_lut_upload_noddi', 'amx_lut_upload_freewater', 'amx_lut_upload_sandi', 'amx_lut_destroy',
           'amx_dir_to_lut_idx', 'amx_noddi_fit', 'amx_freewater_fit', 'amx_sandi_fit',
           'amx_noddi_fit_device', 'amx_freewater_fit_device', 'amx_sandi_fit_device', 'amx_sync_status',
           'amx_noddi_fit_device_f32', 'amx_freewater_fit_device_f32', 'amx_sandi_fit_device_f32', 'amx_czb_fit_device_f32',
           'amx_set_debug_x', 'amx_debug_fetch', 'amx_lut_upload_czb', 'amx_czb_fit', 'amx_czb_fit_f32', 'amx_czb_fit_device', 'amx_noddi_fit_f32', 'amx_freewater_fit_f32', 'amx_sandi_fit_f32', 'amx_set_progress',
           'amx_set_profiling', 'amx_last_kernel_ms', 'amx_last_stats', 'amx_last_seed_stats', 'amx_last_host_narrowed', 'amx_last_path', 'amx_host_pool_info', 'amx_selftest',
           'amx_dti_create', 'amx_dti_destroy', 'amx_dti_directions', 'amx_dti_directions_device', 'amx_dti_directions_device_f32', 'amx_prep_gather_device_f32',
           'amx_prep_create', 'amx_prep_destroy', 'amx_prep_gather', 'amx_prep_gather_device',
           'amx_prep_gather_directions_device', 'amx_prep_gather_directions_device_f32',
           'amx_prep_mean_b0', 'amx_prep_mean_b0_device', 'amx_prep_scatter', 'amx_prep_scatter_device',
           'amx_lut_resample', 'amx_lut_rotate_resample',
           'amx_dict_upload', 'amx_dict_destroy', 'amx_nnls_batched', 'amx_lasso_batched', 'amx_nnls_batched_device', 'amx_lasso_batched_device']

_lib = None
c_vp, c_dp, c_fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
c_i16p, c_i32p, c_i64p = C.POINTER(C.c_int16), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_int64, C.c_int64, c_vp)


def source_id():
    """sha256[:16] of the library's sources as they are in the tree now (same recipe as amico_amd/csrc/Makefile)"""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, 'csrc')
    files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hpp')), key=os.path.basename)
    files.append(os.path.join(_HERE, '..', 'include', 'amico_amd.h'))
    h = hashlib.sha256()
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_id():
    """what the loaded library says it was built from: 'amico_amd <version> csrc <hash>'"""
    return lib().amx_build_id().decode()


def build_is_current():
    return build_id().split()[-1] == source_id()


class AmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def lib():
    """Load libamico_amd.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'amico_amd: HIP library {LIB_PATH} not found -- run `python -c "import '
                           f'__graft_entry__ as g; g.build()"` (or `make -C amico_amd/csrc -j`). '
                           f'There is no CPU fallback.')
    # torch ships its own copy of the ROCm runtime under the same SONAMEs as /opt/rocm's, and the copy that is loaded
    # first serves the whole process; torch only finds its GPUs on its own copy.  Load torch's first when torch is
    # installed, so that device buffers / streams / torch.distributed keep working next to this library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.amx_version.restype = C.c_int
    L.amx_build_id.restype = C.c_char_p
    L.amx_ctx_create.argtypes = [C.c_int, C.POINTER(c_vp)]
    L.amx_ctx_destroy.argtypes = [c_vp]
    L.amx_ctx_destroy.restype = None
    L.amx_last_error.argtypes = [c_vp]
    L.amx_last_error.restype = C.c_char_p
    L.amx_lut_upload_noddi.argtypes = [c_vp, c_fp, c_fp, c_dp, c_fp, c_fp, c_i16p, c_i64p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.POINTER(c_vp)]
    L.amx_lut_upload_freewater.argtypes = [c_vp, c_fp, c_fp, c_i16p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(c_vp)]
    L.amx_lut_upload_sandi.argtypes = [c_vp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(c_vp)]
    L.amx_lut_destroy.argtypes = [c_vp]
    L.amx_lut_destroy.restype = None
    L.amx_dir_to_lut_idx.argtypes = [c_vp, c_vp, c_dp, C.c_int64, c_i32p]
    L.amx_noddi_fit.argtypes = [c_vp, c_vp, c_dp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_uint,
                                c_dp, c_dp, c_dp, c_dp]
    L.amx_freewater_fit.argtypes = [c_vp, c_vp, c_dp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_int,
                                    C.c_uint, c_dp, c_dp, c_dp, c_dp]
    L.amx_sandi_fit.argtypes = [c_vp, c_vp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp]
    L.amx_noddi_fit_device.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, C.c_uint,
                                       c_vp, c_vp, c_vp, c_vp, c_vp]
    L.amx_freewater_fit_device.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, C.c_int,
                                           C.c_uint, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.amx_sandi_fit_device.argtypes = [c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, C.c_uint,
                                       c_vp, c_vp, c_vp, c_vp]
    L.amx_sync_status.argtypes = [c_vp, c_vp]
    L.amx_set_debug_x.argtypes = [c_vp, c_vp]
    L.amx_debug_fetch.argtypes = [c_vp, c_vp, C.c_int, c_vp, C.c_size_t]
    L.amx_noddi_fit_f32.argtypes = [c_vp, c_vp, c_fp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp, c_dp]
    L.amx_freewater_fit_f32.argtypes = [c_vp, c_vp, c_fp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_uint,
                                        c_dp, c_dp, c_dp, c_dp]
    L.amx_sandi_fit_f32.argtypes = [c_vp, c_vp, c_fp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp]
    L.amx_set_progress.argtypes = [c_vp, PROGRESS_CB, c_vp]
    L.amx_lut_upload_czb.argtypes = [c_vp, c_fp, c_fp, c_fp, c_dp, c_i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(c_vp)]
    L.amx_czb_fit.argtypes = [c_vp, c_vp, c_dp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp]
    L.amx_czb_fit_f32.argtypes = [c_vp, c_vp, c_fp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp]
    L.amx_czb_fit_device.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, C.c_uint, c_vp, c_vp, c_vp, c_vp]
    for name in ('noddi', 'freewater', 'sandi', 'czb'):
        getattr(L, 'amx_%s_fit_device_f32' % name).argtypes = getattr(L, 'amx_%s_fit_device' % name).argtypes
        getattr(L, 'amx_%s_fit_device_f32' % name).restype = C.c_int
    L.amx_set_profiling.argtypes = [c_vp, C.c_int]
    L.amx_last_kernel_ms.argtypes = [c_vp, C.c_int, C.POINTER(C.c_float)]
    L.amx_last_stats.argtypes = [c_vp, c_i64p]
    L.amx_last_seed_stats.argtypes = [c_vp, c_i64p]
    L.amx_last_host_narrowed.argtypes = [c_vp]
    L.amx_last_path.argtypes = [c_vp, C.c_char_p, C.c_int]
    L.amx_host_pool_info.argtypes = [c_vp, C.POINTER(C.c_int)]
    L.amx_set_call_voxels.argtypes = [c_vp, C.c_int64]
    L.amx_dict_upload.argtypes = [c_vp, c_dp, C.c_int, C.c_int, C.c_int, C.POINTER(c_vp)]
    L.amx_dict_destroy.argtypes = [c_vp]
    L.amx_dict_destroy.restype = None
    L.amx_nnls_batched.argtypes = [c_vp, c_vp, c_i32p, c_dp, C.c_int64, c_dp, c_dp]
    L.amx_lasso_batched.argtypes = [c_vp, c_vp, c_i32p, c_dp, C.c_int64, C.c_double, C.c_double, c_dp]
    L.amx_nnls_batched_device.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp]
    L.amx_lasso_batched_device.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, c_vp, c_vp]
    L.amx_selftest.argtypes = [c_vp, c_dp]
    L.amx_dti_create.argtypes = [c_vp, c_dp, C.c_int, C.c_double, C.POINTER(c_vp)]
    L.amx_dti_destroy.argtypes = [c_vp]
    L.amx_dti_destroy.restype = None
    L.amx_dti_directions.argtypes = [c_vp, c_vp, c_dp, C.c_int64, c_dp]
    L.amx_dti_directions_device.argtypes = [c_vp, c_vp, c_vp, C.c_int64, c_vp, c_vp]
    L.amx_dti_directions_device_f32.argtypes = [c_vp, c_vp, c_vp, C.c_int64, c_vp, c_vp]
    L.amx_dti_directions_device_f32.restype = C.c_int
    L.amx_prep_create.argtypes = [c_vp, c_i64p, c_i64p, C.c_int, c_i32p, C.c_int64, c_i32p, c_i32p, C.c_int,
                                  c_i32p, C.c_int, C.c_int, C.POINTER(c_vp)]
    L.amx_prep_destroy.argtypes = [c_vp]
    L.amx_prep_destroy.restype = None
    L.amx_prep_gather.argtypes = [c_vp, c_vp, c_fp, C.c_int, C.c_float, c_dp, c_fp]
    L.amx_prep_gather_device.argtypes = [c_vp, c_vp, c_vp, C.c_int, C.c_float, c_vp, c_vp, c_vp]
    L.amx_prep_gather_device_f32.argtypes = [c_vp, c_vp, c_vp, C.c_int, C.c_float, c_vp, c_vp, c_vp]
    L.amx_prep_gather_device_f32.restype = C.c_int
    for f_ in (L.amx_prep_gather_directions_device, L.amx_prep_gather_directions_device_f32):
        f_.argtypes = [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_float, c_vp, c_vp, c_vp, c_vp]      # ctx, plan, tensor helper, img, normalize, thr, y, mean_b0, dirs, stream
        f_.restype = C.c_int
    L.amx_prep_mean_b0.argtypes = [c_vp, c_vp, c_fp, c_fp]
    L.amx_prep_mean_b0_device.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp]
    L.amx_prep_scatter.argtypes = [c_vp, c_vp, c_dp, C.c_int, c_fp]
    L.amx_prep_scatter_device.argtypes = [c_vp, c_vp, c_vp, C.c_int, c_vp, c_vp]
    L.amx_lut_resample.argtypes = [c_vp, c_fp, C.c_int64, C.c_int, c_fp, c_i32p, C.c_int, C.c_int, c_fp]
    L.amx_lut_rotate_resample.argtypes = [c_vp, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_i32p, C.c_int, C.c_int, c_fp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ('amx_version', 'amx_build_id'):
            fn.restype = C.c_int
    _lib = L
    return L


def _p(a, ct):
    return a.ctypes.data_as(ct) if a is not None else None


def device_count():
    """gfx950 devices this process can see (amx_device_count; 0 without a usable GPU)"""
    return max(0, int(lib().amx_device_count()))


class Context:
    """amx_ctx: one per process and GPU."""

    def __init__(self, device=-1):
        self._h = c_vp()
        rc = lib().amx_ctx_create(int(device), C.byref(self._h))
        if rc != AMX_OK:
            self._h = None
            raise AmxError(rc, 'amico_amd: no usable MI355X (gfx950) device -- the fit path has no CPU fallback'
                           if rc == AMX_E_NODEVICE else f'amx_ctx_create failed ({rc})')

    def close(self):
        if getattr(self, '_h', None):
            lib().amx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == AMX_OK:
            return
        msg = lib().amx_last_error(self._h).decode('utf-8', 'replace')
        if rc == AMX_E_BADARG:
            raise ValueError(msg or 'bad argument')
        raise AmxError(rc, msg or f'amico_amd error {rc}')       # RuntimeError, like lut.pyx:352-354

    def set_call_voxels(self, total):
        """the host-buffer fits that follow are shards of a call of `total` voxels (amx_set_call_voxels; 0 = off)"""
        self.check(lib().amx_set_call_voxels(self._h, int(total)))

    def sync(self, stream=None):
        self.check(lib().amx_sync_status(self._h, c_vp(stream or 0)))

    def set_progress(self, fn=None):
        """fn(done, total) is called while a host-buffer fit runs (batches of voxels complete); None unregisters"""
        self._progress_cb = PROGRESS_CB(lambda done, total, _u: fn(int(done), int(total))) if fn else PROGRESS_CB()
        self.check(lib().amx_set_progress(self._h, self._progress_cb, None))

    def set_profiling(self, on=True, only=None):
        """HIP events around the kernel groups of every fit (amx_last_kernel_ms); only=w: the pair of group w alone (each event is a packet
        of the stream: the full set costs a NODDI fit ~70 us)"""
        self.check(lib().amx_set_profiling(self._h, (2 + int(only)) if (on and only is not None) else int(bool(on))))

    def last_kernel_ms(self, which=0):
        ms = C.c_float()
        self.check(lib().amx_last_kernel_ms(self._h, int(which), C.byref(ms)))
        return ms.value

    def last_stats(self):
        out = (C.c_int64 * 4)()
        self.check(lib().amx_last_stats(self._h, out))
        return {'rerun_voxels': out[0], 'itercap_voxels': out[1], 'overflow_voxels': out[2],
                'guard_trips': out[3] >> 32, 'guard_last': out[3] & 0xffffffff}

    def last_host_narrowed(self):
        """batches of the last host-buffer call whose float64 signals crossed PCIe as float32, losslessly (amx_last_host_narrowed)"""
        return int(lib().amx_last_host_narrowed(self._h))

    def host_pool_info(self):
        """host threads of the float32 transport of this context (amx_host_pool_info)"""
        out = (C.c_int * 4)()
        self.check(lib().amx_host_pool_info(self._h, out))
        return {'threads': out[0], 'first_cpu': out[1], 'physical_cores': out[2], 'device': out[3]}

    def last_path(self):
        """the kernels the last fit on this context enqueued, in launch order (amx_last_path)"""
        buf = C.create_string_buffer(2048)
        self.check(lib().amx_last_path(self._h, buf, 2048))
        return buf.value.decode()

    def last_seed_stats(self):
        """how the NODDI voxels since the previous sync were settled (amx_last_seed_stats): certification rates of the three stages"""
        out = (C.c_int64 * 8)()
        self.check(lib().amx_last_seed_stats(self._h, out))
        n = int(out[0])
        d = {'seeded_voxels': n, 'leftover_stage1': int(out[1]), 'leftover_lasso': int(out[2]), 'leftover_stage3': int(out[3]),
             'clipped_stage2': int(out[4])}
        if n > 0:
            d['certified'] = [1.0 - out[k] / n for k in (1, 2, 3)]
        return d

    def selftest(self):
        out = np.zeros((12, 64))
        self.check(lib().amx_selftest(self._h, _p(out, c_dp)))
        return out


class Lut:
    """amx_lut: device-resident dictionary."""

    def __init__(self, ctx, handle, model, nS, n_atoms, n_maps=None):
        self.ctx, self._h, self.model, self.nS, self.n_atoms, self.n_maps = ctx, handle, model, nS, n_atoms, n_maps

    def close(self):
        if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
            lib().amx_lut_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_noddi(ctx, kernels, htable, dwi_idx, is_exvivo=False):
    wm = np.ascontiguousarray(kernels['wm'], dtype=np.float32)
    if wm.ndim != 3:
        raise ValueError("KERNELS['wm'] must be [n_wm, ndirs, nS]")
    n_wm, ndirs, nS = wm.shape
    iso = np.ascontiguousarray(kernels['iso'], dtype=np.float32)
    norms = np.ascontiguousarray(kernels['norms'], dtype=np.float64)
    icvf = np.ascontiguousarray(kernels['icvf'], dtype=np.float32)
    kappa = np.ascontiguousarray(kernels['kappa'], dtype=np.float32)
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    dwi = np.ascontiguousarray(dwi_idx, dtype=np.int64)
    if iso.shape != (nS,) or icvf.shape != (n_wm,) or kappa.shape != (n_wm,) or ht.size != 181 * 181 \
            or norms.shape != (len(dwi), n_wm):
        raise ValueError('NODDI KERNELS / htable have inconsistent shapes')
    h = c_vp()
    ctx.check(lib().amx_lut_upload_noddi(ctx._h, _p(wm, c_fp), _p(iso, c_fp), _p(norms, c_dp), _p(icvf, c_fp),
                                         _p(kappa, c_fp), _p(ht, c_i16p), _p(dwi, c_i64p), n_wm, ndirs, nS,
                                         len(dwi), int(bool(is_exvivo)), C.byref(h)))
    return Lut(ctx, h, 'NODDI', nS, n_wm + 1 + (1 if is_exvivo else 0), 3 + (1 if is_exvivo else 0))


def upload_freewater(ctx, kernels, htable):
    D = np.ascontiguousarray(kernels['D'], dtype=np.float32)
    CSF = np.ascontiguousarray(kernels['CSF'], dtype=np.float32)
    if D.ndim != 3 or CSF.ndim != 2 or CSF.shape[1] != D.shape[2]:
        raise ValueError('FreeWater KERNELS have inconsistent shapes')
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    if ht.size != 181 * 181:
        raise ValueError('htable must have 181*181 entries')
    h = c_vp()
    ctx.check(lib().amx_lut_upload_freewater(ctx._h, _p(D, c_fp), _p(CSF, c_fp), _p(ht, c_i16p), D.shape[0],
                                             CSF.shape[0], D.shape[1], D.shape[2], C.byref(h)))
    return Lut(ctx, h, 'FreeWater', D.shape[2], D.shape[0] + CSF.shape[0])


def upload_sandi(ctx, kernels, Rs, d_in, d_isos):
    sig = np.asfortranarray(kernels['signal'], dtype=np.float64)
    norms = np.ascontiguousarray(kernels['norms'], dtype=np.float64)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64)
    d_in = np.ascontiguousarray(d_in, dtype=np.float64)
    d_isos = np.ascontiguousarray(d_isos, dtype=np.float64)
    nS, n_atoms = sig.shape
    if n_atoms != len(Rs) + len(d_in) + len(d_isos) or norms.shape != (n_atoms,):
        raise ValueError('SANDI KERNELS have inconsistent shapes')
    h = c_vp()
    ctx.check(lib().amx_lut_upload_sandi(ctx._h, _p(sig, c_dp), _p(norms, c_dp), _p(Rs, c_dp), _p(d_in, c_dp),
                                         _p(d_isos, c_dp), nS, len(Rs), len(d_in), len(d_isos), C.byref(h)))
    return Lut(ctx, h, 'SANDI', nS, n_atoms, 6)


def upload_czb(ctx, kernels, Rs, htable):
    wmr = np.ascontiguousarray(kernels['wmr'], dtype=np.float32)
    wmh = np.ascontiguousarray(kernels['wmh'], dtype=np.float32)
    iso = np.ascontiguousarray(kernels['iso'], dtype=np.float32)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64)
    if wmr.ndim != 3 or wmh.ndim != 3 or iso.ndim != 2 or wmh.shape[1:] != wmr.shape[1:] or iso.shape[1] != wmr.shape[2] \
            or Rs.shape != (wmr.shape[0],):
        raise ValueError('CylinderZeppelinBall KERNELS / Rs have inconsistent shapes')
    ht = np.ascontiguousarray(htable, dtype=np.int16)
    if ht.size != 181 * 181:
        raise ValueError('htable must have 181*181 entries')
    h = c_vp()
    ctx.check(lib().amx_lut_upload_czb(ctx._h, _p(wmr, c_fp), _p(wmh, c_fp), _p(iso, c_fp), _p(Rs, c_dp), _p(ht, c_i16p),
                                       wmr.shape[0], wmh.shape[0], iso.shape[0], wmr.shape[1], wmr.shape[2], C.byref(h)))
    return Lut(ctx, h, 'CylinderZeppelinBall', wmr.shape[2], wmr.shape[0] + wmh.shape[0] + iso.shape[0], 3)


def _check_y(y, nS):
    """float32 signals stay float32 (amx_*_fit_f32: the image dtype of the reference, half the PCIe bytes, same results);
    anything else is passed as float64 like evaluation.y"""
    y = np.ascontiguousarray(y, dtype=np.float32 if getattr(y, 'dtype', None) == np.float32 else np.float64)
    if y.ndim != 2 or y.shape[1] != nS:
        raise ValueError(f'y must be [n_vox, {nS}] float64 (or float32)')
    return y


def _yp(y):
    return (True, _p(y, c_fp)) if y.dtype == np.float32 else (False, _p(y, c_dp))


def _check_dirs(dirs, n):
    dirs = np.ascontiguousarray(dirs, dtype=np.float64)      # a copy is never modified (lut.pyx:335-338 quirk)
    if dirs.shape != (n, 3):
        raise ValueError('DIRs must be [n_vox, 3]')
    return dirs


def _outs(out, n, specs):
    """result arrays of a host fit: fresh zeros, or the caller's (`out`: a tuple like the fit's return value -- C-contiguous float64
    arrays of the right shape, e.g. row slices of the arrays a multi-device fit hands to its shards; None where a result is off)"""
    res = []
    for k, (shape, on) in enumerate(specs):
        if not on:
            res.append(None)
            continue
        a = None if out is None else out[k]
        if a is None:
            a = np.zeros((n,) + shape, dtype=np.float64, order='C')
        elif a.dtype != np.float64 or a.shape != (n,) + shape or not a.flags['C_CONTIGUOUS'] or not a.flags['WRITEABLE']:
            raise ValueError('out[%d] must be a writable C-contiguous float64 array of shape %s' % (k, ((n,) + shape,)))
        res.append(a)
    return res


def noddi_fit(ctx, lut, y, dirs, lambda1, lambda2, n_maps, rmse=False, nrmse=False, mod=False, out=None):
    y = _check_y(y, lut.nS)
    n = y.shape[0]
    dirs = _check_dirs(dirs, n)
    if lut.n_maps is not None and n_maps != lut.n_maps:      # (the library writes what the DICTIONARY says: a short buffer would be overrun)
        raise ValueError(f'the dictionary writes {lut.n_maps} maps per voxel, the model expects {n_maps} (isExvivo changed?)')
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | (F_MODULATED if mod else 0)
    est, r, nr, md = _outs(out, n, [((n_maps,), True), ((), rmse), ((), nrmse), ((2,), mod)])
    f32, yp = _yp(y)
    fn = lib().amx_noddi_fit_f32 if f32 else lib().amx_noddi_fit
    ctx.check(fn(ctx._h, lut._h, yp, _p(dirs, c_dp), n, float(lambda1), float(lambda2),
                 flags, _p(est, c_dp), _p(r, c_dp), _p(nr, c_dp), _p(md, c_dp)))
    return est, r, nr, md


def freewater_fit(ctx, lut, y, dirs, lambda1, lambda2, is_mouse, rmse=False, nrmse=False, corrected=False, out=None):
    y = _check_y(y, lut.nS)
    n = y.shape[0]
    dirs = _check_dirs(dirs, n)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | (F_CORRECTED if corrected else 0)
    est, r, nr, yc = _outs(out, n, [((4 if is_mouse else 2,), True), ((), rmse), ((), nrmse), ((lut.nS,), corrected)])
    f32, yp = _yp(y)
    fn = lib().amx_freewater_fit_f32 if f32 else lib().amx_freewater_fit
    ctx.check(fn(ctx._h, lut._h, yp, _p(dirs, c_dp), n, float(lambda1), float(lambda2), int(bool(is_mouse)), flags,
                 _p(est, c_dp), _p(r, c_dp), _p(nr, c_dp), _p(yc, c_dp)))
    return est, r, nr, yc


def sandi_fit(ctx, lut, y, lambda1, lambda2, rmse=False, nrmse=False, out=None):
    y = _check_y(y, lut.nS)
    n = y.shape[0]
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0)
    est, r, nr = _outs(out, n, [((6,), True), ((), rmse), ((), nrmse)])
    f32, yp = _yp(y)
    fn = lib().amx_sandi_fit_f32 if f32 else lib().amx_sandi_fit
    ctx.check(fn(ctx._h, lut._h, yp, n, float(lambda1), float(lambda2), flags, _p(est, c_dp), _p(r, c_dp), _p(nr, c_dp)))
    return est, r, nr


def czb_fit(ctx, lut, y, dirs, lambda1, lambda2, rmse=False, nrmse=False, out=None):
    y = _check_y(y, lut.nS)
    n = y.shape[0]
    dirs = _check_dirs(dirs, n)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0)
    est, r, nr = _outs(out, n, [((3,), True), ((), rmse), ((), nrmse)])
    f32, yp = _yp(y)
    fn = lib().amx_czb_fit_f32 if f32 else lib().amx_czb_fit
    ctx.check(fn(ctx._h, lut._h, yp, _p(dirs, c_dp), n, float(lambda1), float(lambda2), flags, _p(est, c_dp), _p(r, c_dp),
                 _p(nr, c_dp)))
    return est, r, nr


# ---- the same three fits on DEVICE-resident inputs (torch tensors used as plain device buffers); outputs are torch
#      tensors on the same device, enqueued on `stream`; the caller synchronises with ctx.sync(stream)
def debug_fetch(ctx, lut, which, shape, dtype):
    """workspace / dictionary tables of the support seeds (include/amico_amd.h: amx_debug_fetch) as a numpy array"""
    out = np.empty(shape, dtype=dtype)
    ctx.check(lib().amx_debug_fetch(ctx._h, lut._h if lut is not None else None, int(which), out.ctypes.data_as(c_vp), out.nbytes))
    return out


def _dptr(t):
    return c_vp(t.data_ptr()) if t is not None else None


def _dev_fn(model, y_t):
    """amx_<model>_fit_device for float64 tensors, amx_<model>_fit_device_f32 for float32 ones"""
    import torch
    return getattr(lib(), 'amx_%s_fit_device%s' % (model, '_f32' if y_t.dtype == torch.float32 else ''))


def _check_dev(lut, y_t, dirs_t=None):
    """the kernels index `y` with the dictionary's nS as the row stride and `DIRs` with stride 3: anything else
    would read foreign memory and return garbage maps without an error"""
    import torch
    if y_t.dtype not in (torch.float64, torch.float32) or y_t.dim() != 2 or y_t.shape[1] != lut.nS or not y_t.is_contiguous():
        raise ValueError(f'y must be a contiguous float64 (or float32) device tensor [n_vox, {lut.nS}] (the dictionary was built for '
                         f'{lut.nS} volumes per voxel)')
    if dirs_t is not None and (dirs_t.dtype != torch.float64 or tuple(dirs_t.shape) != (y_t.shape[0], 3)
                               or not dirs_t.is_contiguous() or dirs_t.device != y_t.device):
        raise ValueError('DIRs must be a contiguous float64 device tensor [n_vox, 3] on the device of y')


def _debug_x(ctx, shape, like, want):
    """registers a zeroed coefficient buffer for AMX_F_DEBUG_X (include/amico_amd.h) and returns (tensor, flag)"""
    import torch
    if not want:
        return None, 0
    x = torch.zeros(shape, dtype=torch.float64, device=like.device)
    ctx.check(lib().amx_set_debug_x(ctx._h, _dptr(x)))
    return x, F_DEBUG_X


def noddi_fit_device(ctx, lut, y_t, dirs_t, lambda1, lambda2, n_maps, rmse=False, nrmse=False, mod=False, stream=None,
                     return_x=False):
    import torch
    _check_dev(lut, y_t, dirs_t)
    if lut.n_maps is not None and n_maps != lut.n_maps:
        raise ValueError(f'the dictionary writes {lut.n_maps} maps per voxel, the model expects {n_maps} (isExvivo changed?)')
    n, f64 = y_t.shape[0], dict(dtype=torch.float64, device=y_t.device)
    xd, fx = _debug_x(ctx, (n, 3, lut.n_atoms), y_t, return_x)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | (F_MODULATED if mod else 0) | fx
    est = torch.empty((n, n_maps), **f64)
    r = torch.empty(n, **f64) if rmse else None
    nr = torch.empty(n, **f64) if nrmse else None
    md = torch.empty((n, 2), **f64) if mod else None
    ctx.check(_dev_fn('noddi', y_t)(ctx._h, lut._h, _dptr(y_t), _dptr(dirs_t), n, float(lambda1), float(lambda2), flags,
                                         _dptr(est), _dptr(r), _dptr(nr), _dptr(md), c_vp(stream or 0)))
    return (est, r, nr, md, xd) if return_x else (est, r, nr, md)


def freewater_fit_device(ctx, lut, y_t, dirs_t, lambda1, lambda2, is_mouse, rmse=False, nrmse=False, corrected=False,
                         stream=None, return_x=False):
    import torch
    _check_dev(lut, y_t, dirs_t)
    n, f64 = y_t.shape[0], dict(dtype=torch.float64, device=y_t.device)
    xd, fx = _debug_x(ctx, (n, lut.n_atoms), y_t, return_x)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | (F_CORRECTED if corrected else 0) | fx
    est = torch.empty((n, 4 if is_mouse else 2), **f64)
    r = torch.empty(n, **f64) if rmse else None
    nr = torch.empty(n, **f64) if nrmse else None
    yc = torch.empty((n, lut.nS), **f64) if corrected else None
    ctx.check(_dev_fn('freewater', y_t)(ctx._h, lut._h, _dptr(y_t), _dptr(dirs_t), n, float(lambda1), float(lambda2),
                                             int(bool(is_mouse)), flags, _dptr(est), _dptr(r), _dptr(nr), _dptr(yc),
                                             c_vp(stream or 0)))
    return (est, r, nr, yc, xd) if return_x else (est, r, nr, yc)


def czb_fit_device(ctx, lut, y_t, dirs_t, lambda1, lambda2, rmse=False, nrmse=False, stream=None, return_x=False):
    import torch
    _check_dev(lut, y_t, dirs_t)
    n, f64 = y_t.shape[0], dict(dtype=torch.float64, device=y_t.device)
    xd, fx = _debug_x(ctx, (n, lut.n_atoms), y_t, return_x)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | fx
    est = torch.empty((n, 3), **f64)
    r = torch.empty(n, **f64) if rmse else None
    nr = torch.empty(n, **f64) if nrmse else None
    ctx.check(_dev_fn('czb', y_t)(ctx._h, lut._h, _dptr(y_t), _dptr(dirs_t), n, float(lambda1), float(lambda2), flags,
                                       _dptr(est), _dptr(r), _dptr(nr), c_vp(stream or 0)))
    return (est, r, nr, xd) if return_x else (est, r, nr)


def sandi_fit_device(ctx, lut, y_t, lambda1, lambda2, rmse=False, nrmse=False, stream=None, return_x=False):
    import torch
    _check_dev(lut, y_t)
    n, f64 = y_t.shape[0], dict(dtype=torch.float64, device=y_t.device)
    xd, fx = _debug_x(ctx, (n, lut.n_atoms), y_t, return_x)
    flags = (F_RMSE if rmse else 0) | (F_NRMSE if nrmse else 0) | fx
    est = torch.empty((n, 6), **f64)
    r = torch.empty(n, **f64) if rmse else None
    nr = torch.empty(n, **f64) if nrmse else None
    ctx.check(_dev_fn('sandi', y_t)(ctx._h, lut._h, _dptr(y_t), n, float(lambda1), float(lambda2), flags, _dptr(est),
                                         _dptr(r), _dptr(nr), c_vp(stream or 0)))
    return (est, r, nr, xd) if return_x else (est, r, nr)


class Dict:
    """amx_dict: the dictionaries of the batched solver entry points (cyspams.interfaces.nnls / lasso, models.pyx:18, batched).
    A: [n_dicts, m, n] (or [m, n]) in numpy's own layout -- the column-major m x n slices the C ABI wants are made here."""

    def __init__(self, ctx, A):
        A = np.asarray(A, dtype=np.float64)
        if A.ndim == 2:
            A = A[None]
        if A.ndim != 3:
            raise ValueError('dictionary must be [m, n] or [n_dicts, m, n]')
        self.ctx, (self.n_dicts, self.m, self.n) = ctx, A.shape
        cm = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))          # [n_dicts][n][m]: column-major m x n per dictionary
        h = c_vp()
        ctx.check(lib().amx_dict_upload(ctx._h, _p(cm, c_dp), self.m, self.n, self.n_dicts, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                lib().amx_dict_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _batched_inputs(dic, y, idx):
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
    if y.shape[1] != dic.m:
        raise ValueError(f'y has {y.shape[1]} samples, the dictionary {dic.m}')
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if idx.shape != (y.shape[0],):
            raise ValueError('dict_idx must have one entry per voxel')
    elif dic.n_dicts != 1:
        raise ValueError('dict_idx is required with more than one dictionary')
    return y, idx


def nnls_batched(ctx, dic, y, dict_idx=None, return_rnorm=False):
    """x_v = argmin_{x >= 0} ||A_d x - y_v||, d = dict_idx[v]: (X [n_vox, n], rnorm [n_vox]) -- models.pyx:911, 940"""
    y, idx = _batched_inputs(dic, y, dict_idx)
    x = np.zeros((y.shape[0], dic.n))
    rn = np.zeros(y.shape[0]) if return_rnorm else None
    ctx.check(lib().amx_nnls_batched(ctx._h, dic._h, _p(idx, c_i32p) if idx is not None else None, _p(y, c_dp), y.shape[0], _p(x, c_dp),
                                     _p(rn, c_dp) if rn is not None else None))
    return (x, rn) if return_rnorm else x


def lasso_batched(ctx, dic, y, lambda1, lambda2, dict_idx=None):
    """x_v = argmin_{x >= 0} 1/2 ||y_v - A_d x||^2 + lambda1 sum(x) + lambda2 / 2 ||x||^2 -- models.pyx:615, 926, 1238, 1569"""
    y, idx = _batched_inputs(dic, y, dict_idx)
    x = np.zeros((y.shape[0], dic.n))
    ctx.check(lib().amx_lasso_batched(ctx._h, dic._h, _p(idx, c_i32p) if idx is not None else None, _p(y, c_dp), y.shape[0],
                                      float(lambda1), float(lambda2), _p(x, c_dp)))
    return x


def dir_to_lut_idx(ctx, lut, dirs):
    dirs = np.ascontiguousarray(np.atleast_2d(dirs), dtype=np.float64)
    out = np.zeros(dirs.shape[0], dtype=np.int32)
    ctx.check(lib().amx_dir_to_lut_idx(ctx._h, lut._h, _p(dirs, c_dp), dirs.shape[0], _p(out, c_i32p)))
    return out


class Dti:
    """amx_dti: principal-direction estimator of one acquisition scheme (include/amico_amd.h, row f1)."""

    def __init__(self, ctx, inv_design, min_signal=1e-4):
        w = np.ascontiguousarray(inv_design, dtype=np.float64)
        if w.ndim != 2 or w.shape[0] != 7:
            raise ValueError('inv_design must be pinv(design matrix), shape [7, nS]')
        self.ctx, self.nS = ctx, w.shape[1]
        self._h = c_vp()
        ctx.check(lib().amx_dti_create(ctx._h, _p(w, c_dp), self.nS, float(min_signal), C.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
            lib().amx_dti_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def directions(self, y):
        """y f64[n_vox, nS] (host) -> f64[n_vox, 3]."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.ndim != 2 or y.shape[1] != self.nS:
            raise ValueError('y must be [n_vox, %d]' % self.nS)
        out = np.zeros((y.shape[0], 3))
        self.ctx.check(lib().amx_dti_directions(self.ctx._h, self._h, _p(y, c_dp), y.shape[0], _p(out, c_dp)))
        return out

    def directions_device(self, d_y, n_vox, d_dirs, stream=None, f32=False):
        """device pointers (ints), enqueued on `stream`; check with ctx.sync(stream).  f32: d_y holds float32 signals."""
        fn = lib().amx_dti_directions_device_f32 if f32 else lib().amx_dti_directions_device
        self.ctx.check(fn(self.ctx._h, self._h, c_vp(d_y), int(n_vox), c_vp(d_dirs), c_vp(stream or 0)))


class Prep:
    """amx_prep: signal-preparation / result-scatter plan of one image geometry + mask + volume grouping."""

    def __init__(self, ctx, shape, strides, rank, groups, b0_idx, overwrite_in_order=False):
        """shape (X, Y, Z, nS); strides = element strides of the float32 image; rank int32[X, Y, Z] (C order);
        groups = list of index lists (one per output volume)."""
        self.ctx = ctx
        self.shape = tuple(int(v) for v in shape)
        self.strides = tuple(int(v) for v in strides)
        dims = np.asarray(self.shape[:3], dtype=np.int64)
        st = np.asarray(self.strides, dtype=np.int64)
        rank = np.ascontiguousarray(rank, dtype=np.int32)
        if rank.shape != self.shape[:3]:
            raise ValueError('rank must have the spatial shape of the image')
        self.n_vox = int((rank >= 0).sum())
        gptr = np.zeros(len(groups) + 1, dtype=np.int32)
        gptr[1:] = np.cumsum([len(g) for g in groups])
        gidx = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int32).ravel() for g in groups]), dtype=np.int32)
        b0 = np.ascontiguousarray(b0_idx, dtype=np.int32)
        self.n_out = len(groups)
        self.extent = 1 + sum((d - 1) * s for d, s in zip(self.shape, self.strides))
        self._h = c_vp()
        ctx.check(lib().amx_prep_create(ctx._h, _p(dims, c_i64p), _p(st, c_i64p), self.shape[3], _p(rank, c_i32p),
                                        self.n_vox, _p(gptr, c_i32p), _p(gidx, c_i32p), self.n_out,
                                        _p(b0, c_i32p) if len(b0) else None, len(b0), int(bool(overwrite_in_order)),
                                        C.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
            lib().amx_prep_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _img_buffer(self, img):
        """the float32 image as the flat element buffer the strides refer to (no copy for C / Fortran arrays)"""
        if img.dtype != np.float32 or img.shape != self.shape or \
                tuple(s // 4 for s in img.strides) != self.strides:
            raise ValueError('image does not match the plan (dtype float32, shape, strides)')
        flat = np.lib.stride_tricks.as_strided(img, shape=(self.extent,), strides=(4,))
        return flat

    def gather(self, img, normalize=True, b0_threshold=0.0):
        buf = self._img_buffer(img)
        y = np.zeros((self.n_vox, self.n_out))
        mb0 = np.zeros(self.n_vox, dtype=np.float32)
        self.ctx.check(lib().amx_prep_gather(self.ctx._h, self._h, _p(buf, c_fp), int(bool(normalize)),
                                             float(b0_threshold), _p(y, c_dp), _p(mb0, c_fp)))
        return y, (mb0 if normalize else None)

    def mean_b0(self, img):
        buf = self._img_buffer(img)
        out = np.zeros(self.shape[:3], dtype=np.float32)
        self.ctx.check(lib().amx_prep_mean_b0(self.ctx._h, self._h, _p(buf, c_fp), _p(out, c_fp)))
        return out

    def scatter(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            v = v[:, None]
        if v.shape[0] != self.n_vox:
            raise ValueError('values must have one row per masked voxel')
        out = np.zeros(self.shape[:3] + (v.shape[1],), dtype=np.float32)
        self.ctx.check(lib().amx_prep_scatter(self.ctx._h, self._h, _p(v, c_dp), v.shape[1], _p(out, c_fp)))
        return out


def lut_resample(ctx, lm, ylm_out, idx_out, nS):
    """lm f32[..., n_sh] -> f32[..., nS] with ones outside idx_out (resample_kernel, lut.pyx:274-311, batched)"""
    lm = np.ascontiguousarray(lm, dtype=np.float32)
    y = np.ascontiguousarray(ylm_out, dtype=np.float32)
    idx = np.ascontiguousarray(idx_out, dtype=np.int32)
    if y.ndim != 2 or lm.shape[-1] != y.shape[1] or idx.shape != (y.shape[0],):
        raise ValueError('Outdated LUT. Call "generate_kernels( regenerate=True )" to update the LUT')   # lut.pyx:301
    rows = int(np.prod(lm.shape[:-1], dtype=np.int64))
    out = np.empty(lm.shape[:-1] + (int(nS),), dtype=np.float32)
    ctx.check(lib().amx_lut_resample(ctx._h, _p(lm, c_fp), rows, lm.shape[-1], _p(y, c_fp), _p(idx, c_i32p),
                                     y.shape[0], int(nS), _p(out, c_fp)))
    return out


def lut_rotate_resample(ctx, zonal, ylm_rot, ylm_out, idx_out, nS):
    """zonal f32[n_atoms, n_shells * nSH] (const * Klm[idx_m0] per shell), ylm_rot f32[ndirs, nSH] -> f32[n_atoms, ndirs, nS]:
    rotate_kernel + resample_kernel of lut.pyx:227-311 in one GEMM whose left operand is formed on the fly"""
    z = np.ascontiguousarray(zonal, dtype=np.float32)
    r = np.ascontiguousarray(ylm_rot, dtype=np.float32)
    y = np.ascontiguousarray(ylm_out, dtype=np.float32)
    idx = np.ascontiguousarray(idx_out, dtype=np.int32)
    if z.ndim != 2 or r.ndim != 2 or y.ndim != 2 or z.shape[1] % r.shape[1] or y.shape[1] != z.shape[1] or idx.shape != (y.shape[0],):
        raise ValueError('Outdated LUT. Call "generate_kernels( regenerate=True )" to update the LUT')
    out = np.empty((z.shape[0], r.shape[0], int(nS)), dtype=np.float32)
    ctx.check(lib().amx_lut_rotate_resample(ctx._h, _p(z, c_fp), z.shape[0], _p(r, c_fp), r.shape[0], r.shape[1],
                                            z.shape[1] // r.shape[1], _p(y, c_fp), _p(idx, c_i32p), y.shape[0], int(nS), _p(out, c_fp)))
    return out
