"""Device-resident chain of the per-voxel steps: raw float32 image (in HBM) -> float32 map volumes (in HBM).

    amx_prep_gather_directions_device   b0 normalisation (+ merge / shell average), mask gather, clip     core.py:209-268, 451-452
                                        AND the principal directions (log-linear tensor fit)            core.py:428-436, 456-458
    amx_noddi_fit_device        NNLS -> LASSO -> NNLS, maps                                        models.pyx:816-991
    amx_prep_scatter_device     maps / directions into float32 volumes                             core.py:472-498

Everything is enqueued on one HIP stream; the only host synchronisation is `amx_sync_status` at the end.  torch is
used for device buffers only.  `amico_amd.core.Evaluation` is the host-array (numpy in / numpy out) face of the
same chain.
"""
import numpy as np

from . import _capi, dti as _dti, prep as _prep
from .models import get_context


class NoddiVolumePipeline:
    def __init__(self, scheme, img_like, mask, kernels, htable, lambda1=0.5, lambda2=1e-3, do_normalize=True,
                 b0_min_signal=0.0, device=None, fused=True):
        import torch
        self.fused = bool(fused)           # False: gather, then the tensor fit as its own pass over y (the round-4 chain; A/B)
        self.torch = torch
        self.ctx = get_context()
        self.dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.scheme = scheme
        self.prep = _prep.SignalPreparation(scheme, img_like, mask, do_normalize=do_normalize,
                                            b0_min_signal=b0_min_signal, ctx=self.ctx)
        if b0_min_signal != 0.0:
            raise NotImplementedError('b0_min_signal needs the whole-volume b0 mean on the host: use Evaluation')
        self.tensor = _dti.TensorDirections.from_scheme(scheme, ctx=self.ctx)
        self.lut = _capi.upload_noddi(self.ctx, kernels, htable, scheme.dwi_idx)
        self.lambda1, self.lambda2 = float(lambda1), float(lambda2)
        n = self.prep.n_vox
        self.n_vox, self.shape = n, tuple(img_like.shape[:3])
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.y = torch.empty((n, scheme.nS), dtype=torch.float32, device=self.dev)      # float32 like the image (core.py:136)
        self.dirs = torch.empty((n, 3), **f64)
        self.est = torch.empty((n, 3), **f64)
        self.mean_b0 = torch.empty(n, dtype=torch.float32, device=self.dev)
        self.maps = torch.empty(self.shape + (3,), dtype=torch.float32, device=self.dev)
        self.dirs_vol = torch.empty(self.shape + (3,), dtype=torch.float32, device=self.dev)

    def enqueue(self, d_img, stream=None):
        """d_img: torch float32 tensor holding the image's element buffer (same strides as `img_like`)"""
        L, c, p = _capi.lib(), self.ctx, self.prep._plan
        s = _capi.c_vp(stream or 0)
        if self.fused:
            # one pass over the image: the tensor fit rides on the gather's LDS tile (amx_prep_gather_directions_device_f32)
            c.check(L.amx_prep_gather_directions_device_f32(c._h, p._h, self.tensor._dti._h, d_img.data_ptr(), int(self.prep.do_normalize), 0.0,
                                                            self.y.data_ptr(), self.mean_b0.data_ptr(), self.dirs.data_ptr(), s))
        else:
            c.check(L.amx_prep_gather_device_f32(c._h, p._h, d_img.data_ptr(), int(self.prep.do_normalize), 0.0,
                                                 self.y.data_ptr(), self.mean_b0.data_ptr(), s))
            self.tensor.fit_device(self.y.data_ptr(), self.n_vox, self.dirs.data_ptr(), stream, f32=True)
        c.check(L.amx_noddi_fit_device_f32(c._h, self.lut._h, self.y.data_ptr(), self.dirs.data_ptr(), self.n_vox,
                                       self.lambda1, self.lambda2, 0, self.est.data_ptr(), None, None, None, s))
        c.check(L.amx_prep_scatter_device(c._h, p._h, self.est.data_ptr(), 3, self.maps.data_ptr(), s))
        c.check(L.amx_prep_scatter_device(c._h, p._h, self.dirs.data_ptr(), 3, self.dirs_vol.data_ptr(), s))

    def run(self, d_img, stream=None):
        self.enqueue(d_img, stream)
        self.ctx.sync(stream)
        return self.maps, self.dirs_vol
