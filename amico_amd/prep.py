"""Signal preparation around the fit (SURVEY section 8 f, rows 2-3), host side.

What `Evaluation.load_data` does to the 4-D image after reading it (core.py:209-268) and what `Evaluation.fit`
does to get `y` (core.py:451-452) and to store the results (core.py:472-498), planned once per (image geometry,
mask, scheme, options) and executed on the GPU through `amx_prep_*` (include/amico_amd.h).  No CPU fallback.
"""
import numpy as np

from . import _capi


def volume_groups(scheme, do_merge_b0=False, do_directional_average=False):
    """output volume -> input volumes whose float32 mean it is.
    plain: identity; doMergeB0 (core.py:225-227): [b0 volumes] + each DWI volume;
    doDirectionalAverage (core.py:229-252): [b0 volumes] + one group per shell, shells sorted by b-value."""
    b0 = [int(i) for i in scheme.b0_idx]
    if do_directional_average:
        shells = scheme.shells
        order = np.argsort([sh['b'] for sh in shells])
        return [b0] + [[int(i) for i in shells[k]['idx']] for k in order]
    if do_merge_b0:
        return [b0] + [[int(i)] for i in scheme.dwi_idx]
    return [[i] for i in range(scheme.nS)]


def directional_average_table(scheme):
    """the 7-column scheme table of the shell-averaged data (core.py:232-252): b0 row + one x-gradient per shell"""
    shells = scheme.shells
    order = np.argsort([sh['b'] for sh in shells])
    rows = [[1, 0, 0, 0, 0, 0, 0]]
    for k in order:
        sh = shells[k]
        rows.append([1, 0, 0, sh['G'], sh['Delta'], sh['delta'], sh['TE']])
    return np.array(rows, dtype=np.float64)


class SignalPreparation:
    """image [X, Y, Z, nS] float32 (any strides) + mask -> y f64[n_vox, n_out]; per-voxel results -> volumes."""

    def __init__(self, scheme, img_like, mask, do_normalize=True, do_merge_b0=False, do_directional_average=False,
                 b0_min_signal=0.0, ctx=None):
        from .models import get_context
        if img_like.ndim != 4 or img_like.dtype != np.float32:
            raise ValueError('DWI image must be a 4D float32 array')
        if img_like.shape[3] != scheme.nS:
            raise ValueError('Scheme does not match with DWI data')                     # core.py:177-178
        if mask.shape != img_like.shape[:3]:
            raise ValueError('MASK geometry does not match with DWI data')              # core.py:191-192
        if any(s % 4 or s <= 0 for s in img_like.strides):
            raise ValueError('image strides must be positive multiples of the element size')
        if do_normalize and scheme.b0_count == 0:
            raise RuntimeError('No b0 volume to normalize signal with')                 # core.py:214-215
        self.scheme = scheme
        self.do_normalize = bool(do_normalize)
        self.b0_min_signal = float(b0_min_signal)
        self.groups = volume_groups(scheme, do_merge_b0, do_directional_average)
        self.sel = np.asarray(mask) == 1                                                # core.py:451: == 1, not != 0
        rank = np.full(self.sel.shape, -1, dtype=np.int32)
        rank[self.sel] = np.arange(int(self.sel.sum()), dtype=np.int32)                 # C-order enumeration
        self.ctx = ctx if ctx is not None else get_context()
        self._plan = _capi.Prep(self.ctx, img_like.shape, tuple(s // 4 for s in img_like.strides), rank, self.groups,
                                scheme.b0_idx, overwrite_in_order=bool(do_directional_average))
        self.n_vox, self.n_out = self._plan.n_vox, self._plan.n_out
        self.mean_b0s = None

    def b0_threshold(self, img):
        """right-hand side of core.py:217; needs the b0 mean of every voxel only when b0_min_signal != 0"""
        if not self.do_normalize or self.b0_min_signal == 0.0:
            return np.float32(0.0)
        mean_b0s = self._plan.mean_b0(img)
        return self.b0_min_signal * mean_b0s[mean_b0s > 0].mean()

    def gather(self, img):
        """-> (y f64[n_vox, n_out], mean_b0 f32[n_vox] of the masked voxels or None)"""
        y, mb0 = self._plan.gather(img, self.do_normalize, float(self.b0_threshold(img)))
        self.mean_b0s = mb0
        return y, mb0

    def scatter(self, values):
        """per-voxel values [n_vox(, k)] -> float32 volume [X, Y, Z(, k)], zero outside the mask (core.py:472-498)"""
        vol = self._plan.scatter(values)
        return vol[..., 0] if np.ndim(values) == 1 else vol
