"""Principal fibre directions from the log-linear diffusion-tensor fit (SURVEY section 8 f, row 1).

Mirrors what `Evaluation.fit` does before `model.fit` (core.py:431-436, 456-458):

    gtab = gradient_table(bvals=scheme.b, bvecs=scheme.raw[:, :3])
    DTI  = dti.TensorModel(gtab, fit_method='OLS')
    DIRs = np.squeeze(DTI.fit(y).directions)

The one-off part (gradient table, design matrix, pseudo-inverse: a 7 x nS matrix per scheme) is host numpy,
like the reference; the per-voxel part (log, contraction, 3x3 eigen-decomposition) runs on the GPU through
`amx_dti_directions*` (include/amico_amd.h).  There is no CPU fallback.
"""
import numpy as np

from . import _capi

MIN_POSITIVE_SIGNAL = 1e-4      # dipy.reconst.dti.MIN_POSITIVE_SIGNAL (TensorModel.fit default)


def gradient_table(bvals, bvecs, b0_threshold=50.0, atol=1e-2):
    """(bvals, bvecs) as dipy.core.gradients.gradient_table_from_bvals_bvecs + GradientTable see them: vectors
    that are not unit length are only legal on b0 volumes, where they (and the b-value) are zeroed; the table
    then carries bvals = |b g| and bvecs = b g / |b g|."""
    bvals = np.asarray(bvals, dtype=np.float64)
    bvecs = np.asarray(bvecs, dtype=np.float64)
    if bvecs.shape != (bvals.shape[0], 3):
        raise ValueError('bvecs must be [nS, 3]')
    bvecs = np.where(np.isnan(bvecs), 0.0, bvecs)
    unit = np.abs(np.sqrt((bvecs * bvecs).sum(1)) - 1.0) <= atol
    if not np.all(unit[bvals > b0_threshold]):
        raise ValueError('The vectors in bvecs should be unit')
    bvecs = np.where(unit[:, None], bvecs, 0.0)
    grads = (bvals * unit)[:, None] * bvecs
    b = np.sqrt((grads * grads).sum(1))
    g = np.zeros_like(grads)
    nz = b > 0
    g[nz] = grads[nz] / b[nz, None]
    return b, g


def design_matrix(bvals, bvecs):
    """dipy.reconst.dti.design_matrix: rows -[b gx^2, 2 b gx gy, b gy^2, 2 b gx gz, 2 b gy gz, b gz^2, 1]."""
    B = np.zeros((bvals.shape[0], 7))
    B[:, 0] = bvecs[:, 0] * bvecs[:, 0] * bvals
    B[:, 1] = bvecs[:, 0] * bvecs[:, 1] * 2.0 * bvals
    B[:, 2] = bvecs[:, 1] * bvecs[:, 1] * bvals
    B[:, 3] = bvecs[:, 0] * bvecs[:, 2] * 2.0 * bvals
    B[:, 4] = bvecs[:, 1] * bvecs[:, 2] * 2.0 * bvals
    B[:, 5] = bvecs[:, 2] * bvecs[:, 2] * bvals
    B[:, 6] = 1.0
    return -B


class TensorDirections:
    """`TensorModel(gtab, fit_method='OLS').fit(y).directions` of the reference's call site, on the GPU."""

    def __init__(self, bvals, bvecs, min_signal=None, ctx=None):
        from .models import get_context
        if min_signal is not None and min_signal <= 0:
            raise ValueError('The `min_signal` key-word argument needs to be strictly positive.')   # dipy's check
        self.bvals, self.bvecs = gradient_table(bvals, bvecs)
        self.design = design_matrix(self.bvals, self.bvecs)
        self.inv_design = np.linalg.pinv(self.design)
        self.min_signal = MIN_POSITIVE_SIGNAL if min_signal is None else float(min_signal)
        self.ctx = ctx if ctx is not None else get_context()
        self._dti = _capi.Dti(self.ctx, self.inv_design, self.min_signal)

    @classmethod
    def from_scheme(cls, scheme, do_merge_b0=False, **kw):
        """core.py:428-432: with doMergeB0 the table is [one b0] + the DWI volumes."""
        b = np.asarray(scheme.b, dtype=np.float64)
        g = np.asarray(scheme.raw, dtype=np.float64)[:, :3]
        if do_merge_b0:
            idx = np.asarray(scheme.dwi_idx)
            b = np.hstack((0.0, b[idx]))
            g = np.vstack((np.zeros((1, 3)), g[idx]))
        return cls(b, g, **kw)

    def fit(self, y):
        """y [n_vox, nS] -> directions f64[n_vox, 3]."""
        return self._dti.directions(y)

    def fit_device(self, d_y, n_vox, d_dirs, stream=None, f32=False):
        self._dti.directions_device(d_y, n_vox, d_dirs, stream, f32=f32)
