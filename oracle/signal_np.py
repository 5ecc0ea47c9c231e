"""ORACLE (test infrastructure only) -- numpy restatement of the steps either side of model.fit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(amico_amd/) never does.

PARITY UNPINNED: the arithmetic of this step lives in a third-party dependency that is absent from
/root/reference and from this image -- dipy (pinned `dipy>=1.4.1`, requirements.txt:3 / setup.cfg:39).  The
reference holds no test or golden vector for it.  What follows restates dipy's published algorithm
(dipy/reconst/dti.py: design_matrix, TensorModel.fit, ols_fit_tensor, decompose_tensor; dipy/core/gradients.py:
gradient_table_from_bvals_bvecs, GradientTable) with the same numpy/LAPACK calls dipy makes, anchored on the
reference's call site core.py:428-436, 456-458; it is pinned by hand-checkable cases (noise-free tensor signals
whose principal axis is known in closed form) in tests/test_oracle.py.
"""
import numpy as np

MIN_POSITIVE_SIGNAL = 0.0001


def gradient_table(bvals, bvecs, b0_threshold=50.0, atol=1e-2):
    # gradient_table_from_bvals_bvecs: non-unit vectors are zeroed together with their b-value (legal on b0s only)
    bvals = np.asarray(bvals, dtype=float)
    bvecs = np.where(np.isnan(bvecs), 0, np.asarray(bvecs, dtype=float))
    close_to_1 = abs(np.linalg.norm(bvecs, axis=1) - 1) <= atol
    if not np.all(close_to_1[bvals > b0_threshold]):
        raise ValueError('The vectors in bvecs should be unit')
    bvecs = np.where(close_to_1[:, None], bvecs, 0)
    bvals = bvals * close_to_1
    gradients = bvals[:, None] * bvecs
    # GradientTable: bvals / bvecs are derived back from the gradients
    b = np.linalg.norm(gradients, axis=1)
    with np.errstate(divide='ignore', invalid='ignore'):
        g = np.where(b[:, None] > 0, gradients / b[:, None], 0.0)
    return b, g


def design_matrix(b, g):
    B = np.zeros((len(b), 7))
    B[:, 0] = g[:, 0] * g[:, 0] * 1. * b    # Bxx
    B[:, 1] = g[:, 0] * g[:, 1] * 2. * b    # Bxy
    B[:, 2] = g[:, 1] * g[:, 1] * 1. * b    # Byy
    B[:, 3] = g[:, 0] * g[:, 2] * 2. * b    # Bxz
    B[:, 4] = g[:, 1] * g[:, 2] * 2. * b    # Byz
    B[:, 5] = g[:, 2] * g[:, 2] * 1. * b    # Bzz
    B[:, 6] = np.ones(len(b))
    return -B


def dti_directions(y, bvals, bvecs, min_signal=None, return_evals=False):
    """np.squeeze(TensorModel(gtab, fit_method='OLS').fit(y).directions) of core.py:456-458."""
    b, g = gradient_table(bvals, bvecs)
    B = design_matrix(b, g)
    data = np.maximum(np.asarray(y, dtype=float).reshape(-1, B.shape[0]),
                      MIN_POSITIVE_SIGNAL if min_signal is None else min_signal)
    p = np.einsum('ij,...j', np.linalg.pinv(B), np.log(data))           # ols_fit_tensor
    D = np.empty((p.shape[0], 3, 3))
    lt = np.array([[0, 1, 3], [1, 2, 4], [3, 4, 5]])                     # from_lower_triangular
    D[:] = p[:, lt]
    evals, evecs = np.linalg.eigh(D)                                     # decompose_tensor: ascending ...
    order = np.argsort(evals, axis=1)[:, ::-1]                           # ... re-sorted in descending order
    first = order[:, 0]
    dirs = evecs[np.arange(len(first)), :, first]                        # evecs[..., :, 0] after the re-sort
    if return_evals:
        return dirs, np.take_along_axis(evals, order, axis=1)
    return dirs


# ----------------------------------------------------------------------------- signal preparation (numpy IS the
# reference's arithmetic here: core.py:209-268 and 451-452 are numpy expressions on the float32 image)
def prepare_signal(img, mask, b0_idx, dwi_idx, shells=None, do_normalize=True, do_merge_b0=False,
                   do_directional_average=False, b0_min_signal=0):
    """float32 image [X,Y,Z,nS] -> (y f64[n_vox, n_out], mean_b0s float32 volume or None), following load_data's
    preprocessing (core.py:209-252) and fit's gather (core.py:451-452) statement by statement."""
    img = np.array(img, dtype=np.float32, copy=True, order='K')
    mean_b0s = None
    if do_normalize:
        if len(b0_idx) == 0:
            raise RuntimeError('No b0 volume to normalize signal with')
        mean_b0s = np.mean(img[:, :, :, b0_idx], axis=3)
        norm_factor = mean_b0s.copy()
        idx = norm_factor <= b0_min_signal * norm_factor[norm_factor > 0].mean()
        norm_factor[idx] = 1
        norm_factor = 1 / norm_factor
        norm_factor[idx] = 0
        for i in range(img.shape[3]):
            img[:, :, :, i] *= norm_factor
    if do_merge_b0:
        mean = np.expand_dims(np.mean(img[:, :, :, b0_idx], axis=3), axis=3)
        img = np.concatenate((mean, img[:, :, :, dwi_idx]), axis=3)
    if do_directional_average:
        n_sh = len(shells)
        avg = img[:, :, :, :(n_sh + 1)]
        avg[:, :, :, 0] = np.mean(img[:, :, :, b0_idx], axis=3)
        for k, s in enumerate(np.argsort([sh['b'] for sh in shells])):
            avg[:, :, :, k + 1] = np.mean(img[:, :, :, shells[s]['idx']], axis=3)
        img = avg.astype(np.float32)
    y = img[np.asarray(mask) == 1, :].astype(np.double)
    y[y < 0] = 0
    return y, mean_b0s


def scatter_results(values, mask, dtype=np.float32):
    values = np.asarray(values)
    out = np.zeros(mask.shape + values.shape[1:], dtype=dtype)
    out[np.asarray(mask) == 1] = values
    return out
