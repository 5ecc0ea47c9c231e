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
