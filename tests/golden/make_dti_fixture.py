"""Golden vectors for the principal-direction step (SURVEY section 8 f, row 1; core.py:428-436, 456-458).

dipy is not installed in this image, so the expected directions come from a route that is independent of both
oracle/signal_np.py and the HIP kernel: scipy.linalg.lstsq (LAPACK gelsd) on the log-signal for the tensor
parameters and scipy.linalg.eigh (evr driver) per voxel.  Inputs: the 160 NODDI voxels of noddi_fixture.npz
(reference physics, Rician noise) + 64 noise-free single-tensor voxels whose principal axis is known in closed form.
Run:  python tests/golden/make_dti_fixture.py
"""
import os
import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    f = np.load(os.path.join(HERE, 'noddi_fixture.npz'))
    scheme = f['scheme']
    b, g = scheme[:, 3], scheme[:, :3]
    rng = np.random.default_rng(7)
    # closed-form voxels: D = R diag(l1, l2, l3) R^T with l1 > l2 > l3
    n_t = 64
    axes = rng.normal(size=(n_t, 3, 3))
    y_t = np.zeros((n_t, len(b)))
    e1 = np.zeros((n_t, 3))
    for i in range(n_t):
        q, _ = np.linalg.qr(axes[i])
        lam = np.sort(rng.uniform(0.2e-3, 2.0e-3, 3))[::-1] * np.array([1.3, 1.0, 0.8])
        D = (q * lam) @ q.T
        e1[i] = q[:, 0]
        y_t[i] = rng.uniform(0.5, 1.5) * np.exp(-b * np.einsum('ij,jk,ik->i', g, D, g))
    y = np.vstack([f['y'], y_t])
    # independent expected route
    B = -np.column_stack([b * g[:, 0] ** 2, 2 * b * g[:, 0] * g[:, 1], b * g[:, 1] ** 2, 2 * b * g[:, 0] * g[:, 2],
                          2 * b * g[:, 1] * g[:, 2], b * g[:, 2] ** 2, np.ones_like(b)])
    p = sl.lstsq(B, np.log(np.maximum(y, 1e-4)).T, lapack_driver='gelsd')[0].T
    dirs = np.zeros((len(y), 3))
    evals = np.zeros((len(y), 3))
    for i, pi in enumerate(p):
        D = np.array([[pi[0], pi[1], pi[3]], [pi[1], pi[2], pi[4]], [pi[3], pi[4], pi[5]]])
        w, v = sl.eigh(D, driver='evr')
        dirs[i] = v[:, 2]
        evals[i] = w[::-1]
    assert np.all(np.abs(np.abs((dirs[160:] * e1).sum(1)) - 1) < 1e-12)
    np.savez_compressed(os.path.join(HERE, 'dti_fixture.npz'), scheme=scheme, y=y, dirs=dirs, evals=evals,
                        closed_form_axis=e1)
    print('dti_fixture.npz:', y.shape, 'min eigen-gap', float((evals[:, 0] - evals[:, 1]).min()))


if __name__ == '__main__':
    main()
