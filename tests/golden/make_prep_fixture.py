"""Golden vectors for the signal-preparation step (SURVEY section 8 f rows 2-3; core.py:209-268, 451-452, 472-498).

The reference's arithmetic for this step is a handful of numpy statements on the float32 image; they are executed
here literally (numpy 2.2, this container) on a small random volume and the inputs + outputs are stored.
Run:  python tests/golden/make_prep_fixture.py
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(11)
    shape = (9, 7, 5)
    # 2 b0 + 3 shells stored out of b-value order (4000, 1000, 2500), 4 directions each
    b = np.array([0, 0] + [4000] * 4 + [1000] * 4 + [2500] * 4, dtype=np.float64)
    b0_idx = np.where(b == 0)[0]
    dwi_idx = np.where(b > 0)[0]
    shells = [np.where(b == v)[0] for v in (4000, 1000, 2500)]          # order of first appearance (scheme.py:88-120)
    img = rng.uniform(0, 900, shape + (len(b),)).astype(np.float32)
    img[..., b0_idx] += 600
    img[0, 0, 0] = 0
    img[1, 1, 1, 5] = -3.25
    mask = (rng.uniform(size=shape) < 0.6).astype(np.uint8)
    mask[0, 0, 0] = mask[1, 1, 1] = 1
    mask[2, 2, 2] = 2
    out = {'img': img, 'mask': mask, 'b': b}

    def normalise(x, b0_min_signal):
        x = x.copy()
        mean_b0s = np.mean(x[:, :, :, b0_idx], axis=3)
        norm_factor = mean_b0s.copy()
        idx = norm_factor <= b0_min_signal * norm_factor[norm_factor > 0].mean()
        norm_factor[idx] = 1
        norm_factor = 1 / norm_factor
        norm_factor[idx] = 0
        for i in range(x.shape[3]):
            x[:, :, :, i] *= norm_factor
        return x, mean_b0s

    def gather(x):
        y = x[mask == 1, :].astype(np.double)
        y[y < 0] = 0
        return y

    x, mb0 = normalise(img, 0)
    out['mean_b0s'] = mb0
    out['y_plain'] = gather(x)
    out['y_raw'] = gather(img)
    x2, _ = normalise(img, 0.9)
    out['y_b0min'] = gather(x2)
    merged = np.concatenate((np.expand_dims(np.mean(x[:, :, :, b0_idx], axis=3), axis=3), x[:, :, :, dwi_idx]), axis=3)
    out['y_merge'] = gather(merged)
    xa = x.copy()
    avg = xa[:, :, :, :4]                                   # a VIEW, as in core.py:231
    avg[:, :, :, 0] = np.mean(xa[:, :, :, b0_idx], axis=3)
    for k, s in enumerate(np.argsort([4000, 1000, 2500])):
        avg[:, :, :, k + 1] = np.mean(xa[:, :, :, shells[s]], axis=3)
    out['y_diravg'] = gather(avg.astype(np.float32))
    vals = rng.normal(size=(int((mask == 1).sum()), 3))
    vol = np.zeros(shape + (3,), dtype=np.float32)
    vol[mask == 1, :] = vals
    out['values'] = vals
    out['volume'] = vol
    np.savez_compressed(os.path.join(HERE, 'prep_fixture.npz'), **out)
    print('prep_fixture.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
