import os
import sys
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a voxel that stops at the iteration cap of an active-set solver is a test failure, not a warning (round-2 verdict:
    # the exact-atom voxel of the NODDI fixture used to end there; the solver now leaves at a zero residual)
    config.addinivalue_line('filterwarnings', 'error:amico_amd.*iteration cap:RuntimeWarning')


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture
def amx_env(monkeypatch):
    """amx_env(AMX_X='1', AMX_Y=None): set / unset library switches for ONE test.  The library reads its environment once per
    context (amx_ctx_create), so the process-wide context is replaced now and again when the test is over."""
    from amico_amd import reset_context

    def setter(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        reset_context()
    yield setter
    monkeypatch.undo()
    reset_context()


@pytest.fixture(scope='session')
def htable500():
    return load_npz('htable500.npz')


def expand_lut(slices, lut_ids, ndirs=500):
    """[n_atoms, n_ids, nS] slices -> full [n_atoms, ndirs, nS] LUT (zeros elsewhere)."""
    full = np.zeros((slices.shape[0], ndirs, slices.shape[2]), dtype=np.float32)
    full[:, lut_ids, :] = slices
    return full


@pytest.fixture(scope='session')
def noddi_fix():
    f = load_npz('noddi_fixture.npz')
    f['kernels'] = {'model': 'NODDI', 'wm': expand_lut(f['wm_slices'], f['lut_ids']), 'iso': f['iso'],
                    'norms': f['norms'], 'icvf': f['icvf'], 'kappa': f['kappa']}
    return f


@pytest.fixture(scope='session')
def fw_fix():
    f = load_npz('freewater_fixture.npz')
    f['kernels'] = {'model': 'FreeWater', 'D': expand_lut(f['D_slices'], f['lut_ids']), 'CSF': f['CSF']}
    return f


@pytest.fixture(scope='session')
def sandi_fix():
    f = load_npz('sandi_fixture.npz')
    f['kernels'] = {'model': 'SANDI', 'signal': np.asfortranarray(f['signal']), 'norms': f['norms']}
    return f


@pytest.fixture(scope='session')
def czb_fix():
    f = load_npz('czb_fixture.npz')
    f['kernels'] = {'model': 'CylinderZeppelinBall', 'wmr': expand_lut(f['wmr_slices'], f['lut_ids']),
                    'wmh': expand_lut(f['wmh_slices'], f['lut_ids']), 'iso': f['iso']}
    return f
