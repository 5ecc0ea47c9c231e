"""The N-GPU path on the one GPU a test box has: RCCL ("nccl") with a single rank exercises init_process_group, the packed
all_gather_into_tensor on device memory and bench.py's step / timing helpers; `bench.py --gpus 2` must start its own ranks
and fail with a clear message when the GPUs are not there."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _nccl_world1(port):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1)
    return dist


def test_sharded_step_and_fit_sharded_under_rccl_world1():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from amico_amd import _capi, get_context, synthetic as S
    from amico_amd.parallel import fit_sharded, gather_maps
    dist = _nccl_world1(36500 + os.getpid() % 2000)
    try:
        dev = torch.device('cuda', 0)
        ctx = get_context()
        dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
        sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
        n = 4096
        y_h, d_h = S.noddi_signals(n, K, ht, sch, seed=11)
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
        y = torch.from_numpy(y_h).to(dev); d = torch.from_numpy(d_h).to(dev)
        est = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        gathered = torch.full((n, 3), -1.0, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        L = _capi.lib()

        def fit():
            ctx.check(L.amx_noddi_fit_device(ctx._h, lut._h, y.data_ptr(), d.data_ptr(), n, 0.5, 1e-3, 0, est.data_ptr(), None, None, None, stream))
        step = bench.sharded_step(fit, est, gathered, 1)                 # fit + the one collective, on device memory
        el = bench.timed_steps(step, lambda: ctx.sync(stream), 2, 1, 1, dev)
        assert el > 0
        assert torch.equal(gathered, est)
        ref = _capi.noddi_fit(ctx, lut, y_h, d_h, 0.5, 1e-3, 3)[0]
        assert np.abs(gathered.cpu().numpy() - ref).max() < 1e-12
        # the packed gather of fit_sharded (unequal-shard code path, here one shard)
        got = gather_maps(est, n)
        assert got.is_cuda and torch.equal(got, est)

        class Model:
            def fit(self, ev):
                return {'estimates': _capi.noddi_fit(ctx, lut, ev.y, ev.DIRs, 0.5, 1e-3, 3)[0]}

        class Ev:
            pass
        ev = Ev(); ev.y = y_h; ev.DIRs = d_h
        out = fit_sharded(Model(), ev, n_total=n)
        assert np.abs(out['estimates'] - ref).max() < 1e-12
    finally:
        dist.destroy_process_group()


def test_fit_sharded_one_million_voxel_shard_under_rccl():
    """config 5's per-GPU share: a 1 M-voxel shard handed to fit_sharded(n_total=...) with the process group on RCCL, the maps
    gathered straight from HBM by the packed all_gather_into_tensor (world 1: the only world a test box has)"""
    import torch
    import amico_amd
    from amico_amd import synthetic as S
    from amico_amd.parallel import fit_sharded, shard_range
    from oracle import oracle
    dist = _nccl_world1(38600 + os.getpid() % 2000)
    try:
        dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
        sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
        n = 1_000_000
        assert shard_range(8 * n, 3, 8) == (3 * n, 4 * n)               # models.pyx:204-211: rank 3 of 8 owns [3 M, 4 M)
        y_h, d_h = S.noddi_signals_parallel(n, K, ht, sch, seed=31)
        m = amico_amd.NODDI(); m.scheme = sch

        class Ev:
            def get_config(self, key):
                return False
        ev = Ev(); ev.y = y_h; ev.DIRs = d_h; ev.KERNELS = K; ev.htable = ht; ev.nthreads = 8
        out = fit_sharded(m, ev, n_total=n, to_host=False)
        est = out['estimates']
        assert est.is_cuda and est.shape == (n, 3)
        pick = np.arange(0, n, 100)
        ref = oracle.noddi_fit(np.ascontiguousarray(y_h[pick]), np.ascontiguousarray(d_h[pick]), K, ht, sch.dwi_idx, nthreads=os.cpu_count() or 1)['estimates']
        diff = np.abs(est.cpu().numpy()[pick] - ref).max(axis=1)
        assert diff.max() < 1e-4 and (diff > 1e-6).sum() <= 1
    finally:
        dist.destroy_process_group()


def test_bench_gpus2_starts_its_own_ranks():
    """one visible GPU: the command form of the driver's scaling run must get as far as the ranks and say what is missing"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--voxels', '4096',
                        '--no-cpu-baseline', '--no-other-configs'], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert '2 GPUs requested, 1 visible' in (p.stdout + p.stderr)


def test_noddi_fit_is_bitwise_repeatable():
    """The seed solvers run one voxel per lane next to 63 others drawn from a ticket: which voxels share a wavefront changes
    from call to call, the result of a voxel must not (a neighbour-dependent branch once made ~20 of 900 000 seeds differ
    between calls).  Coefficient vectors of all three stages, bit for bit, over repeated calls."""
    import torch
    sys.path.insert(0, ROOT)
    from amico_amd import _capi, get_context, synthetic as S
    ctx = get_context()
    dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
    n = 300_000
    y_h, d_h = S.noddi_signals(n, K, ht, sch, seed=21)
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    y = torch.from_numpy(y_h).cuda(); d = torch.from_numpy(d_h).cuda()
    ref = None
    for rep in range(4):
        out = _capi.noddi_fit_device(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True, return_x=True)
        ctx.sync()
        cur = [out[0].clone(), out[1].clone(), out[-1].clone()]
        if ref is None:
            ref = cur
            continue
        for a, b in zip(ref, cur):
            assert torch.equal(a, b)


@pytest.mark.parametrize('n,share', [(150_001, 0.9), (64_123, 1.0), (300_000, 0.5)])
def test_noddi_skewed_orientation_histograms(n, share, amx_env):
    """Work sharing of the lane kernels (SeedFeed / BlockFeed, csrc/amx_seed.hpp): `share` of the voxels point along THREE
    directions, so three orientations hold (almost) everything, are cut into many chunks, and the workgroups of the ~500 empty
    orientations have nothing but other chunks to join; odd voxel counts; the rescue pass forced on.  Every voxel is fitted once
    -- the maps equal the oracle's on a sample and the wavefront-per-voxel path's (AMX_NO_SEED=1) everywhere -- and twice the same."""
    import torch
    sys.path.insert(0, ROOT)
    from amico_amd import _capi, get_context, synthetic as S
    from oracle import oracle
    dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
    rng = np.random.default_rng(5)
    y_h, d_h = S.noddi_signals(n, K, ht, sch, seed=77)
    # re-point `share` of the voxels (keeping their signals: a wrong direction is just a harder voxel) at three fixed directions
    three = S.random_unit_vectors(3, rng)
    sel = rng.uniform(size=n) < share
    d_h[sel] = three[rng.integers(0, 3, int(sel.sum()))]
    amx_env(AMX_SEED_MIN_VOXELS='0', AMX_RESCUE_FROM='0')
    ctx = get_context()
    lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
    y = torch.from_numpy(y_h).cuda(); d = torch.from_numpy(d_h).cuda()
    out = _capi.noddi_fit_device(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
    ctx.sync()
    st, ss = ctx.last_stats(), ctx.last_seed_stats()
    assert st['itercap_voxels'] == 0 and st['guard_trips'] == 0 and st['overflow_voxels'] == 0
    assert ss['seeded_voxels'] == n
    est, rmse = out[0].cpu().numpy(), out[1].cpu().numpy()
    again = _capi.noddi_fit_device(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
    ctx.sync()
    assert torch.equal(out[0], again[0]) and torch.equal(out[1], again[1])
    pick = np.unique(np.linspace(0, n - 1, 4000).astype(np.int64))
    ref = oracle.noddi_fit(np.ascontiguousarray(y_h[pick]), np.ascontiguousarray(d_h[pick]), K, ht, sch.dwi_idx, nthreads=os.cpu_count() or 1)
    assert np.abs(est[pick] - ref['estimates']).max() < 1e-6
    amx_env(AMX_NO_SEED='1')
    ctx2 = get_context()
    lut2 = _capi.upload_noddi(ctx2, K, ht, sch.dwi_idx)
    cold = _capi.noddi_fit_device(ctx2, lut2, y, d, 0.5, 1e-3, 3, rmse=True)
    ctx2.sync()
    assert np.abs(cold[0].cpu().numpy() - est).max() < 1e-7 and np.abs(cold[1].cpu().numpy() - rmse).max() < 1e-7


def test_order_of_the_left_over_lists_does_not_show_in_the_maps(amx_env):
    """The NNLS left-over kernels walk their lists twice (voxels with a wrong or no seed first, conditioning-only refusals after:
    csrc/amx_kernels.hpp k_noddi), and the seed solvers give voxels up after a number of trips.  A voxel's result depends on the voxel
    alone: the maps are bit-identical with the lists walked in the order they were written (AMX_NO_HARD_FIRST=1), and the maps with
    trip caps of 64 agree with the default's to solver precision (a given-up voxel is solved by Lawson-Hanson from its signal)."""
    import torch
    sys.path.insert(0, ROOT)
    from amico_amd import _capi, get_context, synthetic as S
    dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
    n = 120_000
    y_h, d_h = S.noddi_signals(n, K, ht, sch, seed=21, snr=50.0)             # clean data: the most conditioning-only refusals
    y = torch.from_numpy(y_h).cuda(); d = torch.from_numpy(d_h).cuda()
    res = {}
    for name, env in (('default', {}), ('list order', dict(AMX_NO_HARD_FIRST='1')), ('caps 64', dict(AMX_SEED_TRIPCAP='64,64,64'))):
        amx_env(**{**dict(AMX_NO_HARD_FIRST=None, AMX_SEED_TRIPCAP=None), **env})
        ctx = get_context()
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
        out = _capi.noddi_fit_device(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
        ctx.sync()
        ss = ctx.last_seed_stats()
        assert ss['seeded_voxels'] == n and ss['leftover_stage1'] > n // 50, ss
        res[name] = (out[0].cpu().numpy(), out[1].cpu().numpy())
    assert np.array_equal(res['default'][0], res['list order'][0]) and np.array_equal(res['default'][1], res['list order'][1])
    assert np.abs(res['default'][0] - res['caps 64'][0]).max() < 1e-8
