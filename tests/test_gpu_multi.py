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


def test_forked_fit_matches(amx_env):
    """AMX_FORK=2 (round 6; profiles/r06_fork_negative.txt): the LASSO left-overs finished on a side stream (k_noddi<4> -> seedless k_noddi<3>)
    while the stage-3 lane kernels skip them.  Not the default (it is not faster) -- but it is a fit: every voxel within rounding of the
    unforked chain and of the oracle, repeatable bit for bit, host-buffer batches included."""
    from amico_amd import _capi, synthetic as S
    from oracle import oracle
    dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
    n = 60000
    y, d = S.noddi_signals(n, K, ht, sch, seed=21)
    y[17, 5] = np.nan                                     # a non-finite voxel goes down the side stream as well
    outs = {}
    for fork in ('0', '2'):
        amx_env(AMX_FORK=fork)
        ctx = _capi.Context(-1)
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
        a = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
        b = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3, rmse=True)
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)
        st, ss = ctx.last_stats(), ctx.last_seed_stats()
        assert st['itercap_voxels'] == 0 and st['overflow_voxels'] == 0 and ss['seeded_voxels'] == n
        outs[fork] = a
        lut.close(); ctx.close()
    ok = np.ones(n, bool); ok[17] = False
    assert np.isnan(outs['2'][0][17]).all() and np.isnan(outs['0'][0][17]).all()
    assert np.abs(outs['2'][0][ok] - outs['0'][0][ok]).max() < 1e-9
    assert np.abs(outs['2'][1][ok] - outs['0'][1][ok]).max() < 1e-9
    assert (outs['2'][0][ok] != outs['0'][0][ok]).any(axis=1).mean() < 0.02       # only the forked voxels take another path
    ref = oracle.noddi_fit(y[:20000], d[:20000], K, ht, sch.dwi_idx, nthreads=os.cpu_count() or 1)
    diff = np.abs(outs['2'][0][:20000] - ref['estimates']).max(axis=1)[ok[:20000]]
    assert (diff < 1e-6).mean() > 0.998 and diff.max() < 1e-4


def test_in_process_multi_device_fit_is_bit_identical(htable500, amx_env):
    """`model.fit(evaluation)` on a device SET (round 6; VERDICT r05 missing 2): one context + host thread per device, contiguous shards
    (models.pyx:204-211), results straight into the caller's arrays.  The one-GPU box names its device twice -- two contexts, two threads,
    two PCIe streams of copies -- and every model's maps must equal the single-context call bit for bit (each shard takes the kernel paths
    the WHOLE call's size asks for: amx_set_call_voxels)."""
    import amico_amd
    from amico_amd import NODDI, FreeWater, SANDI, synthetic as S

    class Holder:
        def __init__(self, y, dirs, htable, kernels, **cfg):
            self.y, self.DIRs, self.htable, self.KERNELS, self.nthreads, self._cfg = y, dirs, htable, kernels, 4, cfg

        def get_config(self, k):
            return self._cfg.get(k, False)
    ht, dirs = htable500['htable'], htable500['dirs']
    sch = S.make_scheme(seed=0)
    K = S.noddi_kernels(sch, dirs)
    n = 90001                                   # shards of 45 000 / 45 001: below the two-wavefront builds' threshold, the call above it
    y, d = S.noddi_signals(n, K, ht, sch, seed=31)
    y = y.astype(np.float32).astype(np.float64)         # (what evaluation.y is: float32 values -- each shard narrows with its own pool)
    m = NODDI(); m.scheme = sch
    try:
        amico_amd.set_devices(None)
        one = m.fit(Holder(y, d, ht, K, doComputeRMSE=True, doSaveModulatedMaps=True))
        amico_amd.set_devices([0, 0])
        assert len(amico_amd.get_contexts()) == 2
        two = m.fit(Holder(y, d, ht, K, doComputeRMSE=True, doSaveModulatedMaps=True))
        three = None
        amico_amd.set_devices([0, 0, 0])
        three = m.fit(Holder(y, d, ht, K, doComputeRMSE=True, doSaveModulatedMaps=True))
        for k in ('estimates', 'rmse', 'estimates_mod'):
            assert np.array_equal(one[k], two[k]), k
            assert np.array_equal(one[k], three[k]), k
        # a bad direction in the second shard is reported with the CALLER's voxel number
        amico_amd.set_devices([0, 0])
        db = d.copy(); db[70000] = np.nan
        with pytest.raises(RuntimeError, match=r'index out of bounds.*\[voxel 70000\]'):
            m.fit(Holder(y, db, ht, K))
        # the other models through the same helper
        sf = S.make_scheme(1, ((1000.0, 64),), seed=3)
        Kf = S.freewater_kernels(sf, dirs)
        yf, df = S.freewater_signals(30011, Kf, ht, sf, seed=5)
        full = S.make_sandi_scheme(); avg = S.directional_average_scheme(full)
        Ks, Rs, d_in, d_isos = S.sandi_kernels(avg)
        ys = S.sandi_signals(20003, Ks, avg, seed=6)
        ms = SANDI(); ms.set(Rs=Rs, d_in=d_in, d_isos=d_isos)
        mf = FreeWater()
        res = {}
        for devs in (None, [0, 0]):
            amico_amd.set_devices(devs)
            res[str(devs)] = (mf.fit(Holder(yf, df, ht, Kf, doComputeNRMSE=True)), ms.fit(Holder(ys, None, None, Ks, doComputeRMSE=True)))
        for a, b in zip(res['None'], res['[0, 0]']):
            for k in a:
                assert np.array_equal(a[k], b[k]), k
    finally:
        amico_amd.set_devices(None)


def test_host_thread_pools_of_sibling_devices_take_disjoint_cores(amx_env):
    """amx_stage::Pool: the pools of the devices on one NUMA node stripe their threads over disjoint shares of the node's
    cores.  AMX_HOST_SIBLINGS=i/n forces what a node with n devices computes for its i-th one."""
    from amico_amd import _capi, synthetic as S
    dirs = S.fibonacci_hemisphere(500); ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0); K = S.noddi_kernels(sch, dirs)
    y, d = S.noddi_signals(30000, K, ht, sch, seed=2)
    y = y.astype(np.float32).astype(np.float64)
    info = {}
    for sib in ('0/1', '0/4', '1/4', '3/4'):
        amx_env(AMX_HOST_SIBLINGS=sib)
        ctx = _capi.Context(-1)
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx)
        est = _capi.noddi_fit(ctx, lut, y, d, 0.5, 1e-3, 3)[0]
        assert ctx.last_host_narrowed() >= 1
        info[sib] = (ctx.host_pool_info(), est)
        lut.close(); ctx.close()
    base = info['0/1'][0]
    if base['first_cpu'] < 0 or base['physical_cores'] < 8:
        pytest.skip('NUMA topology of the device unknown on this box: nothing to divide')
    q = [info[k][0] for k in ('0/4', '1/4', '3/4')]
    assert all(x['physical_cores'] == base['physical_cores'] // 4 for x in q)
    assert len({x['first_cpu'] for x in q}) == 3 and all(x['threads'] >= 4 and 2 * x['threads'] <= max(8, x['physical_cores']) for x in q)
    for k in info:
        assert np.array_equal(info[k][1], info['0/1'][1])
