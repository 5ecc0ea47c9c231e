"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol (no
compute without a GPU), the header and the binding agree, sharding + gather work on gloo."""
import os
import re
import sys
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_library_exports_every_declared_symbol():
    from amico_amd import _capi
    hdr = open(os.path.join(ROOT, 'include', 'amico_amd.h')).read()
    declared = sorted(set(re.findall(r'\b(amx_[a-z0-9_]+)\s*\(', hdr)))
    assert declared, 'no declarations found'
    assert sorted(_capi.SYMBOLS) == declared
    L = _capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.amx_version() >= 100


def test_no_cpu_fallback():
    """without a gfx950 device the product path refuses to run (it never routes to the oracle)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from amico_amd import _capi
    with pytest.raises(RuntimeError, match='no usable MI355X'):
        _capi.Context(-1)
    src = ''
    for f in ('_capi.py', 'models.py', 'core.py', 'parallel.py', '__init__.py'):
        src += open(os.path.join(ROOT, 'amico_amd', f)).read()
    assert 'oracle' not in src.replace('never routes to the oracle', '')


def test_model_surface_matches_reference():
    from amico_amd import NODDI, FreeWater, SANDI
    m = NODDI()
    assert m.id == 'NODDI' and m.maps_name == ['NDI', 'ODI', 'FWF']
    assert m.solver_params == {'lambda1': 0.5, 'lambda2': 1e-3}
    assert len(m.IC_VFs) == 12 and len(m.IC_ODs) == 12
    m.set(isExvivo=True)
    assert m.maps_name[-1] == 'dot'
    f = FreeWater()
    assert f.name == 'Free-Water' and f.solver_params == {'lambda1': 0.0, 'lambda2': 1e-3}
    assert len(f.d_perps) == 10 and f.d_isos == [2.5e-3]
    f.set(type='Mouse')
    assert f.maps_name == ['FiberVolume', 'FW', 'FW_blood', 'FW_csf'] and f.d_isos == [1.5e-3, 3e-3]
    s = SANDI()
    assert s.maps_name == ['fsoma', 'fneurite', 'fextra', 'Rsoma', 'Din', 'De']
    assert s.solver_params == {'lambda1': 0.0, 'lambda2': 5e-3}
    assert set(s.get_params()) == {'id', 'name', 'd_is', 'Rs', 'd_in', 'd_isos'}


def test_shard_range_follows_reference_chunking():
    from amico_amd.parallel import shard_range
    for n, w in [(11, 4), (1000, 8), (8, 8), (1_000_003, 8), (5, 1)]:
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[k][1] == rs[k + 1][0] for k in range(w - 1))
        c = n // w
        assert all(j - i == c for i, j in rs[:-1])


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from amico_amd.parallel import shard_range, gather_maps
    full = np.load(os.path.join(tmp, 'full.npy'))
    n = full.shape[0]
    i, j = shard_range(n, rank, world)
    got = gather_maps(torch.from_numpy(full[i:j].copy()), n).numpy()
    np.save(os.path.join(tmp, f'got{rank}.npy'), got)
    dist.destroy_process_group()


def test_gather_maps_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(0)
    full = rng.random((1001, 3))               # odd length: the last shard is longer
    np.save(tmp_path / 'full.npy', full)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'got{r}.npy'), full)


def _worker_fit(rank, world, port, tmp):
    """N>1 path end to end on CPU: each rank 'fits' its shard (the oracle stands in for the
    GPU library in this test only) and the maps are gathered in voxel order."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from amico_amd.parallel import fit_sharded
    from oracle import oracle
    from conftest import load_npz
    f = load_npz('sandi_fixture.npz')

    class StandIn:
        def fit(self, ev):
            return {'estimates': oracle.sandi_fit(ev.y, {'signal': f['signal'], 'norms': f['norms']}, f['Rs'],
                                                  f['d_in'], f['d_isos'])['estimates']}

    class Ev:
        y, DIRs = f['y'], None

    class Directions:                      # stands in for amico_amd.dti.TensorDirections on the shard
        def fit(self, y):
            return np.cumsum(y[:, :3], axis=1)
    out = fit_sharded(StandIn(), Ev(), directions=Directions())
    assert np.array_equal(out['DIRs'], np.cumsum(f['y'][:, :3], axis=1))
    np.save(os.path.join(tmp, f'fit{rank}.npy'), out['estimates'])
    dist.destroy_process_group()


def test_fit_sharded_world2_gloo(tmp_path, sandi_fix):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_fit, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f'fit{r}.npy')
        assert got.shape == (200, 6)
        assert np.allclose(got, sandi_fix['estimates'], rtol=1e-7, atol=1e-7)


def _worker_presharded(rank, world, port, tmp):
    """every rank holds ONLY its shard (n_total given); all results travel in one packed all_gather; then bench.py's own
    step / timing helpers are driven with a stand-in fit (the GPU library cannot run here)"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from amico_amd.parallel import fit_sharded, shard_range
    full = np.load(os.path.join(tmp, 'full.npy'))
    n = full.shape[0]
    i, j = shard_range(n, rank, world)

    class StandIn:
        def fit(self, ev):
            return {'estimates': ev.y[:, :3] * 2.0, 'rmse': ev.y[:, 3].copy(), 'y_corrected': ev.y + 1.0}

    class Ev:
        y, DIRs = full[i:j].copy(), full[i:j, :3].copy()          # this rank's rows only
    out = fit_sharded(StandIn(), Ev(), n_total=n)
    assert np.array_equal(out['estimates'], full[:, :3] * 2.0) and out['estimates'].shape == (n, 3)
    assert np.array_equal(out['rmse'], full[:, 3]) and out['rmse'].shape == (n,)
    assert np.array_equal(out['y_corrected'], full + 1.0)
    try:
        fit_sharded(StandIn(), Ev(), n_total=n + world)          # shard length does not match: refused, not mis-gathered
        raise AssertionError('mismatched shard accepted')
    except ValueError:
        pass
    # bench.py: one rank = one shard of equal size, ONE collective per step, max-over-ranks timing
    import bench
    m = 257
    est = torch.zeros((m, 3), dtype=torch.float64)
    gathered = torch.zeros((world * m, 3), dtype=torch.float64)
    calls = []

    def fit():
        calls.append(1)
        est[:] = float(rank + 1) * len(calls)
    step = bench.sharded_step(fit, est, gathered, world)
    el = bench.timed_steps(step, lambda: None, 3, 2, world, torch.device('cpu'))
    assert len(calls) == 5
    for r in range(world):
        assert torch.all(gathered[r * m:(r + 1) * m] == float(r + 1) * 5)
    t = torch.tensor([el], dtype=torch.float64)
    lo = t.clone(); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    assert float(lo) == el                                       # every rank reports the same (maximum) time
    assert len(bench.timed_steps.per_rank) == world and max(bench.timed_steps.per_rank) == el    # ... and every rank's own clock beside it
    who = bench.rank_records(world, rank, torch.device('cpu'))    # the bench line's multi_gpu record (devices gathered through the group)
    assert who['rccl_ranks'] == world and who['backend'] == 'gloo' and [r['rank'] for r in who['ranks']] == list(range(world))
    assert len({r['pid'] for r in who['ranks']}) == world        # one process per rank
    np.save(os.path.join(tmp, f'ok{rank}.npy'), np.array([el]))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_presharded_fit_and_bench_step_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(world)
    # 1003 = 3 * 334 + 1 = 2 * 501 + 1: odd remainders; world 8 (config 5's real world size): 8 x 4096 voxels + 5 -- the last of
    # the eight shards is the longer one (models.pyx:204-211) and the packed gather runs at the world size the driver will use
    np.save(tmp_path / 'full.npy', rng.random((8 * 4096 + 5 if world == 8 else 1003, 7)))
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_presharded, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / f'ok{r}.npy')


def test_build_id_matches_the_sources():
    """amx_build_id carries the hash of the sources the library was built from; _capi.source_id() takes the same hash of the tree"""
    from amico_amd import _capi
    bid = _capi.build_id()
    assert bid.startswith('amico_amd ') and ' csrc ' in bid and len(bid.split()[-1]) == 16
    assert _capi.build_is_current(), 'libamico_amd.so is older than its sources: run make -C amico_amd/csrc'


def test_dictionary_cache_is_checked_behind_the_fit(monkeypatch):
    """BaseModel._lut / _verified_fit: the digest of KERNELS runs beside the fit; a fit that ran on a stale upload (KERNELS edited in
    place since) is discarded and run again on the rebuilt dictionary -- the reference re-reads KERNELS on every fit (models.pyx:840-847)"""
    from amico_amd import models as M

    class Fake(M.BaseModel):
        def __init__(self):
            self.id, self.scheme, self.builds, self.fits = 'Fake', None, 0, 0

        def set(self): pass
        def get_params(self): return {}
        def set_solver(self): pass
        def generate(self, *a): pass
        def resample(self, *a): pass

        def _builder(self, ev):
            def build():
                self.builds += 1
                return float(ev.KERNELS['wm'].sum())
            return build

        @M._verified_fit
        def fit(self, ev):
            lut = self._lut(ev, self._builder(ev))
            self.fits += 1
            return {'v': lut}

    monkeypatch.setattr(M, 'get_context', lambda: 'ctx')

    class Ev:
        pass
    ev = Ev()
    ev.KERNELS, ev.htable = {'model': 'Fake', 'wm': np.ones((4, 5), np.float32)}, None
    m = Fake()
    assert m.fit(ev) == {'v': 20.0} and (m.builds, m.fits) == (1, 1)
    assert m.fit(ev) == {'v': 20.0} and (m.builds, m.fits) == (1, 2)          # cached: one fit, no upload
    ev.KERNELS['wm'][1, 1] = 3.0                                               # edited in place: same object, shape, dtype
    assert m.fit(ev) == {'v': 22.0} and (m.builds, m.fits) == (2, 4)          # the stale fit was discarded
    assert m.fit(ev) == {'v': 22.0} and (m.builds, m.fits) == (2, 5)
    assert m._lut(ev, m._builder(ev)) == 22.0 and m.builds == 2                # outside a fit: checked on the spot
    ev.KERNELS['wm'][0, 0] = 5.0
    assert m._lut(ev, m._builder(ev)) == 26.0 and m.builds == 3
    ev.KERNELS['model'] = 'Other'
    ev.KERNELS['wm'] = np.ones((4, 6), np.float32)
    with pytest.raises(ValueError, match='same model'):
        m.fit(ev)
    assert m._lut_pending is False                                             # the wrapper leaves no pending check behind


def test_fit_on_devices_shards_like_the_reference_chunks(monkeypatch):
    """BaseModel._fit_on_devices (round 6): contiguous shards by the chunk rule of models.pyx:204-211, one (fake) context per device, results
    into row slices of the caller's arrays, the whole call's size announced to every context, a shard's error reported with the caller's
    voxel number; no GPU needed."""
    import amico_amd.models as M
    from amico_amd import _capi
    from amico_amd.parallel import shard_range

    class Ctx:
        def __init__(self, k):
            self.k, self.sizes = k, []

        def set_call_voxels(self, t):
            self.sizes.append(t)

        def last_stats(self):
            return {'itercap_voxels': 0}

    ctxs = [Ctx(0), Ctx(1), Ctx(2)]
    monkeypatch.setattr(M, 'get_contexts', lambda: ctxs)
    monkeypatch.setattr(M, '_CTXS', ctxs)

    class Fake(M.BaseModel):
        def __init__(self):
            self.id, self.scheme = 'Fake', None

        def set(self): pass
        def get_params(self): pass
        def set_solver(self): pass
        def fit(self, ev): pass

    class Ev:
        KERNELS, htable = {'model': 'Fake', 'wm': np.ones((2, 3), np.float32)}, None
    m = Fake()
    n = 1003
    seen = []

    def fit_shard(ctx, lut, lo, hi, out):
        assert lut == 'lut%d' % ctx.k
        seen.append((ctx.k, lo, hi))
        out[0][:] = np.arange(lo, hi)[:, None] * np.ones(2)
        assert out[1] is None
        out[2][:] = ctx.k
    est, none, who = m._fit_on_devices(Ev(), n, lambda c: 'lut%d' % c.k, fit_shard, (np.zeros((n, 2)), None, np.zeros(n)))
    assert none is None and np.array_equal(est[:, 0], np.arange(n))
    assert sorted(seen) == [(r,) + shard_range(n, r, 3) for r in range(3)]
    assert [int(who[i]) for i in (0, 333, 334, 667, 668, 1002)] == [0, 0, 1, 1, 2, 2]
    assert all(c.sizes == [n, 0] for c in ctxs)                       # announced, then withdrawn
    # the second call finds the dictionaries cached per context
    calls = []
    m._fit_on_devices(Ev(), n, lambda c: calls.append(c.k) or 'again', fit_shard, (np.zeros((n, 2)), None, np.zeros(n)))
    assert calls == []

    def bad(ctx, lut, lo, hi, out):
        if ctx.k == 2:
            raise _capi.AmxError(-3, '"amico.lut.dir_to_lut_idx" index out of bounds (7, 9) [voxel 5]')
    with pytest.raises(RuntimeError, match=r'\[voxel 673\]'):          # 668 + 5
        m._fit_on_devices(Ev(), n, lambda c: 'x', bad, (np.zeros((n, 2)), None, np.zeros(n)))
    assert all(c.sizes[-1] == 0 for c in ctxs)
