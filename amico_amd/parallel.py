"""Multi-GPU sharding of the fit path: voxels are independent, so the only
data-path collective is ONE gather of the per-voxel results at the end (RCCL over xGMI when the
process group backend is "nccl"; gloo in the CPU tests).  One process per GPU.

Shards are the contiguous chunks of ``BaseModel.fit`` (amico/models.pyx:204-211: c = n //
world, the last chunk absorbs the remainder) so that rank order == voxel order.

No scaling curve has been measured on hardware yet (the builder has one GPU; the driver runs the
N = 1, 2, 4, 8 bench): the path is kept correct by construction and covered by multi-process gloo
tests (world sizes 2 and 3, odd remainders) in tests/test_host_cpu.py.
"""
import numpy as np


def shard_range(n, rank, world):
    """[i, j) of `rank` among `world` contiguous shards of n voxels (models.pyx:204-211 rule)."""
    if world <= 1:
        return 0, n
    c = n // world
    i = rank * c
    j = n if rank == world - 1 else (rank + 1) * c
    return i, j


def gather_packed(blocks, n_total, group=None):
    """ONE all_gather for every per-voxel result of the fit.

    blocks: dict name -> torch tensor [n_local, ...] (this rank's shard, same dtype and device for all names -- they
    stay where the fit left them: HBM under nccl).  The tensors are packed side by side into one [longest, K] buffer
    (shards may differ in length: the last one absorbs the remainder, the others are padded to it), gathered with a single
    all_gather_into_tensor and cut apart again; every rank gets name -> [n_total, ...] in voxel order, on the same
    device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    names = sorted(blocks)
    first = blocks[names[0]]
    n_local = first.shape[0]
    i, j = shard_range(n_total, dist.get_rank(group), world)
    if n_local != j - i:
        raise ValueError(f'rank {dist.get_rank(group)} holds {n_local} voxels, its shard of {n_total} has {j - i}')
    widths = [int(np.prod(blocks[k].shape[1:])) for k in names]          # (explicit: reshape(0, -1) is ambiguous for an empty shard)
    flat = [blocks[k].reshape(n_local, w) for k, w in zip(names, widths)]
    longest = n_total - (world - 1) * (n_total // world)
    buf = torch.zeros((longest, sum(widths)), dtype=first.dtype, device=first.device)
    col = 0
    for t, w in zip(flat, widths):
        buf[:n_local, col:col + w] = t
        col += w
    out = torch.empty((world * longest, sum(widths)), dtype=first.dtype, device=first.device)
    dist.all_gather_into_tensor(out, buf, group=group)            # the single collective of the path
    rows = torch.cat([out[r * longest: r * longest + (shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0])]
                      for r in range(world)], dim=0)
    res, col = {}, 0
    for k, w in zip(names, widths):
        res[k] = rows[:, col:col + w].reshape((n_total,) + tuple(blocks[k].shape[1:]))
        col += w
    return res


def gather_maps(local, n_total, group=None):
    """one [n_local, k] block per rank -> [n_total, k] in voxel order on every rank (see gather_packed)"""
    return gather_packed({'x': local}, n_total, group)['x']


def gather_equal(local, out, group=None):
    """equal shards (the weak-scaling bench: every rank fits the same number of voxels): `out` [world * n, k] receives
    the blocks of all ranks in rank order, straight from / into device memory"""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def fit_sharded(model, evaluation, group=None, directions=None, n_total=None, to_host=True):
    """``model.fit`` with the voxels split over the ranks of `group`; every rank returns the dict of the single-GPU call.

    * ``n_total=None``: `evaluation` holds ALL voxels on every rank and each rank fits its ``shard_range`` of them.
    * ``n_total=N``: `evaluation.y` / `.DIRs` hold only THIS rank's shard (``shard_range(N, rank, world)`` rows) -- a rank
      never needs the other ranks' signals.
    `directions` (optional): an estimator with ``.fit(y) -> dirs`` (amico_amd.dti.TensorDirections) run on the rank's
    shard when ``evaluation.DIRs`` is None -- the step before the fit shards the same way; the result then carries 'DIRs'.
    The results the fit left in HBM (``evaluation._dev['out']``, device-resident path) are gathered from there: no host
    round trip before the collective; ``to_host=False`` returns the gathered torch tensors instead of numpy arrays."""
    import copy
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ev = copy.copy(evaluation)
    if getattr(evaluation, '_dev', None) is not None:
        ev._dev = dict(evaluation._dev)          # the shard's device state must not leak into the caller's evaluation
    if n_total is None:
        n_total = evaluation.y.shape[0]
        i, j = shard_range(n_total, rank, world)
        ev.y = evaluation.y[i:j]
        ev.DIRs = None if evaluation.DIRs is None else evaluation.DIRs[i:j]
    else:
        i, j = shard_range(n_total, rank, world)
        if evaluation.y.shape[0] != j - i:
            raise ValueError(f'rank {rank} was given {evaluation.y.shape[0]} voxels, its shard of {n_total} has {j - i}')
    estimated = ev.DIRs is None and directions is not None
    if estimated:
        ev.DIRs = directions.fit(ev.y)
    res = dict(model.fit(ev))
    if estimated:
        res['DIRs'] = ev.DIRs
    on_gpu = dist.get_backend(group) == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    left = (getattr(ev, '_dev', None) or {}).get('out', {}) if on_gpu else {}
    blocks = {}
    for key, val in res.items():
        t = left.get(key)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(val)).to(dev)
        blocks[key] = t.to(torch.float64)
    out = gather_packed(blocks, n_total, group)
    if not to_host:
        return out
    return {k: v.cpu().numpy().astype(res[k].dtype, copy=False) for k, v in out.items()}
