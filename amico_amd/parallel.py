"""Multi-GPU sharding of the fit path: voxels are independent, so the only
data-path collective is ONE gather of the per-voxel maps at the end (RCCL over xGMI when the
process group backend is "nccl"; gloo in the CPU tests).  One process per GPU.

Shards are the contiguous chunks of ``BaseModel.fit`` (amico/models.pyx:204-211: c = n //
world, the last chunk absorbs the remainder) so that rank order == voxel order.
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


def gather_maps(local, n_total, group=None):
    """All ranks contribute their [n_local, k] block (torch tensor, any device the backend
    supports); every rank gets the full [n_total, k] array in voxel order.  Shards may differ
    in length (last one absorbs the remainder): blocks are padded to the longest shard so a
    single all_gather moves them."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k = local.shape[1]
    c = n_total // world
    longest = n_total - (world - 1) * c
    buf = torch.zeros((longest, k), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = torch.empty((world * longest, k), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        i, j = shard_range(n_total, r, world)
        parts.append(out[r * longest: r * longest + (j - i)])
    assert rank < world
    return torch.cat(parts, dim=0)


def fit_sharded(model, evaluation, group=None, directions=None):
    """`model.fit(evaluation)` with the voxels of `evaluation` split over the ranks of `group`;
    returns the same dict as the single-GPU call on every rank.  `directions` (optional): an estimator with
    `.fit(y) -> dirs` (amico_amd.dti.TensorDirections) run on each rank's shard when `evaluation.DIRs` is
    None -- the step before the fit shards the same way; the gathered result then carries 'DIRs' too."""
    import copy
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = evaluation.y.shape[0]
    i, j = shard_range(n, rank, world)
    ev = copy.copy(evaluation)
    ev.y = evaluation.y[i:j]
    ev.DIRs = None if evaluation.DIRs is None else evaluation.DIRs[i:j]
    estimated = ev.DIRs is None and directions is not None
    if estimated:
        ev.DIRs = directions.fit(ev.y)
    res = dict(model.fit(ev))
    if estimated:
        res['DIRs'] = ev.DIRs
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    out = {}
    for key, val in res.items():
        t = torch.from_numpy(np.ascontiguousarray(val.reshape(val.shape[0], -1))).to(dev)
        g = gather_maps(t, n, group).cpu().numpy()
        out[key] = g.reshape((n,) + val.shape[1:])
    return out
