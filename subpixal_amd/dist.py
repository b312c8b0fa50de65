"""Multi-GPU driver: the cutout batch shards by contiguous index blocks, one
process per GPU; the only collective is a gather of the per-cutout (dx, dy)
shifts (float64 [N/R, 2]) to rank 0.  With
``torch.distributed`` backend "nccl" this is RCCL over xGMI; the same code runs
under "gloo" on CPU tensors for the unit tests.
"""
import torch
import torch.distributed as dist

__all__ = ['shard_range', 'gather_shifts', 'xcorr_refine_sharded']


def shard_range(n, rank, world):
    """Half-open index range [lo, hi) of rank ``rank`` of ``world`` for ``n`` items
    (contiguous blocks, remainders spread over the first ranks)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_shifts(local, n_total=None, dst=0, group=None):
    """Gather per-rank shift blocks ``local [n_r, 2]`` onto ``dst`` in rank order.
    Returns the concatenated ``[n_total, 2]`` tensor on ``dst`` and None elsewhere.
    Blocks may differ in length by one (see :func:`shard_range`)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_total is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(cnt, group=group)
        n_total = int(cnt.item())
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    padded = local
    if local.shape[0] < longest:           # equal-size buffers for the collective
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]),
                          dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], dim=0)
    padded = padded.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, gather_list=bufs, dst=dst, group=group)
        return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
    dist.gather(padded, gather_list=None, dst=dst, group=group)
    return None


def xcorr_refine_sharded(make_local_batch, n_total, upsample=1, cc_type='CC', dst=0,
                         compute=None):
    """Each rank builds its own block of the batch with
    ``make_local_batch(lo, hi) -> (ref, img)`` (no input exchange), runs the pair
    kernel on it and the shifts are gathered on ``dst``.  ``compute`` defaults to
    :func:`subpixal_amd.cc.xcorr_refine_batch` (tests inject a stand-in to run the
    sharding/gather logic under gloo without a GPU)."""
    if compute is None:
        from .cc import xcorr_refine_batch as compute
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_total, rank, world)
    ref, img = make_local_batch(lo, hi)
    local = compute(ref, img, upsample=upsample, cc_type=cc_type)
    return gather_shifts(local, n_total=n_total, dst=dst)
