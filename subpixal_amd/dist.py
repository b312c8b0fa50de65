"""Multi-GPU driver: the cutout batch shards by contiguous index blocks, one
process per GPU; the only collective is a gather of the per-cutout (dx, dy)
shifts (float64 [N/R, 2]) to rank 0.  With
``torch.distributed`` backend "nccl" this is RCCL over xGMI; the same code runs
under "gloo" on CPU tensors for the unit tests.
"""
import torch
import torch.distributed as dist

__all__ = ['shard_range', 'gather_shifts', 'PendingGather', 'xcorr_refine_sharded']


def shard_range(n, rank, world):
    """Half-open index range [lo, hi) of rank ``rank`` of ``world`` for ``n`` items
    (contiguous blocks, remainders spread over the first ranks)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PendingGather:
    """A gather of shift blocks that has been enqueued but not waited for (``gather_shifts(...,
    async_op=True)``).  ``result()`` waits (RCCL: the current stream waits, not the host) and returns
    what :func:`gather_shifts` returns.  Holds the buffers the collective works on until then."""

    def __init__(self, work, bufs, sizes, keep, local=None):
        self._work, self._bufs, self._sizes, self._keep, self._local = work, bufs, sizes, keep, local

    def result(self):
        if self._local is not None:               # single process: nothing was exchanged
            return self._local
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        if self._bufs is None:
            return None
        return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(self._bufs, self._sizes)], dim=0)


def gather_shifts(local, n_total=None, dst=0, group=None, async_op=False):
    """Gather per-rank shift blocks ``local [n_r, 2]`` onto ``dst`` in rank order.
    Returns the concatenated ``[n_total, 2]`` tensor on ``dst`` and None elsewhere.
    Blocks may differ in length by one (see :func:`shard_range`).

    ``async_op=True`` returns a :class:`PendingGather` instead: the collective runs on the
    backend's own stream while the caller enqueues the next batch's kernel; call ``result()``
    before the gathered shifts are needed (``bench.py`` keeps one gather in flight)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return PendingGather(None, None, None, None, local=local) if async_op else local
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
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    work = dist.gather(padded, gather_list=bufs, dst=dst, group=group, async_op=async_op)
    pending = PendingGather(work if async_op else None, bufs, sizes, padded)
    return pending if async_op else pending.result()


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
