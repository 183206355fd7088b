"""Multi-GPU plumbing: frames shard across ranks with no data-path collective;
the one exchange step is the gather of the encoded streams.
Backend-agnostic (RCCL via "nccl" on GPUs, "gloo" in the CPU tests)."""


def shard_frames(total, rank, world):
    """Contiguous block of frame indices [first, first+count) owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_streams(streams, lens, dst=0, group=None, async_op=False):
    """Gather variable-length byte streams to rank `dst`.

    streams: uint8 tensor [n, stride] (row i holds lens[i] valid bytes), lens: int64 [n],
    same n on every rank.  Returns (list of per-rank uint8 tensors [n, width], int64
    tensor [world*n]) on dst and (None, lens_all) elsewhere.  Two collectives: an
    all_gather of the lengths, then one gather of rows cut to the longest stream
    (rounded up to 8 bytes).  With async_op the gather runs on the collective's own stream
    and (bufs, all_lens, work, send_buffer) is returned."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = lens.numel()
    all_lens = torch.empty((world * n,), dtype=torch.int64, device=lens.device)
    dist.all_gather_into_tensor(all_lens, lens.contiguous(), group=group)
    width = min(streams.shape[1], (int(all_lens.max().item()) + 7) // 8 * 8)
    mine = streams[:, :width].contiguous()
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    work = dist.gather(mine, bufs, dst=dst, group=group, async_op=async_op)
    if async_op:
        # the caller overlaps the transfer with its own decode and calls work.wait() at the end of the step
        return bufs, all_lens, work, mine
    return bufs, all_lens
