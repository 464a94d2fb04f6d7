"""Multi-GPU plumbing for the batched-crop path (BASELINE cfg #5): the crop list shards across
ranks (one process per GPU), each rank runs K1 into ITS slice of the full [N,C,H,W] tensor, and one all-gather
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" in the CPU tests) assembles the tensor on every
rank.  The reference has no multi-GPU code: this is new work defined by BASELINE.json."""


def shard_bounds(n_items, world, rank):
    """Contiguous block partition of n_items over `world` ranks; the remainder goes to the first ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0] for r in range(world)]


def gather_shards(full, n_items, dist, group=None):
    """All-gather the per-rank slices of `full` (first dimension = items) in place.

    Equal shards: one in-place all_gather_into_tensor (send buffer = this rank's slice of the receive buffer).
    Unequal shards (n_items % world != 0): all_gather with per-rank views of the same buffer."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return None
    sizes = shard_sizes(n_items, world)
    lo, hi = shard_bounds(n_items, world, rank)
    if len(set(sizes)) == 1:
        return dist.all_gather_into_tensor(full[:n_items], full[lo:hi], group=group)
    outs, pos = [], 0
    for s in sizes:
        outs.append(full[pos:pos + s])
        pos += s
    # all_gather needs equally shaped tensors: pad every shard to the largest and trim afterwards
    m = max(sizes)
    if any(s != m for s in sizes):
        import torch
        send = torch.zeros((m,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        send[:hi - lo] = full[lo:hi]
        recv = [torch.empty_like(send) for _ in range(world)]
        work = dist.all_gather(recv, send, group=group)
        for r, s in enumerate(sizes):
            outs[r].copy_(recv[r][:s])
        return work
    return dist.all_gather(outs, full[lo:hi], group=group)
