"""Multi-GPU layout of the batch engine: streams are independent, so a node runs
one process per GPU, each owning a CONTIGUOUS block of stream indices; the data path has no
collective.  The only cross-rank traffic is bookkeeping (a barrier and a max-reduce of the elapsed
time in bench.py, a gather of per-stream byte outputs if the caller wants them in one place)."""


def stream_shard(n_streams, rank, world_size):
    """(first_stream, count) of `rank`: contiguous blocks, sizes differing by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(n_streams, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def all_shards(n_streams, world_size):
    return [stream_shard(n_streams, r, world_size) for r in range(world_size)]


def max_over_ranks(value, dist=None, device=None):
    """Elapsed-time reduction the bench contract asks for: MAX over ranks (identity without a group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stream_outputs(local_outputs, dist=None):
    """Concatenate per-stream outputs of all ranks in stream order (rank order == stream order)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_outputs)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(local_outputs))
    out = []
    for p in parts:
        out.extend(p)
    return out
