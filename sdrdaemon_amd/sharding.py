"""Stream sharding across GPUs: streams are independent, stream s lives on rank
s // streams_per_gpu of a one-process-per-GPU job; no data-path collective.  The only
collectives are the reporting reductions below (a few bytes)."""


def stream_ids(rank, world_size, streams_per_gpu):
    """Global stream ids owned by `rank` (weak scaling: every rank owns streams_per_gpu)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))


def stream_ids_strong(rank, world_size, total_streams):
    """Strong scaling: a fixed set of streams, stream s on rank s mod G."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, total_streams, world_size))


def owner(stream_id, streams_per_gpu):
    return stream_id // streams_per_gpu


def aggregate(elapsed_s, samples, dist=None, device=None, force=False):
    """-> (max elapsed over ranks, total samples over ranks).  dist = torch.distributed or None.  force: run the two
    reductions on a world of one as well (the RCCL smoke test of the N = 1 box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(elapsed_s), float(samples)
    import torch

    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    n = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())
