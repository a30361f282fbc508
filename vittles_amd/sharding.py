"""Row sharding across GPUs (one process per GPU, torch.distributed over RCCL/xGMI).

The path shards by data rows: rank r owns the contiguous rows
[r*N/R, (r+1)*N/R) of (X, y, w); theta-hat, H and its Cholesky factor are replicated; the
sensitivities S stay sharded like X.  The ONLY data-path collectives are sum all-reduces of
the D x D partial Hessian and the D-vector partial gradient / HVP.  The prior (which is not
a sum over rows) is contributed by exactly one rank.
"""
import torch


def shard_rows(n_rows, rank, world):
    """Contiguous row range [start, stop) of rank ``rank`` out of ``world``."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    start = (n_rows * rank) // world
    stop = (n_rows * (rank + 1)) // world
    return start, stop


def group_rank(group):
    if group is None:
        return 0
    import torch.distributed as dist
    return dist.get_rank(group)


def group_size(group):
    if group is None:
        return 1
    import torch.distributed as dist
    return dist.get_world_size(group)


def contributes_prior(group):
    """True on the single rank that adds the (unsharded) prior terms."""
    return group_rank(group) == 0


def all_reduce_sum(tensors, group):
    """In-place sum all-reduce of one tensor or a list of tensors (RCCL for CUDA tensors with
    the nccl backend, gloo for CPU tensors).  Several small tensors are flattened into ONE
    message: the collective is latency-bound (80 KB for H at D = 100)."""
    if group is None:
        return tensors
    import torch.distributed as dist
    single = isinstance(tensors, torch.Tensor)
    ts = [tensors] if single else list(tensors)
    # a CPU-only backend (gloo: tests, one-GPU rehearsals of the multi-rank path) gets device
    # tensors staged through the host; RCCL reduces them in place over xGMI
    staged = ts[0].is_cuda and dist.get_backend(group) != "nccl"
    if len(ts) == 1 and not staged:
        dist.all_reduce(ts[0], op=dist.ReduceOp.SUM, group=group)
    else:
        flat = torch.cat([t.reshape(-1) for t in ts])
        if staged:
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat = host.to(ts[0].device)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return tensors


def agree_scalars(values, group, device):
    """Return rank 0's copy of a list of Python floats on every rank of ``group``.

    Host-side control flow that depends on reduced quantities (an optimiser's stopping and
    line-search tests) must take the same branch on every rank, or the ranks' collectives stop
    matching.  All-reduce results are normally bit-identical across ranks; this makes the
    agreement explicit at the cost of one tiny broadcast."""
    if group is None:
        return list(values)
    import torch.distributed as dist
    staged = dist.get_backend(group) != "nccl"
    t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if staged else device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
    return [float(v) for v in t.tolist()]


def all_reduce_max(values, group, device):
    """Maximum over the ranks of a list of Python floats (status codes, norms): every rank gets the result."""
    if group is None:
        return list(values)
    import torch.distributed as dist
    staged = dist.get_backend(group) != "nccl"
    t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if staged else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.tolist()]
