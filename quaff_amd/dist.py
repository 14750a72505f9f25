"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) — gloo on CPU for
tests.  Read x reference pairs shard embarrassingly: align/overlap need no data-path collective; the
only exchange is the E-step reduction of `train` (QuaffCountingScheduler::finalCounts/finalLogLike,
src/qmodel.cpp:2416-2422): one all-reduce(sum, fp64) of the flattened counts + log-likelihood per EM iteration."""
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    rank, world, local = env_rank()
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend)
    return rank, world, local


def _dev():
    import torch
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def shard_range(n, rank, world):
    """Contiguous block of items [lo, hi) owned by `rank` (read ownership is fixed across EM iterations)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        import torch
        if dist.get_backend() == "nccl":
            torch.cuda.synchronize()
        dist.barrier()
        if dist.get_backend() == "nccl":
            torch.cuda.synchronize()


def allreduce_sum(vec):
    """Sum a float64 vector (flattened counts + log-likelihood) over all ranks; returns a numpy array."""
    import torch.distributed as dist
    v = np.ascontiguousarray(vec, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return v.copy()
    import torch
    t = torch.from_numpy(v.copy()).to(_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_max(x):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def estep_allreduce(counts, loglike):
    """The train E-step exchange: returns (global counts, global log-likelihood)."""
    out = allreduce_sum(np.concatenate([np.asarray(counts, np.float64), [loglike]]))
    return out[:-1], float(out[-1])


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
