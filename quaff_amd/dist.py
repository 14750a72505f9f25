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
    if world > 1 or "WORLD_SIZE" in os.environ:   # under a launcher: also a one-rank job has its process group (rehearsals)
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            # RCCL prints a version banner on stdout when its first communicator comes up; bench.py's stdout carries one JSON
            # line and nothing else, so the banner goes to stderr: fd 1 points at fd 2 until the first collective has run
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if backend == "nccl":
                    dist.init_process_group(backend, device_id=torch.device("cuda", local))
                    dist.barrier(device_ids=[local])
                else:
                    dist.init_process_group(backend)
                    dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
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
            dist.barrier(device_ids=[torch.cuda.current_device()])
            torch.cuda.synchronize()
        else:
            dist.barrier()


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


def attach_rccl(ctx):
    """Give the context (one per rank, one rank per GPU) the library's own RCCL communicator: rank 0's unique id travels
    over the process group that launched the ranks.  Not possible when several ranks share one GPU (the one-GPU rehearsal
    with gloo): RCCL wants a GPU per rank, and estep_allreduce then goes through torch.distributed instead."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    if dist.get_backend() != "nccl":
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init_rank(box[0], rank, world)
    return True


def rccl_report(ctx, single_device=False):
    """What the ranks of this job really were: every rank's (host, PCI bus id of its context's device) gathered over the process
    group, and -- under the nccl backend -- a library communicator (qf_comm_init_rank) brought up on the context and exercised with
    one all-reduce of a 1 per rank, so that the printed rank count is RCCL's own.  Returns a dict on every rank; raises on rank 0 if
    the ranks do not sit on `world` distinct devices (unless single_device: the one-GPU rehearsal)."""
    import socket
    import torch.distributed as dist
    rank, world, _ = env_rank()
    me = (socket.gethostname(), ctx.device_bus_id())
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks": 1, "distinct_devices": 1, "backend": None, "devices": ["%s/%s" % me]}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    rep = {"ranks": dist.get_world_size(), "distinct_devices": len(set(everyone)), "backend": dist.get_backend(),
           "devices": sorted("%s/%s" % d for d in set(everyone))}
    if dist.get_backend() == "nccl":
        if ctx.comm_size() <= 1 and world > 1:
            attach_rccl(ctx)
        if ctx.comm_size() > 0:
            ones, _ = ctx.allreduce_counts(np.ones(1), 0.0)
            rep["library_comm_size"] = ctx.comm_size()
            rep["library_allreduce_of_ones"] = float(ones[0])
    if rank == 0 and not single_device and rep["distinct_devices"] != rep["ranks"]:
        raise RuntimeError("%d ranks on %d distinct devices: %s" % (rep["ranks"], rep["distinct_devices"], rep["devices"]))
    return rep


def estep_allreduce(counts, loglike, ctx=None):
    """The train E-step exchange (QuaffCountingScheduler::finalCounts / finalLogLike, src/qmodel.cpp:2416-2422): returns
    (global counts, global log-likelihood).  Through the library's qf_allreduce_counts when the context carries a
    communicator (attach_rccl), else through the process group (gloo rehearsals on CPU / one GPU)."""
    if ctx is not None and ctx.comm_size() > 0:
        return ctx.allreduce_counts(counts, loglike)
    out = allreduce_sum(np.concatenate([np.asarray(counts, np.float64), [loglike]]))
    return out[:-1], float(out[-1])


def estep_allreduce_exact(counts_exact, loglike_exact, ctx=None):
    """The same exchange on the order-free 128-bit fixed-point words (qf_count_result.counts_exact / loglike_exact): returns
    (global counts, global log-likelihood, global words) with the doubles converted once from the exact totals -- identical on
    every rank and for every number of ranks.  RCCL through the library's qf_allreduce_counts_exact when the context carries a
    communicator; otherwise 32-bit limbs in int64 through the process group (gloo rehearsals)."""
    from . import api
    fx = np.concatenate([np.ascontiguousarray(counts_exact, np.uint64).reshape(-1, 2),
                         np.ascontiguousarray(loglike_exact, np.uint64).reshape(1, 2)])
    if ctx is not None and ctx.comm_size() > 0:
        tot = ctx.allreduce_counts_exact(fx)
    else:
        import torch.distributed as dist
        tot = fx
        if dist.is_available() and dist.is_initialized():
            import torch
            marker = (fx[:, 1] == np.uint64(1 << 63)) & (fx[:, 0] == 0)
            limbs = np.stack([fx[:, 0] & np.uint64(0xFFFFFFFF), fx[:, 0] >> np.uint64(32), fx[:, 1] & np.uint64(0xFFFFFFFF),
                              fx[:, 1] >> np.uint64(32), marker.astype(np.uint64)], axis=1)
            limbs[marker, :4] = 0
            t = torch.from_numpy(limbs.astype(np.int64)).to(_dev())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            L = t.cpu().numpy().astype(object)
            tot = np.zeros_like(fx)
            for k in range(len(fx)):
                if L[k, 4]:
                    tot[k] = (0, 1 << 63)
                else:
                    v = (int(L[k, 0]) + (int(L[k, 1]) << 32) + (int(L[k, 2]) << 64) + (int(L[k, 3]) << 96)) & ((1 << 128) - 1)
                    tot[k] = (v & ((1 << 64) - 1), v >> 64)
    vals = api.exact_to_double(tot)
    return vals[:-1], float(vals[-1]), tot


def balanced_blocks(weights, world):
    """Cut items 0..n-1 (rows of the overlap pair triangle, reads of a full-DP batch) into `world` contiguous blocks of
    nearly equal total weight: block r = [cuts[r], cuts[r+1]).  The reference balances dynamically (a shared task queue,
    src/qoverlap.cpp:528-547, src/qmodel.cpp:2870-2882); ranks that share nothing at run time balance by weight instead."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    targets = cum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(cum, targets, side="left")
    # the boundary item goes to whichever side leaves the blocks closer to the target
    for k, t in enumerate(targets):
        c = int(cuts[k])
        if c > 0 and abs(cum[c - 1] - t) <= abs(cum[min(c, len(w))] - t):
            cuts[k] = c - 1
    cuts = np.concatenate([[0], np.maximum.accumulate(cuts), [len(w)]]).astype(np.int64)
    return cuts


def overlap_row_plan(n_originals, n_seqs, rows, rank, world, block_pairs):
    """Rows [0, rows) of QuaffOverlapScheduler's enumeration (row nx = pairs (nx, ny), ny = nx + 1 ... n_seqs - 1;
    src/qoverlap.cpp:475-480) cut over `world` ranks into contiguous row ranges of equal pair count, and this rank's range into
    sub-blocks of about block_pairs pairs (the unit its contexts pull off their shared list, one qf_overlap_rows call each).
    Returns ((r0, r1), [(b0, b1), ...], pairs of the rank)."""
    assert 0 <= rows <= n_originals - 1
    row_len = (n_seqs - 1 - np.arange(rows)).astype(np.float64)
    cuts = balanced_blocks(row_len, world)
    r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
    blocks, b0, acc = [], r0, 0
    for r in range(r0, r1):
        acc += int(row_len[r])
        if acc >= max(1, block_pairs) or r == r1 - 1:
            blocks.append((b0, r + 1))
            b0, acc = r + 1, 0
    return (r0, r1), blocks, int(row_len[r0:r1].sum())


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
