"""Data parallelism: one process per GPU (torchrun), replicas of the bucketed minibatch, ONE all-reduce of the flat
float32 gradient arena per step over RCCL/xGMI (backend "nccl" on ROCm), scaled by 1/world so that the per-replica
1/B_local cross-entropy means compose to the global 1/B mean.  All ranks seed Python's `random`
identically (nn.py:54), so the teacher-forcing flags agree; BatchNorm uses per-replica statistics.
CPU tests run the same code over gloo."""
import os

import torch
import torch.distributed as td


def is_distributed():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialises torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or td.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    td.init_process_group(backend=backend)


def allreduce_flat(buf):
    """In-place mean of one flat buffer over all ranks."""
    if not is_distributed():
        return buf
    td.all_reduce(buf, op=td.ReduceOp.SUM)
    buf.mul_(1.0 / world_size())
    return buf


def allreduce_grads(arena):
    return allreduce_flat(arena.grad)


def broadcast_params(arena, src=0):
    if is_distributed():
        td.broadcast(arena.data, src=src)


def shard_rows(n_rows, rank_, world):
    """Contiguous row shard [lo, hi) of a bucketed batch for `rank_` (equal T => equal work)."""
    per = n_rows // world
    return rank_ * per, (rank_ + 1) * per
