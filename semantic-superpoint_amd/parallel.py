"""Data parallelism: one process per GPU, replicated parameters, one all-reduce (RCCL over xGMI when the backend
is "nccl"; "gloo" in the CPU tests) of the flat fp32 gradient bucket per optimizer step, then the mean.
The reference has no distributed layer; BatchNorm statistics stay per replica, which is
the semantics of its unsynchronised nn.BatchNorm2d at per-device batch size."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat):
    """In-place mean over ranks of one flat bucket (gradients incl. the 3 eta entries). No-op for world 1."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(w)
    return flat


def broadcast_(flat, src=0):
    """Make replicas identical (parameters, BN running statistics) at start-up."""
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def shard_batch(n_items, rank=None, world=None):
    """Contiguous shard [lo, hi) of `n_items` independent units (image pairs / export images) for this rank."""
    world = world or world_size()
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pair_step_overlapped(eng, sample, lr, optimizer_step=True, **step_kwargs):
    """One data-parallel pair step with the gradient all-reduce overlapped with the tail of the backward pass
    (SURVEY.md section 8e).  The flat gradient vector is split at `eng.early_offset`:

      phase 1 of the step (forward, losses, backward of the heads and encoder layers 7..2)
      all-reduce(SUM) of grads[early_offset:]  -- 97.7 % of the bytes; asynchronous: RCCL runs it on its own stream,
                                                  ordered after phase 1 by an event, while ...
      phase 2 (backward of the two 240x320 layers, ~40 % of the backward time) runs on the compute stream
      all-reduce(SUM) of grads[:early_offset]  -- 151 KB
      fused Adam on grads / world

    With world size 1 (or optimizer_step=False: a gradient-accumulation micro-batch) no collective is issued.
    Returns the device scalars of the step (no host synchronisation)."""
    w = world_size()
    if w == 1 or not optimizer_step:
        sc = eng.pair_step(sample, **step_kwargs)
        if optimizer_step:
            eng.adam_step(lr)
        return sc
    off = eng.early_offset
    sc = eng.pair_step(sample, phase=1, **step_kwargs)
    w1 = dist.all_reduce(eng.grads[off:], op=dist.ReduceOp.SUM, async_op=True)
    eng.pair_step(sample, phase=2, **step_kwargs)
    w2 = dist.all_reduce(eng.grads[:off], op=dist.ReduceOp.SUM, async_op=True)
    w1.wait()
    w2.wait()
    eng.adam_step(lr, grad_scale=1.0 / w)
    return sc
