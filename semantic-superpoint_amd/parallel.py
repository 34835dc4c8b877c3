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


class StepDiag:
    """Self-diagnosis of the overlapped data-parallel step (the first multi-GPU run has to explain itself: no 8-GPU node was ever
    available to the build).  Every `every`-th step is bracketed: `phase2_ms` = the backward of the two 240x320 layers (the window
    the early bucket's all-reduce hides under), `allreduce_exposed_ms` = what the compute stream waits for the two collectives
    AFTER phase 2 has finished (0 when both are hidden; the late 151 KB bucket is issued after phase 2 and is always exposed).
    CUDA events on the compute stream (`cuda=True`) or host clocks (`cuda=False`: gloo on the CPU, where wait() blocks the host)."""

    def __init__(self, every=4, cuda=True):
        self.every, self.cuda, self.n, self.rec = max(1, int(every)), cuda, 0, []

    def sampled(self):
        self.n += 1
        return (self.n - 1) % self.every == 0

    def mark(self):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def add(self, t_p1, t_p2, t_done):
        self.rec.append((t_p1, t_p2, t_done))

    def summary(self):
        """{"samples", "phase2_ms", "allreduce_exposed_ms"} (means over the bracketed steps); call after a device synchronisation."""
        if not self.rec:
            return {"samples": 0, "phase2_ms": None, "allreduce_exposed_ms": None}
        if self.cuda:
            p2 = [a.elapsed_time(b) for a, b, _ in self.rec]
            ex = [b.elapsed_time(c) for _, b, c in self.rec]
        else:
            p2 = [(b - a) * 1e3 for a, b, _ in self.rec]
            ex = [(c - b) * 1e3 for _, b, c in self.rec]
        return {"samples": len(self.rec), "phase2_ms": round(sum(p2) / len(p2), 4), "allreduce_exposed_ms": round(sum(ex) / len(ex), 4)}


# what the first 8-GPU run should show: the early bucket (6.39 MB) needs 0.15-0.25 ms on a ring over xGMI and has
# the whole of phase 2 to hide under; exposed per step: the late 151 KB bucket (one latency-bound all-reduce) and RCCL's CUs
DP_EXPECTED = {"allreduce_exposed_ms": "0.03-0.1 (the late 151 KB bucket; the early 6.4 MB bucket hides under phase 2)",
               # per-GPU batch 32 (the bench configuration), from the single-GPU step timelines (tools/timeline_step.sh): fp32 = weight
               # gradient 1.70 + data gradient 1.40 of the 64 -> 64 layer at 240x320 + first-layer kernels 0.41; phase 2 scales with
               # the per-rank batch: the two-ranks-on-one-device trace profiles/r04_dp_readiness.txt ran 16 pairs per rank (2.07 / 0.85)
               "phase2_ms": "batch 32 per GPU: fp32 ~3.5 of a 13.7 ms step, bf16 path ~1.5-1.7 of 7.2 ms (batch 16 per rank: 2.1 / 0.85)",
               "weak_scaling_efficiency_8gpu": "0.97-0.99",
               "rccl": "ring or tree, Simple / LL128 protocol for the 6.4 MB bucket; LL for the 151 KB bucket"}

_NCCL_ALGO = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree"}
_NCCL_PROTO = {0: "LL", 1: "LL128", 2: "Simple"}


def parse_nccl_log(text):
    """Algorithm / protocol choices from an NCCL_DEBUG=INFO (NCCL_DEBUG_SUBSYS=INIT,TUNING or COLL) log of rank 0:
    {"version": ..., "choices": {"<bytes> B": "<algo>/<proto> x <count>"}} - best effort, None when nothing is recognised."""
    import re
    if not text:
        return None
    out = {}
    m = re.search(r"(?:NCCL|RCCL) version ([0-9][^\s]*)", text)
    if m:
        out["version"] = m.group(1)
    ch = {}
    for m in re.finditer(r"(\d+) Bytes -> Algo (\d+) proto (\d+)", text):
        key = "%s B" % m.group(1)
        name = "%s/%s" % (_NCCL_ALGO.get(int(m.group(2)), m.group(2)), _NCCL_PROTO.get(int(m.group(3)), m.group(3)))
        ch.setdefault(key, {}).setdefault(name, 0)
        ch[key][name] += 1
    if ch:
        out["choices"] = {k: ", ".join("%s x %d" % kv for kv in sorted(v.items())) for k, v in ch.items()}
    rings = re.findall(r"Connected all (rings|trees)", text)
    if rings:
        out["connected"] = sorted(set(rings))
    return out or None


def dp_diagnostics(diag_summary, per_rank_ms, nccl_log_text=None):
    """The self-diagnosis block of an N > 1 bench line."""
    return {"allreduce_exposed_ms": diag_summary.get("allreduce_exposed_ms"), "phase2_ms": diag_summary.get("phase2_ms"),
            "bracketed_steps": diag_summary.get("samples"),
            "ms_per_step_ranks": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "all": [round(v, 3) for v in per_rank_ms]},
            "rccl": parse_nccl_log(nccl_log_text), "expected": DP_EXPECTED}


def pair_step_overlapped(eng, sample, lr, optimizer_step=True, diag=None, **step_kwargs):
    """One data-parallel pair step with the gradient all-reduce overlapped with the tail of the backward pass
    (SURVEY.md section 8e).  The flat gradient vector is split at `eng.early_offset`:

      phase 1 of the step (forward, losses, backward of the heads and encoder layers 7..2)
      all-reduce(SUM) of grads[early_offset:]  -- 97.7 % of the bytes; asynchronous: RCCL runs it on its own stream,
                                                  ordered after phase 1 by an event, while ...
      phase 2 (backward of the two 240x320 layers, ~40 % of the backward time) runs on the compute stream
      all-reduce(SUM) of grads[:early_offset]  -- 151 KB
      fused Adam on grads / world

    With world size 1 (or optimizer_step=False: a gradient-accumulation micro-batch) no collective is issued.
    diag: a StepDiag that brackets every n-th step (phase-2 time, exposed all-reduce time).
    Returns the device scalars of the step (no host synchronisation)."""
    w = world_size()
    if w == 1 or not optimizer_step:
        sc = eng.pair_step(sample, **step_kwargs)
        if optimizer_step:
            eng.adam_step(lr)
        return sc
    off = eng.early_offset
    rec = diag is not None and diag.sampled()
    sc = eng.pair_step(sample, phase=1, **step_kwargs)
    t1 = diag.mark() if rec else None
    w1 = dist.all_reduce(eng.grads[off:], op=dist.ReduceOp.SUM, async_op=True)
    eng.pair_step(sample, phase=2, **step_kwargs)
    t2 = diag.mark() if rec else None
    w2 = dist.all_reduce(eng.grads[:off], op=dist.ReduceOp.SUM, async_op=True)
    w1.wait()
    w2.wait()
    if rec:
        diag.add(t1, t2, diag.mark())
    eng.adam_step(lr, grad_scale=1.0 / w)
    return sc
