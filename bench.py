#!/usr/bin/env python3
"""bench.py - image-pairs/s of the Semantic-SuperPoint pair training step on N MI355X (one process per GPU).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment, RCCL backend).  A step is one
full pair-training step on one batch of synthetic pairs already resident in HBM: 2 forwards (separate BatchNorm
statistics), label ops, detector / sparse-descriptor (/ segmentation) losses with on-device index sampling,
multi-task loss, backward, gradient all-reduce (N > 1), Adam.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1]: SuperPointNet_gauss2, 240x320, batch 32 per GPU, fp32 (use --arch ssp for
configs[2]).  `roofline`: conv_wino_pipe_kernel (all 3x3 forward + data-gradient launches, 2/3 of the step's FLOPs, run as
Winograd F(2x2,3x3) in fp32), algorithmic FLOPs / HIP-event time measured live on the launch stream.  `cpu_baseline`: the oracle
(oracle/cpu_ref.py, a restatement pinned against the reference) timed on this host's cores, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PAIR = {"SuperPointNet_gauss2": 77.98, "SuperPointNet_gauss2_ssmall": 82.72}
PEAK_FP32_MFMA_TF = 157.3  # MI355X_MICROARCH.md
PEAK_BF16_MFMA_TF = 2500.0  # dense bf16 (only used for the opt-in --conv-algo 3 line)


def cpu_baseline(arch, H, W, batch=8, steps=1):
    """Oracle pair step on the host cores (bounded sample).  PyTorch's CPU kernels stop scaling (and the
    oracle's Python loops thrash) far below the 256 hardware threads of the GPU host, so the baseline uses
    min(cores, 32) threads - measured faster than 256 - and reports that number as `cores`."""
    import torch
    from oracle import cpu_ref as C
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    semantic = arch.endswith("ssmall")
    sample = C.make_synthetic_pair(batch, H, W, seed=1, semantic=semantic)
    tr = C.Trainer(arch, C.init_state_dict(arch, seed=0), lr=0.001)
    tr.train_val_sample(sample, n_iter=0, train=True)  # warm-up
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_val_sample(sample, n_iter=i + 1, train=True)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch / dt, 4), "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle/cpu_ref.py Trainer, %s %dx%d batch %d, 1 warm-up + %d timed steps (%.2f s/step)"
                      % (arch, H, W, batch, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="sp", choices=["sp", "ssp"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--conv-algo", type=int, default=1, choices=[0, 1, 2, 3, 5],
                    help="ssp_set_conv_algo: 1 = fp32 Winograd (default, the headline), 0 = fp32 direct, 2 = fp32 Winograd "
                         "un-pipelined, 5 = fp32 Winograd pipelined with LDS-staged weights, 3 = Winograd with bf16 matrix-core operands (reduced precision: reported as dtype bf16)")
    ap.add_argument("--desc-loss", default="sparse", choices=["sparse", "dense"],
                    help="descriptor loss of the step: sparse (shipped configs, the headline) or dense (model.dense_loss)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # debugging aid for 1-GPU boxes: SSP_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses
    # two ranks on one device); the measured configuration is always one rank per GPU over RCCL ("nccl").
    single_dev = os.environ.get("SSP_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import semantic_superpoint_amd as ssp
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import Engine, layer_table, SCALAR_NAMES

    arch = "SuperPointNet_gauss2" if args.arch == "sp" else "SuperPointNet_gauss2_ssmall"
    B, H, W = args.batch, args.height, args.width
    ssp.lib.set_conv_algo(args.conv_algo)
    dense = {"descriptor_dist": 4, "lambda_d": 800} if args.desc_loss == "dense" else None
    eng = Engine(arch, B, H, W, dev, dense_loss=dense is not None)
    eng.load_state_dict(synth.default_init_state_dict(layer_table(arch), seed=0))  # identical replicas
    sample = synth.make_pair(B, H, W, dev, seed=100 + rank, semantic=arch.endswith("ssmall"))
    torch.cuda.synchronize()

    def step(it):
        eng.zero_grad()
        eng.pair_step(sample, indices=None, seed=(it * 1000003 + rank * 7919 + 1), train=True, lambda_loss=1.0,
                      lamda_d=1.0, multi_task=True, dense=dense)
        if world > 1:  # data parallel: one all-reduce of the flat fp32 gradient bucket (incl. eta), then mean
            dist.all_reduce(eng.grads)
            eng.grads.div_(world)
        eng.adam_step(args.lr)

    for it in range(args.warmup):
        step(it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.no_roofline and rank == 0:
        eng.profile_enable("conv3x3_all")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(args.warmup + it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        scal = dict(zip(SCALAR_NAMES, eng.scalars.cpu().tolist()))
        pairs_s = world * B * args.steps / dt
        out = {"metric": "image-pairs/sec at %dx%d bs%d (pair training step)" % (H, W, B), "value": round(pairs_s, 2),
               "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if args.conv_algo == 3 else "f32", "data": "synthetic",
               "config": {"workload": "%s pair step %dx%d, batch %d per GPU, %s, %s, Adam"
                                      % (arch, H, W, B, "fp32" if args.conv_algo != 3 else
                                         "bf16 matrix-core operands / fp32 accumulate + master (NOT the headline precision)",
                                         "sparse loss 1000x100" if dense is None else
                                         "dense descriptor loss (1200x1200 per image)"), "parallelism": "dp%d" % world,
                          "global_batch": world * B},
               "step_tflops": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3, 2),
               "step_frac_of_fp32_mfma_peak": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3 / (PEAK_FP32_MFMA_TF * world), 4),
               "final_loss": round(scal["loss"], 4)}
        if not args.no_roofline:
            pr = eng.profile_read()
            if pr["launches"] > 0 and pr["ms"] > 0:
                ach = pr["flops"] / (pr["ms"] * 1e-3) / 1e12
                traffic = None  # HBM bytes / launch from the committed rocprofv3 PMC passes (cannot be read live)
                tpath = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")
                if args.arch == "sp" and (B, H, W) == (32, 240, 320) and args.conv_algo == 1 and os.path.exists(tpath):
                    tj = json.load(open(tpath))  # only valid for the launch structure it was profiled with
                    if abs(tj.get("flops_per_launch_avg_gflop", 0) - pr["flops"] / pr["launches"] / 1e9) < 0.05 * tj.get(
                            "flops_per_launch_avg_gflop", 1):
                        traffic = round(tj["hbm_bytes_per_launch"])
                peak = PEAK_BF16_MFMA_TF if args.conv_algo == 3 else PEAK_FP32_MFMA_TF
                out["roofline"] = {"bound": "mfma", "kernel": "conv_wino_pipe_kernel (3x3 forward + data-gradient, Winograd "
                                                              "F(2x2,3x3) on v_mfma_f32_32x32x2_f32)",
                                   "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4),
                                   "note": "achieved = ALGORITHMIC (direct-convolution) FLOPs / time; Winograd executes "
                                           "16/36 of them on the matrix cores, so frac may exceed 1",
                                   "executed_tflops": round(ach * 16.0 / 36.0, 2),
                                   "executed_frac": round(ach * 16.0 / 36.0 / peak, 4), "traffic": traffic,
                                   "algorithmic_bytes_per_launch": round(pr["bytes"] / pr["launches"]),
                                   "launches": pr["launches"], "avg_launch_ms": round(pr["ms"] / pr["launches"], 4),
                                   "flops_per_launch_avg": round(pr["flops"] / pr["launches"] / 1e9, 3)}
            eng.profile_enable("none")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arch, H, W)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
