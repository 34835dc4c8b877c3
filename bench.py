#!/usr/bin/env python3
"""bench.py - image-pairs/s of the Semantic-SuperPoint pair training step on N MI355X (one process per GPU).

Contract: `python bench.py --gpus N --steps K --warmup W`.
  * launched under torch.distributed.run (WORLD_SIZE set): this process is one rank; WORLD_SIZE must equal --gpus;
  * launched bare with --gpus N > 1: the parent - BEFORE anything touches the GPU - starts N fresh child processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relays rank 0's JSON line and exits non-zero if
    any child fails (it never re-executes itself: children are ordinary subprocesses);
  * N = 1 runs in-process, so `rocprofv3 ... -- python3 bench.py` has no hop between the profiler and the kernels.

Workload = BASELINE.json's north star, configs[2]: SuperPointNet_gauss2_ssmall (Semantic-SuperPoint: encoder + detector
+ descriptor + segmentation heads, uncertainty-weighted multi-task loss), 240x320, batch 32 per GPU, fp32
(`--arch sp` = configs[1]).  A step is one full pair-training step on one batch of synthetic pairs already resident in
HBM: 2 forwards (separate BatchNorm statistics), label ops, detector / sparse-descriptor / segmentation losses with
on-device index sampling, multi-task loss, backward, gradient all-reduce (N > 1, overlapped with the tail of the
backward pass), Adam.  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel family conv_wino_pipe_kernel (all 3x3 forward + data-gradient launches, 2/3 of the
step's FLOPs, Winograd F(2x2,3x3) in fp32): algorithmic FLOPs / HIP-event time measured live on the launch stream.
`roofline.traffic`: HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE)
collected in THIS run by two short profiled child runs of the same command (`--traffic live`, the default at N = 1 when
rocprofv3 is on PATH; null if that fails).  `cpu_baseline`: the oracle (oracle/cpu_ref.py, a restatement pinned against
the reference) timed on this host's cores, rank 0, N = 1 only: batch 32, 1 warm-up + 3 timed steps.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PAIR = {"SuperPointNet_gauss2": 77.98, "SuperPointNet_gauss2_ssmall": 82.72}
PEAK_FP32_MFMA_TF = 157.3  # MI355X_MICROARCH.md
PEAK_BF16_MFMA_TF = 2500.0  # dense bf16 (only used for the opt-in --conv-algo 3 line)
# kernels of the 3x3 forward + data-gradient launches per --conv-algo (algo 1 runs the small 30x40 maps on the p2 kernel)
DOMINANT_KERNEL = {0: ("conv_mfma_kernel<3",), 1: ("conv_wino4_kernel", "conv_wino_pipe_kernel", "conv_wino_p2_kernel"), 2: ("conv_wino_kernel",),
                   3: ("conv_wino_bf16_kernel",), 5: ("conv_wino_pipe_kernel",), 6: ("conv_wino_p2_kernel",),
                   7: ("conv_wino_bf16_kernel",), 8: ("conv_wino_bf16_kernel",),
                   9: ("conv_wino_pipe_kernel", "conv_wino_p2_kernel"), 10: ("conv_wino4_kernel", "conv_wino_p2_kernel")}


def cpu_baseline(arch, H, W, batch=32, steps=3):
    """Oracle pair step on the host cores (bounded sample: 1 warm-up + `steps` timed steps at the benchmark batch).
    PyTorch's CPU kernels stop scaling (and the oracle's Python loops thrash) far below the 256 hardware threads of the
    GPU host, so the baseline uses min(cores, 32) threads - measured faster than 256 - and reports that as `cores`."""
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="ssp", choices=["sp", "ssp"],
                    help="ssp = SuperPointNet_gauss2_ssmall (north star, configs[2]); sp = SuperPointNet_gauss2 (configs[1])")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--conv-algo", type=int, default=1, choices=[0, 1, 2, 3, 5, 6, 7, 8, 9, 10],
                    help="ssp_set_conv_algo: 1 = fp32 Winograd (default, the headline), 0 = fp32 direct, 2 = fp32 Winograd "
                         "un-pipelined, 5 = fp32 Winograd pipelined with LDS-staged weights, 3 = Winograd with bf16 "
                         "matrix-core operands (reduced precision: reported as dtype bf16), 6 = fp32 Winograd, two 4-wave "
                         "workgroups per CU, 7 = Winograd with split-bf16 (hi + lo, 16 significant bits) operands: reported "
                         "as dtype bf16x2")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="f32 = --conv-algo 1 (the headline); bf16 = --conv-algo 8, the validated mixed bf16 mode of BASELINE "
                         "configs[3] (forward convs split-bf16, backward convs bf16, everything else fp32)")
    ap.add_argument("--desc-loss", default="sparse", choices=["sparse", "dense"],
                    help="descriptor loss of the step: sparse (shipped configs, the headline) or dense (model.dense_loss)")
    ap.add_argument("--graph", action="store_true", help="replay the pair step as a hipGraph (ssp_pair_step_graph)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: one blocking all-reduce after the whole backward instead of the overlapped split bucket")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "none"],
                    help="roofline.traffic: live = two rocprofv3 --pmc child runs of this command (FETCH_SIZE, WRITE_SIZE) "
                         "before the timed run; auto = live when N = 1, rocprofv3 is on PATH and the roofline leg is on")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the profiled child of --traffic live
    args = ap.parse_args(argv)
    if args.dtype is not None:
        args.conv_algo = {"f32": 1, "bf16": 8}[args.dtype]
    return args


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (the parent never touches the GPU)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# roofline.traffic measured in this run: rocprofv3 PMC passes over a short child run of the same workload
# ------------------------------------------------------------------------------------------------
def live_traffic(args):
    """HBM bytes per launch of the dominant kernel: (2 x FETCH_SIZE + WRITE_SIZE) KB -> bytes, separate PMC passes, FETCH
    doubled (gfx950 counts wide coalesced reads at half: MI355X_MICROARCH.md section HBM).  Returns (bytes or None, note)."""
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 not on PATH"
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "1", "--gpus", "1",
             "--arch", args.arch, "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
             "--conv-algo", str(args.conv_algo), "--desc-loss", args.desc_loss, "--no-cpu-baseline", "--no-roofline",
             "--traffic", "none"]
    tot, launches = {}, {}
    tmp = tempfile.mkdtemp(prefix="ssp_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out" % ctr
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "no counter_collection.csv from rocprofv3"
            s, seen = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if any(k in row["Kernel_Name"] for k in DOMINANT_KERNEL[args.conv_algo]) and row["Counter_Name"] == ctr:
                    s += float(row["Counter_Value"])
                    seen.add(row["Dispatch_Id"])
            tot[ctr], launches[ctr] = s, len(seen)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not launches.get("FETCH_SIZE") or launches["FETCH_SIZE"] != launches.get("WRITE_SIZE"):
        return None, "PMC passes saw different launch counts: %s" % launches
    n = launches["FETCH_SIZE"]
    hbm = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / n
    return hbm, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of a 1+1-step child run in this job, %d launches); "
                 "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches" % n)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))  # nothing below has run: the parent never initialises the GPU
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    traffic, traffic_note = None, "not collected"
    want_live = args.traffic == "live" or (args.traffic == "auto" and not args.no_roofline and not args.pmc_child)
    if want_live and world == 1:
        traffic, traffic_note = live_traffic(args)  # child processes; this process has not touched the GPU yet

    import torch
    import torch.distributed as dist

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
    from semantic_superpoint_amd import parallel, synth
    from semantic_superpoint_amd.lib import Engine, layer_table, SCALAR_NAMES

    arch = "SuperPointNet_gauss2" if args.arch == "sp" else "SuperPointNet_gauss2_ssmall"
    B, H, W = args.batch, args.height, args.width
    ssp.lib.set_conv_algo(args.conv_algo)
    dense = {"descriptor_dist": 4, "lambda_d": 800} if args.desc_loss == "dense" else None
    eng = Engine(arch, B, H, W, dev, dense_loss=dense is not None)
    eng.load_state_dict(synth.default_init_state_dict(layer_table(arch), seed=0))  # identical replicas
    sample = synth.make_pair(B, H, W, dev, seed=100 + rank, semantic=arch.endswith("ssmall"))
    torch.cuda.synchronize()
    rccl_ranks = 1
    if world > 1:  # prove the collective spans all ranks before timing anything
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        rccl_ranks = int(round(float(t.item())))
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    stream = torch.cuda.Stream(device=dev) if args.graph else None  # stream capture needs a non-default stream

    def step(it):
        kw = dict(indices=None, seed=(it * 1000003 + rank * 7919 + 1), train=True, lambda_loss=1.0, lamda_d=1.0,
                  multi_task=True, dense=dense, graph=args.graph)
        eng.zero_grad()
        if world > 1 and not args.no_overlap:
            parallel.pair_step_overlapped(eng, sample, args.lr, **kw)
        else:
            eng.pair_step(sample, **kw)
            if world > 1:  # one blocking all-reduce of the flat fp32 gradient bucket (incl. eta), then the mean
                dist.all_reduce(eng.grads)
                eng.adam_step(args.lr, grad_scale=1.0 / world)
            else:
                eng.adam_step(args.lr)

    def run(n, first):
        if stream is None:
            for it in range(n):
                step(first + it)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                for it in range(n):
                    step(first + it)
            torch.cuda.current_stream().wait_stream(stream)

    run(args.warmup, 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    profiled = not args.no_roofline and rank == 0 and not args.graph  # hipEvents cannot be recorded into a capture
    if profiled:
        eng.profile_enable("conv3x3_all")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
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
        reduced = args.conv_algo in (3, 7, 8)
        prec = {3: ("bf16", "bf16 matrix-core operands / fp32 accumulate + master (NOT the headline precision)"),
                7: ("bf16x2", "split-bf16 (hi + lo = 16 significant bits) matrix-core operands, three bf16 MFMAs per product / "
                              "fp32 accumulate + master (NOT the headline precision)"),
                8: ("bf16", "mixed bf16: forward convolutions with split-bf16 (hi + lo) operands, data / weight gradients with "
                            "bf16 operands; fp32 accumulate, BatchNorm, master weights, Adam (NOT the headline precision)")
                }.get(args.conv_algo, ("f32", "fp32"))
        out = {"metric": "image-pairs/sec at %dx%d bs%d (pair training step)" % (H, W, B), "value": round(pairs_s, 2),
               "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": prec[0], "data": "synthetic",
               "config": {"workload": "%s pair step %dx%d, batch %d per GPU, %s, %s, Adam%s"
                                      % (arch, H, W, B, prec[1],
                                         "sparse loss 1000x100" if dense is None else
                                         "dense descriptor loss (1200x1200 per image)",
                                         ", hipGraph replay" if args.graph else ""),
                          "parallelism": "dp%d" % world, "global_batch": world * B,
                          "allreduce": ("none" if world == 1 else "one bucket after backward" if args.no_overlap else
                                        "split bucket, early part overlapped with the backward of the 240x320 layers")},
               "rccl_ranks": rccl_ranks,
               "step_tflops": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3, 2),
               "step_frac_of_fp32_mfma_peak": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3 / (PEAK_FP32_MFMA_TF * world), 4),
               "final_loss": round(scal["loss"], 4)}
        if profiled:
            pr = eng.profile_read()
            if pr["launches"] > 0 and pr["ms"] > 0:
                ach = pr["flops"] / (pr["ms"] * 1e-3) / 1e12
                peak = PEAK_BF16_MFMA_TF if reduced else PEAK_FP32_MFMA_TF
                # Winograd executes 16 of 36 multiplies; the split-bf16 mode three bf16 MFMAs per product block
                # (the library counts the multiplies each launch executes: 1/4 of the algorithmic ones for F(4x4,3x3), 16/36 for
                # F(2x2,3x3))
                exec_ratio = {7: 3.0, 8: 2.0}.get(args.conv_algo, 1.0) * pr["exec_flops"] / pr["flops"]
                out["roofline"] = {"bound": "mfma", "kernel": "%s (3x3 forward + data-gradient, %s on v_mfma_f32_32x32x2_f32)"
                                                              % (" + ".join(DOMINANT_KERNEL[args.conv_algo]), "direct implicit GEMM"
                                                                 if args.conv_algo == 0 else "Winograd F(4x4,3x3) on the "
                                                                 "240x320 / 120x160 maps, F(2x2,3x3) below" if args.conv_algo == 1
                                                                 else "Winograd F(4x4,3x3)" if args.conv_algo == 10
                                                                 else "Winograd F(2x2,3x3)"),
                                   "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4),
                                   "note": "achieved = ALGORITHMIC (direct-convolution) FLOPs / time; Winograd executes "
                                           "16/36 (F(2x2,3x3)) or 1/4 (F(4x4,3x3)) of them on the matrix cores, so frac may "
                                           "exceed 1: executed_frac is the hardware fraction of the matrix-core peak",
                                   "executed_tflops": round(ach * exec_ratio, 2),
                                   "executed_frac": round(ach * exec_ratio / peak, 4),
                                   "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_note,
                                   "algorithmic_bytes_per_launch": round(pr["bytes"] / pr["launches"]),
                                   "launches": pr["launches"], "avg_launch_ms": round(pr["ms"] / pr["launches"], 4),
                                   "flops_per_launch_avg": round(pr["flops"] / pr["launches"] / 1e9, 3)}
            if world == 1 and args.conv_algo in (0, 1, 2, 5, 6, 9, 10):
                # second family, outside the timed region: the 3x3 weight-gradient launches (3 extra steps)
                eng.profile_enable("conv3x3_wgrad")
                run(3, args.warmup + args.steps)
                torch.cuda.synchronize()
                pw = eng.profile_read()
                if pw["launches"] > 0 and pw["ms"] > 0:
                    achw = pw["flops"] / (pw["ms"] * 1e-3) / 1e12
                    rw = 1.0 if args.conv_algo == 0 else 16.0 / 36.0
                    out["roofline_wgrad"] = {"bound": "mfma", "kernel": "wgrad_mfma_kernel" if args.conv_algo == 0 else
                                             "wgrad_wino_kernel (3x3 weight gradient, Winograd F(3x3,2x2))",
                                             "achieved": round(achw, 2), "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s",
                                             "frac": round(achw / PEAK_FP32_MFMA_TF, 4),
                                             "executed_tflops": round(achw * rw, 2),
                                             "executed_frac": round(achw * rw / PEAK_FP32_MFMA_TF, 4),
                                             "launches": pw["launches"], "avg_launch_ms": round(pw["ms"] / pw["launches"], 4),
                                             "note": "measured over 3 extra steps after the timed region"}
            eng.profile_enable("none")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arch, H, W, batch=B)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
