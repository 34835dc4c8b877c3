#!/usr/bin/env python3
"""bench_export.py - images/s of the homography-adaptation export (BASELINE.json
configs[4]) on N MI355X, one process per GPU, images sharded over ranks with no collective ("weak" scaling).

A step = TWO images (one ssp_export_points call): for each image n_views warped copies + valid masks are produced on
the device from the resident image (datasets/Coco.py:258-292), the detector head runs over them as one train-mode
BatchNorm batch, softmax -> depth-to-space, masked un-warp accumulation (export.py:49-60), greedy NMS, soft-argmax
refinement and top-k (models/model_wrap.py:129-293) -- the body of export.py:274-318 without the file I/O.  The point
lists stay on the device; one host read of the two counts closes the step (the reference reads the whole heatmap).

`python bench_export.py [--gpus N] [--steps K] [--warmup W] [--height 240 --width 320 --views 100]` prints ONE JSON
line on rank 0 (same fields as bench.py).  bench.py remains the headline (pair training step) benchmark.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PEAK_FP32_MFMA_TF = 157.3
GFLOP_FWD_DET_240x320 = 12.1609  # encoder + convPa + convPb of one 240x320 view


def cpu_baseline(arch, H, W, views, thr):
    """The oracle's export_points on ONE image on the host cores (bounded sample)."""
    import numpy as np
    import torch
    from oracle import cpu_ref as C
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    rs = np.random.RandomState(3)
    v = min(views, 25)  # bounded: the forward dominates and is linear in the number of views
    img = torch.from_numpy(rs.uniform(0, 1, (H, W)).astype(np.float32))
    t0 = time.perf_counter()
    sample = C.homo_adapt_sample(img, v, rs)
    sd = C.to_torch(C.init_state_dict(arch, seed=0))
    C.export_points(sd, sample, arch, conf_thresh=np.float32(thr), nms_dist=4, top_k=600, subpixel=True)
    dt = (time.perf_counter() - t0) * views / v
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle/cpu_ref.py homo_adapt_sample + export_points, %s %dx%d, %d of %d views timed and scaled "
                      "linearly (%.1f s/image)" % (arch, H, W, v, views, dt)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default="sp", choices=["sp", "ssp"])
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--height", type=int, default=480)  # BASELINE.json configs[4]: 480x640, 100 homographies per image
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--thresh", type=float, default=0.0155)  # random-init logits: softmax ~ 1/65 = 0.01538
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also time the stages separately (extra, untimed pass)")
    ap.add_argument("--images-per-step", type=int, default=8,
                    help="images per timed step (BASELINE configs[4]: batch 64 over 8 GPUs = 8 per GPU; the library call takes 2)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "none"], help="PMC child passes (FETCH_SIZE, WRITE_SIZE) for roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


EXPORT_PMC_KERNELS = ("conv_wino4_kernel", "conv_wino_pipe_kernel", "conv_wino_p2_kernel", "conv0_direct_kernel", "conv1x1_group_kernel")


def export_pmc(arch_key, views, H, W):
    """HBM bytes per launch of the export's convolution kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a
    1 + 1-step child run of this script (one call, two images), collected and corrected by bench.live_pmc.  {} on failure."""
    import bench
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "1", "--gpus", "1", "--arch", arch_key,
             "--views", str(views), "--height", str(H), "--width", str(W), "--no-cpu-baseline", "--traffic", "none"]
    try:
        pmc, _ = bench.live_pmc(None, counter_sets=(("FETCH_SIZE",), ("WRITE_SIZE",)), child=child, kernels=EXPORT_PMC_KERNELS)
    except Exception:
        return {}
    return pmc


def measure_export(dev, arch, n, H, W, thresh, steps, warmup, rank=0, world=1, dist=None, stages=False, images_per_step=2, pmc=None):
    """`steps` timed steps of `images_per_step` images (ssp_export_points handles two images per call: a step is
    images_per_step / 2 calls back to back) on `dev` after `warmup` untimed ones; barriers and the max over ranks when world > 1.
    `pmc` = {kernel: {"traffic": bytes per launch}} from bench.live_pmc (HBM bytes of the dominant kernel), or None.  Returns the result fields (rank 0: incl. the roofline of the 3x3 forward launches).
    Used by main() below and by bench.py's `export` block (BASELINE configs[4] beside the headline line)."""
    import numpy as np
    import torch
    from semantic_superpoint_amd import lib as L
    from semantic_superpoint_amd import synth

    eng = L.Engine(arch, n, H, W, dev, with_grad=False)
    eng.load_state_dict(synth.default_init_state_dict(L.layer_table(arch), seed=0))
    rs = np.random.RandomState(1000 + rank)
    g = torch.Generator().manual_seed(1000 + rank)
    if images_per_step < 2 or images_per_step % 2:
        raise ValueError("images_per_step must be a positive multiple of 2 (the library call takes two images)")
    calls = images_per_step // 2
    imgs = [torch.rand(H, W, generator=g).to(dev) for _ in range(images_per_step)]

    def homographies():
        hs = np.stack([np.linalg.inv(synth.sample_homography(rs, **synth.WARP_PARAMS)) for _ in range(n)])
        hs[0] = np.identity(3)
        hs = torch.from_numpy(hs.astype(np.float32))
        return hs.to(dev), torch.inverse(hs).contiguous().to(dev)

    hom = [homographies() for _ in range(images_per_step)]  # host-side sampling stays outside the timed region (as in the loader)
    torch.cuda.synchronize()

    def step():
        counts = []
        for c in range(calls):
            ks = (2 * c, 2 * c + 1)
            vm = [L.op_homoadapt_views(imgs[k], hom[k][1]) for k in ks]
            outs = eng.export_points([v for v, _ in vm], [m for _, m in vm], [hom[k][0] for k in ks],
                                     conf_thresh=thresh, nms_dist=4, top_k=600, subpixel=True)
            counts += [int(o["count"].item()) for o in outs]  # the host needs the counts to slice the point lists
        return counts

    counts = None
    for _ in range(warmup):
        counts = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rank == 0:
        eng.profile_enable("conv3x3_all")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        counts = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"images_per_s": world * images_per_step * steps / dt, "ms_per_step": 1e3 * dt / steps, "points_last_step": counts,
           "images_per_step": images_per_step}
    if rank == 0:
        pr = eng.profile_read()
        if pr["launches"] > 0 and pr["ms"] > 0:
            alg = pr["flops"] / (pr["ms"] * 1e-3) / 1e12
            ex = pr["exec_flops"] / (pr["ms"] * 1e-3) / 1e12  # multiplies executed on the matrix cores (per launch, library)
            kern = eng.profile_read_kernels()
            res["roofline"] = {"bound": "mfma",
                               "kernel": " + ".join(kern) + " (3x3 forward of the encoder and the detector head; Winograd F(4x4,3x3) "
                                         "on the large maps / F(2x2,3x3) below, v_mfma_f32_32x32x2_f32)",
                               "achieved": round(ex, 2), "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s",
                               "frac": round(ex / PEAK_FP32_MFMA_TF, 4),
                               "note": "achieved / frac = multiplies EXECUTED on the matrix cores (1/4 of the direct-convolution ones "
                                       "for F(4x4,3x3), 16/36 for F(2x2,3x3)); algorithmic_* = direct-convolution FLOPs / time",
                               "algorithmic_tflops": round(alg, 2), "algorithmic_frac": round(alg / PEAK_FP32_MFMA_TF, 4),
                               "executed_tflops": round(ex, 2), "executed_frac": round(ex / PEAK_FP32_MFMA_TF, 4),
                               "traffic": None, "traffic_kernel": None,
                               "algorithmic_bytes_per_launch": round(pr["bytes"] / pr["launches"]),
                               "launches": pr["launches"], "avg_launch_ms": round(pr["ms"] / pr["launches"], 4)}
            if pmc:  # HBM bytes per launch of the kernel that carries the export (PMC child passes of bench.live_pmc)
                dom = max((k for k in pmc if pmc[k].get("traffic")), key=lambda k: pmc[k]["traffic"] * pmc[k]["launches"], default=None)
                if dom is not None:
                    res["roofline"]["traffic"] = round(pmc[dom]["traffic"])
                    res["roofline"]["traffic_kernel"] = "%s, %d launches in the counted call" % (dom, pmc[dom]["launches"])
                    res["roofline"]["traffic_all"] = {k: {"bytes_per_launch": round(v["traffic"]), "launches": v["launches"]}
                                                      for k, v in pmc.items() if v.get("traffic")}
        eng.profile_enable("none")
        if stages:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            vm = [L.op_homoadapt_views(imgs[k], hom[k][1]) for k in range(2)]
            ev[1].record()
            eng.export_points([v for v, _ in vm], [m for _, m in vm], [hom[k][0] for k in range(2)],
                              conf_thresh=thresh, nms_dist=4, top_k=600, subpixel=True)
            ev[2].record()
            torch.cuda.synchronize()
            res["stage_ms"] = {"views": round(ev[0].elapsed_time(ev[1]), 3),
                               "export_points": round(ev[1].elapsed_time(ev[2]), 3)}
    del eng
    return res


def main():
    args = parse_args()
    # same launch contract as bench.py: bare `--gpus N` spawns N ranks of THIS script before anything touches the GPU;
    # under a launcher WORLD_SIZE must equal --gpus.  Images are sharded over the ranks with no collective on the data path.
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world_env is None and args.gpus > 1:
        from bench import spawn_ranks
        sys.exit(spawn_ranks(args, script=os.path.abspath(__file__)))
    if int(world_env or "1") != args.gpus:
        raise SystemExit("bench_export.py: --gpus %d does not match WORLD_SIZE=%s of the launcher" % (args.gpus, world_env))

    pmc = None
    if args.gpus == 1 and args.traffic == "auto" and not args.pmc_child:  # child processes: this one has not touched the GPU yet
        pmc = export_pmc(args.arch, args.views, args.height, args.width)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench_export.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    arch = "SuperPointNet_gauss2" if args.arch == "sp" else "SuperPointNet_gauss2_ssmall"
    n, H, W = args.views, args.height, args.width
    res = measure_export(dev, arch, n, H, W, args.thresh, args.steps, args.warmup, rank, world, dist, stages=args.stages,
                         images_per_step=2 if args.pmc_child else args.images_per_step, pmc=pmc)

    if rank == 0:
        ips = res["images_per_s"]
        gf = GFLOP_FWD_DET_240x320 * (H * W) / (240.0 * 320.0) * n
        out = {"metric": "images/sec, homography-adaptation export (%d views/image, %dx%d)" % (n, H, W),
               "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(res["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s homography-adaptation export, %d views of %dx%d per image, %d images per step "
                                      "(%d ssp_export_points calls of 2 images), threshold %.4f, nms 4, top-k 600, soft-argmax"
                                      % (arch, n, H, W, res["images_per_step"], res["images_per_step"] // 2, args.thresh),
                          "parallelism": "images sharded over %d rank(s), no collective" % world},
               "views_per_s": round(ips * n, 1), "forward_tflops": round(ips * gf / 1e3, 2),
               "points_last_step": res["points_last_step"]}
        for k in ("roofline", "stage_ms"):
            if k in res:
                out[k] = res[k]
        if world == 1 and not args.no_cpu_baseline and not args.pmc_child:
            out["cpu_baseline"] = cpu_baseline(arch, H, W, n, args.thresh)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
