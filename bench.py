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

`roofline`: the dominant kernel of the step, conv_wino4_kernel (3x3 forward + data-gradient convolutions of the 240x320 /
120x160 / 60x80 maps, Winograd F(4x4,3x3) in fp32), timed live with HIP events on the launch stream inside the library
(ssp_profile_enable) over the timed region.  `achieved` / `frac` are the multiplies EXECUTED on the matrix cores per second
against the fp32 MFMA peak - the hardware fraction; `algorithmic_tflops` / `algorithmic_frac` count direct-convolution
FLOPs (Winograd executes 1/4 resp. 16/36 of them, so that ratio may exceed 1).  `roofline.kernels` holds the same numbers
for every 3x3 kernel of the step (conv_wino4, conv_wino_pipe, conv_wino_p2, wgrad_wino...), `mfma_busy` the PMC ratio
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs : GRBM_GUI_ACTIVE / 8 XCDs, `traffic` the HBM bytes per launch (2 x FETCH_SIZE +
WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md) - all three from rocprofv3 PMC child runs of this command made BEFORE
this process touches the GPU (`--traffic live`, the default at N = 1 when rocprofv3 is on PATH; null if that fails).
`cpu_baseline`: the oracle (oracle/cpu_ref.py, a restatement pinned against the reference) timed on this host's cores,
rank 0, N = 1 only: batch 32, 1 warm-up + 3 timed steps.  `bf16` (fp32 line, N = 1): BASELINE configs[3] per GPU - the same step on
the bf16 path (conv algorithm 12), 5 + 20 steps after the fp32 measurement, with its own `roofline` (bound "hbm": algorithmic bytes
of the dominant kernel against 8 TB/s, PMC traffic from two more child passes); `--dtype bf16` makes that path the line.  `export`: BASELINE configs[4] beside the headline - 3 timed
steps of 8 images (4 ssp_export_points calls of 2 images x 100 views, 480x640) after everything else; never part of `value`.
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
# 3x3 kernels whose PMC counters are reported (name substrings of the rocprofv3 kernel names)
PMC_KERNELS = ("conv_wino4_kernel", "conv_wino_pipe_kernel", "conv_wino_p2_kernel", "wgrad_wino4_kernel", "wgrad_wino_kernel",
               "conv_wino_bf16_kernel", "wgrad_wino_bf16_kernel", "conv_mfma_kernel<3", "wgrad_mfma_kernel<3", "conv_wino_kernel",
               "conv_bf16_ws_kernel", "conv_bf16_kernel<3", "wgrad_bf16_kernel<3")
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured with a float4 copy)
N_SIMD, N_XCD = 1024, 8  # MI355X: 256 CUs x 4 SIMDs in 8 XCDs


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


def under_profiler():
    """True when this process runs under rocprofv3 (its tool library is preloaded and has initialised the GPU before main()): no
    child processes may be started from here, and the clock-probe launches would only pollute the kernel statistics."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)


def power_state():
    """Socket power cap / current draw of GPU 0 as rocm-smi reports them (a child process, run BEFORE this process touches the GPU;
    None where the tool or the field is missing): the boxes of a pool differ in the clock their power controller grants."""
    exe = shutil.which("rocm-smi")
    if exe is None or under_profiler():
        return None
    try:
        r = subprocess.run([exe, "-d", "0", "--showmaxpower", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
    except Exception:
        return None
    out = {}
    for k, v in card.items():
        kl = k.lower()
        try:
            if "max" in kl and "power" in kl:
                out["power_cap_w"] = float(v)
            elif "power" in kl and ("socket" in kl or "average" in kl or "current" in kl):
                out["power_idle_w"] = float(v)
        except (TypeError, ValueError):
            pass
    return out or None


# the REAL reference (Train_model_heatmap_all.train_val_sample imported with stub modules) timed in the
# survey container (8 host threads, PyTorch CPU): SSp / SP pair step at 240x320, batch 32
REFERENCE_CPU_CONTAINER = {"SuperPointNet_gauss2_ssmall": {"value": 0.62, "s_per_step": 51.8}, "SuperPointNet_gauss2": {"value": 0.90, "s_per_step": 35.7}}


def roofline_block(prof_kernels, pmc, pmc_note, conv_algo, n_prof_steps, steps):
    """`roofline` of a measured line from the per-kernel HIP-event timings of the library (Engine.profile_read_kernels) and the
    PMC child runs.  fp32 kernels: bound "mfma" - multiplies EXECUTED on the matrix cores per second against the fp32 MFMA peak.
    Kernels of the bf16 path (conv_bf16*, wgrad_bf16): bound "hbm" - ALGORITHMIC bytes (each tensor once: bf16 input + output, resp.
    input + dY) per second against 8 TB/s; their matrix-core rate against the dense bf16 peak rides along as mfma_tflops / mfma_frac."""
    what = {"conv_wino4_kernel": "3x3 forward + data gradient, Winograd F(4x4,3x3): 1/4 of the direct multiplies",
            "conv_wino_pipe_kernel": "3x3 forward + data gradient, Winograd F(2x2,3x3): 16/36",
            "conv_wino_p2_kernel": "3x3 forward + data gradient on the 30x40 maps, Winograd F(2x2,3x3): 16/36",
            "wgrad_wino_kernel": "3x3 weight gradient, Winograd F(3x3,2x2): 16/36",
            "wgrad_wino4_kernel": "3x3 weight gradient, Winograd F(3x3,4x4): 1/4",
            "other": "direct implicit GEMM / bf16-operand Winograd kernels",
            "conv_bf16_kernel": "3x3 forward + data gradient of the bf16 path: direct implicit GEMM on v_mfma_f32_32x32x16_bf16, bf16 tensors",
            "wgrad_bf16_kernel": "3x3 weight gradient of the bf16 path: K = pixels through ds_read_b64_tr_b16"}
    pmc_name = {"conv_bf16_kernel": "conv_bf16_kernel<3", "wgrad_bf16_kernel": "wgrad_bf16_kernel<3"}
    reduced = conv_algo in (3, 7, 8)
    mult = {7: 3.0}.get(conv_algo, 1.0)  # the split-bf16 mode 7 issues 3 bf16 MFMAs per product block
    kernels = {}
    for name, k in prof_kernels.items():
        if k["ms"] <= 0:
            continue
        sec = k["ms"] * 1e-3
        c = pmc.get(pmc_name.get(name, name), {})
        e = {"what": what.get(name, ""), "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 4),
             "ms_per_step": round(k["ms"] / n_prof_steps, 3), "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"]),
             "traffic": None if c.get("traffic") is None else round(c["traffic"])}
        if name in ("conv_bf16_kernel", "wgrad_bf16_kernel"):
            gbs = k["bytes"] / sec / 1e9
            tf = k["flops"] / sec / 1e12
            e.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                      "mfma_tflops": round(tf, 1), "mfma_frac": round(tf / PEAK_BF16_MFMA_TF, 4),
                      "traffic_gbs": None if c.get("traffic") is None else round(c["traffic"] * k["launches"] / sec / 1e9, 1)})
        else:
            alg, ex = k["flops"] / sec / 1e12, (mult if name == "other" else 1.0) * k["exec_flops"] / sec / 1e12
            # per-kernel peak: the fp32 kernel families price against the fp32 MFMA peak under every algorithm (the mixed bf16
            # mode runs its forward on them); only the bf16-operand bucket of a reduced-precision line uses the bf16 peak
            kpeak = PEAK_BF16_MFMA_TF if (reduced and name == "other") else PEAK_FP32_MFMA_TF
            e.update({"bound": "mfma", "peak": kpeak, "unit": "TFLOP/s", "algorithmic_tflops": round(alg, 2),
                      "algorithmic_frac": round(alg / kpeak, 4), "executed_tflops": round(ex, 2), "executed_frac": round(ex / kpeak, 4),
                      "achieved": round(ex, 2), "frac": round(ex / kpeak, 4),
                      "mfma_busy": None if c.get("mfma_busy") is None else round(c["mfma_busy"], 4)})
        kernels[name] = e
    if not kernels:
        return None
    dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"])  # the kernel with the most time per step
    d = kernels[dom]
    rl = {"bound": d["bound"], "kernel": "%s (%s)" % (dom, d["what"]), "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"],
          "frac": d["frac"], "traffic": d["traffic"], "pmc_source": pmc_note,
          "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "launches": d["launches"],
          "avg_launch_ms": d["avg_launch_ms"], "ms_per_step": d["ms_per_step"],
          "bracketed_steps": "%d of the %d timed steps" % (n_prof_steps, steps), "kernels": kernels}
    if d["bound"] == "mfma":
        rl["note"] = ("achieved / frac = multiplies EXECUTED on the matrix cores per second / fp32 MFMA peak (the hardware fraction); "
                      "algorithmic_* = direct-convolution FLOPs / time, which Winograd undercuts by 4x resp. 36/16, so that ratio may exceed 1")
        for f in ("algorithmic_tflops", "algorithmic_frac", "executed_tflops", "executed_frac", "mfma_busy"):
            rl[f] = d[f]
    else:
        rl["note"] = ("achieved = ALGORITHMIC bytes of the launches (bf16 input + output of each 3x3 convolution, once) / their time, "
                      "against the 8 TB/s HBM3E spec; traffic = HBM bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE); "
                      "mfma_* = direct-convolution FLOPs / time against the dense bf16 MFMA peak")
        rl["mfma_tflops"], rl["mfma_frac"], rl["traffic_gbs"] = d["mfma_tflops"], d["mfma_frac"], d["traffic_gbs"]
    return rl


PRECISION = {
    3: ("bf16", "bf16 matrix-core operands / fp32 accumulate + master (NOT the headline precision)"),
    7: ("bf16x2", "split-bf16 (hi + lo = 16 significant bits) matrix-core operands, three bf16 MFMAs per product / "
                  "fp32 accumulate + master (NOT the headline precision)"),
    12: ("bf16", "bf16 path: bf16 NHWC activations / activation gradients in HBM, every convolution (forward, data and weight "
                 "gradient) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; fp32 BatchNorm statistics, losses, master "
                 "weights, Adam (BASELINE configs[3] per GPU; NOT the fp32 headline precision)"),
    8: ("bf16", "mixed bf16: fp32 forward, data / weight gradients of the 3x3 layers with bf16 matrix-core operands; "
                "fp32 tensors, accumulate, BatchNorm, master weights, Adam (NOT the headline precision)"),
}


def workload_name(arch, H, W, B, conv_algo, dense, graph=False):
    return "%s pair step %dx%d, batch %d per GPU, %s, %s, Adam%s" % (
        arch, H, W, B, PRECISION.get(conv_algo, ("f32", "fp32"))[1],
        "sparse loss 1000x100" if dense is None else "dense descriptor loss (1200x1200 per image)", ", hipGraph replay" if graph else "")


def side_block(Engine, what, arch, conv_algo, B, H, W, dev, sample, args, headline_pairs_s, pmc_side, pmc_note, dense, cpu_ref_line=None,
               steps=20, warmup=5):
    """Another single-GPU configuration of BASELINE.json measured beside the headline in the same process (never part of `value`):
    the same pair step, same inputs and device-sampled indices, on another architecture / conv algorithm.  Carries the headline's
    fields: metric, value, unit, dtype, steps, warmup, ms_per_step, config.workload, roofline (HIP events on every 4th step; PMC
    traffic / matrix-pipe occupancy from child passes of that configuration), cpu_baseline (a reference to the one measured here:
    the oracle is fp32, it is not timed twice)."""
    import torch
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import layer_table, SCALAR_NAMES
    eng = Engine(arch, B, H, W, dev, dense_loss=dense is not None)
    eng.set_conv_algo(conv_algo)
    eng.load_state_dict(synth.default_init_state_dict(layer_table(arch), seed=0))

    def step(it):
        eng.zero_grad()
        eng.pair_step(sample, indices=None, seed=(it * 1000003 + 1), train=True, lambda_loss=1.0, lamda_d=1.0, multi_task=True, dense=dense)
        eng.adam_step(args.lr)
    for it in range(warmup):
        step(it)
    torch.cuda.synchronize()
    eng.profile_enable("conv3x3_every")
    t0 = time.perf_counter()
    for it in range(steps):
        eng.profile_pause(it % 4 != 0)
        step(warmup + it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs_s = B * steps / dt
    scal = dict(zip(SCALAR_NAMES, eng.scalars.cpu().tolist()))
    blk = {"metric": "image-pairs/sec at %dx%d bs%d (pair training step), %s" % (H, W, B, what), "value": round(pairs_s, 2),
           "unit": "image-pairs/s", "n_gpus": 1, "dtype": PRECISION.get(conv_algo, ("f32", "fp32"))[0], "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 3), "vs_headline": round(pairs_s / headline_pairs_s, 3), "data": "synthetic",
           "config": {"workload": workload_name(arch, H, W, B, conv_algo, dense), "parallelism": "dp1", "global_batch": B},
           "step_tflops": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3, 2), "final_loss": round(scal["loss"], 4)}
    if conv_algo == 12:
        blk["vs_fp32_line"] = blk["vs_headline"]
    n_prof = len([i for i in range(steps) if i % 4 == 0])
    rl = roofline_block(eng.profile_read_kernels(), pmc_side, pmc_note, conv_algo, n_prof, steps)
    if rl is not None:
        blk["roofline"] = rl
    if cpu_ref_line is not None:
        blk["cpu_baseline"] = cpu_ref_line
    eng.profile_enable("none")
    del eng
    torch.cuda.empty_cache()
    return blk


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
    ap.add_argument("--conv-algo", type=int, default=1, choices=[0, 1, 6, 9, 10, 11, 12],
                    help="ssp_set_conv_algo: 1 = fp32 Winograd (default, the headline), 0 = fp32 direct implicit GEMM, 6 = fp32 "
                         "Winograd F(2x2,3x3), two 4-wave workgroups per CU, 9 = F(2x2,3x3) only, 10 = F(4x4,3x3) wherever legal, 11 = "
                         "algorithm 1 with the F(3x3,4x4) weight gradient, 12 = the bf16 path (--dtype bf16).  (2 / 3 / 5 / 7 / 8: "
                         "experiments of rounds 1-3, compiled out of the shipped library - SSP_LEGACY_ALGOS)")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="f32 = --conv-algo 1 (the headline); bf16 = --conv-algo 12, the bf16 path of BASELINE configs[3] (bf16 NHWC "
                         "activations in HBM, every convolution on the bf16 matrix cores, fp32 accumulate / statistics / master weights)")
    ap.add_argument("--desc-loss", default="sparse", choices=["sparse", "dense"],
                    help="descriptor loss of the step: sparse (shipped configs, the headline) or dense (model.dense_loss)")
    ap.add_argument("--graph", action="store_true", help="replay the pair step as a hipGraph (ssp_pair_step_graph)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: one blocking all-reduce after the whole backward instead of the overlapped split bucket")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "none"],
                    help="roofline.traffic: live = two rocprofv3 --pmc child runs of this command (FETCH_SIZE, WRITE_SIZE) "
                         "before the timed run; auto = live when N = 1, rocprofv3 is on PATH and the roofline leg is on")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the profiled child of --traffic live
    ap.add_argument("--no-export", action="store_true", help="skip the `export` block (BASELINE configs[4], 3 calls at 100 x 480x640)")
    ap.add_argument("--no-sp", action="store_true",
                    help="skip the `sp` block of the SSp fp32 line (BASELINE configs[1]: SuperPointNet_gauss2, the same step without the "
                         "segmentation head)")
    ap.add_argument("--no-bf16", action="store_true",
                    help="skip the `bf16` block of the fp32 line (BASELINE configs[3] per GPU: the same step on the bf16 path, conv algorithm 12)")
    args = ap.parse_args(argv)
    if args.dtype is not None:
        args.conv_algo = {"f32": 1, "bf16": 12}[args.dtype]
    return args


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (the parent never touches the GPU)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args, script=None, argv=None, timeout=None):
    """Start args.gpus fresh rank processes of `script` (default: this file) with `argv` (default: this process's own),
    relay rank 0's stdout, and WATCH them: the first rank that exits non-zero (or the overall timeout, SSP_BENCH_TIMEOUT,
    default 3600 s) gets the others terminated - a dead rank would otherwise leave its siblings in the RCCL rendezvous /
    a collective until torch's timeout.  Children are ordinary subprocesses started BEFORE this process touches the GPU."""
    script = os.path.abspath(script or __file__)
    argv = list(sys.argv[1:] if argv is None else argv)
    timeout = float(os.environ.get("SSP_BENCH_TIMEOUT", "3600")) if timeout is None else timeout
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    t0, failed = time.monotonic(), None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = "rank %d exited with code %d" % (r, p.returncode)
                break
        else:
            if time.monotonic() - t0 > timeout:
                failed = "timeout after %.0f s" % timeout
            else:
                time.sleep(0.2)
    if failed is not None:  # stop the survivors: exactly the processes started above, by PID
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    out0.seek(0)
    text = out0.read()
    out0.close()
    if text:
        sys.stdout.write(text)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or failed:
        sys.stderr.write("%s: ranks failed: %s (%s)\n" % (os.path.basename(script), bad, failed))
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# PMC counters measured in this run: rocprofv3 passes over a short child run of the same workload
# ------------------------------------------------------------------------------------------------
def live_pmc(args, conv_algo=None, counter_sets=(("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")),
             child=None, kernels=None):
    """Per 3x3 kernel: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB -> bytes (separate PMC passes, FETCH doubled:
    gfx950 counts wide coalesced reads at half, MI355X_MICROARCH.md section HBM) and the matrix-pipe occupancy
    SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs : GRBM_GUI_ACTIVE / 8 XCDs (third pass).
    Returns ({kernel: {"traffic": bytes or None, "mfma_busy": ratio or None, "launches": n}}, note)."""
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return {}, "rocprofv3 not on PATH"
    kernels = PMC_KERNELS if kernels is None else kernels
    if child is None:  # (default: one step of THIS benchmark; bench_export passes its own child command)
        child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "1", "--gpus", "1",
             "--arch", args.arch, "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
             "--conv-algo", str(args.conv_algo if conv_algo is None else conv_algo), "--desc-loss", args.desc_loss,
             "--no-cpu-baseline", "--no-roofline", "--traffic", "none", "--no-export", "--no-bf16", "--no-sp"]
    tot = {}  # kernel -> counter -> [sum, set(dispatch ids)]
    tmp = tempfile.mkdtemp(prefix="ssp_pmc_", dir="/tmp")
    notes = []
    try:
        for ctrs in counter_sets:
            d = os.path.join(tmp, ctrs[0])
            cmd = [rocprof, "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                notes.append("rocprofv3 --pmc %s timed out" % " ".join(ctrs))
                continue
            if r.returncode != 0:
                notes.append("rocprofv3 --pmc %s failed (rc %d)" % (" ".join(ctrs), r.returncode))
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                notes.append("no counter_collection.csv for %s" % " ".join(ctrs))
                continue
            for row in csv.DictReader(open(files[0])):
                kn = row["Kernel_Name"].replace("wgrad_wino_fused_kernel", "wgrad_wino_kernel")
                kn = kn.replace("conv_bf16_ws_kernel", "conv_bf16_kernel<3")   # (one bucket: the 3x3 forward / data-gradient launches)
                k = next((k for k in kernels if k in kn), None)
                if k is None or row["Counter_Name"] not in ctrs:
                    continue
                e = tot.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
                e[0] += float(row["Counter_Value"])
                e[1].add(row["Dispatch_Id"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, c in tot.items():
        e = {"traffic": None, "mfma_busy": None, "launches": 0}
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        if f and w and len(f[1]) == len(w[1]) > 0:
            e["traffic"] = (2.0 * f[0] + w[0]) * 1024.0 / len(f[1])
            e["launches"] = len(f[1])
        m, g = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
        if m and g and g[0] > 0:
            e["mfma_busy"] = (m[0] / N_SIMD) / (g[0] / N_XCD)
            e["launches"] = e["launches"] or len(m[1])
        out[k] = e
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (three separate passes of a "
            "1+1-step child run in this job); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches; mfma_busy = "
            "MFMA_BUSY / 1024 SIMDs : GUI_ACTIVE / 8 XCDs")
    if notes:
        note += "; FAILED: " + "; ".join(notes)
    return out, note


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

    power = power_state() if (rank == 0 and not args.pmc_child) else None   # (a child process: nothing here has touched the GPU)
    pmc, pmc_note = {}, "not collected"
    want_live = args.traffic == "live" or (args.traffic == "auto" and not args.no_roofline and not args.pmc_child)
    pmc16 = {}
    pmc_export = None
    want_bf16_block = (world == 1 and rank == 0 and args.conv_algo == 1 and not args.no_bf16 and not args.pmc_child)
    # BASELINE configs[1] (SuperPointNet_gauss2, no semantic head) beside the configs[2] headline
    want_sp_block = (world == 1 and rank == 0 and args.conv_algo == 1 and args.arch == "ssp" and not args.no_sp and not args.pmc_child)
    pmc_sp = {}
    if want_live and world == 1:
        pmc, pmc_note = live_pmc(args)  # child processes; this process has not touched the GPU yet
        if want_bf16_block:  # HBM traffic of the bf16 path's kernels: two more passes
            pmc16, _ = live_pmc(args, conv_algo=12, counter_sets=(("FETCH_SIZE",), ("WRITE_SIZE",)))
        if want_sp_block:    # the SP step's own counters (same kernels, one 3x3 head less per launch group)
            sp_args = argparse.Namespace(**dict(vars(args), arch="sp"))
            pmc_sp, _ = live_pmc(sp_args)
        if not args.no_export and rank == 0:  # ... and of the export block's convolutions (one call, two images)
            import bench_export
            pmc_export = bench_export.export_pmc("sp", 100, 480, 640)

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
        if rank == 0 and not single_dev and "NCCL_DEBUG" not in os.environ:
            # rank 0 logs RCCL's init and algorithm / protocol choices to a file that the line quotes (`dp.rccl`)
            os.environ["SSP_NCCL_LOG"] = os.path.join(tempfile.gettempdir(), "ssp_rccl_rank0_%d.log" % os.getpid())
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING", NCCL_DEBUG_FILE=os.environ["SSP_NCCL_LOG"])
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

    dp_diag = parallel.StepDiag(every=4, cuda=True) if world > 1 else None  # N > 1: the line diagnoses its own all-reduce overlap
    stream = torch.cuda.Stream(device=dev) if args.graph else None  # stream capture needs a non-default stream
    profiling = [False]

    PROF_EVERY = 4  # HIP events bracket the 3x3 launches of every 4th timed step (all of them cost ~1.7 % of the step)

    def step(it):
        if profiling[0]:
            eng.profile_pause((it - args.warmup) % PROF_EVERY != 0)
        kw = dict(indices=None, seed=(it * 1000003 + rank * 7919 + 1), train=True, lambda_loss=1.0, lamda_d=1.0,
                  multi_task=True, dense=dense, graph=args.graph)
        eng.zero_grad()
        if world > 1 and not args.no_overlap:
            parallel.pair_step_overlapped(eng, sample, args.lr, diag=dp_diag if it >= args.warmup else None, **kw)
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
        eng.profile_enable("conv3x3_every")  # every 3x3 forward / data-gradient / weight-gradient launch, split by kernel
        profiling[0] = True
    torch.cuda.synchronize()
    probe_clock = rank == 0 and not under_profiler() and not args.pmc_child
    clock_before = ssp.lib.clock_probe(5.0) if probe_clock else None   # (outside the timed region; ~5 ms of fp32 MFMAs on every CU)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    clock_after = ssp.lib.clock_probe(5.0) if probe_clock else None
    per_rank_ms = [1e3 * dt / args.steps]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [1e3 * float(v.item()) / args.steps for v in allt]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        scal = dict(zip(SCALAR_NAMES, eng.scalars.cpu().tolist()))
        pairs_s = world * B * args.steps / dt
        prec = PRECISION.get(args.conv_algo, ("f32", "fp32"))
        out = {"metric": "image-pairs/sec at %dx%d bs%d (pair training step)" % (H, W, B), "value": round(pairs_s, 2),
               "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": prec[0], "data": "synthetic",
               "config": {"workload": workload_name(arch, H, W, B, args.conv_algo, dense, args.graph),
                          "parallelism": "dp%d" % world, "global_batch": world * B,
                          "allreduce": ("none" if world == 1 else "one bucket after backward" if args.no_overlap else
                                        "split bucket, early part overlapped with the backward of the 240x320 layers")},
               "rccl_ranks": rccl_ranks,
               "step_tflops": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3, 2),
               # direct-convolution (algorithmic) FLOPs of the whole step against the fp32 MFMA peak: an ALGORITHMIC ratio (Winograd
               # executes 1/4 resp. 16/36 of these multiplies, so it may exceed 1); the hardware fraction is roofline.frac
               "step_algorithmic_tflops_vs_fp32_mfma_peak": round(pairs_s * GFLOP_PER_PAIR[arch] / 1e3 / (PEAK_FP32_MFMA_TF * world), 4),
               "final_loss": round(scal["loss"], 4), "build_id": ssp.lib.build_id()[:16]}
        # what explains this box (boxes of one pool differ by +-3 % in pairs/s): the shader clock the device sustains under fp32
        # matrix-core load right before and right after the timed region (ssp_clock_probe: 5 ms on every CU, 2400 MHz is the part's
        # maximum) and the socket's power cap / draw at start as rocm-smi reports them
        if clock_before is not None and clock_after is not None:
            out["gpu_clock_mhz"] = {"before": round(clock_before, 1), "after": round(clock_after, 1), "max": 2400.0,
                                    "pairs_per_s_per_ghz": round(pairs_s / (0.5e-3 * (clock_before + clock_after)), 2)}
        if power is not None:
            out["power"] = power
        if world > 1:
            log = None
            try:
                with open(os.environ.get("SSP_NCCL_LOG", "")) as f:
                    log = f.read()
            except OSError:
                pass
            out["dp"] = parallel.dp_diagnostics(dp_diag.summary() if dp_diag is not None else {}, per_rank_ms, log)
        if profiled:
            n_prof_steps = len([i for i in range(args.steps) if i % PROF_EVERY == 0])  # the bracketed steps of the timed region
            rl = roofline_block(eng.profile_read_kernels(), pmc, pmc_note, args.conv_algo, n_prof_steps, args.steps)
            if rl is not None:
                out["roofline"] = rl
            eng.profile_enable("none")
            profiling[0] = False
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arch, H, W, batch=B)
            if (H, W, B) == (240, 320, 32) and arch in REFERENCE_CPU_CONTAINER:
                ref = REFERENCE_CPU_CONTAINER[arch]
                out["cpu_baseline"]["reference_cpu_container"] = {
                    "value": ref["value"], "unit": "image-pairs/s", "cores": 8, "kind": "reference",
                    "sample": "BASELINE.md section 2: the reference's own train_val_sample (stub modules for absent packages) in the build "
                              "container, 8 threads, %.1f s per step - measured there once, NOT in this run" % ref["s_per_step"]}
        cpu_line = None
        if "cpu_baseline" in out:
            cpu_line = dict(out["cpu_baseline"], note="the fp32 oracle step timed once in this run (top-level cpu_baseline)")
        if want_sp_block:
            # BASELINE configs[1] beside the headline (never part of `value`): SuperPointNet_gauss2, the same step without the
            # segmentation head, fp32 default kernels, same inputs
            try:
                out["sp"] = side_block(Engine, "SuperPointNet_gauss2 (BASELINE configs[1])", "SuperPointNet_gauss2", 1, B, H, W, dev,
                                       sample, args, pairs_s, pmc_sp, pmc_note, dense)
            except Exception as e:  # the headline line must survive a failure of the side measurement
                out["sp"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if want_bf16_block:
            # BASELINE configs[3] per GPU beside the fp32 headline (never part of `value`): the same step, same inputs, on the bf16
            # path (conv algorithm 12: bf16 activations in HBM, bf16 matrix cores, fp32 accumulate / statistics / master / Adam)
            try:
                out["bf16"] = side_block(Engine, "bf16 path (BASELINE configs[3] per GPU)", arch, 12, B, H, W, dev, sample, args,
                                         pairs_s, pmc16, pmc_note, dense, cpu_ref_line=cpu_line)
            except Exception as e:  # the headline line must survive a failure of the side measurement
                out["bf16"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_export and not args.pmc_child:
            # BASELINE configs[4] next to the headline (never part of `value`): the training engine is released first
            try:
                import bench_export
                del eng, sample
                torch.cuda.empty_cache()
                ex = bench_export.measure_export(dev, "SuperPointNet_gauss2", 100, 480, 640, 0.0155, steps=3, warmup=1,
                                                 images_per_step=8, pmc=pmc_export)
                out["export"] = {"metric": "images/sec, homography-adaptation export (100 views/image, 480x640)",
                                 "value": round(ex["images_per_s"], 2), "unit": "images/s", "steps": 3, "warmup": 1,
                                 "ms_per_step": round(ex["ms_per_step"], 2), "images_per_step": ex["images_per_step"],
                                 "points_last_step": ex["points_last_step"], "roofline": ex.get("roofline")}
            except Exception as e:  # the headline line must survive a failure of the side measurement
                out["export"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
