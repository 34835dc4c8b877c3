"""CPU: the drop-in boundary without a GPU - the C-ABI library loads and exports every symbol declared in
include/ssp_hip.h, the Python shims expose the reference's state_dict layout, refuse to run on the CPU, and the
host-side logic (reference-faithful index sampler, data-parallel helpers over gloo) is correct."""
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as C
from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import semantic_superpoint_amd as ssp
    hdr = open(os.path.join(ROOT, "include", "ssp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ssp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = ssp.load_library()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(ssp.lib.EXPORTS) == declared


def test_library_is_built_from_the_checked_out_sources():
    """The build id compiled into the in-tree library (sha256 over csrc/*.hip, csrc/*.hip.h, include/ssp_hip.h) equals the hash of
    the sources as checked out: the binary under test is HEAD's, by content (not by modification time)."""
    import semantic_superpoint_amd as ssp
    ssp.build()  # rebuilds iff the ids differ
    want = ssp.hipbuild.source_id()
    assert ssp.hipbuild.library_id() == want
    assert ssp.lib.build_id() == want


def test_store_data_hazard_lint():
    """hipbuild.store_data_hazards: the ISA lint behind verify_binary (round 5: hipcc for gfx950 leaves no wait state between a 12 /
    16-byte store and a vector instruction that overwrites one of its data registers; the hardware needs one).  A synthetic
    disassembly with the pattern that corrupted conv_bf16_ws_kernel, its harmless neighbours, and the in-tree library (clean)."""
    import tempfile
    import semantic_superpoint_amd as ssp
    hb = ssp.hipbuild
    dis = (
        "0000000000001000 <kernel_a>:\n"
        "\tbuffer_store_dwordx4 v[6:9], v55, s[36:39], s62 offen      // 000000001000: E07C1000\n"
        "\tv_lshlrev_b32_e32 v6, 16, v7                               // 000000001008: 240C0E90\n"      # the round-5 bug
        "\tglobal_store_dwordx3 v[2:3], v[10:12], off                  // 000000001010: DC7C8000\n"
        "\tv_pk_add_f32 v[12:13], v[20:21], v[22:23]                  // 000000001018: D3B2400C\n"      # overlaps v12
        "\tbuffer_store_dwordx4 v[14:17], v55, s[36:39], s20 offen    // 000000001020: E07C1000\n"
        "\ts_nop 0                                                    // 000000001028: BF800000\n"      # padded: fine
        "\tv_lshlrev_b32_e32 v14, 16, v15                             // 00000000102C: 241C1E90\n"
        "\tbuffer_store_dwordx4 v[18:21], v55, s[36:39], s20 offen    // 000000001030: E07C1000\n"
        "\tv_add_f32_e32 v30, v18, v19                                // 000000001038: 023C2712\n"      # reads only: fine
        "\tbuffer_store_dwordx2 v[40:41], v55, s[36:39], s20 offen    // 000000001040: E0741000\n"
        "\tv_mov_b32_e32 v40, 0                                       // 000000001048: 7E500280\n"      # 8-byte store: no hazard
        "\tscratch_store_dwordx4 off, v[50:53], off offset:16         // 000000001050: DC7C0010\n"      # the `off,` scratch form ...
        "<L7>:\n"                                                                                        # ... across a branch-target label
        "\tv_mov_b32_e32 v52, 0                                       // 000000001058: 7E680280\n"
        "\tglobal_store_dwordx4 v[2:3], a[4:7], off                   // 000000001060: DC7C8000\n"      # accumulation-register data
        "\tv_accvgpr_write_b32 a5, v3                                 // 000000001068: D3D94005\n"
        "0000000000002000 <kernel_b>:\n"
        "\tbuffer_store_dwordx4 v[6:9], v55, s[36:39], s62 offen      // 000000002000: E07C1000\n"
        "\tv_cmp_gt_f32_e32 vcc, v6, v7                               // 000000002008: 7C880F06\n"      # writes no vector register
    )
    found = hb.store_data_hazards(dis)
    assert [(k, c.split()[0], c.split()[1].rstrip(",")) for k, _, c in found] == [("kernel_a", "v_lshlrev_b32_e32", "v6"),
                                                                                ("kernel_a", "v_pk_add_f32", "v[12:13]"),
                                                                                ("kernel_a", "v_mov_b32_e32", "v52"),
                                                                                ("kernel_a", "v_accvgpr_write_b32", "a5")], found
    ssp.build()
    with tempfile.TemporaryDirectory(prefix="ssp_isa_", dir="/tmp") as tmp:
        real, _ = hb.disassemble(hb.LIB, tmp)
    assert len(real) > 1_000_000 and hb.store_data_hazards(real) == []


def test_create_without_gpu_reports_layout():
    """ssp_create / counts need no device memory: parameter and BN layout match the oracle's spec."""
    import ctypes as Ct
    import semantic_superpoint_amd as ssp
    lib = ssp.load_library()
    for arch, aid in (("SuperPointNet_gauss2", 0), ("SuperPointNet_gauss2_ssmall", 1)):
        cfg = ssp.lib.SspConfig(aid, 133, 2, 64, 96, 1000, 100)
        h = Ct.c_void_p()
        assert lib.ssp_create(Ct.byref(cfg), Ct.byref(h)) == 0, lib.ssp_last_error()
        n = sum(int(np.prod(s)) for k, s, _ in C.state_spec(arch) if k in C.param_keys(arch))
        assert lib.ssp_param_count(h) == n
        assert lib.ssp_bn_layer_count(h) == sum(1 for _, bn, _, _, _ in C.layer_table(arch) if bn)
        assert lib.ssp_workspace_bytes(h) > 0
        lib.ssp_destroy(h)
    bad = ssp.lib.SspConfig(0, 133, 2, 60, 96, 1000, 100)  # H not a multiple of 8
    assert lib.ssp_create(Ct.byref(bad), Ct.byref(h)) != 0 and b"multiples of 8" in lib.ssp_last_error()


def test_retired_conv_algorithms_are_refused_and_the_bench_flags_follow():
    """Conv algorithms 2 / 3 / 5 / 7 / 8 (experiments of rounds 1-3) are compiled out of the shipped library: the per-handle and the
    process-wide selectors refuse them with a message, the shipped ones are accepted; bench.py offers only what the library runs and
    its side blocks (`sp`, `bf16`) can be switched off."""
    import ctypes as Ct
    import semantic_superpoint_amd as ssp
    lib = ssp.load_library()
    cfg = ssp.lib.SspConfig(1, 133, 2, 64, 96, 1000, 100)
    h = Ct.c_void_p()
    assert lib.ssp_create(Ct.byref(cfg), Ct.byref(h)) == 0
    for algo in (2, 3, 5, 7, 8):
        assert lib.ssp_handle_set_conv_algo(h, algo) != 0 and b"compiled out" in lib.ssp_last_error(), algo
        assert lib.ssp_set_conv_algo(algo) != 0 and b"compiled out" in lib.ssp_last_error(), algo
    for algo in (0, 1, 6, 9, 10, 11, 12):
        assert lib.ssp_handle_set_conv_algo(h, algo) == 0, (algo, lib.ssp_last_error())
    assert lib.ssp_handle_set_conv_algo(h, 4) != 0 and lib.ssp_handle_set_conv_algo(h, 13) != 0
    assert lib.ssp_set_conv_algo(1) == 0
    lib.ssp_destroy(h)
    b = _load_bench()
    for algo in ("2", "3", "5", "7", "8"):
        with pytest.raises(SystemExit):
            b.parse_args(["--conv-algo", algo])
    a = b.parse_args(["--no-sp", "--no-bf16", "--conv-algo", "10"])
    assert a.no_sp and a.no_bf16 and a.conv_algo == 10 and not b.parse_args([]).no_sp


@pytest.mark.parametrize("arch", ["SuperPointNet_gauss2", "SuperPointNet_gauss2_ssmall"])
def test_module_state_dict_is_the_reference_wire_format(arch):
    from semantic_superpoint_amd import models
    net = getattr(models, arch)()
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(s)) for k, s, _ in C.state_spec(arch)]
    assert [k for k, _ in net.named_parameters()] == C.param_keys(arch)
    sd = C.init_state_dict(arch, seed=1)
    net.load_state_dict({k: torch.as_tensor(np.array(v)) for k, v in sd.items()})  # golden weights load unchanged
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 8, 8))


def test_trainer_refuses_cpu_and_unsupported_configs():
    from semantic_superpoint_amd.Train_model_heatmap_all import Train_model_heatmap_all as T
    cfg = {"data": {"semantic": False, "gaussian_label": {"enable": True}, "warped_pair": {"enable": True}},
           "model": {"name": "SuperPointNet_gauss2", "params": {}, "batch_size": 2, "real_batch_size": 2,
                     "learning_rate": 1e-3, "lambda_loss": 1, "multi_task_loss": True,
                     "dense_loss": {"enable": False}, "sparse_loss": {"enable": True, "params": {"method": "2d", "dist": "cos"}}},
           "validation_interval": 10}
    with pytest.raises(RuntimeError, match="HIP device"):
        T(cfg, device="cpu")


def test_trainer_tensorboard_helpers_follow_the_reference_names():
    """tb_images_dict / tb_hist_dict / printLosses of the base class (Train_model_frontend_all.py:535-582): tags
    "<task>-<element>/<idx>", at most max_img images per element, semantic class maps reduced to their argmax, step n_iter // r."""
    import types
    from semantic_superpoint_amd.Train_model_heatmap_all import Train_model_heatmap_all as T
    calls = []
    writer = types.SimpleNamespace(add_image=lambda tag, img, step: calls.append(("image", tag, np.asarray(img).shape, step)),
                                   add_histogram=lambda tag, v, step: calls.append(("hist", tag, step)))
    me = types.SimpleNamespace(_writer=writer, config={"semantic": True}, n_iter=12, r=4)
    imgs = {"heat": np.zeros((3, 1, 8, 8)), "sem_pred": np.random.rand(2, 5, 8, 8), "warp_sem_pred": np.random.rand(2, 5, 8, 8)}
    expect = np.argmax(imgs["sem_pred"], axis=1)
    T.tb_images_dict(me, "training", imgs, max_img=2)
    tags = [c[1] for c in calls]
    assert tags == ["training-heat/0", "training-heat/1", "training-sem_pred/0", "training-sem_pred/1",
                    "training-warp_sem_pred/0", "training-warp_sem_pred/1"]
    assert all(c[3] == 3 for c in calls) and calls[2][2] == (1, 8, 8)
    assert np.array_equal(imgs["sem_pred"][:, 0], expect)
    calls.clear()
    T.tb_hist_dict(me, "val", {"a": np.arange(4.0)})
    assert calls == [("hist", "val-a", 3)]
    me._writer = None   # no writer: silently nothing, like tb_scalar_dict
    T.tb_images_dict(me, "val", {"x": np.zeros((1, 1, 2, 2))})
    T.tb_hist_dict(me, "val", {"a": np.arange(4.0)})


def test_host_sampler_reproduces_reference_indices():
    """`ssp_sampler: reference` consumes numpy/torch RNG like the reference: same seeds => G4's indices."""
    from semantic_superpoint_amd.Train_model_heatmap_all import sample_sparse_indices_host
    g = G.load("g4_sparse_loss_small.npz")
    np.random.seed(123)
    torch.manual_seed(321)
    ma, mb, nm = sample_sparse_indices_host(torch.from_numpy(g["H"]), 4, 6, 1000, 100)
    for i in range(ma.shape[0]):
        assert np.array_equal(ma[i].numpy(), (g["uv_a%d" % i][:, 0] + g["uv_a%d" % i][:, 1] * 6).astype(np.int32))
        assert np.array_equal(mb[i].numpy(), (g["uv_b%d" % i][:, 0] + g["uv_b%d" % i][:, 1] * 6).astype(np.int32))
        assert np.array_equal(nm[i].numpy(), g["nm_b%d" % i].astype(np.int32))


def test_synthetic_pair_generator_matches_oracle_warps():
    """semantic-superpoint_amd/synth.py (product side) against the oracle's restatement of the dataset warps."""
    from semantic_superpoint_amd import synth
    s = synth.make_pair(2, 40, 56, "cpu", seed=3, semantic=True)
    inv = s["inv_homographies"]
    w = C.inv_warp_image_batch(s["image"], inv)
    assert (w - s["warped_img"]).abs().max() < 1e-5
    vm = C.compute_valid_mask((40, 56), inv, erosion_radius=3).view(2, 1, 40, 56)
    assert float((vm != s["warped_valid_mask"]).float().mean()) < 2e-3
    for i in range(2):
        pts = torch.nonzero(s["labels_2D"][i, 0]).flip(1)
        assert torch.equal(C.warp_labels(pts, 40, 56, s["homographies"][i]), s["warped_labels"][i])
    assert s["semantic"].dtype == torch.int64 and int(s["warped_sem"].max()) <= 133


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from semantic_superpoint_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    g = torch.full((1000,), float(rank + 1))
    parallel.allreduce_mean_(g)
    p = torch.full((10,), float(rank))
    parallel.broadcast_(p, src=0)
    q.put((rank, float(g[0]), float(p[0]), parallel.shard_batch(7)))
    dist.barrier()
    dist.destroy_process_group()


class _FakeEngine:
    """pair_step_overlapped's view of an Engine, on the CPU: a flat gradient vector with an early / late split."""

    def __init__(self, rank):
        self.grads = torch.zeros(1000)
        self.early_offset = 100
        self.rank = rank
        self.calls = []

    def pair_step(self, sample, phase=0, **kw):
        self.calls.append(phase)
        if phase == 1:
            self.grads[100:] = float(self.rank + 1)
        elif phase == 2:
            self.grads[:100] = 10.0 * (self.rank + 1)
        return torch.zeros(16)

    def adam_step(self, lr, grad_scale=None):
        self.scale = grad_scale


def _dp_diag_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from semantic_superpoint_amd import parallel
    parallel.init_from_env(backend="gloo")
    eng, diag = _FakeEngine(rank), parallel.StepDiag(every=2, cuda=False)
    for it in range(4):
        parallel.pair_step_overlapped(eng, None, 0.001, diag=diag)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([1.0 + rank], dtype=torch.float64))
    blk = parallel.dp_diagnostics(diag.summary(), [float(v) for v in per_rank],
                                  "NCCL version 2.22.3\nrank 0 AllReduce: 6537304 Bytes -> Algo 1 proto 2 time 75.0\nConnected all rings")
    q.put((rank, float(eng.grads[0]), float(eng.grads[999]), eng.calls[:2], eng.scale, blk))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_step_and_its_self_diagnosis_gloo_world2():
    """parallel.pair_step_overlapped over gloo with 2 CPU processes (a stand-in engine): both buckets are summed over the ranks,
    Adam gets 1 / world, and the N > 1 bench line's `dp` block carries the keys the first multi-GPU run is read by (exposed all-reduce
    time, phase-2 time, per-rank step times, RCCL's algorithm / protocol, the expected values of DESIGN.md section 6)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_diag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g_late, g_early, calls, scale, blk in res:
        assert g_late == 30.0 and g_early == 3.0 and calls == [1, 2] and scale == 0.5
        assert set(blk) == {"allreduce_exposed_ms", "phase2_ms", "bracketed_steps", "ms_per_step_ranks", "rccl", "expected"}
        assert blk["bracketed_steps"] == 2 and blk["allreduce_exposed_ms"] >= 0 and blk["phase2_ms"] >= 0
        assert blk["ms_per_step_ranks"] == {"min": 1.0, "max": 2.0, "all": [1.0, 2.0]}
        assert blk["rccl"] == {"version": "2.22.3", "choices": {"6537304 B": "Ring/Simple x 1"}, "connected": ["rings"]}
        assert "allreduce_exposed_ms" in blk["expected"] and "weak_scaling_efficiency_8gpu" in blk["expected"]


def test_data_parallel_helpers_gloo_world2():
    """N > 1 path on CPU: gradient bucket mean, replica broadcast and unit sharding with 2 gloo processes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 1.5          # mean of (1, 2)
    assert res[0][2] == res[1][2] == 0.0          # broadcast from rank 0
    assert res[0][3] == (0, 4) and res[1][3] == (4, 7)


def test_bench_gpus_flag_is_checked_and_spawns_ranks():
    """bench.py: --gpus must match the launcher's WORLD_SIZE; bare `--gpus 2` spawns 2 ranks itself (here, without a
    GPU, both exit with the 'needs an MI355X' message and the parent reports the failure with a non-zero code)."""
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--traffic", "none"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    if not torch.cuda.is_available():
        assert "ranks failed" in r.stderr and "needs an MI355X" in r.stderr


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_bench_export_gpus_flag_spawns_the_export_script():
    """bench_export.py --gpus 2 must start ranks of bench_export.py itself (round 2 relayed bench.py's TRAINING line): without a
    GPU both ranks die with bench_export's own message, and the parent names bench_export.py."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_export.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--views", "4", "--height", "64", "--width", "96"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench_export.py: ranks failed" in r.stderr
    if not torch.cuda.is_available():
        assert "bench_export.py needs an MI355X" in r.stderr and "bench.py needs" not in r.stderr


def test_spawn_ranks_watchdog_stops_the_survivors(tmp_path):
    """One rank dies at start-up while the other would block forever (RCCL rendezvous in real life): the parent terminates
    the survivor and returns non-zero within seconds; a healthy pair relays rank 0's stdout and returns 0."""
    import time
    import types
    b = _load_bench()
    bad = tmp_path / "ranks_bad.py"
    bad.write_text("import os, sys, time\n"
                   "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                   "print('rank0 alive', flush=True)\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = b.spawn_ranks(types.SimpleNamespace(gpus=2), script=str(bad), argv=[])
    assert rc == 1 and time.monotonic() - t0 < 60
    good = tmp_path / "ranks_good.py"
    good.write_text("import os\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                    "print('rank', os.environ['RANK'])\n")
    assert b.spawn_ranks(types.SimpleNamespace(gpus=2), script=str(good), argv=[]) == 0
    hang = tmp_path / "ranks_hang.py"
    hang.write_text("import time\ntime.sleep(600)\n")
    t0 = time.monotonic()
    assert b.spawn_ranks(types.SimpleNamespace(gpus=2), script=str(hang), argv=[], timeout=2.0) == 1
    assert time.monotonic() - t0 < 60


def test_bench_dtype_flag_selects_the_conv_algorithm():
    """bench.py --dtype: f32 = the headline Winograd fp32 path, bf16 = the bf16 path (conv algorithm 12, BASELINE configs[3])."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.parse_args([]).conv_algo == 1 and b.parse_args(["--dtype", "f32", "--conv-algo", "0"]).conv_algo == 1
    assert b.parse_args(["--dtype", "bf16"]).conv_algo == 12
    assert b.parse_args([]).arch == "ssp" and b.parse_args([]).batch == 32  # the north-star workload is the default


def test_lds_layouts_are_conflict_free_under_the_lane_group_model():
    """The LDS layouts of the pipelined Winograd conv against the ds_read_b128 lane-group model of MI355X_MICROARCH.md
    (tools/lds_conflicts.py restates the kernel's offset formulas): fragment reads and the 8x32-tile transform reads take
    the conflict-free 4 LDS cycles, the round-1 swizzle took 8 (the hardware counter agrees: profiles/r02_pmc_before_after.txt);
    round 6: also the 32x8 tiles of conv_wino_pipe_kernel and both tile shapes of conv_wino_p2_kernel."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_conflicts as M
    assert [M.cycles(M.a_frag(4, mt)) for mt in (0, 1)] == [4, 4]
    assert [M.cycles(M.a_frag(2, mt)) for mt in (0, 1)] == [8, 8]
    wide = [M.cycles(M.transform(True, w, j, ra, M.pipe_rot(True))) for w in range(2) for j in range(4) for ra in range(4)]
    narrow = [M.cycles(M.transform(False, w, j, ra, M.pipe_rot(False))) for w in range(2) for j in range(4) for ra in range(4)]
    assert max(wide) == 4 and max(narrow) == 4
    # conv_wino_p2_kernel (round 6; the round-5 counters showed 1.37 conflict cycles per LDS instruction on the 30x40 maps): the
    # 16x8 tiles need the column XOR, the 8x16 tiles are conflict-free without
    for wide_p2 in (True, False):
        assert max(M.cycles(M.p2_transform(wide_p2, j, r)) for j in range(4) for r in range(4)) == 4
    assert min(M.cycles(M.p2_transform(False, j, r, xor=False)) for j in range(4) for r in range(4)) == 8


def test_optimizer_state_dict_has_torch_adam_layout():
    """saveModel's optimizer_state_dict loads into the reference's optimizer: torch.optim.Adam(net.parameters() + [eta])."""
    from semantic_superpoint_amd import lib as L

    class FakeEngine:  # the wire format is host logic: no device needed
        pass
    arch = "SuperPointNet_gauss2"
    e = FakeEngine()
    e.layout, e.n_params = L.param_layout(arch)
    e.adam_t = 3
    e.adam_m = torch.arange(e.n_params + 3, dtype=torch.float32)
    e.adam_v = torch.arange(e.n_params + 3, dtype=torch.float32) * 2
    e.device = torch.device("cpu")
    e.params = torch.zeros(e.n_params + 3)
    osd = L.optimizer_state_dict(e, 0.001)
    from semantic_superpoint_amd import models
    net = models.SuperPointNet_gauss2()
    eta = torch.nn.Parameter(torch.tensor([1.0, 2.0, 1.0]))
    opt = torch.optim.Adam(list(net.parameters()) + [eta], lr=0.5, betas=(0.9, 0.999))
    opt.load_state_dict(osd)  # raises on a layout mismatch
    st = opt.state_dict()["state"]
    assert len(st) == len(e.layout) + 1 and float(st[0]["step"]) == 3.0
    assert opt.param_groups[0]["lr"] == 0.001
    assert torch.equal(st[len(e.layout)]["exp_avg"], e.adam_m[e.n_params:])
    e2 = FakeEngine()
    e2.layout, e2.n_params, e2.device = e.layout, e.n_params, e.device
    e2.adam_m, e2.adam_v, e2.params, e2.adam_t = torch.zeros_like(e.adam_m), torch.zeros_like(e.adam_v), torch.zeros_like(e.params), 0
    L.load_optimizer_state(e2, osd, eta=[0.1, 0.2, 0.3])
    assert torch.equal(e2.adam_m, e.adam_m) and torch.equal(e2.adam_v, e.adam_v) and e2.adam_t == 3
    assert torch.allclose(e2.params[e.n_params:], torch.tensor([0.1, 0.2, 0.3]))


def test_scaled_homographies_and_device_warp_order_match_the_reference_ops():
    """Index parity: (a) lib.scaled_homographies == the oracle's restatement of
    scale_homography_torch bit for bit; (b) the device's warp order fma(p2, 1, fma(p1, y, p0 * x)) followed by a
    correctly rounded division is exactly what torch's CPU `homographies @ points^T` + `/` give (N >= 12 points: below
    that oneMKL takes another path) - emulated here in float64 with explicit float32 roundings."""
    from semantic_superpoint_amd import lib as L
    rs = np.random.RandomState(3)
    Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(6)]).astype(np.float32))
    for (hh, ww) in ((30, 40), (240, 320), (60, 80)):
        mine = L.scaled_homographies(Hs, hh, ww)
        for i in range(Hs.shape[0]):
            assert torch.equal(mine[i], C.scale_homography(Hs[i], (hh, ww)))
    f32 = lambda v: v.float().double()  # noqa: E731
    for n in (12, 230, 1200, 4800):
        pts = torch.cat([torch.from_numpy(rs.randint(0, 320, (n, 2))).float(), torch.ones(n, 1)], 1)
        for P in L.scaled_homographies(Hs, 240, 320):
            ref = (P.view(3, 3) @ pts.t()).t()
            ref_xy = ref[:, :2] / ref[:, 2:]
            a, x, y = P.double(), pts[:, 0].double(), pts[:, 1].double()
            rows = []
            for r in range(3):
                t = f32(a[r, 0] * x)
                t = f32(a[r, 1] * y + t)          # fma: one rounding
                rows.append(f32(t + a[r, 2]))     # fma(p2, 1, t) == t + p2 rounded once
            wx, wy = (rows[0] / rows[2]).float(), (rows[1] / rows[2]).float()
            assert torch.equal(wx, ref_xy[:, 0]) and torch.equal(wy, ref_xy[:, 1]), n


def test_f4x4_kernel_accumulators_are_private_to_its_inline_asm(tmp_path):
    """conv_wino4_kernel keeps 16 accumulators in FIXED accumulation registers a[0:255] that only its inline asm touches
    (conv_wino4.hip.h).  The compiler does not know they are occupied: if register pressure ever made it park a value in
    an accumulation register (or spill), the convolution would be silently wrong.  Compile the kernel to ISA and check:
    every accumulation-register access is one of the asm forms (bracket syntax a[N]), nothing goes to scratch, and the
    kernel descriptor reserves all 256 accumulation registers behind the vector registers."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semantic-superpoint_amd", "csrc")
    tu = tmp_path / "tu.hip"
    tu.write_text('#include <hip/hip_runtime.h>\n#include "conv_wino4.hip.h"\n' + "".join(
        "template __global__ void sspk::conv_wino4_kernel<%d, %s>(const sspk::ConvArgs);\n" % (m, w)
        for m in (0, 1) for w in ("true", "false")))
    out = tmp_path / "tu.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-munsafe-fp-atomics",
                    "-I", csrc, str(tu), "-o", str(out)], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN4sspk17conv_wino4_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) == 4
    for name, body in kernels:
        assert not re.search(r"v_accvgpr_(write|read)_b32 [^\n]*\ba\d+\b", body), name   # compiler-allocated aN
        assert "v_accvgpr_mov" not in body and "scratch_" not in body, name
        assert len(re.findall(r"v_mfma_f32_32x32x2_f32 a\[", body)) == 208, name         # (2 task kinds x 40 + 24) x 2 stage instances: hipbuild.FIXED_AGPR_KERNELS
        assert len(re.findall(r"v_mfma_f32_32x32x2_f32 v\[", body)) == 16, name          # the pair in vector registers
    for accum, nxt in zip(re.findall(r"\.amdhsa_accum_offset (\d+)", text), re.findall(r"\.amdhsa_next_free_vgpr (\d+)", text)):
        if int(nxt) > 256:  # the four conv_wino4_kernel instances
            assert int(nxt) == int(accum) + 256


def test_shipped_binary_keeps_the_accumulation_register_contract():
    """hipbuild.verify_binary on the in-tree .so (the file that ships with the gpurun snapshot): the gfx950 code object is
    unbundled and disassembled; conv_wino4_kernel must contain exactly its inline asm's accumulation-register instructions
    (128 MFMAs, 2 x 256 clears, 2 x 256 reads; wgrad_wino4_kernel: 256 / 256 / 256), no scratch, no spilled vector registers, 256 reserved accumulation registers.
    A tampered contract (one instruction less expected) must be rejected."""
    from semantic_superpoint_amd import hipbuild
    if not os.path.exists(os.path.join(hipbuild.LLVM_BIN, "llvm-objdump")):
        pytest.skip("llvm binutils not available")
    lib = hipbuild.build()
    rep = hipbuild.verify_binary(lib)
    assert len([k for k in rep if "conv_wino4_kernel" in k]) == 4 and len([k for k in rep if "wgrad_wino4_kernel" in k]) == 4
    assert hipbuild.verified(lib)
    wrong = {"conv_wino4_kernel": dict(hipbuild.FIXED_AGPR_KERNELS["conv_wino4_kernel"], v_accvgpr_read_b32=511)}
    with pytest.raises(RuntimeError, match="differ from the inline-asm contract"):
        hipbuild.verify_binary(lib, expected=wrong)
