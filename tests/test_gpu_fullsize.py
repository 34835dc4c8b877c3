"""GPU: the pair step at the BENCHMARK size (240x320; B = 2 against the real reference's golden step, B = 32 through
size-independent properties), the proof that end-to-end gradient differences are ReLU / max-pool gate flips only,
the split (data-parallel overlap) and captured (hipGraph) forms of the step, and the loss gradients G3 / G5."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import cpu_ref as C
from tests import golden_util as G

pytestmark = pytest.mark.gpu
ARCHS = {"sp": "SuperPointNet_gauss2", "ssp": "SuperPointNet_gauss2_ssmall"}
SCALARS = ("loss", "loss_det", "loss_det_warp", "loss_desc", "loss_sem", "loss_sem_warp", "positive_dist", "negative_dist")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


def _engine(arch, B, H, W, sd, **kw):
    from semantic_superpoint_amd.lib import Engine
    e = Engine(arch, B, H, W, _dev(), **kw)
    e.load_state_dict(sd)
    return e


def _to_dev(sample):
    return {k: v.to(_dev()).contiguous() for k, v in sample.items()}


def _idx_to_dev(idx, Wc):
    ma = torch.stack([(i["uv_a"][:, 0] + i["uv_a"][:, 1] * Wc) for i in idx]).to(torch.int32)
    mb = torch.stack([(i["uv_b"][:, 0] + i["uv_b"][:, 1] * Wc) for i in idx]).to(torch.int32)
    nm = torch.stack([i["nm_b"] for i in idx]).to(torch.int32)
    return ma.to(_dev()).contiguous(), mb.to(_dev()).contiguous(), nm.to(_dev()).contiguous()


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _noisy(arch):  # conv biases feeding a BatchNorm: exact gradient 0, both sides hold rounding noise
    return {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}


# ------------------------------------------------------------------------------------------------
# (i) the real reference's step at 240x320 (G12)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", [1, 10])
@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_full_size_step_golden(tag, algo):
    """G12: forward checksums, two optimizer steps and the first step's gradients of the REAL reference at 240x320, B = 2.
    algo 1 = the default predicate (at B = 2 every 3x3 layer runs F(2x2,3x3): 600 tile-block items < 1024); algo 10 = the
    BENCHMARKED kernel set: conv_wino4_kernel (Winograd F(4x4,3x3)) with its in-step epilogues - pooled raw output, fused
    BatchNorm-backward sums, two views per launch, XCD split, nt stores - on every 3x3 layer of the reference's own step."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch = ARCHS[tag]
    g = G.load("g12_step_%s_240x320.npz" % tag)
    sample = C.compact_from_npz(g)
    B, _, H, W = sample["image"].shape
    assert (H, W) == (240, 320)
    sd = C.init_state_dict(arch, seed=29)
    ds = _to_dev(sample)
    # forward: per-channel checksums (sums over 1200 cells / 76800 pixels) and strided slices
    e = _engine(arch, B, H, W, sd, with_grad=False)
    e.set_conv_algo(algo)
    want = ("semi", "desc", "sem") if tag == "ssp" else ("semi", "desc")
    o = e.forward(ds["image"], slot=0, train=True, want=want)
    torch.cuda.synchronize()
    assert (o["semi"].cpu()[:, ::4, ::3, ::4] - torch.from_numpy(g["fwd/semi_s"])).abs().max() < 1e-3
    assert (o["desc"].cpu()[:, ::16, ::3, ::4] - torch.from_numpy(g["fwd/desc_s"])).abs().max() < 1e-3
    cs = o["semi"].double().sum(dim=(2, 3)).cpu()
    assert (cs - torch.from_numpy(g["fwd/semi_chsum"])).abs().max() < 1e-3 * 1200 ** 0.5 * 4
    cs = o["desc"].double().sum(dim=(2, 3)).cpu()
    assert (cs - torch.from_numpy(g["fwd/desc_chsum"])).abs().max() < 1e-3
    if tag == "ssp":
        assert (o["sem"].cpu()[:, ::19, ::24, ::32] - torch.from_numpy(g["fwd/sem_s"])).abs().max() < 1e-3
        cs = o["sem"].double().sum(dim=(2, 3)).cpu()
        ref = torch.from_numpy(g["fwd/sem_chsum"])
        assert (cs - ref).abs().max() < 2e-5 * 76800, float((cs - ref).abs().max())
    del e
    # the step
    e = _engine(arch, B, H, W, sd)
    e.set_conv_algo(algo)
    idx = _idx_to_dev(G.indices_from(g, "idx/", B), W // 8)
    for it in range(2):
        e.zero_grad()
        sc = e.pair_step(ds, indices=idx, train=True, lambda_loss=1.0, lamda_d=1.0, multi_task=True)
        torch.cuda.synchronize()
        sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
        for name in SCALARS:
            ref = float(g["step%d/%s" % (it, name)])
            assert abs(sc[name] - ref) < (2e-4 if it == 0 else 1e-3) * max(1.0, abs(ref)), (it, name, sc[name], ref)
        if it == 0:
            gd = e.grad_dict()
            worst = 0.0
            for k in C.param_keys(arch):
                if k in _noisy(arch):
                    continue
                n_ref = float(g["grad_norm/" + k])
                mine = gd[k].cpu().reshape(-1)
                assert abs(float(mine.norm()) - n_ref) < 2e-3 * n_ref + 1e-7, (k, float(mine.norm()), n_ref)
                sl = torch.from_numpy(g["grad_slice/" + k])
                err = float((mine[:64] - sl).abs().max()) / (float(mine.abs().max()) + 1e-30)
                worst = max(worst, err)
                # measured worst slice error (GPU box, round 4): algorithm 1 5.6e-3 (sp) / 4.8e-3 (ssp), algorithm 10 (F(4x4,3x3)
                # everywhere) 1.5e-2 / 9.1e-3 -> 2x the measured value resp. the former 2e-2 where that is tighter
                assert err < (1.2e-2 if algo == 1 else 2e-2), (k, err)
            print("G12 %s algo %d: worst 64-element slice error %.2e of max|grad|" % (tag, algo, worst))
            assert (gd["eta"].cpu() - torch.from_numpy(g["grad/eta"])).abs().max() < 2e-4
        e.adam_step(0.001)
    torch.cuda.synchronize()
    assert (e.eta.cpu() - torch.from_numpy(g["post/eta"])).abs().max() < 1e-5


# ------------------------------------------------------------------------------------------------
# (iii) gate flips are the ONLY source of the end-to-end gradient differences
# ------------------------------------------------------------------------------------------------
def _hip_gates(e, arch, slot, B, H, W):
    """ReLU gates and max-pool winners of the HIP forward in `slot`, recomputed from its raw convolution outputs and
    BatchNorm affine (float64 gives the exact sign of the fp32 fma; rounding to fp32 reproduces the pooled values)."""
    t = C.layer_table(arch)
    relu, pool = {}, {}
    nheads = 3 if arch.endswith("ssmall") else 2
    res = [(H, W), (H, W), (H // 2, W // 2), (H // 2, W // 2), (H // 4, W // 4), (H // 4, W // 4), (H // 8, W // 8),
           (H // 8, W // 8)]
    for l in range(8):
        hh, ww = res[l]
        c = t[l][3]
        y = e.debug_buffer(slot, "Y%d" % l, (B, hh, ww, c)).double()
        z = (y * e.debug_buffer(slot, "scale%d" % l, (c,)).double() + e.debug_buffer(slot, "shift%d" % l, (c,)).double())
        z = z.permute(0, 3, 1, 2).cpu()  # NCHW
        relu[t[l][0]] = (z > 0)
        if l in (1, 3, 5):  # pooled on the way into layer l + 1
            a = torch.relu(z.float())
            win = a.view(B, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, c, hh // 2, ww // 2, 4)
            pool[l + 1] = win.argmax(dim=4)  # first maximum, like torch's max_pool2d and the HIP routing
    hc, wc = H // 8, W // 8
    yh = e.debug_buffer(slot, "Y8", (B, hc, wc, 256 * nheads)).double()
    for k, (name, l) in enumerate((("convPa", 8), ("convDa", 10), ("convDS", 12))[:nheads]):
        z = yh[..., 256 * k:256 * (k + 1)] * e.debug_buffer(slot, "scale%d" % l, (256,)).double() + \
            e.debug_buffer(slot, "shift%d" % l, (256,)).double()
        relu[name] = (z.permute(0, 3, 1, 2).cpu() > 0)
    return {"relu": relu, "pool": pool}


def _gate_flip_case(arch, B, H, W, sd, sample, used, algo, loss_kw, plain_grads, forced_tol=1e-4, plain_tol=5e-3):
    """(a) HIP vs the plain oracle gradients `plain_grads`: statistical agreement (gate flips of activations within rounding
    distance of 0 perturb the gradient).  (b) HIP vs the oracle evaluated WITH THE HIP PATH'S ReLU gates and max-pool
    winners: agreement to `forced_tol` relative L2 per tensor (and 10 x that per element of max|ref|) -> the flips are the
    whole difference.  Returns (worst plain, worst forced, worst 64-element slice error against the forced oracle)."""
    single = "warped_img" not in sample
    nv = 1 if single else 2
    e = _engine(arch, B, H, W, sd)
    if algo is not None:
        e.set_conv_algo(algo)
    e.zero_grad()
    e.pair_step(_to_dev(sample), indices=None if used is None else _idx_to_dev(used, W // 8), train=True, **loss_kw)
    torch.cuda.synchronize()
    gd = {k: v.cpu().clone() for k, v in e.grad_dict().items()}
    forced = tuple(_hip_gates(e, arch, v, B, H, W) for v in range(nv))
    nflip = ngates = 0
    for v in range(nv):
        for k, z in _oracle_preacts(sd, sample, arch, v).items():
            nflip += int((forced[v]["relu"][k] != (z > 0)).sum())
            ngates += z.numel()
    tsd = C.to_torch(sd, requires_grad=True)
    eta = torch.tensor([1.0, 2.0, 1.0], requires_grad=True)
    okw = {k: v for k, v in loss_kw.items() if k in ("lambda_loss", "lamda_d", "multi_task", "gaussian")}
    loss, _, _ = C.pair_losses(tsd, eta, sample, arch, indices=used, forced=forced, warped_pair=not single, **okw)
    loss.backward()
    worst_plain, worst_forced, worst_slice = (0.0, ""), (0.0, ""), (0.0, "")
    for k in C.param_keys(arch):
        if k in _noisy(arch) or tsd[k].grad is None:
            continue
        l2p, _ = _rel(gd[k], plain_grads[k])
        l2f, mxf = _rel(gd[k], tsd[k].grad)
        sl = float((gd[k].reshape(-1)[:64] - tsd[k].grad.reshape(-1)[:64]).abs().max() / (tsd[k].grad.abs().max() + 1e-30))
        worst_plain, worst_forced = max(worst_plain, (l2p, k)), max(worst_forced, (max(l2f, 0.1 * mxf), k))
        worst_slice = max(worst_slice, (sl, k))
    print("%s %dx%d algo %s%s: gate flips %d of %d; worst rel-L2: plain %.2e (%s), gates forced %.2e (%s); worst 64-element "
          "slice vs the forced oracle %.2e of max|grad| (%s)"
          % (arch, H, W, algo, " single view" if single else "", nflip, ngates, worst_plain[0], worst_plain[1], worst_forced[0],
             worst_forced[1], worst_slice[0], worst_slice[1]))
    assert worst_plain[0] <= plain_tol, ("plain oracle", worst_plain, "flipped gates: %d" % nflip)
    assert worst_forced[0] <= forced_tol, ("gates forced", worst_forced, "flipped gates: %d" % nflip)
    if eta.grad is not None:
        assert (gd["eta"] - eta.grad).abs().max() < 1e-5
    return worst_plain, worst_forced, worst_slice


@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_gradient_differences_are_gate_flips_only(tag):
    """120x160, B = 2, default kernels (every 3x3 layer on F(2x2,3x3) at this size)."""
    arch = ARCHS[tag]
    B, H, W = 2, 120, 160
    sd = C.init_state_dict(arch, seed=9)
    sample = C.make_synthetic_pair(B, H, W, seed=4, semantic=(tag == "ssp"), kp_prob=0.005)
    tr = C.Trainer(arch, sd, lr=0.001)
    tr.real_batch_size = 10 ** 9
    np.random.seed(50)
    torch.manual_seed(60)
    tr.train_val_sample(sample, n_iter=0, train=True)
    _gate_flip_case(arch, B, H, W, sd, sample, tr.aux["indices"], None, {}, tr.last_grads)


@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_gate_flips_only_under_the_benchmarked_kernels_240x320(tag):
    """The same proof where conv_wino4_kernel (Winograd F(4x4,3x3), 36 % of the fp32 step) really runs: algorithm 10 on the
    G12 inputs (240x320, B = 2, the reference's own sampled indices).  The G12 slice differences against the REAL reference
    (test_full_size_step_golden: up to 1.5e-2 of max|grad|) are thereby gate flips, not kernel error: against the oracle
    evaluated with the HIP path's gates every tensor agrees to 1e-4 and every 64-element slice to 2 x the measured residual."""
    arch = ARCHS[tag]
    g = G.load("g12_step_%s_240x320.npz" % tag)
    sample = C.compact_from_npz(g)
    B, _, H, W = sample["image"].shape
    sd = C.init_state_dict(arch, seed=29)
    used = G.indices_from(g, "idx/", B)
    tr = C.Trainer(arch, sd, lr=0.001, lambda_loss=1.0, multi_task=True)
    tr.real_batch_size = 10 ** 9
    tr.train_val_sample(sample, n_iter=1, train=True, indices=used)
    for k in C.param_keys(arch):  # the oracle's plain gradients ARE the reference's (fixture slices)
        if k not in _noisy(arch):
            r = torch.from_numpy(g["grad_slice/" + k])
            assert (tr.last_grads[k].reshape(-1)[:64] - r).abs().max() <= 1e-3 * float(tr.last_grads[k].abs().max()) + 1e-7, k
    _, _, ws = _gate_flip_case(arch, B, H, W, sd, sample, used, 10, dict(lambda_loss=1.0, lamda_d=1.0, multi_task=True),
                               tr.last_grads, forced_tol=5e-5, plain_tol=2e-2)
    assert ws[0] <= FORCED_SLICE_TOL_W4, ws


# 64-element slices of the algorithm-10 gradients against the forced-gate oracle at 240x320: 2 x the residual measured on the GPU box
FORCED_SLICE_TOL_W4 = 3e-5  # measured 1.44e-5 (sp), 1.18e-5 (ssp); per-tensor rel-L2 2.34e-5


def _oracle_preacts(sd, sample, arch, view):
    """Pre-activation signs of the oracle's own forward (to count the flipped gates)."""
    import torch.nn.functional as F
    tsd = C.to_torch(sd)
    x = sample["image"] if view == 0 else sample["warped_img"]
    t = C.layer_table(arch)
    out, h = {}, x
    with torch.no_grad():
        for i, (conv, bn, cin, cout, k) in enumerate(t[:8]):
            if i in (2, 4, 6):
                h = F.max_pool2d(h, 2)
            y = F.conv2d(h, tsd[conv + ".weight"], tsd[conv + ".bias"], padding=1)
            z = F.batch_norm(y, None, None, tsd[bn + ".weight"], tsd[bn + ".bias"], training=True, eps=1e-5)
            out[conv] = z
            h = F.relu(z)
        for conv, bn in (("convPa", "bnPa"), ("convDa", "bnDa"), ("convDS", "bnS1")):
            if conv + ".weight" in tsd:
                y = F.conv2d(h, tsd[conv + ".weight"], tsd[conv + ".bias"], padding=1)
                out[conv] = F.batch_norm(y, None, None, tsd[bn + ".weight"], tsd[bn + ".bias"], training=True, eps=1e-5)
    return out


# ------------------------------------------------------------------------------------------------
# (ii) properties at the benchmarked size B = 32, 240x320
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_bench_size_properties(tag):
    """B = 32 at 240x320 (BASELINE configs[1] / [2]): the persistent XCD-aware tile loops, the skewed 5 GB workspace,
    both-views-per-launch and the deferred weight-gradient slabs take their real shape only here.
      * Winograd (default) and direct implicit GEMM agree on every scalar and on the flat gradient;
      * BatchNorm running statistics of two sampled channels per checked layer match a B = 32 oracle forward;
      * 20 optimizer steps on one batch: every scalar finite, the loss falls."""
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import SCALAR_NAMES, layer_table
    arch = ARCHS[tag]
    B, H, W = 32, 240, 320
    dev = _dev()
    sd = synth.default_init_state_dict(layer_table(arch), seed=0)
    sample = synth.make_pair(B, H, W, dev, seed=100, semantic=(tag == "ssp"))
    e = _engine(arch, B, H, W, sd)
    res = {}
    for algo in (1, 0):
        e.set_conv_algo(algo)
        e.load_state_dict(sd)  # also resets the running statistics
        e.zero_grad()
        sc = e.pair_step(sample, indices=None, seed=7, train=True)
        torch.cuda.synchronize()
        res[algo] = (dict(zip(SCALAR_NAMES, sc.cpu().tolist())), e.grads.clone(), e.state_dict())
    e.set_conv_algo(1)
    for name in SCALARS:
        a, b = res[1][0][name], res[0][0][name]
        assert np.isfinite(a) and abs(a - b) < 1e-4 * max(1.0, abs(a)), (name, a, b)
    assert bool(torch.isfinite(res[1][1]).all())
    l2, mx = _rel(res[1][1], res[0][1])
    assert l2 < 3e-3 and mx < 3e-2, (l2, mx)  # different summation orders + gate flips, 157 M activations per layer
    # running statistics after ONE forward of the first view... the pair step runs both views: the running buffers hold
    # momentum-0.1 updates of view 0 then view 1 (Train_model_heatmap_all.py:258,262); the oracle does the same
    osd = C.to_torch({k: np.asarray(v.cpu()) if torch.is_tensor(v) else np.asarray(v) for k, v in sd.items()})
    with torch.no_grad():
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        C.forward(osd, sample["image"].cpu(), arch)
        C.forward(osd, sample["warped_img"].cpu(), arch)
    mine = res[1][2]
    for k in ("inc.conv.conv.1", "inc.conv.conv.4", "down1.mpconv.1.conv.4", "down3.mpconv.1.conv.4", "bnPa", "bnDb"):
        for ch in (3, 41):
            for stat in ("running_mean", "running_var"):
                a, b = float(mine[k + "." + stat][ch]), float(osd[k + "." + stat][ch])
                assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (k, stat, ch, a, b)
    # soak
    e.load_state_dict(sd)
    first = last = None
    for it in range(20):
        e.zero_grad()
        sc = e.pair_step(sample, indices=None, seed=1000 + it, train=True)
        e.adam_step(0.001)
        if it in (0, 19):
            v = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
            assert all(np.isfinite(x) for x in v.values()), v
            first, last = (v, last) if it == 0 else (first, v)
    assert last["loss"] < first["loss"] - 0.05, (first["loss"], last["loss"])
    assert bool(torch.isfinite(e.params).all())


def _oracle_indices(idx, Wc):
    """Device-sampled (match_a, match_b, nonmatch_b) -> the per-image index dicts of cpu_ref.Trainer."""
    ma, mb, nm = (t.cpu().long() for t in idx)
    out = []
    for i in range(ma.shape[0]):
        out.append({"uv_a": torch.stack((ma[i] % Wc, ma[i] // Wc), dim=1).float(),
                    "uv_b": torch.stack((mb[i] % Wc, mb[i] // Wc), dim=1).float(), "nm_b": nm[i]})
    return out


def _np_sd(sd):
    return {k: (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}


def _compare_step_with_oracle(tag, e, sd, sample, sc, idx, scal_tol, norm_tol, flat_tol, what):
    """One oracle step (cpu_ref.Trainer) on the same inputs and the same sparse-loss indices: 8 scalars, per-tensor
    gradient norms, the flat gradient and the eta gradient."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch = ARCHS[tag]
    W = sample["image"].shape[-1]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    tr = C.Trainer(arch, _np_sd(sd), lr=0.001)
    tr.real_batch_size = 10 ** 9  # gradients only
    cpu = {k: v.cpu() for k, v in sample.items() if k != "cell_homographies"}
    tr.train_val_sample(cpu, n_iter=0, train=True, indices=_oracle_indices(idx, W // 8))
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for name in SCALARS:
        ref = tr.scalar_dict[name]
        assert abs(sc[name] - ref) < scal_tol * max(1.0, abs(ref)), (what, name, sc[name], ref)
    gd = e.grad_dict()
    mine_flat, ref_flat, worst = [], [], (0.0, "")
    for k in C.param_keys(arch):
        if k in _noisy(arch):
            continue
        mine, ref = gd[k].cpu().reshape(-1).double(), tr.last_grads[k].reshape(-1).double()
        n_ref = float(ref.norm())
        assert abs(float(mine.norm()) - n_ref) < norm_tol * n_ref + 1e-7, (what, k, float(mine.norm()), n_ref)
        worst = max(worst, (float((mine - ref).norm()) / (n_ref + 1e-30), k))
        mine_flat.append(mine), ref_flat.append(ref)
    l2, mx = _rel(torch.cat(mine_flat), torch.cat(ref_flat))
    print("%s: flat gradient rel-L2 %.2e (max %.2e), worst tensor %.2e (%s)" % (what, l2, mx, worst[0], worst[1]))
    assert l2 < flat_tol, (what, l2)
    assert (gd["eta"].cpu() - tr.last_grads["eta"]).abs().max() < 1e-5, what
    _compare_step_with_oracle.last_trainer = tr   # (the oracle's forward outputs of both views: tr.aux)
    return worst


@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_bench_size_step_vs_oracle(tag):
    """THE benchmarked configuration against the oracle: B = 32, 240x320 (BASELINE configs[1] / [2]), default algorithm -
    conv_wino4_kernel on the 240x320 / 120x160 / 60x80 layers with nprob = 2, fused BatchNorm-backward sums, pooled raw
    outputs, the XCD split and nt stores, F(2x2,3x3) on the 30x40 layers, F(3x3,2x2) / F(3x3,4x4) weight gradients -, inputs
    of bench.py's generator, DEVICE-sampled indices read back and fed to cpu_ref.Trainer (one oracle step ~ 15 s on the
    GPU host).  8 scalars <= 2e-4, per-tensor gradient norm <= 2e-3, flat gradient rel-L2 <= 3e-3, eta gradient 1e-5."""
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import layer_table
    arch = ARCHS[tag]
    B, H, W = 32, 240, 320
    sd = synth.default_init_state_dict(layer_table(arch), seed=0)
    sample = synth.make_pair(B, H, W, _dev(), seed=100, semantic=(tag == "ssp"))
    e = _engine(arch, B, H, W, sd)
    e.zero_grad()
    sc = e.pair_step(sample, indices=None, seed=7, train=True).clone()
    torch.cuda.synchronize()
    _compare_step_with_oracle(tag, e, sd, sample, sc, e._last_idx, 2e-4, 2e-3, 3e-3, "B=32 240x320 %s" % tag)
    # the north star's literal criterion at the benchmark size: detector logits and descriptors of BOTH views, element-wise,
    # within 1e-3 of the oracle's B = 32 forward (the step above did not move the weights: no optimizer step)
    tr = _compare_step_with_oracle.last_trainer
    for view, key in (("image", "out"), ("warped_img", "out_warp")):
        o = e.forward(sample[view], slot=0, train=True, want=("semi", "desc"))
        torch.cuda.synchronize()
        for name in ("semi", "desc"):
            err = float((o[name].cpu() - tr.aux[key][name].detach()).abs().max())
            print("B=32 240x320 %s %s %s: max |HIP - oracle| %.2e" % (tag, view, name, err))
            assert err < 1e-3, (view, name, err)


def test_default_predicate_mixes_kernels_below_max_batch():
    """Engine sized for 6 pairs, step of 4 at 240x320: the default predicate sends the two 240x320 layers through
    conv_wino4_kernel (2 x 4 x 150 = 1200 tile-block items >= 1024) and everything below through F(2x2,3x3) (320 / 80 items),
    i.e. the mixed per-layer weight-image layouts of ONE step (ssp_handle::pk_w4_*) with B < max_batch, against the oracle.
    A second step of 2 pairs on the same engine (no layer eligible any more) must re-pack and agree as well."""
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import layer_table
    tag, arch = "ssp", ARCHS["ssp"]
    H, W = 240, 320
    sd = synth.default_init_state_dict(layer_table(arch), seed=2)
    e = _engine(arch, 6, H, W, sd)
    for B, seed in ((4, 31), (2, 32)):
        sample = synth.make_pair(B, H, W, _dev(), seed=seed, semantic=True)
        e.load_state_dict(sd)
        e.zero_grad()
        sc = e.pair_step(sample, indices=None, seed=seed, train=True).clone()
        torch.cuda.synchronize()
        _compare_step_with_oracle(tag, e, sd, sample, sc, e._last_idx, 2e-4, 3e-3, 5e-3, "default predicate, B=%d of 6" % B)


def test_backward_refuses_stale_weight_images():
    """The packed weight images belong to the last forward of the handle: changing the conv algorithm between a forward and
    its backward, or back-propagating after an inference-only pack, must fail loudly instead of running a kernel on an image
    of another layout."""
    arch = ARCHS["sp"]
    B, H, W = 2, 64, 96
    sd = C.init_state_dict(arch, seed=3)
    e = _engine(arch, B, H, W, sd)
    x = torch.rand(B, 1, H, W, device=_dev())
    out = e.forward(x, slot=0, train=True)
    e.set_conv_algo(10)
    with pytest.raises(RuntimeError, match="conv algorithm changed"):
        e.backward(0, torch.zeros_like(out["semi"]), torch.zeros_like(out["desc"]), None)
    e.set_conv_algo(1)
    # forward of ANOTHER shape on the other slot, then the backward of slot 0: the record, not the shape, picks the kernels.
    # Reference = the oracle evaluated WITH the HIP forward's own ReLU gates and max-pool winners (read back from slot 0): on an
    # 8x12-cell map one flipped gate is 5 % of a gradient, with the gates forced the comparison is exact to fp32 rounding.
    e.forward(x, slot=0, train=True)
    torch.cuda.synchronize()
    forced = _hip_gates(e, arch, 0, B, H, W)
    tsd = C.to_torch(sd, requires_grad=True)
    ref = C.forward(tsd, x.cpu(), arch, forced=forced)
    gs = {k: torch.randn_like(ref[k]) for k in ref}
    sum((ref[k] * gs[k]).sum() for k in ref).backward()
    e.forward(torch.rand(1, 1, 32, 48, device=_dev()), slot=1, train=True)
    e.zero_grad()
    e.backward(0, gs["semi"].to(_dev()), gs["desc"].to(_dev()), None)
    torch.cuda.synchronize()
    gd = e.grad_dict()
    for k in ("convPa.weight", "down2.mpconv.1.conv.3.weight", "inc.conv.conv.3.weight"):
        l2, _ = _rel(gd[k].cpu(), tsd[k].grad)
        assert l2 < 1e-4, (k, l2)


def test_loaded_binary_keeps_the_accumulation_register_contract():
    """The .so THIS process loaded (on the GPU box: the one shipped with the snapshot) is disassembled and checked:
    conv_wino4_kernel's fixed accumulation registers are touched by its inline asm only, no scratch, no spills."""
    from semantic_superpoint_amd import hipbuild
    from semantic_superpoint_amd.lib import load_library
    lib = load_library()
    rep = hipbuild.verify_binary(lib._name)
    assert len([k for k in rep if "conv_wino4_kernel" in k]) == 4
    assert hipbuild.verified(lib._name)


# ------------------------------------------------------------------------------------------------
# split and captured forms of the step
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["sp", "ssp"])
def test_phases_equal_whole_step(tag):
    """ssp_pair_step_phase 1 + 2 == ssp_pair_step, and after phase 1 the early bucket is already final.  (Two runs of
    the step are not bit-identical: the fp32 / fp64 atomics of the statistics and of the descriptor-loss gradient commit
    in a different order; 1e-5 relative is that noise.)"""
    arch = ARCHS[tag]
    B, H, W = 2, 64, 96
    sd = C.init_state_dict(arch, seed=3)
    ds = _to_dev(C.make_synthetic_pair(B, H, W, seed=8, semantic=(tag == "ssp"), kp_prob=0.01))
    e = _engine(arch, B, H, W, sd)
    idx = e.sample_indices(ds["homographies"], 5)
    e.zero_grad()
    s0 = e.pair_step(ds, indices=idx, train=True).clone()
    g0 = e.grads.clone()
    e.load_state_dict(sd)
    e.zero_grad()
    s1 = e.pair_step(ds, indices=idx, train=True, phase=1).clone()
    torch.cuda.synchronize()
    off = e.early_offset
    assert 0 < off < e.n_params
    early = e.grads[off:].clone()
    l2, mx = _rel(early, g0[off:])
    assert l2 < 1e-5 and mx < 1e-4, ("early bucket after phase 1", l2, mx)
    assert float(e.grads[:off].abs().max()) == 0.0
    e.pair_step(ds, indices=idx, train=True, phase=2)
    torch.cuda.synchronize()
    assert torch.equal(e.grads[off:], early), "phase 2 must not touch the early bucket (it is being all-reduced)"
    l2, mx = _rel(e.grads[:off], g0[:off])
    assert l2 < 1e-5 and mx < 1e-4, ("late bucket", l2, mx)
    assert (s0 - s1).abs().max() < 1e-5


def test_graph_replay_equals_eager():
    """ssp_pair_step_graph: capture + three replays with different sampler seeds == the eager steps with the same seeds.
    Both engines start every step from the SAME state (two training runs drift apart on their own: Adam turns the
    commit-order noise of the atomics into parameter differences of up to lr per element)."""
    arch = ARCHS["ssp"]
    B, H, W = 2, 64, 96
    sd = C.init_state_dict(arch, seed=3)
    ds = _to_dev(C.make_synthetic_pair(B, H, W, seed=8, semantic=True, kp_prob=0.01))
    ea, eb = _engine(arch, B, H, W, sd), _engine(arch, B, H, W, sd)
    st = torch.cuda.Stream()
    seen_idx = []
    for it, seed in enumerate((11, 12, 99, 12)):
        for name in ("params", "adam_m", "adam_v", "bn_running", "nbt"):
            getattr(eb, name).copy_(getattr(ea, name))
        eb.adam_t = ea.adam_t
        torch.cuda.synchronize()
        ea.zero_grad()
        sa = ea.pair_step(ds, indices=None, seed=seed, train=True).clone()
        ea.adam_step(0.001)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            eb.zero_grad()
            sb = eb.pair_step(ds, indices=None, seed=seed, train=True, graph=True).clone()
            eb.adam_step(0.001)
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        seen_idx.append(eb._graph_idx[2].cpu().clone())
        assert bool(torch.isfinite(ea.grads).all()), it
        bad = [(k, float(v.abs().max())) for k, v in eb.grad_dict().items() if not bool(torch.isfinite(v).all()) or float(v.abs().max()) > 1e3]
        assert not bad, (it, bad[:10], len(bad))
        assert (sa - sb).abs().max() < 1e-5 * max(1.0, float(sa.abs().max())), (it, sa, sb)
        l2, mx = _rel(ea.grads, eb.grads)
        assert l2 < 1e-5 and mx < 1e-4, (it, l2, mx)  # atomics: not bit-reproducible between two runs
        l2, _ = _rel(ea.bn_running, eb.bn_running)
        assert l2 < 1e-6, (it, l2)
    # the sampler seed lives in device memory: replays with another seed draw other indices, the same seed the same
    assert not torch.equal(seen_idx[0], seen_idx[1]) and torch.equal(seen_idx[1], seen_idx[3])


# ------------------------------------------------------------------------------------------------
# data parallel: two ranks through the engine on ONE device (gloo), overlapped all-reduce
# ------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, out_dir, algo=1):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semantic_superpoint_amd import lib as L, parallel
    from semantic_superpoint_amd.lib import Engine
    arch, B, H, W = ARCHS["ssp"], 2, 64, 96
    dev = torch.device("cuda:0")
    sd = C.init_state_dict(arch, seed=3)
    L.set_deterministic(algo == 12)  # bit-reproducible accumulation: the bf16 path's sum can then be checked like the fp32 one
    e = Engine(arch, B, H, W, dev)
    e.set_conv_algo(algo)
    e.load_state_dict(sd)
    ds = {k: v.to(dev).contiguous() for k, v in C.make_synthetic_pair(B, H, W, seed=20 + rank, semantic=True, kp_prob=0.01).items()}
    for it in range(2):
        e.zero_grad()
        parallel.pair_step_overlapped(e, ds, 0.001, indices=None, seed=100 * it + rank, train=True)
        if it == 0:
            torch.cuda.synchronize()
            torch.save(e.grads.cpu(), os.path.join(out_dir, "gsum%d.pt" % rank))
    torch.cuda.synchronize()
    torch.save(e.params.cpu(), os.path.join(out_dir, "params%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", [1, 12])
def test_two_ranks_one_device_overlapped_allreduce(tmp_path, algo):
    """2 ranks on cuda:0 over gloo through the ENGINE (parallel.pair_step_overlapped): bit-identical parameters on both
    ranks after 2 steps, and the all-reduced gradient of step 1 == the sum of the two ranks' single-rank gradients
    (BatchNorm statistics stay per replica).  algo 12 = the bf16 path of BASELINE configs[3] under data parallelism (bf16
    activations; the gradient bucket that crosses the ranks stays fp32)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), algo), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "params0.pt"), torch.load(tmp_path / "params1.pt")
    assert torch.equal(p0, p1)
    gs0, gs1 = torch.load(tmp_path / "gsum0.pt"), torch.load(tmp_path / "gsum1.pt")
    assert torch.equal(gs0, gs1)
    # single-rank reference: each rank's own gradient, summed on the host
    from semantic_superpoint_amd import lib as L
    arch, B, H, W = ARCHS["ssp"], 2, 64, 96
    sd = C.init_state_dict(arch, seed=3)
    tot = None
    L.set_deterministic(algo == 12)
    try:
        for rank in range(2):
            e = _engine(arch, B, H, W, sd)
            e.set_conv_algo(algo)
            ds = _to_dev(C.make_synthetic_pair(B, H, W, seed=20 + rank, semantic=True, kp_prob=0.01))
            e.zero_grad()
            e.pair_step(ds, indices=None, seed=rank, train=True)
            torch.cuda.synchronize()
            tot = e.grads.cpu().clone() if tot is None else tot + e.grads.cpu()
            del e
    finally:
        L.set_deterministic(False)
    l2, mx = _rel(gs0, tot)
    # fp32: two runs differ by the commit order of the atomics only.
    # The bf16 path (12) re-quantises every activation, so run-to-run noise would decorrelate the roundings through the depth of the
    # network: its ranks and the single-rank runs therefore use the bit-reproducible accumulation (ssp_set_deterministic), under
    # which the single-rank gradients are the SAME BITS the ranks computed and every element of both buckets must be g0 + g1
    if algo == 12:
        assert torch.equal(gs0, tot), (l2, mx)
    else:
        assert l2 < 1e-5 and mx < 1e-4, (l2, mx)


# ------------------------------------------------------------------------------------------------
# (iv) G3 / G5: the loss gradients themselves, read through ssp_debug_buffer
# ------------------------------------------------------------------------------------------------
def test_detector_and_semantic_loss_gradients_vs_autograd():
    """d loss / d semi (G3's quantity) and d loss / d convSout (G5's, through the fused bilinear upsample) of a pair step
    against autograd of the oracle's detector_loss / sem_loss on the HIP path's own logits."""
    import torch.nn.functional as F
    arch = ARCHS["ssp"]
    B, H, W = 2, 64, 96
    Hc, Wc = H // 8, W // 8
    sd = C.init_state_dict(arch, seed=5)
    sample = C.make_synthetic_pair(B, H, W, seed=6, semantic=True, kp_prob=0.01)
    sample["semantic"][0, :9, :] = 133  # ignored pixels
    e = _engine(arch, B, H, W, sd)
    e.zero_grad()
    e.pair_step(_to_dev(sample), indices=None, seed=1, train=True, multi_task=False, lambda_loss=0.0)
    torch.cuda.synchronize()
    for v, (lab, msk, sem) in enumerate((("labels_2D_gaussian", "valid_mask", "semantic"),
                                         ("warped_labels_gaussian", "warped_valid_mask", "warped_sem"))):
        ypb = e.debug_buffer(v, "Y9", (B, Hc, Wc, 80))[..., :65].cpu()
        semi = (ypb * e.debug_buffer(v, "scale9", (65,)).cpu() + e.debug_buffer(v, "shift9", (65,)).cpu())
        semi = semi.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        C.detector_loss(semi, C.labels2Dto3D(sample[lab]).float(), C.get_masks(sample[msk])).backward()
        mine = e.debug_buffer(v, "dsemi", (B, Hc, Wc, 80))[..., :65].permute(0, 3, 1, 2).cpu()
        assert (mine - semi.grad).abs().max() < 1e-6 + 1e-4 * float(semi.grad.abs().max()), v
        cs = e.debug_buffer(v, "Y13", (B, Hc, Wc, 136))[..., :133].permute(0, 3, 1, 2).contiguous().cpu().requires_grad_(True)
        C.sem_loss(F.interpolate(cs, (H, W), mode="bilinear", align_corners=False), sample[sem]).backward()
        mine = e.debug_buffer(v, "dsout", (B, Hc, Wc, 136))[..., :133].permute(0, 3, 1, 2).cpu()
        assert (mine - cs.grad).abs().max() < 1e-6 + 2e-4 * float(cs.grad.abs().max()), v


def test_golden_loss_gradients_g3():
    """G3 (detector_loss value and d/d semi from the REAL reference, incl. the saturated logit that hits the -100 clamp)
    through ssp_op_detector_loss.  The fixture stores the 3-D target; depth-to-space of its 64 cell channels is a 2-D label
    map whose labels2Dto3D is that target again (cells with points sum to 1, empty cells give the dustbin)."""
    from semantic_superpoint_amd import lib as L
    g = G.load("g3_detector_loss.npz")
    tgt = torch.from_numpy(g["target"])
    B, _, Hc, Wc = tgt.shape
    lab2d = tgt[:, :64].view(B, 8, 8, Hc, Wc).permute(0, 3, 1, 4, 2).reshape(B, 1, Hc * 8, Wc * 8).contiguous()
    assert (C.labels2Dto3D(lab2d).float() - tgt).abs().max() < 1e-6
    mask2d = torch.from_numpy(g["mask"]).repeat_interleave(8, 1).repeat_interleave(8, 2).unsqueeze(1).contiguous()
    loss, dsemi = L.op_detector_loss(torch.from_numpy(g["semi"]).to(_dev()), lab2d.to(_dev()), mask2d.to(_dev()))
    assert abs(loss - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    ref = torch.from_numpy(g["dsemi"])
    assert (dsemi.cpu() - ref).abs().max() < 1e-6 + 1e-4 * float(ref.abs().max())
