"""GPU: the HIP path (through the C ABI / Engine) against the CPU oracle on the same seeded inputs and
against the committed golden fixtures generated from the real reference.
Tolerances (north star): detector logits / descriptors within 1e-3 (fp32); label indexing bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as C
from tests import golden_util as G

pytestmark = pytest.mark.gpu
ARCHS = ("SuperPointNet_gauss2", "SuperPointNet_gauss2_ssmall")
TOL = 1e-3


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


# End-to-end gradient tolerances.  Every kernel is exact to fp32 rounding (tests/test_gpu_ops.py, 1e-4; the gates-forced
# comparison of tests/test_gpu_fullsize.py::test_gradient_differences_are_gate_flips_only: 5.6e-6).  What is left end to end
# are ReLU / max-pool GATE FLIPS: an activation within rounding distance of 0 (about one element per 50 k at fp32) may take
# the other branch than in the oracle, which moves dY of that element by O(|dA|) and spreads to every lower layer.
#   * maps of >= 120x160 pixels (>= 15x20 cells): flips average out - measured 2.6e-3 worst per-tensor rel-L2 at 120x160,
#     B = 2 -> DEFAULT bound l2 <= 5e-3, no element off by more than 5 % of max|ref|;
#   * the 8x12-cell (64x96) and smaller test shapes: ONE flip in a 96-cell head map is 5 % of max|dY| at convPa (measured),
#     so those calls pass SMALL explicitly.
SMALL = dict(l2=1.5e-2, mx=0.1)     # 64x96 inputs (8x12 cells)
TINY = dict(l2=5e-2, mx=0.1)        # 32x48 / 32x64 / 40x56 inputs (4x6 .. 5x7 cells)


def _grad_close(mine, ref, what, l2=5e-3, mx=5e-2):
    """relative L2 error <= l2 and no element off by more than mx * max|ref| (see the tolerance note above)."""
    mine, ref = mine.double().reshape(-1), ref.double().reshape(-1)
    n = float(ref.norm())
    e2 = float((mine - ref).norm()) / (n + 1e-30)
    em = float((mine - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert e2 <= l2 and em <= mx, (what, "rel-l2 %.3e rel-max %.3e" % (e2, em))
    return e2


def _engine(arch, B, H, W, sd, **kw):
    from semantic_superpoint_amd.lib import Engine
    e = Engine(arch, B, H, W, _dev(), **kw)
    e.load_state_dict(sd)
    return e


@pytest.mark.parametrize("arch", ARCHS)
def test_forward_golden(arch):
    """G1: two consecutive train-mode forwards + running statistics + eval forward, vs the reference."""
    g = G.load("g1_forward_%s.npz" % arch)
    sd = C.init_state_dict(arch, seed=11)
    e = _engine(arch, 2, 32, 48, sd)
    want = ("semi", "desc", "sem") if arch.endswith("ssmall") else ("semi", "desc")
    dev = _dev()
    o1 = e.forward(torch.from_numpy(g["x1"]).to(dev), slot=0, train=True, want=want)
    o2 = e.forward(torch.from_numpy(g["x2"]).to(dev), slot=1, train=True, want=want)
    torch.cuda.synchronize()
    assert (o1["semi"].cpu() - torch.from_numpy(g["semi1"])).abs().max() < TOL
    assert (o1["desc"].cpu() - torch.from_numpy(g["desc1"])).abs().max() < TOL
    assert (o2["semi"].cpu() - torch.from_numpy(g["semi2"])).abs().max() < TOL
    assert (o2["desc"].cpu() - torch.from_numpy(g["desc2"])).abs().max() < TOL
    if "sem" in o1:
        assert (o1["sem"].cpu()[:, ::7, ::3, ::5] - torch.from_numpy(g["sem1_s"])).abs().max() < TOL
        assert (o2["sem"].cpu()[:, ::7, ::3, ::5] - torch.from_numpy(g["sem2_s"])).abs().max() < TOL
    st = e.state_dict()
    for k, v in g.items():
        if k.startswith("state/"):
            assert (st[k[6:]].cpu().double() - torch.from_numpy(np.asarray(v)).double()).abs().max() < TOL, k
    ge = G.load("g1_eval_%s.npz" % arch)
    oe = e.forward(torch.from_numpy(g["x1"]).to(dev), slot=0, train=False)
    assert (oe["semi"].cpu() - torch.from_numpy(ge["semi"])).abs().max() < TOL
    assert (oe["desc"].cpu() - torch.from_numpy(ge["desc"])).abs().max() < TOL


@pytest.mark.parametrize("arch,H,W,B", [(ARCHS[0], 64, 96, 2), (ARCHS[1], 64, 96, 2), (ARCHS[0], 120, 160, 2),
                                         (ARCHS[0], 240, 320, 2)])
def test_forward_vs_oracle(arch, H, W, B):
    """Covers both tile geometries (W % 32 == 0 layers and the 8-wide ones) and BASELINE's 240x320."""
    sd = C.init_state_dict(arch, seed=3)
    rs = np.random.RandomState(H + W)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
    ref = C.forward(C.to_torch(sd), x, arch)
    e = _engine(arch, B, H, W, sd, with_grad=False)
    want = ("semi", "desc", "sem") if arch.endswith("ssmall") else ("semi", "desc")
    out = e.forward(x.to(_dev()), slot=0, train=True, want=want)
    torch.cuda.synchronize()
    for k in want:
        err = (out[k].cpu() - ref[k]).abs().max()
        assert err < TOL, (k, float(err))


@pytest.mark.parametrize("arch", ARCHS)
def test_backward_vs_oracle(arch):
    """ssp_backward (autograd-compat entry): gradients of a random linear functional of the outputs."""
    B, H, W = 2, 64, 96
    sd = C.init_state_dict(arch, seed=5)
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
    tsd = C.to_torch(sd, requires_grad=True)
    ref = C.forward(tsd, x, arch)
    gs = {k: torch.from_numpy(rs.randn(*ref[k].shape).astype(np.float32)) for k in ref}
    if "sem" in gs:
        gs["sem"] *= 0.05
    loss = sum((ref[k] * gs[k]).sum() for k in ref)
    loss.backward()
    e = _engine(arch, B, H, W, sd)
    dev = _dev()
    e.forward(x.to(dev), slot=0, train=True, want=())
    e.zero_grad()
    e.backward(0, gs["semi"].to(dev), gs["desc"].to(dev), gs["sem"].to(dev) if "sem" in gs else None)
    torch.cuda.synchronize()
    gd = e.grad_dict()
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
    for k in C.param_keys(arch):
        r = tsd[k].grad
        mine = gd[k].cpu()
        scale = float(r.abs().max())
        if k in noisy:  # exact gradient is 0; both sides hold rounding noise of the size of the dY sums
            assert float(mine.abs().max()) < 1e-3 * max(1.0, scale) + 1e-2, k
            continue
        _grad_close(mine, r, k, **SMALL)  # 64x96


@pytest.mark.parametrize("algo", [1, 0, 10])
def test_backward_with_zero_gamma_in_pooled_layers(algo):
    """The pooled layers' BatchNorm-backward sums are taken from the pooled activation (xhat = (z - beta)/gamma) - inside
    the data-gradient conv's epilogue (default algorithm) or by bn_bwd_reduce_pool_kernel (algo 0); channels with
    gamma == 0 must fall back to a scan over Y (all four window values tie at relu(beta))."""
    from semantic_superpoint_amd import lib as L
    L.set_conv_algo(algo)
    try:
        # (F(4x4,3x3) on every layer: six times the rounding noise of F(2x2,3x3), so a few more ReLU gates of the deeper
        # layers flip against the oracle and move the gradient that ARRIVES at the zeroed channels - 2e-4 observed)
        _zero_gamma_case(exact_tol=1e-3 if algo == 10 else 1e-4)
    finally:
        L.set_conv_algo(1)


@pytest.mark.parametrize("algo", [1, 10])
def test_negative_gamma_in_pooled_layers(algo):
    """(conv algorithm 1: conv_wino_pipe_kernel writes the pooled raw copy; 10: conv_wino4_kernel.)  Pooled layers: the producing conv writes the per-channel MAX of every 2x2 window of its raw output for gamma >= 0
    and the MIN for gamma < 0 (maxpool(relu(bn(y))) = relu(bn(pool(y))) with that choice); consumers apply BatchNorm + ReLU
    on load.  Half of the pooled layers' gammas negative: forward outputs and every gradient against the oracle."""
    from semantic_superpoint_amd import lib as L
    L.set_conv_algo(algo)
    try:
        _negative_gamma_case()
    finally:
        L.set_conv_algo(1)


def _negative_gamma_case():
    arch, B, H, W = ARCHS[0], 2, 32, 64
    sd = C.init_state_dict(arch, seed=11)
    rs = np.random.RandomState(6)
    for k in ("inc.conv.conv.4", "down1.mpconv.1.conv.4", "down2.mpconv.1.conv.4"):
        g = np.array(sd[k + ".weight"], dtype=np.float32, copy=True)
        g[::2] *= -1.0
        sd[k + ".weight"] = g
    x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
    tsd = C.to_torch(sd, requires_grad=True)
    ref = C.forward(tsd, x, arch)
    gs = {k: torch.from_numpy(rs.randn(*ref[k].shape).astype(np.float32)) for k in ref}
    sum((ref[k] * gs[k]).sum() for k in ref).backward()
    e = _engine(arch, B, H, W, sd)
    dev = _dev()
    out = e.forward(x.to(dev), slot=0, train=True, want=("semi", "desc"))
    for k in ("semi", "desc"):
        assert (out[k].cpu() - ref[k].detach()).abs().max() < 1e-3, k
    e.zero_grad()
    e.backward(0, gs["semi"].to(dev), gs["desc"].to(dev), None)
    torch.cuda.synchronize()
    gd = e.grad_dict()
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}  # exact gradient 0 (bias before BN)
    for k in C.param_keys(arch):
        if k not in noisy:
            _grad_close(gd[k].cpu(), tsd[k].grad, k, **TINY)  # 32x64


def _zero_gamma_case(exact_tol=1e-4):
    arch, B, H, W = ARCHS[0], 2, 32, 48
    sd = C.init_state_dict(arch, seed=9)
    rs = np.random.RandomState(5)
    for k in ("inc.conv.conv.4", "down1.mpconv.1.conv.4", "down2.mpconv.1.conv.4"):
        g = np.array(sd[k + ".weight"], dtype=np.float32, copy=True)
        b = np.array(sd[k + ".bias"], dtype=np.float32, copy=True)
        g[[1, 6, 7, 33]] = 0.0
        b[[1, 6]] = 0.3     # relu(beta) > 0: the first window element wins the tie
        b[[7, 33]] = -0.2   # dead channel
        sd[k + ".weight"], sd[k + ".bias"] = g, b
    x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
    tsd = C.to_torch(sd, requires_grad=True)
    ref = C.forward(tsd, x, arch)
    gs = {k: torch.from_numpy(rs.randn(*ref[k].shape).astype(np.float32)) for k in ref}
    sum((ref[k] * gs[k]).sum() for k in ref).backward()
    e = _engine(arch, B, H, W, sd)
    dev = _dev()
    e.forward(x.to(dev), slot=0, train=True, want=())
    e.zero_grad()
    e.backward(0, gs["semi"].to(dev), gs["desc"].to(dev), None)
    torch.cuda.synchronize()
    gd = e.grad_dict()
    zeroed = [1, 6, 7, 33]
    for layer in ("inc.conv.conv.4", "down1.mpconv.1.conv.4", "down2.mpconv.1.conv.4"):
        for k in (layer + ".weight", layer + ".bias"):
            r, mine = tsd[k].grad, gd[k].cpu()
            # the zeroed channels exactly (no ReLU flips possible there: z == beta), the rest statistically
            assert (mine[zeroed] - r[zeroed]).abs().max() <= exact_tol * float(r.abs().max()), k
            _grad_close(mine, r, k, **TINY)  # 32x48
    for k in ("inc.conv.conv.3.weight", "inc.conv.conv.0.weight", "down1.mpconv.1.conv.0.weight"):
        _grad_close(gd[k].cpu(), tsd[k].grad, k, **TINY)


def _to_dev(sample):
    return {k: v.to(_dev()).contiguous() for k, v in sample.items()}


# 64-element gradient slices of the G6 goldens (the first 64 elements of every gradient tensor against the reference's).  Measured on
# the GPU box (round 4): 4.8e-6 .. 6.0e-6 of max|grad| under algorithm 1, 2.2e-5 .. 2.5e-5 under algorithm 10 (F(4x4,3x3) carries
# ~5x the rounding noise) - no ReLU gate of these slices flips.  Bound = 4x the worst measured value (was 2e-2).
SLICE_TOL = {"sp_64x96": 1e-4, "ssp_64x96": 1e-4, "pair_lambda0_32x48": 1e-4}


def _idx_to_dev(idx, Wc):
    ma = torch.stack([(i["uv_a"][:, 0] + i["uv_a"][:, 1] * Wc) for i in idx]).to(torch.int32)
    mb = torch.stack([(i["uv_b"][:, 0] + i["uv_b"][:, 1] * Wc) for i in idx]).to(torch.int32)
    nm = torch.stack([i["nm_b"] for i in idx]).to(torch.int32)
    return ma.to(_dev()).contiguous(), mb.to(_dev()).contiguous(), nm.to(_dev()).contiguous()


@pytest.mark.parametrize("tag,arch,lam", [("sp_64x96", ARCHS[0], 1.0), ("ssp_64x96", ARCHS[1], 1.0),
                                          ("pair_lambda0_32x48", ARCHS[0], 0.0)])
@pytest.mark.parametrize("algo", [1, 10])
def test_pair_step_golden(tag, arch, lam, algo):
    """G6: the full step (2 forwards + losses + backward + Adam) against the reference's scalars,
    gradients and post-step eta, with the reference's own sampled indices.  algo 10: every 3x3 layer on conv_wino4_kernel
    (Winograd F(4x4,3x3), the benchmarked kernel of the large maps) with its in-step epilogues."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    g = G.load("g6_step_%s.npz" % tag)
    sample = G.sample_from(g)
    B, _, H, W = sample["image"].shape
    sd = C.init_state_dict(arch, seed=23)
    e = _engine(arch, B, H, W, sd)
    e.set_conv_algo(algo)
    idx = _idx_to_dev(G.indices_from(g, "idx/", B), W // 8) if lam > 0 else None
    e.zero_grad()
    sc = e.pair_step(_to_dev(sample), indices=idx, train=True, lambda_loss=lam, lamda_d=1.0, multi_task=True)
    torch.cuda.synchronize()
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for name in ("loss", "loss_det", "loss_det_warp", "loss_desc", "loss_sem", "loss_sem_warp", "positive_dist",
                 "negative_dist"):
        ref = float(g["step0/" + name])
        assert abs(sc[name] - ref) < TOL * max(1.0, abs(ref)), (name, sc[name], ref)
    gd = e.grad_dict()
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
    worst_slice = 0.0
    for k in C.param_keys(arch):
        if k in noisy or ("grad_norm/" + k) not in g:
            continue
        n_ref = float(g["grad_norm/" + k])
        mine = gd[k].cpu().reshape(-1)
        assert abs(float(mine.norm()) - n_ref) < 5e-3 * n_ref + 1e-6, (k, float(mine.norm()), n_ref)
        sl = torch.from_numpy(g["grad_slice/" + k])
        err = float((mine[:64] - sl).abs().max()) / (float(mine.abs().max()) + 1e-30)
        worst_slice = max(worst_slice, err)
        assert err < SLICE_TOL[tag] + 1e-6, (k, err)
    print("G6 %s algo %d: worst 64-element slice error %.2e of max|grad|" % (tag, algo, worst_slice))
    assert (gd["eta"].cpu() - torch.from_numpy(g["grad/eta"])).abs().max() < 1e-3
    e.adam_step(0.001)
    torch.cuda.synchronize()
    # the reference logs the LIVE eta parameter, i.e. post-step values (oracle/cpu_ref.py Trainer)
    eta = e.eta.cpu()
    assert abs(float(eta[0]) - float(g["step0/eta_det"])) < 1e-5 and abs(float(eta[1]) - float(g["step0/eta_desc"])) < 1e-5
    if "step0/eta_sem" in g:
        assert abs(float(eta[2]) - float(g["step0/eta_sem"])) < 1e-5


@pytest.mark.parametrize("arch", ARCHS)
def test_pair_step_vs_oracle_two_steps(arch):
    """Two optimizer steps on fresh seeded data (120x160, B=2) vs the oracle; also the no-grad (val) path."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    B, H, W = 2, 120, 160
    semantic = arch.endswith("ssmall")
    sd = C.init_state_dict(arch, seed=9)
    sample = C.make_synthetic_pair(B, H, W, seed=4, semantic=semantic, kp_prob=0.005)
    tr = C.Trainer(arch, sd, lr=0.001)
    e = _engine(arch, B, H, W, sd)
    ds = _to_dev(sample)
    for it in range(2):
        np.random.seed(50 + it)
        torch.manual_seed(60 + it)
        tr.train_val_sample(sample, n_iter=it, train=True)
        idx = _idx_to_dev(tr.aux["indices"], W // 8)
        e.zero_grad()
        sc = e.pair_step(ds, indices=idx, train=True)
        if it == 0:  # every gradient tensor of the first step at the tight (>= 120x160) bound
            torch.cuda.synchronize()
            gd = e.grad_dict()
            noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
            worst = max(_grad_close(gd[k].cpu(), tr.last_grads[k], k) for k in C.param_keys(arch) if k not in noisy)
            print("120x160 %s: worst per-tensor gradient rel-L2 %.2e" % (arch, worst))
        e.adam_step(0.001)
        torch.cuda.synchronize()
        sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
        for name in ("loss", "loss_det", "loss_det_warp", "positive_dist", "negative_dist", "loss_sem", "loss_sem_warp"):
            ref = tr.scalar_dict[name]
            assert abs(sc[name] - ref) < (2e-4 if it == 0 else 2e-3) * max(1.0, abs(ref)), (it, name, sc[name], ref)
    assert (e.eta.cpu() - tr.eta.detach()).abs().max() < 1e-4
    rv = e.state_dict()
    for k in ("inc.conv.conv.1.running_var", "bnPb.running_mean", "down3.mpconv.1.conv.4.running_var"):
        assert (rv[k].cpu() - tr.sd[k]).abs().max() < 1e-3 * max(1.0, float(tr.sd[k].abs().max())), k
    # validation path: train=False leaves the gradient buffer untouched
    before = e.grads.clone()
    e.pair_step(ds, indices=idx, train=False)
    torch.cuda.synchronize()
    assert torch.equal(before, e.grads)


def test_sparse_loss_golden_full():
    """G4 at the real 30x40 cell grid: loss values with the reference's indices (descriptor maps are inputs)."""
    from semantic_superpoint_amd import lib as L
    g = G.load("g4_sparse_loss_full.npz")
    d, dw = G.g4_full_inputs()
    B = d.shape[0]
    idx = G.indices_from(g, "", B)
    pos, neg = L.op_sparse_loss(torch.from_numpy(d).to(_dev()), torch.from_numpy(dw).to(_dev()), *_idx_to_dev(idx, 40))
    assert abs(pos - float(g["pos"])) < 1e-4 and abs(neg - float(g["neg"])) < 1e-4


@pytest.mark.parametrize("tag", ["small", "mid"])
@pytest.mark.parametrize("method,dist", [("2d", "cos"), ("1d", "cos"), ("2d", "euclidean"), ("1d", "euclidean")])
def test_sparse_loss_variants_golden(method, dist, tag):
    """G14 (and G4 for the shipped pair of values): loss terms AND gradients of the sparse descriptor loss kernels against the REAL
    reference for every (method, dist) descriptor_loss_sparse accepts (sparse_loss.py:76-77), through ssp_op_sparse_loss."""
    from semantic_superpoint_amd import lib as L
    if (method, dist) == ("2d", "cos"):
        if tag != "small":
            pytest.skip("G4 stores the full gradients at the small size only")
        g = G.load("g4_sparse_loss_small.npz")
    else:
        g = G.load("g14_sparse_loss_%s_%s_%s.npz" % (method, dist, tag))
    d, dw = torch.from_numpy(g["desc"]), torch.from_numpy(g["desc_w"])
    B, _, Hc, Wc = d.shape
    idx = G.indices_from(g, "", B)
    w = g["grad_weights"]   # d (w0 loss + w1 pos + w2 neg), loss = lamda_d pos + neg, lamda_d = 1
    pos, neg, ga, gb = L.op_sparse_loss(d.to(_dev()), dw.to(_dev()), *_idx_to_dev(idx, Wc), method=method, dist=dist,
                                        grad=(float(w[0] + w[1]), float(w[0] + w[2])))
    assert abs(pos - float(g["pos"])) < 2e-5 * max(1.0, abs(float(g["pos"])))
    assert abs(neg - float(g["neg"])) < 2e-5 * max(1.0, abs(float(g["neg"])))
    for mine, ref in ((ga, g["ddesc"]), (gb, g["ddesc_w"])):
        ref = torch.from_numpy(ref)
        assert (mine.cpu() - ref).abs().max() < 1e-7 + 2e-5 * float(ref.abs().max())
    p2, n2 = L.op_sparse_loss(d.to(_dev()), dw.to(_dev()), *_idx_to_dev(idx, Wc), method=method, dist=dist)   # forward only
    assert abs(p2 - pos) < 1e-6 and abs(n2 - neg) < 1e-6


@pytest.mark.parametrize("B,H,W", [(4, 240, 320), (2, 480, 640)])
def test_device_sampler_distribution(B, H, W):
    """ssp_sample_indices: every sampled match is a valid correspondence of the oracle, matches are distinct
    when >= n_match candidates exist, non-matches are uniform over the grid.  480x640 (4800 cells) exercises the
    8192-key sort."""
    from semantic_superpoint_amd.lib import Engine
    Hc, Wc = H // 8, W // 8
    e = Engine("SuperPointNet_gauss2", B, H, W, _dev(), with_grad=False)
    rs = np.random.RandomState(2)
    Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(B)]).astype(np.float32))
    from semantic_superpoint_amd.lib import scaled_homographies
    ma, mb, nm = e.sample_indices(Hs.to(_dev()), seed=1234, cell_homographies=scaled_homographies(Hs, Hc, Wc))
    ma_a, mb_a, _ = e.sample_indices(Hs.to(_dev()), seed=1234)  # analytic T^-1 H T on the device
    torch.cuda.synchronize()
    ma, mb, nm = ma.cpu(), mb.cpu(), nm.cpu()
    for i in range(B):
        uv_a, uv_b = C.cell_matches(Hs[i], Hc, Wc)
        valid = {int(a[0] + a[1] * Wc): int(b[0] + b[1] * Wc) for a, b in zip(uv_a, uv_b)}
        # host-scaled cell homography: EVERY sampled pair is a correspondence of the reference, none is missing
        assert all(valid.get(a, -1) == b for a, b in zip(ma[i].tolist(), mb[i].tolist())), i
        if len(valid) >= 1000:
            assert len(set(ma[i].tolist())) == 1000
        else:
            assert set(ma[i].tolist()) == set(valid.keys())
        bad = sum(1 for a, b in zip(ma_a[i].cpu().tolist(), mb_a[i].cpu().tolist()) if valid.get(a, -1) != b)
        assert bad <= 2 * (Hc * Wc) // 1200, bad  # analytic form: rounding ties at .5 may differ
    assert nm.min() >= 0 and nm.max() < Hc * Wc
    hist = torch.bincount(nm.reshape(-1).long(), minlength=Hc * Wc).float()
    m = float(hist.mean())  # Poisson counts: every cell within 5 sigma of the mean
    assert float(hist.min()) > m - 5 * m ** 0.5 and float(hist.max()) < m + 5 * m ** 0.5
    ma2, _, nm2 = e.sample_indices(Hs.to(_dev()), seed=1234)
    assert torch.equal(ma2.cpu(), ma) and torch.equal(nm2.cpu(), nm)  # deterministic in the seed


def test_uniform_sum_loss_and_lamda_d():
    """model.multi_task_loss: false -> loss = det + det_warp + lambda*(lamda_d*pos + neg)
    (Train_model_heatmap_all.py:363-365) incl. its gradients, with lamda_d != 1."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch, B, H, W = ARCHS[0], 2, 64, 96
    sd = C.init_state_dict(arch, seed=12)
    sample = C.make_synthetic_pair(B, H, W, seed=13, kp_prob=0.01)
    tr = C.Trainer(arch, sd, lr=0.001, multi_task=False, lambda_loss=0.5, lamda_d=3.0)
    tr.real_batch_size = 10 ** 9
    np.random.seed(1)
    torch.manual_seed(2)
    tr.train_val_sample(sample, n_iter=0, train=True)
    e = _engine(arch, B, H, W, sd)
    e.zero_grad()
    sc = e.pair_step(_to_dev(sample), indices=_idx_to_dev(tr.aux["indices"], W // 8), train=True, lambda_loss=0.5,
                     lamda_d=3.0, multi_task=False)
    torch.cuda.synchronize()
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for name in ("loss", "loss_det", "loss_det_warp", "loss_desc", "positive_dist", "negative_dist"):
        ref = tr.scalar_dict[name]
        assert abs(sc[name] - ref) < TOL * max(1.0, abs(ref)), (name, sc[name], ref)
    gd = e.grad_dict()
    assert float(gd["eta"].abs().max()) == 0.0  # eta is not in the graph of the uniform sum
    for k in ("convDb.weight", "convPb.weight", "inc.conv.conv.3.weight"):
        _grad_close(gd[k].cpu(), tr.last_grads[k], k, **SMALL)  # 64x96


def test_gradient_accumulation_over_micro_batches():
    """real_batch_size = 2 * batch_size: gradients of two micro-batches add up un-scaled before ONE Adam step
    (Train_model_heatmap_all.py:406-413)."""
    arch, B, H, W = ARCHS[0], 2, 64, 96
    sd = C.init_state_dict(arch, seed=14)
    s1 = C.make_synthetic_pair(B, H, W, seed=15, kp_prob=0.01)
    s2 = C.make_synthetic_pair(B, H, W, seed=16, kp_prob=0.01)
    tr = C.Trainer(arch, sd, lr=0.001)
    tr.real_batch_size = 2 * B
    e = _engine(arch, B, H, W, sd)
    e.zero_grad()
    for it, s in enumerate((s1, s2)):
        np.random.seed(30 + it)
        torch.manual_seed(40 + it)
        tr.train_val_sample(s, n_iter=it, train=True)  # steps only after the second micro-batch
        e.pair_step(_to_dev(s), indices=_idx_to_dev(tr.aux["indices"], W // 8), train=True)
    torch.cuda.synchronize()
    # the oracle's last_grads holds the ACCUMULATED gradient after micro-batch 2 (cloned before its optimizer step)
    gd = e.grad_dict()
    for k in ("convDb.weight", "convPa.weight", "down1.mpconv.1.conv.0.weight", "bnPb.weight"):
        _grad_close(gd[k].cpu(), tr.last_grads[k], k, **SMALL)  # 64x96
    e.adam_step(0.001)
    torch.cuda.synchronize()
    assert (e.eta.cpu() - tr.eta.detach()).abs().max() < 1e-4


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (3, 120, 160), (2, 40, 56)])
def test_odd_shapes_forward_and_step(B, H, W):
    """batch 1 / odd batch, cell grids that are not multiples of the tile sizes (15x20, 5x7)."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch = ARCHS[0]
    sd = C.init_state_dict(arch, seed=20)
    sample = C.make_synthetic_pair(B, H, W, seed=21, kp_prob=0.01)
    tr = C.Trainer(arch, sd, lr=0.001)
    np.random.seed(5)
    torch.manual_seed(6)
    tr.train_val_sample(sample, n_iter=0, train=True)
    e = _engine(arch, B, H, W, sd)
    e.zero_grad()
    sc = e.pair_step(_to_dev(sample), indices=_idx_to_dev(tr.aux["indices"], W // 8), train=True)
    torch.cuda.synchronize()
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for name in ("loss", "loss_det", "loss_det_warp", "positive_dist", "negative_dist"):
        ref = tr.scalar_dict[name]
        assert abs(sc[name] - ref) < 2e-3 * max(1.0, abs(ref)), (name, sc[name], ref)
    gd = e.grad_dict()
    for k in ("convDb.weight", "inc.conv.conv.0.weight", "down2.mpconv.1.conv.3.weight"):
        if H >= 120:
            _grad_close(gd[k].cpu(), tr.last_grads[k], k)  # 120x160: the tight default
        else:  # batch 1 at 64x96 (96 cells in the whole batch) / 40x56 (5x7 cells): single flips dominate
            _grad_close(gd[k].cpu(), tr.last_grads[k], k, l2=3e-2, mx=0.2)


@pytest.mark.parametrize("algo", [0, 6, 9, 10, 11])
def test_conv_algorithms_agree_on_a_training_step(algo):
    """ssp_set_conv_algo: the direct implicit-GEMM kernels (0), the two-workgroup Winograd kernel (6), F(2x2,3x3) only (9),
    Winograd F(4x4,3x3) on every 3x3 layer (10: pooled raw outputs, fused BatchNorm-backward sums and partial tile blocks
    of conv_wino4_kernel) and the F(3x3,4x4) weight gradient (11) give the
    losses and gradients of the default (pipelined Winograd, 1) on the same step (fp32 everywhere; only the summation
    order / the Winograd transforms differ)."""
    from semantic_superpoint_amd import lib as L
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch, B, H, W = "SuperPointNet_gauss2", 2, 64, 96
    sd = C.init_state_dict(arch, seed=21)
    sample = _to_dev(C.make_synthetic_pair(B, H, W, seed=6, kp_prob=0.005))
    out = {}
    try:
        for a in (1, algo):
            L.set_conv_algo(a)
            e = _engine(arch, B, H, W, sd)
            e.zero_grad()
            sc = e.pair_step(sample, indices=None, seed=3, train=True)
            torch.cuda.synchronize()
            out[a] = (dict(zip(SCALAR_NAMES, sc.cpu().tolist())), e.grads.clone().cpu())
    finally:
        L.set_conv_algo(1)
    for name in ("loss", "loss_det", "loss_det_warp", "positive_dist", "negative_dist"):
        assert abs(out[1][0][name] - out[algo][0][name]) < 1e-4 * max(1.0, abs(out[1][0][name])), name
    _grad_close(out[algo][1], out[1][1], "flat gradient, algo %d vs 1" % algo, **SMALL)  # 64x96


# (The bf16-OPERAND experiments inside the fp32 Winograd kernels - conv algorithms 3 / 7 / 8 of rounds 2-3 - were superseded by the bf16
# path (algorithm 12, tests/test_gpu_bf16_path.py) and are compiled out of the shipped library (-DSSP_LEGACY_ALGOS=1 brings them back);
# their tests left with them: git history of this file at round 4, results in profiles/PERF_LOG_rounds_1-4.md section 10.)


@pytest.mark.parametrize("n_classes", [21, 150])
def test_segmentation_head_with_another_class_count(n_classes):
    """convSout with 21 (one 32-channel n-tile, sout stride 24) and 150 (five n-tiles = 3 + 2) classes through the grouped
    pointwise kernels (forward, data gradient with the fused BatchNorm-backward sums of bnS1, weight gradient) against
    autograd through the oracle's forward: outputs, and the gradients of convSout / convDS / the encoder below."""
    arch = ARCHS[1]
    B, H, W = 2, 96, 128
    sd = C.init_state_dict(arch, seed=5, n_classes=n_classes)
    e = _engine(arch, B, H, W, sd, n_classes=n_classes)
    rs = np.random.RandomState(n_classes)
    x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
    out = e.forward(x.to(_dev()), slot=0, train=True, want=("semi", "desc", "sem"))
    tsd = C.to_torch(sd, requires_grad=True)
    ref = C.forward(tsd, x, arch, n_classes=n_classes)
    for k in ("semi", "desc", "sem"):
        assert (out[k].cpu() - ref[k].detach()).abs().max() < TOL, k
    gs = {k: torch.from_numpy(rs.randn(*ref[k].shape).astype(np.float32)) for k in ("semi", "desc", "sem")}
    sum((ref[k] * gs[k]).sum() for k in gs).backward()
    e.zero_grad()
    e.backward(0, gs["semi"].to(_dev()), gs["desc"].to(_dev()), gs["sem"].to(_dev()))
    torch.cuda.synchronize()
    gd = e.grad_dict()
    for k in ("convSout.weight", "convSout.bias", "convDS.weight", "bnS1.weight", "convPb.weight", "convDb.weight",
              "down3.mpconv.1.conv.3.weight"):
        _grad_close(gd[k].cpu(), tsd[k].grad, k, **SMALL)


def test_pair_step_with_more_than_64_pairs():
    """B = 80 pairs in ONE ssp_pair_step call (the per-image accumulators of the sparse descriptor loss were fixed 64-entry arrays
    up to round 4; SSP_MAX_PAIRS = 128 now): the scalars against the oracle fed with the device-sampled indices."""
    from semantic_superpoint_amd.lib import SCALAR_NAMES
    arch, B, H, W = ARCHS[0], 80, 32, 48
    sd = C.init_state_dict(arch, seed=2)
    sample = C.make_synthetic_pair(B, H, W, seed=3, kp_prob=0.02)
    e = _engine(arch, B, H, W, sd, n_match=200, n_non=20)
    e.zero_grad()
    sc = e.pair_step(_to_dev(sample), indices=None, seed=11, train=True).clone()
    torch.cuda.synchronize()
    ma, mb, nm = (t.cpu().long() for t in e._last_idx)
    Wc = W // 8
    idx = [{"uv_a": torch.stack((ma[b] % Wc, ma[b] // Wc), 1).float(), "uv_b": torch.stack((mb[b] % Wc, mb[b] // Wc), 1).float(),
            "nm_b": nm[b]} for b in range(B)]
    tr = C.Trainer(arch, sd, lr=1e-3, n_match=200, n_non=20)
    tr.real_batch_size = 10 ** 9
    tr.train_val_sample(sample, n_iter=0, train=True, indices=idx)
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for k in ("loss", "loss_det", "loss_det_warp", "loss_desc", "positive_dist", "negative_dist"):
        ref = tr.scalar_dict[k]
        assert abs(sc[k] - ref) < TOL * max(1.0, abs(ref)), (k, sc[k], ref)
    gd = e.grad_dict()
    for k in ("convDb.weight", "convPb.weight", "bnDa.weight"):
        _grad_close(gd[k].cpu(), tr.last_grads[k], k, **TINY)
