"""CPU: the oracle (oracle/cpu_ref.py) against the fixtures generated from the REAL reference
(oracle/make_goldens.py).  No GPU, no HIP."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as C
from tests import golden_util as G

ARCHS = ("SuperPointNet_gauss2", "SuperPointNet_gauss2_ssmall")


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("arch", ARCHS)
def test_g1_forward(arch):
    g = G.load("g1_forward_%s.npz" % arch)
    sd = C.to_torch(C.init_state_dict(arch, seed=11))
    o1 = C.forward(sd, t(g["x1"]), arch)
    o2 = C.forward(sd, t(g["x2"]), arch)
    tol = 1e-5
    assert (o1["semi"] - t(g["semi1"])).abs().max() < tol
    assert (o1["desc"] - t(g["desc1"])).abs().max() < tol
    assert (o2["semi"] - t(g["semi2"])).abs().max() < tol
    assert (o2["desc"] - t(g["desc2"])).abs().max() < tol
    if "sem1_s" in g:
        assert (o1["sem"][:, ::7, ::3, ::5] - t(g["sem1_s"])).abs().max() < tol
        assert (o2["sem"][:, ::7, ::3, ::5] - t(g["sem2_s"])).abs().max() < tol
    for k, v in g.items():
        if k.startswith("state/"):
            assert (sd[k[6:]].double() - t(v).double()).abs().max() < tol, k
    e = G.load("g1_eval_%s.npz" % arch)
    oe = C.forward(sd, t(g["x1"]), arch, train=False)
    assert (oe["semi"] - t(e["semi"])).abs().max() < tol
    assert (oe["desc"] - t(e["desc"])).abs().max() < tol


def test_g2_labels_exact():
    g = G.load("g2_labels.npz")
    for name in ("bin", "gauss"):
        o = C.labels2Dto3D(t(g["labels_" + name]), 8, True).float()
        assert torch.equal(o, t(g["labels3D_" + name]))  # bit-exact indexing + values
    assert torch.equal(C.get_masks(t(g["mask"])), t(g["mask3D"]))


def test_g3_detector_loss():
    g = G.load("g3_detector_loss.npz")
    semi = t(g["semi"]).requires_grad_(True)
    loss = C.detector_loss(semi, t(g["target"]), t(g["mask"]))
    grad, = torch.autograd.grad(loss, semi)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert (grad - t(g["dsemi"])).abs().max() < 1e-6


def test_g5_sem_loss():
    g = G.load("g5_sem_loss.npz")
    rs = np.random.RandomState(13)
    pred = t(rs.randn(2, 133, 16, 24).astype(np.float32)).requires_grad_(True)
    lab = t(rs.randint(0, 134, size=(2, 16, 24)).astype(np.int64))
    assert torch.equal(lab, t(g["label"])) and torch.equal(pred[:, ::9].detach(), t(g["pred_s"]))
    loss = C.sem_loss(pred, lab)
    grad, = torch.autograd.grad(loss, pred)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert (grad[:, ::9] - t(g["dpred_s"])).abs().max() < 1e-7


@pytest.mark.parametrize("tag", ["small", "full"])
def test_g4_sparse_loss(tag):
    g = G.load("g4_sparse_loss_%s.npz" % tag)
    if tag == "small":
        d, dw = g["desc"], g["desc_w"]
    else:
        d, dw = G.g4_full_inputs()
        assert abs(d.astype(np.float64).sum() - float(g["desc_sum"])) < 1e-6
    B = d.shape[0]
    idx = G.indices_from(g, "", B)
    da, db = t(d).requires_grad_(True), t(dw).requires_grad_(True)
    loss, pos, neg, _ = C.batch_descriptor_loss_sparse(da, db, t(g["H"]), idx)
    for a, k in ((loss, "loss"), (pos, "pos"), (neg, "neg")):
        assert abs(float(a) - float(g[k])) < 2e-6 * max(1, abs(float(g[k]))), k
    w = g["grad_weights"]
    ga, gb = torch.autograd.grad(float(w[0]) * loss + float(w[1]) * pos + float(w[2]) * neg, (da, db))
    if tag == "small":
        assert (ga - t(g["ddesc"])).abs().max() < 1e-6 and (gb - t(g["ddesc_w"])).abs().max() < 1e-6
    else:
        assert (ga[:, ::16] - t(g["ddesc_s"])).abs().max() < 1e-6
        assert abs(float(gb.norm()) - float(g["ddesc_w_norm"])) < 1e-4 * float(g["ddesc_w_norm"])


def test_g4_sampler_reproduces_reference_indices():
    """Same numpy/torch RNG streams => the oracle's sampler yields the reference's indices bit-exactly."""
    g = G.load("g4_sparse_loss_small.npz")
    np.random.seed(123)
    torch.manual_seed(321)
    Hs = t(g["H"])
    for i in range(Hs.shape[0]):
        idx = C.sample_sparse_indices(Hs[i], 4, 6)
        assert np.array_equal(idx["uv_a"].numpy().astype(np.int16), g["uv_a%d" % i])
        assert np.array_equal(idx["uv_b"].numpy().astype(np.int16), g["uv_b%d" % i])
        assert np.array_equal(idx["nm_b"].numpy().astype(np.int16), g["nm_b%d" % i])


def test_g7_warps():
    g = G.load("g7_warps.npz")
    Hs = t(g["H"])
    inv = torch.inverse(Hs).contiguous()
    w = C.inv_warp_image_batch(t(g["img"]), inv)
    assert (w - t(g["warped"])).abs().max() < 1e-5
    m = C.compute_valid_mask(g["img"].shape[2:], inv, 0)
    assert float((m != t(g["mask"])).float().mean()) < 1e-3  # nearest ties may flip across CPUs
    for i in range(4):
        lab = C.warp_labels(t(g["pts%d" % i].astype(np.int64)), 40, 56, Hs[i])
        assert torch.equal(lab, t(g["wlabels%d" % i]))
        assert (C.scale_homography(Hs[i], (30, 40)) - t(g["Hcell%d" % i])).abs().max() < 1e-5


@pytest.mark.parametrize("tag,arch,lam", [("sp_64x96", ARCHS[0], 1.0), ("ssp_64x96", ARCHS[1], 1.0),
                                          ("magicpoint_32x48", ARCHS[0], 0.0)])
def test_g6_train_step(tag, arch, lam):
    g = G.load("g6_step_%s.npz" % tag)
    sample = G.sample_from(g)
    sd = C.init_state_dict(arch, seed=23)
    idx = G.indices_from(g, "idx/", 2) if lam > 0 else None
    tr = C.Trainer(arch, sd, lr=0.001, lambda_loss=lam)
    # the step is taken: the reference's scalar_dict holds the LIVE eta parameter, i.e. its logged
    # eta_* are post-step values (Train_model_heatmap_all.py:415-441 after :410-413)
    tr.train_val_sample(sample, n_iter=1, train=True, indices=idx)
    for k, v in g.items():
        if k.startswith("step0/"):
            ref = float(v)
            assert abs(tr.scalar_dict[k[6:]] - ref) < 3e-5 * max(1.0, abs(ref)), k
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
    for k, gr in tr.last_grads.items():
        if k == "eta":
            assert (gr - t(g["grad/eta"])).abs().max() < 1e-5
        elif gr is not None and k not in noisy:
            n = float(g["grad_norm/" + k])
            assert abs(float(gr.norm()) - n) < 1e-3 * n + 1e-7, k
            assert (gr.reshape(-1)[:64] - t(g["grad_slice/" + k])).abs().max() < 1e-3 * float(gr.abs().max()) + 1e-7, k


def test_erode_ellipse_shape():
    k = C.ellipse_kernel(3)
    assert k.shape == (6, 6) and k[3, 3] == 1 and k.sum() > 12
    m = torch.ones(20, 20)
    m[10, 10] = 0
    e = C.erode_ellipse(m, 3)
    assert e.sum() == 400 - k.sum()
