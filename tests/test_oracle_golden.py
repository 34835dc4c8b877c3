"""CPU: the oracle (oracle/cpu_ref.py) against the fixtures generated from the REAL reference
(oracle/make_goldens.py).  No GPU, no HIP."""
import math
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


@pytest.mark.parametrize("tag", ["small", "mid"])
@pytest.mark.parametrize("method,dist", [("1d", "cos"), ("2d", "euclidean"), ("1d", "euclidean")])
def test_g14_sparse_loss_variants(method, dist, tag):
    """descriptor_loss_sparse's other parameter values (method "1d", dist "euclidean": sparse_loss.py:76-77) - the oracle against the
    reference's loss terms and gradients (oracle/make_goldens.py: g14)."""
    g = G.load("g14_sparse_loss_%s_%s_%s.npz" % (method, dist, tag))
    B = g["desc"].shape[0]
    idx = G.indices_from(g, "", B)
    da, db = t(g["desc"]).requires_grad_(True), t(g["desc_w"]).requires_grad_(True)
    loss, pos, neg, _ = C.batch_descriptor_loss_sparse(da, db, t(g["H"]), idx, 1.0, int(g["n_match"]), int(g["n_non"]), dist=dist,
                                                       method=method)
    for a, k in ((loss, "loss"), (pos, "pos"), (neg, "neg")):
        assert abs(float(a) - float(g[k])) < 2e-6 * max(1, abs(float(g[k]))), k
    w = g["grad_weights"]
    ga, gb = torch.autograd.grad(float(w[0]) * loss + float(w[1]) * pos + float(w[2]) * neg, (da, db))
    assert (ga - t(g["ddesc"])).abs().max() < 1e-6 and (gb - t(g["ddesc_w"])).abs().max() < 1e-6


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


DENSE = {"descriptor_dist": 4, "lambda_d": 800}  # the shipped spelling; descriptor_loss swallows it: lamda_d = 250


@pytest.mark.parametrize("tag,arch,lam", [("sp_64x96", ARCHS[0], 1.0), ("ssp_64x96", ARCHS[1], 1.0),
                                          ("pair_lambda0_32x48", ARCHS[0], 0.0), ("sp_dense_64x96", ARCHS[0], 1.0),
                                          ("sp_dense_uniform_64x96", ARCHS[0], 1.0)])
def test_g6_train_step(tag, arch, lam):
    g = G.load("g6_step_%s.npz" % tag)
    sample = G.sample_from(g)
    sd = C.init_state_dict(arch, seed=23)
    dense = "dense" in tag
    idx = G.indices_from(g, "idx/", 2) if lam > 0 and not dense else None
    kw = dict(dense=DENSE, multi_task="uniform" not in tag) if dense else {}
    tr = C.Trainer(arch, sd, lr=0.001, lambda_loss=lam, **kw)
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
            if gr is None:  # uniform sum: eta is not in the graph
                assert "grad/eta" not in g
                continue
            assert (gr - t(g["grad/eta"])).abs().max() < 1e-5
        elif gr is not None and k not in noisy:
            n = float(g["grad_norm/" + k])
            assert abs(float(gr.norm()) - n) < 1e-3 * n + 1e-7, k
            assert (gr.reshape(-1)[:64] - t(g["grad_slice/" + k])).abs().max() < 1e-3 * float(gr.abs().max()) + 1e-7, k


@pytest.mark.parametrize("tag", ["uniform", "kendall"])
def test_g13_single_view_step(tag):
    """BASELINE configs[0]: the shipped magicpoint yaml's `warped_pair.enable: false` branch (one forward, detector loss
    only; Train_model_heatmap_all.py:207,237-262,330-332) - oracle leg against the real reference's two optimizer steps."""
    g = G.load("g13_single_view_%s_120x160.npz" % tag)
    sample = G.g13_sample(g)
    arch = ARCHS[0]
    sd = C.init_state_dict(arch, seed=37)
    tr = C.Trainer(arch, sd, lr=0.001, lambda_loss=0.0, multi_task=tag == "kendall", gaussian=False, warped_pair=False)
    for it in range(2):
        tr.train_val_sample(sample, n_iter=it + 1, train=True)
        for k, v in g.items():
            if k.startswith("step%d/" % it):
                ref = float(v)
                assert abs(tr.scalar_dict[k[6:]] - ref) < 3e-5 * max(1.0, abs(ref)), (k, tr.scalar_dict[k[6:]], ref)
        if it == 0:
            noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
            for k, gr in tr.last_grads.items():
                if k == "eta":
                    assert (gr is None) == ("grad/eta" not in g)
                    if gr is not None:
                        assert (gr - t(g["grad/eta"])).abs().max() < 1e-5
                elif gr is None:  # the descriptor head is not in the graph of a single-view step
                    assert k.startswith(("convD", "bnD")) and "grad_norm/" + k not in g
                elif k not in noisy:
                    n = float(g["grad_norm/" + k])
                    assert abs(float(gr.norm()) - n) < 1e-3 * n + 1e-7, k
                    assert (gr.reshape(-1)[:64] - t(g["grad_slice/" + k])).abs().max() < 1e-3 * float(gr.abs().max()) + 1e-7, k
    assert (tr.eta.detach() - t(g["post/eta"])).abs().max() < 1e-5
    assert float(g["step0/loss_det_warp"]) == 0.0 and float(g["step0/loss_desc"]) == 0.0
    assert "must match the size" in str(g["l2_raises"])  # loss_type l2 raises in the reference itself
    with pytest.raises(AssertionError):  # :343 "need a pair of images"
        C.pair_losses(tr.sd, tr.eta, sample, arch, lambda_loss=1.0, warped_pair=False)


def test_erode_ellipse_shape():
    k = C.ellipse_kernel(3)
    assert k.shape == (6, 6) and k[3, 3] == 1 and k.sum() > 12
    m = torch.ones(20, 20)
    m[10, 10] = 0
    e = C.erode_ellipse(m, 3)
    assert e.sum() == 400 - k.sum()


def test_third_party_definitions_hand_vectors():
    """cv2 / torchgeometry are absent from the image: their restatements are pinned against vectors that do not come
    from this repository - OpenCV's documented 5x5 ellipse, hand-evaluated elements, homographies with a closed form,
    a two-point soft-argmax."""
    # cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (5, 5)) as printed in OpenCV's morphology tutorial
    assert C.structuring_element_ellipse(5, 5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1],
                                                            [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    assert C.structuring_element_ellipse(3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    # (6, 6) = erosion_radius 3 (configs: erosion_radius 3), evaluated by hand from the source: r = c = 3,
    # dx(dy) = round(3 sqrt(1 - dy^2 / 9)) = 0, 2, 3, 3, 3, 2 for dy = -3 .. 2 -> columns 3 - dx .. 3 + dx (clipped)
    assert C.ellipse_kernel(3).tolist() == [[0, 0, 0, 1, 0, 0], [0, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1],
                                            [1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]]
    # erosion with that element: a single zero at (10, 10) spreads to the REFLECTED support (anchor (3, 3))
    m = torch.ones(20, 20)
    m[10, 10] = 0
    e = C.erode_ellipse(m, 3)
    k = torch.tensor(C.ellipse_kernel(3))
    for y in range(20):
        for x in range(20):
            i, j = 10 - y + 3, 10 - x + 3  # out[y, x] = min_k mask[y + i - 3, x + j - 3]
            hit = 0 <= i < 6 and 0 <= j < 6 and bool(k[i, j])
            assert e[y, x] == (0.0 if hit else 1.0), (y, x)
    # getPerspectiveTransform: closed-form homographies are recovered from four correspondences
    sq = np.array([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]])  # the reference's pts1 (homographies.py:59)
    for Hm in (np.eye(3), np.array([[2.0, 0, 3.0], [0, 0.5, -1.0], [0, 0, 1]]),
               np.array([[np.cos(0.3), -np.sin(0.3), 0.2], [np.sin(0.3), np.cos(0.3), -0.1], [0, 0, 1]]),
               np.array([[1.1, 0.05, 0.3], [-0.02, 0.9, 0.1], [0.15, -0.25, 1.0]])):
        q = np.c_[sq, np.ones(4)] @ Hm.T
        got = C.get_perspective_transform(sq, q[:, :2] / q[:, 2:])
        assert np.abs(got - Hm).max() < 1e-12
    # the textbook unit-square -> quadrilateral map (Heckbert 1989, eq. for the projective mapping of a square):
    # square (0,0),(1,0),(1,1),(0,1) -> (0,0),(2,0),(3,2),(0,1): g = 1/3... evaluated by hand below
    src = np.array([[0.0, 0], [1, 0], [1, 1], [0, 1]])
    dst = np.array([[0.0, 0], [2, 0], [3, 2], [0, 1]])
    # sx = x0-x1+x2-x3 = 1, sy = y0-y1+y2-y3 = 1; dx1 = x1-x2 = -1, dx2 = x3-x2 = -3, dy1 = y1-y2 = -2, dy2 = y3-y2 = -1
    # det = dx1 dy2 - dx2 dy1 = 1 - 6 = -5;  g = (sx dy2 - dx2 sy) / det = (-1 + 3) / -5 = -0.4
    # h = (dx1 sy - sx dy1) / det = (-1 + 2) / -5 = -0.2;  a = x1 - x0 + g x1 = 2 - 0.8 = 1.2;  b = x3 - x0 + h x3 = 0
    # d = y1 - y0 + g y1 = 0;  e = y3 - y0 + h y3 = 1 - 0.2 = 0.8
    want = np.array([[1.2, 0.0, 0.0], [0.0, 0.8, 0.0], [-0.4, -0.2, 1.0]])
    assert np.abs(C.get_perspective_transform(src, dst) - want).max() < 1e-12
    # SpatialSoftArgmax2d (torchgeometry 0.1.2, normalized_coordinates=False): two cells carry all the mass
    x = torch.full((1, 1, 5, 5), -1e30)
    x[0, 0, 2, 1] = math.log(0.25)
    x[0, 0, 2, 3] = math.log(0.75)
    ex, ey = C.spatial_soft_argmax2d(x)[0, 0].tolist()
    inv = 1.0 / (1.0 / 3.0 + 1.0 + 1e-6)  # exp(x - max) = {1/3, 1}; the eps enters the normaliser
    assert abs(ex - (1.0 / 3.0 + 3.0) * inv) < 1e-6 and abs(ey - 2.0 * (4.0 / 3.0) * inv) < 1e-6
    assert abs(ex - 2.5) < 1e-5 and abs(ey - 2.0) < 1e-5


G8 = ("sp_64x96_v6", "ssp_48x64_v5", "sp_120x160_v4")


def g8_sample(g):
    return {"image": t(g["views"]), "valid_mask": t(g["valid_mask"]), "homographies": t(g["homographies"]),
            "inv_homographies": t(g["inv_homographies"])}


@pytest.mark.parametrize("name", G8)
def test_g8_export(name):
    """Homography-adaptation export: oracle vs the reference's SuperPointFrontend_torch/combine_heatmap outputs."""
    g = G.load("g8_export_%s.npz" % name)
    arch, thr, top_k = str(g["arch"]), float(g["thr"]), int(g["top_k"])
    # dataset side (Coco.py:258-292): views and masks from the image and the homographies
    inv = t(g["inv_homographies"])
    H, W = g["img"].shape
    views = C.inv_warp_image_batch(t(g["img"]).view(1, 1, H, W).repeat(inv.shape[0], 1, 1, 1), inv)
    assert (views - t(g["views"])).abs().max() < 1e-6
    assert torch.equal(C.compute_valid_mask((H, W), inv, int(g["erosion"])).view(-1, 1, H, W), t(g["valid_mask"]))
    sd = C.to_torch(C.init_state_dict(arch, seed=int(g["seed"])))
    o = C.export_points(sd, g8_sample(g), arch, conf_thresh=thr, nms_dist=4, top_k=top_k, subpixel=True)
    assert (o["views_heatmap"] - t(g["views_heatmap"])).abs().max() < 1e-6
    assert (o["heatmap"].squeeze() - t(g["aggregate"])).abs().max() < 1e-6
    assert (sd["bnPb.running_var"] - t(g["bnPb_running_var"])).abs().max() < 1e-5
    # point extraction on the reference's own aggregate: bit-exact
    nms = C.get_pts_from_heatmap(g["aggregate"], thr, 4)
    assert np.array_equal(nms, g["pts_nms"])
    sub = C.soft_argmax_points(g["aggregate"], nms).transpose()
    if top_k and sub.shape[0] > top_k:
        sub = sub[:top_k]
    assert np.array_equal(sub, g["pts"])


def test_nms_fast_edge_cases():
    """models/model_wrap.py:151-155 (0 / 1 corners), border removal and the all-below-threshold heatmap."""
    assert C.get_pts_from_heatmap(np.zeros((16, 24), np.float32), 0.015, 4).shape == (3, 0)
    hm = np.zeros((16, 24), np.float32)
    hm[8, 10] = 0.5
    assert np.array_equal(C.get_pts_from_heatmap(hm, 0.015, 4), np.array([[10.0], [8.0], [0.5]]))
    hm[2, 10] = 0.9  # inside the 4-pixel border: survives NMS (and suppresses nothing here) but is dropped
    assert np.array_equal(C.get_pts_from_heatmap(hm, 0.015, 4), np.array([[10.0], [8.0], [0.5]]))
    hm[6, 12] = 0.7  # suppressed by the border point (10, 2) BEFORE that one is dropped, so (10, 8) still survives
    assert np.array_equal(C.get_pts_from_heatmap(hm, 0.015, 4)[:2].T, np.array([[10.0, 8.0]]))
    hm[2, 10] = 0.0  # without it (12, 6) wins over (10, 8), which is within Chebyshev distance 4
    assert np.array_equal(C.get_pts_from_heatmap(hm, 0.015, 4)[:2].T, np.array([[12.0, 6.0]]))
    hm[:] = np.nan  # 0/0 of combine_heatmap where no view covers a pixel
    assert C.get_pts_from_heatmap(hm, 0.015, 4).shape == (3, 0)


def test_g9_logging_branch():
    """heatmap_to_nms + batch_precision_recall (Train_model_heatmap_all.py:574-622,693-707) vs the reference."""
    g = G.load("g9_logging.npz")
    heat = C.flatten_detection(t(g["semi"]))
    assert (heat - t(g["heat"])).abs().max() < 1e-7
    nms = np.stack([C.heatmap_nms(h) for h in g["heat"]])
    assert np.array_equal(nms, g["nms"].astype(np.float32))
    pr = C.batch_precision_recall(t(nms[:, None]), t(g["labels"]))
    assert abs(pr["precision"] - float(g["precision"])) < 1e-7 and abs(pr["recall"] - float(g["recall"])) < 1e-7


def test_g10_dense_descriptor_loss():
    """Dense descriptor loss (utils/utils.py:779-893) and its autograd gradients vs the reference."""
    g = G.load("g10_dense_loss_small.npz")
    a = t(g["desc"]).clone().requires_grad_(True)
    b = t(g["desc_w"]).clone().requires_grad_(True)
    loss, mask, pos, neg = C.descriptor_loss_dense(a, b, t(g["homographies"]), t(g["mask_valid"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-6 and abs(float(pos) - float(g["pos_sum"])) < 1e-6
    assert abs(float(neg) - float(g["neg_sum"])) < 1e-8
    assert np.array_equal(mask.numpy().astype(np.uint8), g["mask"])
    ga, gb = torch.autograd.grad(loss, (a, b), retain_graph=True)
    assert (ga - t(g["g_loss_a"])).abs().max() < 1e-7 and (gb - t(g["g_loss_b"])).abs().max() < 1e-7
    ga, gb = torch.autograd.grad(0.5 * (pos + neg), (a, b))
    assert (ga - t(g["g_mt_a"])).abs().max() < 1e-7 and (gb - t(g["g_mt_b"])).abs().max() < 1e-7
    f = G.load("g10_dense_loss_full.npz")
    d, dw = G.g10_inputs(int(f["seed"]), 2, 30, 40)
    loss, mask, pos, neg = C.descriptor_loss_dense(t(d), t(dw), t(f["homographies"]), t(f["mask_valid"]))
    assert abs(float(loss) - float(f["loss"])) < 1e-6 and abs(float(pos) - float(f["pos_sum"])) < 1e-6
    assert float(mask.sum()) == float(f["mask_sum"])


def test_g11_pair_labels():
    """warpLabels(bilinear=True) products vs the reference (collision-free point sets)."""
    g = G.load("g11_pair_labels.npz")
    for k in range(3):
        H, W = g["labels%d" % k].shape[1:]
        lab, res, bi = C.warp_labels_full(t(g["pts%d" % k].astype(np.int64)), H, W, t(g["H%d" % k]))
        assert torch.equal(lab, t(g["labels%d" % k])) and torch.equal(res, t(g["res%d" % k]))
        assert torch.equal(bi, t(g["bi%d" % k]))


@pytest.mark.parametrize("tag,arch", [("sp", ARCHS[0]), ("ssp", ARCHS[1])])
def test_g12_full_size_step(tag, arch):
    """G12: the real reference's pair step at the benchmark resolution 240x320 (B = 2): forward checksums, the first
    step's scalars and gradients (the compact fixture round-trips its inputs exactly)."""
    g = G.load("g12_step_%s_240x320.npz" % tag)
    sample = C.compact_from_npz(g)
    assert tuple(sample["image"].shape) == (2, 1, 240, 320)
    regen = C.make_compact_pair(2, 240, 320, seed=41, semantic=(tag == "ssp"))
    assert torch.equal(regen["labels_2D"], sample["labels_2D"]) and torch.equal(regen["image"], sample["image"])
    sd_np = C.init_state_dict(arch, seed=29)
    o = C.forward(C.to_torch(sd_np), sample["image"], arch)
    assert (o["semi"].double().sum(dim=(2, 3)) - t(g["fwd/semi_chsum"])).abs().max() < 1e-2
    assert (o["desc"][:, ::16, ::3, ::4] - t(g["fwd/desc_s"])).abs().max() < 1e-5
    tr = C.Trainer(arch, sd_np, lr=0.001, lambda_loss=1.0, multi_task=True)
    tr.real_batch_size = 10 ** 9
    tr.train_val_sample(sample, n_iter=1, train=True, indices=G.indices_from(g, "idx/", 2))
    for k in ("loss", "loss_det", "loss_det_warp", "loss_desc", "positive_dist", "negative_dist", "loss_sem", "loss_sem_warp"):
        ref = float(g["step0/" + k])
        assert abs(tr.scalar_dict[k] - ref) < 5e-5 * max(1.0, abs(ref)), (k, tr.scalar_dict[k], ref)
    for k in ("inc.conv.conv.0.weight", "inc.conv.conv.3.weight", "down2.mpconv.1.conv.0.weight", "convPb.weight", "bnDb.weight"):
        gr = tr.last_grads[k]
        assert abs(float(gr.norm()) - float(g["grad_norm/" + k])) < 1e-3 * float(g["grad_norm/" + k])
        assert (gr.reshape(-1)[:64] - t(g["grad_slice/" + k])).abs().max() < 1e-3 * float(gr.abs().max()) + 1e-7
    assert (tr.last_grads["eta"] - t(g["grad/eta"])).abs().max() < 1e-5


def test_forced_gates_hook_is_transparent_with_own_gates():
    """forward(forced=...) with the network's OWN ReLU gates and max-pool winners is the plain forward (values and
    gradients): the hook used by the GPU gate-flip test changes nothing but who decides the gates."""
    import torch.nn.functional as F
    arch = ARCHS[1]
    sd_np = C.init_state_dict(arch, seed=2)
    x = t(np.random.RandomState(1).uniform(0, 1, (2, 1, 32, 48)).astype(np.float32))
    sd = C.to_torch(sd_np, requires_grad=True)
    ref = C.forward(sd, x, arch)
    # own gates: recompute the pre-activations layer by layer
    relu, pool, h = {}, {}, x
    tb = C.layer_table(arch)
    with torch.no_grad():
        for i, (conv, bn, cin, cout, k) in enumerate(tb[:8]):
            if i in (2, 4, 6):
                N, Cc, Hh, Ww = h.shape
                win = h.view(N, Cc, Hh // 2, 2, Ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, Cc, Hh // 2, Ww // 2, 4)
                pool[i] = win.argmax(dim=4)
                h = F.max_pool2d(h, 2)
            z = F.batch_norm(F.conv2d(h, sd[conv + ".weight"], sd[conv + ".bias"], padding=1), None, None,
                             sd[bn + ".weight"], sd[bn + ".bias"], training=True, eps=1e-5)
            relu[conv] = z > 0
            h = F.relu(z)
        for conv, bn in (("convPa", "bnPa"), ("convDa", "bnDa"), ("convDS", "bnS1")):
            z = F.batch_norm(F.conv2d(h, sd[conv + ".weight"], sd[conv + ".bias"], padding=1), None, None,
                             sd[bn + ".weight"], sd[bn + ".bias"], training=True, eps=1e-5)
            relu[conv] = z > 0
    sd2 = C.to_torch(sd_np, requires_grad=True)
    out = C.forward(sd2, x, arch, forced={"relu": relu, "pool": pool})
    for k in ref:
        assert (out[k] - ref[k]).abs().max() < 1e-6, k
    w = {k: torch.randn_like(v) for k, v in ref.items()}
    sum((ref[k] * w[k]).sum() for k in ref).backward()
    sum((out[k] * w[k]).sum() for k in out).backward()
    for k in ("inc.conv.conv.0.weight", "down1.mpconv.1.conv.3.weight", "convDS.weight"):
        assert (sd[k].grad - sd2[k].grad).abs().max() < 1e-5 * float(sd[k].grad.abs().max()) + 1e-7, k
