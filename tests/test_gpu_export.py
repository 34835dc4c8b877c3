"""GPU: homography-adaptation export -- the HIP kernels through the C ABI against the
CPU oracle and the G8 fixtures generated from the real reference (SuperPointFrontend_torch / combine_heatmap).
Bars: point extraction (threshold, greedy NMS, border, order, top-k) bit-exact on identical heatmaps; heatmaps within
1e-5 of the reference (detector probabilities, fp32); soft-argmax offsets within 1e-5."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as C
from tests import golden_util as G

pytestmark = pytest.mark.gpu
G8 = ("sp_64x96_v6", "ssp_48x64_v5", "sp_120x160_v4")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.asarray(a))


def _oracle_pts(hm, thr, dist=4, border=4, top_k=0, subpixel=False):
    pts = C.get_pts_from_heatmap(hm, np.float32(thr), dist, border)
    if subpixel:
        pts = C.soft_argmax_points(hm, pts)
    pts = pts.transpose()
    return pts[:top_k] if top_k and pts.shape[0] > top_k else pts


@pytest.mark.parametrize("name", G8)
def test_views_and_masks(name):
    from semantic_superpoint_amd import lib as L
    g = G.load("g8_export_%s.npz" % name)
    views, masks = L.op_homoadapt_views(t(g["img"]).to(_dev()), t(g["inv_homographies"]))
    # white-noise image: |d img / d pixel| ~ 1, fp32 source coordinates carry a few 1e-6 pixels of rounding
    assert (views.cpu() - t(g["views"])).abs().max() < 5e-5
    if int(g["erosion"]) > 0:
        masks = L.op_erode(masks, int(g["erosion"]))
    assert float((masks.cpu() != t(g["valid_mask"])).float().mean()) < 1e-3  # nearest ties at .5 may flip


@pytest.mark.parametrize("name", G8)
def test_flatten_and_combine(name):
    from semantic_superpoint_amd import lib as L
    g = G.load("g8_export_%s.npz" % name)
    dev = _dev()
    sd = C.to_torch(C.init_state_dict(str(g["arch"]), seed=int(g["seed"])))
    with torch.no_grad():
        semi = C.forward(sd, t(g["views"]), str(g["arch"]), train=True)["semi"]
    heat = L.op_flatten_detection(semi.to(dev).contiguous())
    assert (heat.cpu() - t(g["views_heatmap"])).abs().max() < 1e-6
    mask = t(g["valid_mask"]).to(dev)
    masked = L.op_flatten_detection(semi.to(dev).contiguous(), mask)
    assert torch.equal(masked, heat * mask)
    agg = L.op_combine_heatmap(masked, mask, t(g["homographies"]))
    # random-weight heatmaps jump by ~0.1 between neighbouring pixels; fp32 source coordinates differ by ~1e-5 px
    assert (agg.cpu() - t(g["aggregate"])).abs().max() < 5e-6


@pytest.mark.parametrize("name", G8)
def test_points_from_reference_aggregate(name):
    """Same heatmap in, same points out: (x, y, conf) bit-exact, soft-argmax within 1e-5, same top-k cut."""
    from semantic_superpoint_amd import lib as L
    g = G.load("g8_export_%s.npz" % name)
    hm = t(g["aggregate"]).to(_dev())
    nms = L.op_heatmap_points(hm, float(g["thr"]), 4, 4)
    assert np.array_equal(nms, g["pts_nms"].T)
    sub = L.op_heatmap_points(hm, float(g["thr"]), 4, 4, top_k=int(g["top_k"]), subpixel=True)
    assert sub.shape == g["pts"].shape
    assert np.array_equal(sub[:, 2], g["pts"][:, 2])
    assert np.abs(sub[:, :2] - g["pts"][:, :2]).max() < 1e-5


def _heatmaps(rs):
    H, W = 240, 320
    peaky = np.exp(3.0 * rs.randn(H, W)).astype(np.float32)
    peaky /= peaky.max()
    yield "peaky", peaky, 0.015, 4, 4
    yield "dense_uniform", rs.uniform(0, 1, (H, W)).astype(np.float32), 0.5, 4, 4
    yield "all_candidates", rs.uniform(0.1, 1, (H, W)).astype(np.float32), 0.015, 4, 4
    ramp = (np.arange(H)[:, None] * W + np.arange(W)[None, :]).astype(np.float32) / (H * W) + 0.1  # one long chain
    yield "ramp", ramp, 0.015, 4, 4
    yield "constant_ties", np.full((H, W), 0.25, np.float32), 0.015, 4, 4  # ties: lower row-major index wins
    q = np.round(rs.uniform(0, 1, (H, W)) * 8).astype(np.float32) / 8  # many ties inside NMS windows
    yield "quantised_ties", q, 0.2, 4, 4
    yield "dist1_border0", rs.uniform(0, 1, (48, 64)).astype(np.float32), 0.3, 1, 0
    yield "dist8", rs.uniform(0, 1, (96, 128)).astype(np.float32), 0.3, 8, 2
    yield "big_480x640", rs.uniform(0, 1, (480, 640)).astype(np.float32), 0.7, 4, 4
    nanmap = rs.uniform(0, 1, (64, 96)).astype(np.float32)
    nanmap[rs.uniform(size=nanmap.shape) < 0.3] = np.nan  # 0/0 of combine_heatmap where no view covers a pixel
    yield "with_nan", nanmap, 0.3, 4, 4
    yield "empty", np.zeros((64, 96), np.float32), 0.015, 4, 4
    one = np.zeros((64, 96), np.float32)
    one[30, 40] = 0.9
    yield "single", one, 0.015, 4, 4


def test_greedy_nms_matches_sequential_oracle():
    """The parallel fixed-point NMS reproduces nms_fast's sequential greedy result, including ties (stable order)."""
    from semantic_superpoint_amd import lib as L
    rs = np.random.RandomState(5)
    for name, hm, thr, dist, border in _heatmaps(rs):
        mine = L.op_heatmap_points(t(hm).to(_dev()), thr, dist, border)
        ref = _oracle_pts(hm, thr, dist, border)
        assert mine.shape == ref.shape, (name, mine.shape, ref.shape)
        assert np.array_equal(mine, ref), name
        # size-independent properties: kept points pairwise farther than dist, every candidate is covered
        if len(mine):
            xy = mine[:, :2].astype(np.int64)
            assert np.all(np.diff(mine[:, 2]) <= 0), name
            if len(xy) < 4000:
                d = np.abs(xy[:, None, :] - xy[None, :, :]).max(-1)
                np.fill_diagonal(d, 10 ** 6)
                assert d.min() > dist, name


def test_top_k_and_subpixel_on_random_heatmap():
    from semantic_superpoint_amd import lib as L
    rs = np.random.RandomState(6)
    hm = np.exp(2.0 * rs.randn(120, 160)).astype(np.float32)
    hm /= hm.max()
    ref = _oracle_pts(hm, 0.015, 4, 4, top_k=100, subpixel=True)
    mine = L.op_heatmap_points(t(hm).to(_dev()), 0.015, 4, 4, top_k=100, subpixel=True)
    assert mine.shape == ref.shape == (100, 3)
    assert np.array_equal(mine[:, 2], ref[:, 2])
    assert np.abs(mine[:, :2] - ref[:, :2]).max() < 1e-5


# Winograd F(4x4,3x3) carries ~6x the rounding noise of F(2x2,3x3) (1.5e-6 vs 2.4e-7 rel-L2 on a layer's data,
# profiles/r02_wino_error_probe.txt): forced onto every 3x3 layer (algo 10: the export's real kernel set at 480x640, which
# the small fixtures would otherwise never reach) the aggregated heat map sits 1.6e-5 from the reference instead of
# < 1e-5; the north-star tolerance is 1e-3.
HEAT_TOL = {1: 1e-5, 10: 3e-5}


@pytest.mark.parametrize("algo", [1, 10])
@pytest.mark.parametrize("name", G8)
def test_fused_export_against_reference(name, algo):
    """Engine.export_points (forward + flatten + combine + points in one call) on the fixture's views."""
    from semantic_superpoint_amd.lib import Engine, points_to_numpy
    g = G.load("g8_export_%s.npz" % name)
    dev = _dev()
    arch, thr, top_k = str(g["arch"]), float(g["thr"]), int(g["top_k"])
    n, _, H, W = g["views"].shape
    e = Engine(arch, n, H, W, dev, with_grad=False)
    e.set_conv_algo(algo)
    e.load_state_dict(C.init_state_dict(arch, seed=int(g["seed"])))
    views, masks, hms = (t(g[k]).to(dev).contiguous() for k in ("views", "valid_mask", "homographies"))
    out = e.export_points([views], [masks], [hms], conf_thresh=thr, nms_dist=4, top_k=top_k, subpixel=True,
                          want_heatmap=True)[0]
    agg = out["heatmap"].cpu().numpy()
    assert np.abs(agg - g["aggregate"]).max() < HEAT_TOL[algo]
    # BatchNorm ran in train mode over the views: running statistics move exactly as in the reference
    assert (e.state_dict()["bnPb.running_var"].cpu() - t(g["bnPb_running_var"])).abs().max() < 1e-4
    # points: exactly what the reference's extraction gives on THIS aggregate ...
    mine = points_to_numpy(out["pts"], out["count"], True)
    ref = _oracle_pts(agg, thr, 4, 4, top_k=top_k, subpixel=True)
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, 2], ref[:, 2])
    assert np.abs(mine[:, :2] - ref[:, :2]).max() < 1e-5
    # ... and nearly all of the reference's exported points (fp32 noise may flip a near-tie or a threshold case)
    gold = {(int(round(x)), int(round(y))) for x, y, _ in g["pts"]}
    hit = sum((int(round(x)), int(round(y))) in gold for x, y, _ in mine)
    assert hit >= 0.9 * len(gold), (hit, len(gold))


def test_two_images_per_call_match_single_calls():
    """The pair launch (one BatchNorm batch per image) gives each image the result of a call of its own."""
    from semantic_superpoint_amd.lib import Engine, points_to_numpy
    dev = _dev()
    arch, n, H, W = "SuperPointNet_gauss2", 6, 64, 96
    rs = np.random.RandomState(9)
    samples = [C.homo_adapt_sample(t(rs.uniform(0, 1, (H, W)).astype(np.float32)), n, rs) for _ in range(2)]
    views = [s["image"].to(dev).contiguous() for s in samples]
    masks = [s["valid_mask"].to(dev).contiguous() for s in samples]
    hms = [s["homographies"].to(dev).contiguous() for s in samples]
    sd = C.init_state_dict(arch, seed=3)
    e = Engine(arch, n, H, W, dev, with_grad=False)
    e.load_state_dict(sd)
    both = e.export_points(views, masks, hms, conf_thresh=0.0152, top_k=0, subpixel=False, want_heatmap=True)
    for k in range(2):
        e.load_state_dict(sd)
        one = e.export_points(views[k:k + 1], masks[k:k + 1], hms[k:k + 1], conf_thresh=0.0152, top_k=0, subpixel=False,
                              want_heatmap=True)[0]
        assert (one["heatmap"] - both[k]["heatmap"]).abs().max() < 5e-6
        a, b = points_to_numpy(one["pts"], one["count"], False), points_to_numpy(both[k]["pts"], both[k]["count"], False)
        sa, sb = {tuple(r[:2]) for r in a}, {tuple(r[:2]) for r in b}
        assert len(sa & sb) >= 0.95 * max(len(sa), 1)
        # oracle end to end on the same views
        o = C.export_points(C.to_torch(C.init_state_dict(arch, seed=3)), samples[k], arch, conf_thresh=np.float32(0.0152),
                            top_k=0, subpixel=False)
        assert (o["heatmap"].squeeze() - both[k]["heatmap"].cpu()).abs().max() < 1e-5


@pytest.mark.parametrize("algo", [1, 10])
def test_dropin_frontend_and_combine_heatmap(algo):
    """The reference's call sequence (export.py:296-309) through the drop-in names (algo 10: F(4x4,3x3) on every 3x3 layer)."""
    from semantic_superpoint_amd import lib as L
    L.set_conv_algo(algo)  # process default: the nn.Module creates its engine lazily
    try:
        _dropin_frontend_case(HEAT_TOL[algo])
    finally:
        L.set_conv_algo(1)


def _dropin_frontend_case(heat_tol):
    from semantic_superpoint_amd import export as X
    from semantic_superpoint_amd.models.SuperPointNet_gauss2 import SuperPointNet_gauss2
    g = G.load("g8_export_sp_64x96_v6.npz")
    dev = _dev()
    thr, top_k = float(g["thr"]), int(g["top_k"])
    cfg = {"model": {"name": "SuperPointNet_gauss2", "params": {}, "subpixel": {"enable": True}}}
    fe = X.SuperPointFrontend_torch(cfg, "", nms_dist=4, conf_thresh=thr, nn_thresh=0.7, device=dev, load=False)
    net = SuperPointNet_gauss2()
    net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in C.init_state_dict("SuperPointNet_gauss2", seed=int(g["seed"])).items()})
    fe.net = net.to(dev)
    img, mask = t(g["views"]).to(dev), t(g["valid_mask"]).to(dev)
    heat = fe.run(img, onlyHeatmap=True, train=False)
    assert (heat.cpu() - t(g["views_heatmap"])).abs().max() < heat_tol
    outputs = X.combine_heatmap(heat, t(g["homographies"]).unsqueeze(0).to(dev), mask, device=dev)
    assert outputs.shape == (1, 64, 96)
    assert (outputs.cpu().squeeze() - t(g["aggregate"])).abs().max() < heat_tol
    # identical heatmap in -> identical points out, through the reference's method names
    fe.heatmap = t(g["aggregate"])
    pts = fe.getPtsFromHeatmap(t(g["aggregate"]))
    assert np.array_equal(pts, g["pts_nms"])
    sub = fe.soft_argmax_points([pts])[0].transpose()[:top_k]
    assert np.abs(sub - g["pts"]).max() < 1e-5
    with pytest.raises(NotImplementedError):
        fe.run(img, onlyHeatmap=False)
    # nms_fast on an explicit corner list
    rs = np.random.RandomState(2)
    flat = rs.choice(64 * 96, 500, replace=False)
    corners = np.stack([flat % 96, flat // 96, rs.uniform(0.01, 1, 500)]).astype(np.float64)
    corners[2] = corners[2].astype(np.float32)
    kept, inds = fe.nms_fast(corners, 64, 96, 4)
    assert np.array_equal(kept, C.nms_fast(corners, 64, 96, 4))
    assert np.array_equal(corners[:, inds], kept)
    # the fused exporter on the same sample
    ex = X.HomoAdaptExporter(fe.net, dev, thr, 4, top_k, True)
    pts2, hms = ex([{"image": img, "valid_mask": mask, "homographies": t(g["homographies"])}], want_heatmap=True)
    assert (hms[0].cpu() - t(g["aggregate"])).abs().max() < heat_tol
    assert pts2[0].shape[1] == 3 and len(pts2[0]) <= top_k


def test_export_refuses_bad_arguments():
    from semantic_superpoint_amd.lib import Engine
    dev = _dev()
    e = Engine("SuperPointNet_gauss2", 4, 32, 48, dev, with_grad=False)
    v = torch.zeros(6, 1, 32, 48, device=dev)
    with pytest.raises(RuntimeError):  # more views than the engine was sized for
        e.export_points([v], [v], [torch.eye(3, device=dev).repeat(6, 1, 1)])
    with pytest.raises(RuntimeError):  # host tensors: no CPU fallback
        e.export_points([v.cpu()[:4]], [v.cpu()[:4]], [torch.eye(3).repeat(4, 1, 1)])


def test_logging_branch_nms_map_and_precision_recall():
    """heatmap_to_nms + batch_precision_recall against the reference's G9 arrays."""
    from semantic_superpoint_amd import lib as L
    g = G.load("g9_logging.npz")
    dev = _dev()
    heat = L.op_flatten_detection(t(g["semi"]).to(dev))
    assert (heat.cpu() - t(g["heat"])).abs().max() < 1e-6
    nms, pr = L.op_heatmap_nms(t(g["heat"]).to(dev), t(g["labels"]).to(dev))
    assert np.array_equal(nms.cpu().numpy(), g["nms"].astype(np.float32))
    prm = pr.cpu().numpy().mean(axis=0)
    assert abs(prm[0] - float(g["precision"])) < 1e-6 and abs(prm[1] - float(g["recall"])) < 1e-6
    nms2, none = L.op_heatmap_nms(t(g["heat"]).to(dev))
    assert none is None and torch.equal(nms2, nms)


def test_trainer_logs_precision_recall():
    """The drop-in trainer's logging branch: precision / recall of the un-warped view vs the oracle on the same step."""
    from semantic_superpoint_amd.Train_model_heatmap_all import Train_model_heatmap_all
    dev = _dev()
    B, H, W = 2, 64, 96
    cfg = {"data": {"dataset": "Coco", "semantic": False, "gaussian_label": {"enable": False},
                    "warped_pair": {"enable": True}},
           "model": {"name": "SuperPointNet_gauss2", "params": {}, "batch_size": B, "real_batch_size": B,
                     "learning_rate": 1e-3, "lambda_loss": 1, "multi_task_loss": True, "detector_loss": {"loss_type": "softmax"},
                     "sparse_loss": {"enable": True, "params": {"num_matching_attempts": 1000,
                                                                "num_masked_non_matches_per_match": 100,
                                                                "lamda_d": 1, "dist": "cos", "method": "2d"}}},
           "retrain": True, "reset_iter": True, "train_iter": 10, "validation_interval": 5, "tensorboard_interval": 2,
           "save_interval": 100, "ssp_sampler": "reference"}
    agent = Train_model_heatmap_all(cfg, device=dev)
    agent.loadModel()
    sd = C.init_state_dict("SuperPointNet_gauss2", seed=4)
    agent.net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    agent.dataParallel()
    sample = C.make_synthetic_pair(B, H, W, seed=8)
    agent.train_val_sample(sample, n_iter=1, train=True)
    assert "precision" not in agent.scalar_dict  # 1 % tensorboard_interval != 0
    state = {k: v.detach().cpu().clone() for k, v in agent.net.state_dict().items()}
    agent.train_val_sample(sample, n_iter=2, train=True)
    assert {"precision", "recall"} <= set(agent.scalar_dict)
    # oracle: forward of the un-warped view with the weights the step started from (train-mode BatchNorm)
    with torch.no_grad():
        semi = C.forward(C.to_torch(state), sample["image"], "SuperPointNet_gauss2", train=True)["semi"]
    heat = C.flatten_detection(semi).numpy()
    nms = np.stack([C.heatmap_nms(h) for h in heat])
    mine = agent.images_dict["heatmap_org_nms_batch"][:, 0]
    assert float((mine != nms).mean()) < 2e-3  # fp32 noise at the 0.015 threshold / near-ties may flip a few points
    pr = C.batch_precision_recall(t(nms[:, None]), sample["labels_2D"])
    assert abs(agent.scalar_dict["precision"] - pr["precision"]) < 0.02
    assert abs(agent.scalar_dict["recall"] - pr["recall"]) < 0.02
    # image overlays of the branch (utils/draw.py:50-56 restated): gray x 3, labels on red, NMS / heat map on green.
    # The NMS overlay is ONE image (sample 0: the reference passes heatmap_nms_batch[np.newaxis]), the heat-map overlay
    # covers the batch.
    def overlap(r, g, gray):
        img = np.concatenate((gray, gray, gray), axis=0)
        img[0] += r[0]
        img[1] += g[0]
        return np.clip(img, 0.0, 1.0)
    im = agent.images_dict
    assert im["original_nms_overlap"].shape == (1, 3, H, W) and im["original_heatmap_nms_overlap"].shape == (B, 3, H, W)
    assert {"warped_nms_overlap", "warped_heatmap_nms_overlap", "heatmap_warp_nms_batch"} <= set(im)
    want = overlap(sample["labels_2D"][0].numpy(), mine[0:1], sample["image"][0].numpy())
    assert np.abs(im["original_nms_overlap"][0] - want).max() < 1e-6
    for i in range(B):
        want = overlap(sample["labels_2D"][i].numpy(), heat[i].reshape(1, H, W), sample["image"][i].numpy())
        assert np.abs(im["original_heatmap_nms_overlap"][i] - want).max() < 1e-4
    agent.train_val_sample(sample, n_iter=3, train=False)  # validation always logs
    assert {"precision", "recall"} <= set(agent.scalar_dict)


def test_full_size_export_properties():
    """BASELINE configs[4] size (100 views of 240x320): size-independent properties instead of an oracle run.
    (1) identical calls give identical points; (2) with identity homographies and full masks every view is the same
    image, so the aggregate equals any single view's heatmap; (3) exported points are sorted, inside the border band and
    pairwise farther apart than the NMS distance."""
    from semantic_superpoint_amd import lib as L
    from semantic_superpoint_amd.lib import Engine, points_to_numpy
    dev = _dev()
    arch, n, H, W = "SuperPointNet_gauss2", 100, 240, 320
    e = Engine(arch, n, H, W, dev, with_grad=False)
    sd = C.init_state_dict(arch, seed=12)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(H, W, generator=g).to(dev)
    eye = torch.eye(3, device=dev).repeat(n, 1, 1).contiguous()
    views, masks = L.op_homoadapt_views(img, eye)
    # the identity warp still goes through linspace / unnormalise in fp32: views are copies of each other, not of img
    assert torch.equal(views[0], views[57]) and (views[0, 0] - img).abs().max() < 1e-4 and float(masks.min()) == 1.0
    outs = []
    for _ in range(2):
        e.load_state_dict(sd)
        o = e.export_points([views], [masks], [eye], conf_thresh=0.0156, nms_dist=4, top_k=600, subpixel=True,
                            want_heatmap=True)[0]
        outs.append((points_to_numpy(o["pts"], o["count"], True), o["heatmap"].clone()))
    assert np.array_equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # single-view heatmap of the same BatchNorm batch (all views identical -> batch statistics = this view's)
    e.load_state_dict(sd)
    semi = e.forward(views, train=True, want=("semi",))["semi"]
    single = L.op_flatten_detection(semi[:1].contiguous())[0, 0]
    assert (outs[0][1] - single).abs().max() < 1e-5
    pts = outs[0][0]
    assert 0 < len(pts) <= 600 and np.all(np.diff(pts[:, 2]) <= 0)
    xy = np.round(pts[:, :2] - (pts[:, :2] - np.round(pts[:, :2])))  # integer positions (soft-argmax moves < 2 px)
    nms = points_to_numpy(e.export_points([views], [masks], [eye], conf_thresh=0.0156, nms_dist=4, top_k=600,
                                          subpixel=False)[0]["pts"][:len(pts)], torch.tensor([len(pts)]), False)
    q = nms[:, :2].astype(np.int64)
    assert q[:, 0].min() >= 4 and q[:, 0].max() < W - 4 and q[:, 1].min() >= 4 and q[:, 1].max() < H - 4
    d = np.abs(q[:, None, :] - q[None, :, :]).max(-1)
    np.fill_diagonal(d, 10 ** 6)
    assert d.min() > 4
    assert np.abs(pts[:, :2] - nms[:, :2]).max() <= 2.0 + 1e-6


def test_export_at_config5_size_480x640():
    """BASELINE configs[4] at its stated size: 100 homography views of a 480x640 image per BatchNorm batch (7.9 GB per
    240x320-equivalent layer pair: one buffer descriptor per image).  Properties: the aggregate heatmap is finite where at
    least one view covers the pixel and NaN nowhere else; points are sorted by confidence, cut at top-k, inside the border
    band, pairwise farther apart than the NMS distance, all above the threshold; ONE call with two images == two calls with
    one image each (the two images ride the two BatchNorm problems of every launch independently)."""
    from semantic_superpoint_amd import lib as L
    from semantic_superpoint_amd import synth
    from semantic_superpoint_amd.lib import Engine, points_to_numpy
    dev = _dev()
    arch, n, H, W = "SuperPointNet_gauss2", 100, 480, 640
    thr, nms_d, topk = 0.0155, 4, 600
    e = Engine(arch, n, H, W, dev, with_grad=False)
    sd = C.init_state_dict(arch, seed=12)
    rs = np.random.RandomState(5)
    g = torch.Generator().manual_seed(9)
    imgs = [torch.rand(H, W, generator=g).to(dev) for _ in range(2)]
    hom = []
    for _ in range(2):
        hs = np.stack([np.linalg.inv(synth.sample_homography(rs, **synth.WARP_PARAMS)) for _ in range(n)])
        hs[0] = np.identity(3)
        hs = torch.from_numpy(hs.astype(np.float32))
        hom.append((hs.to(dev), torch.inverse(hs).contiguous().to(dev)))
    vm = [L.op_homoadapt_views(imgs[k], hom[k][1]) for k in range(2)]
    e.load_state_dict(sd)
    both = e.export_points([v for v, _ in vm], [m for _, m in vm], [hom[k][0] for k in range(2)], conf_thresh=thr,
                           nms_dist=nms_d, top_k=topk, subpixel=True, want_heatmap=True)
    both = [(points_to_numpy(o["pts"], o["count"], True), o["heatmap"].clone()) for o in both]
    for k in range(2):
        pts, heat = both[k]
        covered = (vm[k][1].sum(dim=0)[0] > 0)
        assert bool(torch.isfinite(heat[covered]).all()), "aggregate must be finite wherever a view covers the pixel"
        assert 0 < len(pts) <= topk and np.all(np.isfinite(pts))
        assert np.all(np.diff(pts[:, 2]) <= 0) and pts[-1, 2] >= np.float32(thr)
        e.load_state_dict(sd)
        single = e.export_points([vm[k][0]], [vm[k][1]], [hom[k][0]], conf_thresh=thr, nms_dist=nms_d, top_k=topk,
                                 subpixel=False, want_heatmap=True)[0]
        q = points_to_numpy(single["pts"], single["count"], False)
        assert len(q) == len(pts)
        qi = q[:, :2].astype(np.int64)
        assert qi[:, 0].min() >= 4 and qi[:, 0].max() < W - 4 and qi[:, 1].min() >= 4 and qi[:, 1].max() < H - 4
        d = np.abs(qi[:, None, :] - qi[None, :, :]).max(-1)
        np.fill_diagonal(d, 10 ** 6)
        assert d.min() > nms_d
        # the single-image call reproduces the image's half of the two-image call: same BatchNorm batch (its 100 views),
        # same kernels; the statistics atomics commit in another order -> fp32 noise on the heatmap, identical points
        hs_ = single["heatmap"]
        both_ok = torch.isfinite(heat) & torch.isfinite(hs_)
        assert bool((torch.isfinite(heat) == torch.isfinite(hs_)).all())
        assert float((heat[both_ok] - hs_[both_ok]).abs().max()) < 1e-4  # measured 1.1e-5 (softmax of 100-view sums)
        # every refined point sits within 2 px of exactly one un-refined point (the NMS distance keeps them > 4 apart) with
        # the same confidence; the ORDER of near-equal confidences may differ between the two runs (fp32 noise)
        dd = np.abs(pts[:, None, :2] - q[None, :, :2]).max(-1)
        j = dd.argmin(1)
        assert dd.min(1).max() <= 2.0 + 1e-6 and len(set(j.tolist())) == len(pts)
        assert np.abs(q[j, 2] - pts[:, 2]).max() < 1e-4
