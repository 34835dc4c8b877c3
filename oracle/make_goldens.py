"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/ref_harness.py) on seeded inputs, asserting on the way that oracle/cpu_ref.py reproduces it.

Run in the build container only:  python oracle/make_goldens.py
Fixtures are DATA (inputs + the reference's outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as C  # noqa: E402
from oracle import ref_harness as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ARCHS = ("SuperPointNet_gauss2", "SuperPointNet_gauss2_ssmall")


def npy(t):
    return t.detach().cpu().numpy().copy()  # a copy: fixtures are written at the end of a generator, the tensors live on


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, "%s: oracle differs from reference by %g (tol %g)" % (what, err, tol)
    return err


def close_adam(a, b, lr, what):
    """Post-Adam parameters: the first steps move every element by ~lr*sign(g), so elements whose
    true gradient is ~0 (dead ReLU taps) are rounding-noise driven on BOTH sides.  Require: max
    difference <= 2.2*lr and fewer than 5 % of elements off by more than 5 % of lr."""
    d = (torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs()
    frac = float((d > 0.05 * lr).double().mean())
    assert float(d.max()) <= 2.2 * lr and frac < 5e-2, "%s: max %g frac %g" % (what, float(d.max()), frac)


def ref_net(arch, sd):
    import importlib
    R.install()
    mod = importlib.import_module("models." + arch)
    net = getattr(mod, arch)()
    net.load_state_dict({k: torch.as_tensor(np.array(v)) for k, v in sd.items()})
    return net


def g1_forward():
    """G1: model forward (train mode, two consecutive forwards -> running stats)."""
    for arch in ARCHS:
        sd = C.init_state_dict(arch, seed=11)
        net = ref_net(arch, sd)
        rs = np.random.RandomState(5)
        x1 = torch.from_numpy(rs.uniform(0, 1, (2, 1, 32, 48)).astype(np.float32))
        x2 = torch.from_numpy(rs.uniform(0, 1, (2, 1, 32, 48)).astype(np.float32))
        o1 = net(x1)
        o2 = net(x2)
        tsd = C.to_torch(sd)
        p1 = C.forward(tsd, x1, arch)
        p2 = C.forward(tsd, x2, arch)
        for k in o1:
            close(o1[k], p1[k], 1e-6, "G1 %s %s" % (arch, k))
            close(o2[k], p2[k], 1e-6, "G1 %s %s (2nd)" % (arch, k))
        rsd = net.state_dict()
        for k in rsd:
            close(rsd[k], tsd[k], 1e-6, "G1 state " + k)
        save = {"x1": npy(x1), "x2": npy(x2), "semi1": npy(o1["semi"]), "desc1": npy(o1["desc"]),
                "semi2": npy(o2["semi"]), "desc2": npy(o2["desc"])}
        if "sem" in o1:
            save["sem1_s"] = npy(o1["sem"][:, ::7, ::3, ::5])  # strided slice keeps the file small
            save["sem2_s"] = npy(o2["sem"][:, ::7, ::3, ::5])
        for k in rsd:
            if "running" in k or "num_batches" in k:
                save["state/" + k] = npy(rsd[k])
        np.savez_compressed(os.path.join(OUT, "g1_forward_%s.npz" % arch), **save)
        # eval-mode forward (Val_model_heatmap.py:68 calls .eval())
        net.eval()
        oe = net(x1)
        pe = C.forward(tsd, x1, arch, train=False)
        for k in oe:
            close(oe[k], pe[k], 1e-6, "G1 eval %s" % k)
        np.savez_compressed(os.path.join(OUT, "g1_eval_%s.npz" % arch), semi=npy(oe["semi"]), desc=npy(oe["desc"]))


def g2_labels():
    """G2: labels2Dto3D / getMasks - exact arrays."""
    R.install()
    from utils.utils import labels2Dto3D
    import Train_model_frontend_all as TF
    rs = np.random.RandomState(7)
    lab = (rs.uniform(size=(2, 1, 32, 48)) < 0.02).astype(np.float32)
    lab[0, 0, 0:8, 0:8] = 0  # an empty cell -> dustbin 1
    lab[1, 0, 8, 8] = 1; lab[1, 0, 9, 9] = 1  # two points in one cell -> renormalised to 0.5
    gau = lab * rs.uniform(0.2, 1.0, size=lab.shape).astype(np.float32)  # "gaussian" labels in [0,1]
    mask = (rs.uniform(size=(2, 1, 32, 48)) < 0.995).astype(np.float32)
    out = {}
    for name, arr in (("bin", lab), ("gauss", gau)):
        r = labels2Dto3D(torch.from_numpy(arr), 8, add_dustbin=True).float()
        o = C.labels2Dto3D(torch.from_numpy(arr), 8, True).float()
        assert torch.equal(torch.nan_to_num(r), torch.nan_to_num(o)), "G2 labels " + name
        out["labels_" + name] = arr
        out["labels3D_" + name] = npy(r)
    rm = TF.Train_model_frontend_all.getMasks(None, torch.from_numpy(mask), 8)
    om = C.get_masks(torch.from_numpy(mask), 8)
    assert torch.equal(rm, om), "G2 masks"
    out["mask"] = mask
    out["mask3D"] = npy(rm)
    np.savez_compressed(os.path.join(OUT, "g2_labels.npz"), **out)


def g3_detector_loss():
    R.install()
    import Train_model_heatmap_all as T
    rs = np.random.RandomState(9)
    semi = torch.from_numpy((rs.randn(2, 65, 4, 6) * 3).astype(np.float32)).requires_grad_(True)
    semi.data[0, :, 0, 0] = torch.tensor([200.0] + [0.0] * 64)  # saturates: exercises the -100 log clamp
    lab = (rs.uniform(size=(2, 1, 32, 48)) < 0.03).astype(np.float32)
    t = C.labels2Dto3D(torch.from_numpy(lab)).float()
    mask = torch.from_numpy((rs.uniform(size=(2, 4, 6)) < 0.8).astype(np.float32))
    lr = T.Train_model_heatmap_all.detector_loss(None, semi, t, mask, "softmax")
    gr, = torch.autograd.grad(lr, semi)
    s2 = semi.detach().clone().requires_grad_(True)
    lo = C.detector_loss(s2, t, mask)
    go, = torch.autograd.grad(lo, s2)
    close(lr, lo, 1e-6, "G3 loss")
    close(gr, go, 1e-7, "G3 grad")
    np.savez_compressed(os.path.join(OUT, "g3_detector_loss.npz"), semi=npy(semi), target=npy(t), mask=npy(mask),
                        loss=npy(lr), dsemi=npy(gr))


def g5_sem_loss():
    R.install()
    import Train_model_heatmap_all as T
    rs = np.random.RandomState(13)
    pred = torch.from_numpy(rs.randn(2, 133, 16, 24).astype(np.float32)).requires_grad_(True)
    lab = torch.from_numpy(rs.randint(0, 134, size=(2, 16, 24)).astype(np.int64))
    lr = T.Train_model_heatmap_all.sem_loss(None, pred, lab, "cpu")
    gr, = torch.autograd.grad(lr, pred)
    p2 = pred.detach().clone().requires_grad_(True)
    lo = C.sem_loss(p2, lab)
    go, = torch.autograd.grad(lo, p2)
    close(lr, lo, 0, "G5 loss")
    close(gr, go, 0, "G5 grad")
    np.savez_compressed(os.path.join(OUT, "g5_sem_loss.npz"), pred_s=npy(pred[:, ::9]), label=npy(lab), loss=npy(lr),
                        dpred_s=npy(gr[:, ::9]), seed=13)


def _capture_sparse(desc, desc_w, Hs, **params):
    """Run the reference's batch_descriptor_loss_sparse with the two hinge terms monkey-patched to
    record the sampled coordinates / indices."""
    R.install()
    import utils.loss_functions.sparse_loss as SL
    from utils.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss as P
    rec = []
    om, on = P.match_loss, P.non_match_descriptor_loss

    def ml(a, b, ma, mb, **kw):
        rec.append({"ma": ma.detach().clone().view(-1, 2), "mb": mb.detach().clone().view(-1, 2)})
        return om(a, b, ma, mb, **kw)

    def nl(a, b, na, nb, **kw):
        rec[-1]["na"] = na.detach().clone()
        rec[-1]["nb"] = nb.detach().clone()
        return on(a, b, na, nb, **kw)

    P.match_loss, P.non_match_descriptor_loss = staticmethod(ml), staticmethod(nl)
    try:
        out = SL.batch_descriptor_loss_sparse(desc, desc_w, Hs, device="cpu", **params)
    finally:
        P.match_loss, P.non_match_descriptor_loss = staticmethod(om), staticmethod(on)
    return out, rec


def g4_sparse_loss():
    params = {"num_matching_attempts": 1000, "num_masked_non_matches_per_match": 100, "lamda_d": 1, "dist": "cos",
              "method": "2d"}
    for tag, (Hc, Wc) in (("small", (4, 6)), ("full", (30, 40))):
        rs = np.random.RandomState(17)
        B = 2
        d = rs.randn(B, 256, Hc, Wc).astype(np.float32)
        dw = (0.6 * d + 0.8 * rs.randn(B, 256, Hc, Wc)).astype(np.float32)  # correlated: many hard negatives
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        dw /= np.linalg.norm(dw, axis=1, keepdims=True)
        Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(B)]).astype(np.float32))
        desc = torch.from_numpy(d).requires_grad_(True)
        desc_w = torch.from_numpy(dw).requires_grad_(True)
        np.random.seed(123); torch.manual_seed(321)
        (loss, _, pos, neg), rec = _capture_sparse(desc, desc_w, Hs, **params)
        g_d, g_dw = torch.autograd.grad(loss + 0.5 * pos + 0.25 * neg, (desc, desc_w))
        # oracle with the SAME rng streams must reproduce the sampled indices bit-exactly
        np.random.seed(123); torch.manual_seed(321)
        d2 = desc.detach().clone().requires_grad_(True)
        dw2 = desc_w.detach().clone().requires_grad_(True)
        lo, po, no, used = C.batch_descriptor_loss_sparse(d2, dw2, Hs, None, 1.0, 1000, 100, np.random, None)
        for i in range(B):
            wh = torch.tensor([Wc, Hc]).float()
            assert torch.equal(C.norm_pts(used[i]["uv_a"], wh), rec[i]["ma"]), "G4 uv_a"
            assert torch.equal(C.norm_pts(used[i]["uv_b"], wh), rec[i]["mb"]), "G4 uv_b"
            ia = (used[i]["uv_a"][:, 0] + used[i]["uv_a"][:, 1] * Wc).long().repeat_interleave(100)
            assert torch.equal(ia, rec[i]["na"]) and torch.equal(used[i]["nm_b"], rec[i]["nb"]), "G4 non-match idx"
        close(loss, lo, 1e-6, "G4 loss"); close(pos, po, 1e-6, "G4 pos"); close(neg, no, 1e-6, "G4 neg")
        go_d, go_dw = torch.autograd.grad(lo + 0.5 * po + 0.25 * no, (d2, dw2))
        close(g_d, go_d, 1e-7, "G4 d desc"); close(g_dw, go_dw, 1e-7, "G4 d desc_w")
        save = {"H": npy(Hs), "loss": npy(loss), "pos": npy(pos), "neg": npy(neg), "seed": 17,
                "grad_weights": np.array([1.0, 0.5, 0.25], np.float32)}
        if tag == "small":
            save.update({"desc": d, "desc_w": dw, "ddesc": npy(g_d), "ddesc_w": npy(g_dw)})
        else:  # inputs are regenerated from the seed by the test (see tests/golden_util.py)
            save.update({"ddesc_s": npy(g_d[:, ::16]), "ddesc_w_s": npy(g_dw[:, ::16]),
                         "ddesc_norm": np.float32(g_d.norm()), "ddesc_w_norm": np.float32(g_dw.norm()),
                         "desc_sum": np.float64(d.astype(np.float64).sum()),
                         "desc_w_sum": np.float64(dw.astype(np.float64).sum())})
        for i in range(B):
            save["uv_a%d" % i] = npy(used[i]["uv_a"]).astype(np.int16)
            save["uv_b%d" % i] = npy(used[i]["uv_b"]).astype(np.int16)
            save["nm_b%d" % i] = npy(used[i]["nm_b"]).astype(np.int16)
        np.savez_compressed(os.path.join(OUT, "g4_sparse_loss_%s.npz" % tag), **save)
    # by-eye KAT of the reference (sparse_loss.py:335-345): identical descriptors + identity H => pos == 0.
    # It holds for method="1d" (index_select); with the configs' method="2d" the bilinear sample at
    # u*(Wc-1)/Wc blends neighbours, so pos = mean(1-|blend|^2) > 0 - both facts are asserted here.
    dd = torch.from_numpy(d)
    np.random.seed(1); torch.manual_seed(1)
    (l, _, p, n), _ = _capture_sparse(dd, dd, torch.eye(3).repeat(B, 1, 1), **dict(params, method="1d"))
    assert abs(float(p)) < 1e-6, float(p)
    np.random.seed(1); torch.manual_seed(1)
    (l, _, p2, n), _ = _capture_sparse(dd, dd, torch.eye(3).repeat(B, 1, 1), **params)
    np.random.seed(1); torch.manual_seed(1)
    _, po, _, _ = C.batch_descriptor_loss_sparse(dd, dd, torch.eye(3).repeat(B, 1, 1))
    assert float(p2) > 0.01 and abs(float(p2) - float(po)) < 1e-6


def g14_sparse_loss_variants():
    """G14: descriptor_loss_sparse with the parameter values no shipped config selects but the function accepts
    (sparse_loss.py:76-77, pixelwise_contrastive_loss.py:140): method "1d" (index_select at the cell instead of the bilinear
    grid_sample) and dist "euclidean" (squared distance / (max(0, ||a - b|| - 0.2))^2 instead of the hinges on the dot product)."""
    for tag, (Hc, Wc) in (("small", (4, 6)), ("mid", (9, 12))):
        for method, dist in (("1d", "cos"), ("2d", "euclidean"), ("1d", "euclidean")):
            params = {"num_matching_attempts": 200, "num_masked_non_matches_per_match": 20, "lamda_d": 1, "dist": dist, "method": method}
            rs = np.random.RandomState(23)
            B = 2
            d = rs.randn(B, 256, Hc, Wc).astype(np.float32)
            dw = (0.9 * d + 0.45 * rs.randn(B, 256, Hc, Wc)).astype(np.float32)  # close pairs: hard negatives in both metrics
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            dw /= np.linalg.norm(dw, axis=1, keepdims=True)
            Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(B)]).astype(np.float32))
            desc = torch.from_numpy(d).requires_grad_(True)
            desc_w = torch.from_numpy(dw).requires_grad_(True)
            R.install()
            import utils.loss_functions.sparse_loss as SL
            np.random.seed(77); torch.manual_seed(78)
            loss, _, pos, neg = SL.batch_descriptor_loss_sparse(desc, desc_w, Hs, device="cpu", **params)
            g_d, g_dw = torch.autograd.grad(loss + 0.5 * pos + 0.25 * neg, (desc, desc_w))
            np.random.seed(77); torch.manual_seed(78)
            d2 = desc.detach().clone().requires_grad_(True)
            dw2 = desc_w.detach().clone().requires_grad_(True)
            lo, po, no, used = C.batch_descriptor_loss_sparse(d2, dw2, Hs, None, 1.0, 200, 20, np.random, None, dist, method)
            close(loss, lo, 1e-6, "G14 loss"); close(pos, po, 1e-6, "G14 pos"); close(neg, no, 1e-6, "G14 neg")
            go_d, go_dw = torch.autograd.grad(lo + 0.5 * po + 0.25 * no, (d2, dw2))
            close(g_d, go_d, 1e-7, "G14 d desc"); close(g_dw, go_dw, 1e-7, "G14 d desc_w")
            assert float(neg) > 0 and float(pos) > 0, (tag, method, dist, float(pos), float(neg))
            save = {"H": npy(Hs), "loss": npy(loss), "pos": npy(pos), "neg": npy(neg), "seed": 23, "n_match": 200, "n_non": 20,
                    "grad_weights": np.array([1.0, 0.5, 0.25], np.float32), "desc": d, "desc_w": dw, "ddesc": npy(g_d),
                    "ddesc_w": npy(g_dw)}
            for i in range(B):
                save["uv_a%d" % i] = npy(used[i]["uv_a"]).astype(np.int16)
                save["uv_b%d" % i] = npy(used[i]["uv_b"]).astype(np.int16)
                save["nm_b%d" % i] = npy(used[i]["nm_b"]).astype(np.int16)
            np.savez_compressed(os.path.join(OUT, "g14_sparse_loss_%s_%s_%s.npz" % (method, dist, tag)), **save)


def _sample_to_npz(s):
    return {("in/" + k): npy(v) for k, v in s.items()}


def g6_train_step():
    """G6: full train_val_sample (2 forwards + losses + backward + Adam) on the reference vs oracle."""
    cases = [("sp_64x96", "SuperPointNet_gauss2", 64, 96, dict()),
             ("ssp_64x96", "SuperPointNet_gauss2_ssmall", 64, 96, dict()),
             ("pair_lambda0_32x48", "SuperPointNet_gauss2", 32, 48, dict(lambda_loss=0)),  # PAIR step, descriptor loss off
             ("sp_dense_64x96", "SuperPointNet_gauss2", 64, 96, dict(dense=True)),
             ("sp_dense_uniform_64x96", "SuperPointNet_gauss2", 64, 96, dict(dense=True, multi_task=False))]
    only = os.environ.get("SSP_G6_ONLY")
    for tag, arch, H, W, opt in cases:
        if only and only not in tag:
            continue
        semantic = arch.endswith("ssmall")
        lam = opt.get("lambda_loss", 1)
        cfg = R.base_config(semantic=semantic, H=H, W=W, batch=2, lr=0.001, lambda_loss=lam,
                            multi_task=opt.get("multi_task", True))
        kw = dict(lambda_loss=float(lam), multi_task=opt.get("multi_task", True))
        if opt.get("dense"):  # the shipped spelling `lambda_d` is swallowed by descriptor_loss(**config): lamda_d = 250
            cfg["model"]["dense_loss"] = {"enable": True, "params": {"descriptor_dist": 4, "lambda_d": 800}}
            kw["dense"] = {"descriptor_dist": 4, "lambda_d": 800}
        sd = C.init_state_dict(arch, seed=23)
        agent = R.make_trainer(cfg, sd)
        sample = C.make_synthetic_pair(2, H, W, seed=31, semantic=semantic, kp_prob=0.01)
        tr = C.Trainer(arch, sd, lr=0.001, **kw)
        steps = {}
        noisy = {conv + ".bias" for conv, bn, _, _, _ in C.layer_table(arch) if bn is not None}
        for it in range(2):
            np.random.seed(100 + it); torch.manual_seed(200 + it)
            # n_iter=1,2: skips the tensorboard branch (uses removed np.int) - SURVEY.md 8c
            agent.optimizer.zero_grad() if False else None
            grads_before = None
            l_ref = agent.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=it + 1, train=True)
            sc_ref = {k: float(v) for k, v in agent.scalar_dict.items()}
            np.random.seed(100 + it); torch.manual_seed(200 + it)
            l_or = tr.train_val_sample(sample, n_iter=it + 1, train=True)
            for k in sc_ref:
                close(sc_ref[k], tr.scalar_dict[k], 2e-5 * max(1.0, abs(sc_ref[k])), "G6 %s scalar %s it%d" % (tag, k, it))
            steps[it] = sc_ref
            # post-step parameters (weights / BN affine / eta; NOT the conv biases feeding a BN:
            # their true gradient is 0 and Adam amplifies rounding noise - SURVEY.md section 7)
            for k, p in agent.net.named_parameters():
                if k in noisy:
                    continue
                close_adam(p, tr.sd[k], 0.001, "G6 %s post-step %s it%d" % (tag, k, it))
            close(agent.multi_task_loss.eta, tr.eta, 1e-5, "G6 eta")
        # gradients of a fresh step without optimizer step: recompute on a new agent for storage
        agent2 = R.make_trainer(cfg, sd)
        np.random.seed(100); torch.manual_seed(200)
        agent2.real_batch_size = 10 ** 9  # never step: leaves .grad in place
        agent2.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=1, train=True)
        tr2 = C.Trainer(arch, sd, lr=0.001, **kw)
        tr2.real_batch_size = 10 ** 9
        np.random.seed(100); torch.manual_seed(200)
        tr2.train_val_sample(sample, n_iter=1, train=True)
        save = _sample_to_npz(sample)
        for k, p in agent2.net.named_parameters():
            g = p.grad
            go = tr2.last_grads[k]
            if g is None:  # lambda_loss == 0: the descriptor head is not in the graph
                assert go is None, k
                continue
            scale = max(1e-6, float(g.abs().max()))
            if k in noisy:  # true gradient is 0: both sides hold rounding noise only
                assert float(g.abs().max()) < 1e-3 and float(go.abs().max()) < 1e-3, k
            else:
                close(g, go, 2e-4 * scale + 1e-6, "G6 %s grad %s" % (tag, k))
            save["grad_norm/" + k] = np.float32(g.norm().item())
            save["grad_slice/" + k] = npy(g.reshape(-1)[:64])
        if agent2.multi_task_loss.eta.grad is not None:
            save["grad/eta"] = npy(agent2.multi_task_loss.eta.grad)
            close(agent2.multi_task_loss.eta.grad, tr2.last_grads["eta"], 1e-5, "G6 eta grad")
        else:
            assert tr2.last_grads["eta"] is None
        for i, idx in enumerate(tr2.aux["indices"] or []):
            save["idx/uv_a%d" % i] = npy(idx["uv_a"]).astype(np.int16)
            save["idx/uv_b%d" % i] = npy(idx["uv_b"]).astype(np.int16)
            save["idx/nm_b%d" % i] = npy(idx["nm_b"]).astype(np.int16)
        for it in steps:
            for k, v in steps[it].items():
                save["step%d/%s" % (it, k)] = np.float32(v)
        save["post/eta"] = npy(agent.multi_task_loss.eta)
        for k in ("inc.conv.conv.3.weight", "down3.mpconv.1.conv.4.weight", "convPb.weight", "bnDb.bias"):
            save["post_slice/" + k] = npy(dict(agent.net.named_parameters())[k].reshape(-1)[:64])
        rsd = agent.net.state_dict()
        for k in rsd:
            if "running_var" in k:
                save["post_state/" + k] = npy(rsd[k])
        np.savez_compressed(os.path.join(OUT, "g6_step_%s.npz" % tag), **save)
        print("G6", tag, "ok; loss", steps[0]["loss"], "->", steps[1]["loss"])


def g7_warps():
    R.install()
    from utils.utils import inv_warp_image_batch, compute_valid_mask
    from utils.homographies import scale_homography_torch
    from datasets.data_tools import warpLabels
    rs = np.random.RandomState(41)
    H, W = 40, 56
    Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(4)]).astype(np.float32))
    inv = torch.inverse(Hs).contiguous()
    img = torch.from_numpy(rs.uniform(0, 1, (4, 1, H, W)).astype(np.float32))
    wr = inv_warp_image_batch(img, inv, mode="bilinear")
    wo = C.inv_warp_image_batch(img, inv, mode="bilinear")
    close(wr, wo, 1e-6, "G7 warp")
    mr = compute_valid_mask(torch.tensor([H, W]), inv, erosion_radius=0)
    mo = C.compute_valid_mask((H, W), inv, 0)
    assert torch.equal(mr, mo), "G7 mask"
    save = {"H": npy(Hs), "img": npy(img), "warped": npy(wr), "mask": npy(mr)}
    for i in range(4):
        pts = torch.nonzero(torch.from_numpy((rs.uniform(size=(H, W)) < 0.02))).flip(1)
        lr = warpLabels(pts, H, W, Hs[i])["labels"]
        lo = C.warp_labels(pts, H, W, Hs[i])
        assert torch.equal(lr, lo), "G7 warpLabels"
        close(scale_homography_torch(Hs[i], (30, 40)), C.scale_homography(Hs[i], (30, 40)), 0, "G7 scaleH")
        save["pts%d" % i] = npy(pts).astype(np.int16)
        save["wlabels%d" % i] = npy(lr)
        save["Hcell%d" % i] = npy(scale_homography_torch(Hs[i], (30, 40)))
    np.savez_compressed(os.path.join(OUT, "g7_warps.npz"), **save)


def g10_dense_loss():
    """Dense descriptor loss (utils/utils.py:779-893) with the reference's autograd gradients."""
    R.install()
    from utils.utils import descriptor_loss
    for name, B, Hc, Wc, seed in (("small", 2, 6, 8, 3), ("full", 2, 30, 40, 5)):
        rs = np.random.RandomState(seed)
        d = rs.randn(B, 256, Hc, Wc).astype(np.float32)
        dw = (0.8 * d + 0.6 * rs.randn(B, 256, Hc, Wc)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        dw /= np.linalg.norm(dw, axis=1, keepdims=True)
        Hs = torch.from_numpy(np.stack([np.linalg.inv(C.sample_homography(rs)) for _ in range(B)]).astype(np.float32))
        mv = torch.from_numpy((rs.uniform(size=(B, 1, Hc, Wc)) > 0.15).astype(np.float32))
        out = {}
        for who, fn in (("ref", lambda a, b: descriptor_loss(a, b, Hs, mask_valid=mv, device="cpu", descriptor_dist=4,
                                                             lambda_d=800)),
                        ("oracle", lambda a, b: C.descriptor_loss_dense(a, b, Hs, mv, lamda_d=250.0, descriptor_dist=4.0))):
            a = torch.from_numpy(d).requires_grad_(True)
            b = torch.from_numpy(dw).requires_grad_(True)
            loss, mask, pos, neg = fn(a, b)
            g_loss = torch.autograd.grad(loss, (a, b), retain_graph=True)
            g_mt = torch.autograd.grad(0.5 * (pos + neg), (a, b))  # what the multi-task loss differentiates
            out[who] = (loss, mask, pos, neg, g_loss, g_mt)
        r, o = out["ref"], out["oracle"]
        for i in (0, 2, 3):
            close(r[i], o[i], 1e-6 * max(1.0, abs(float(r[i]))), "G10 %s scalar %d" % (name, i))
        assert torch.equal(r[1], o[1]), "G10 mask"
        for k in range(2):
            close(r[4][k], o[4][k], 1e-7, "G10 grad loss")
            close(r[5][k], o[5][k], 1e-7, "G10 grad multi-task")
        save = dict(seed=seed, homographies=npy(Hs), mask_valid=npy(mv), loss=float(r[0]), pos_sum=float(r[2]),
                    neg_sum=float(r[3]), mask_sum=float(r[1].sum()))
        if name == "small":
            save.update(desc=d, desc_w=dw, mask=npy(r[1]).astype(np.uint8), g_loss_a=npy(r[4][0]), g_loss_b=npy(r[4][1]),
                        g_mt_a=npy(r[5][0]), g_mt_b=npy(r[5][1]))
        else:  # inputs are regenerated from the seed (tests/golden_util.py:g10_inputs); gradients as norms + slices
            for nm, t_ in (("g_loss_a", r[4][0]), ("g_loss_b", r[4][1]), ("g_mt_a", r[5][0]), ("g_mt_b", r[5][1])):
                save[nm + "_norm"] = np.float32(t_.norm().item())
                save[nm + "_slice"] = npy(t_.reshape(-1)[::977][:256])
        np.savez_compressed(os.path.join(OUT, "g10_dense_loss_%s.npz" % name), **save)
        print("  G10", name, "loss %.5f pos %.5f neg %.3e positives %d" % (float(r[0]), float(r[2]), float(r[3]), int(r[1].sum())))


def g11_pair_labels():
    """warpLabels(bilinear=True) products (labels, res, labels_bi) on point sets without pixel collisions."""
    R.install()
    from datasets.data_tools import warpLabels
    rs = np.random.RandomState(19)
    H, W = 48, 64
    save = {}
    k = 0
    while k < 3:
        Hm = torch.from_numpy(np.linalg.inv(C.sample_homography(rs)).astype(np.float32))
        pts = torch.nonzero(torch.from_numpy(rs.uniform(size=(H, W)) < 0.01)).flip(1)
        ref = warpLabels(pts, H, W, Hm, bilinear=True)
        # collision-free: every point and every splat target lands on its own pixel (scatter order is undefined else)
        wp = ref["warped_pnts"]
        q = wp.round().long()
        flat = (q[:, 1] * W + q[:, 0]).numpy()
        lab_o, res_o, bi_o = C.warp_labels_full(pts, H, W, Hm)
        # the 4 splat targets of all points must be distinct pixels too
        from utils.utils import warp_points as _wp, homography_scaling_torch as _hs, filter_points as _fp
        wall = _wp(torch.stack((pts[:, 0], pts[:, 1]), dim=1).long(), _hs(Hm, H, W))
        pi = wall.long().float()
        ext = torch.cat((pi, torch.stack((pi[:, 0], pi[:, 1] + 1), 1), torch.stack((pi[:, 0] + 1, pi[:, 1]), 1), pi + 1), 0)
        ext = _fp(ext, torch.tensor([W, H]))
        eflat = (ext[:, 1].long() * W + ext[:, 0].long()).numpy()
        if len(np.unique(flat)) != len(flat) or len(np.unique(eflat)) != len(eflat):
            continue
        assert torch.equal(ref["labels_bi"], bi_o), "G11 labels_bi"
        assert torch.equal(ref["labels"], lab_o), "G11 labels"
        assert torch.equal(ref["res"].permute(2, 0, 1), res_o), "G11 res"
        save["H%d" % k] = npy(Hm); save["pts%d" % k] = npy(pts).astype(np.int16)
        save["labels%d" % k] = npy(ref["labels"]); save["res%d" % k] = npy(ref["res"].permute(2, 0, 1))
        save["bi%d" % k] = npy(ref["labels_bi"])
        k += 1
    np.savez_compressed(os.path.join(OUT, "g11_pair_labels.npz"), **save)


def g8_export():
    """Homography-adaptation export (export.py:274-318) through the real SuperPointFrontend_torch /
    combine_heatmap / getPtsFromHeatmap / soft_argmax_points on random-init weights."""
    R.install()
    import export as E
    from models.model_wrap import SuperPointFrontend_torch
    from utils.utils import flattenDetection
    cases = (("sp_64x96_v6", "SuperPointNet_gauss2", 64, 96, 6, 0.0152, 0, 40, 3),
             ("ssp_48x64_v5", "SuperPointNet_gauss2_ssmall", 48, 64, 5, 0.015, 2, 0, 4),
             ("sp_120x160_v4", "SuperPointNet_gauss2", 120, 160, 4, 0.0155, 0, 600, 5))
    for name, arch, H, W, nv, thr, erosion, top_k, seed in cases:
        rs = np.random.RandomState(100 + seed)
        img = torch.from_numpy(rs.uniform(0, 1, (H, W)).astype(np.float32))
        sample = C.homo_adapt_sample(img, nv, rs, erosion_radius=erosion)
        sd0 = C.init_state_dict(arch, seed=seed)
        cfg = {"model": {"subpixel": {"enable": True}}}
        fe = SuperPointFrontend_torch(config=cfg, weights_path="", nms_dist=4, conf_thresh=thr, nn_thresh=0.7,
                                      cuda=False, device="cpu", load=False)
        fe.net = ref_net(arch, sd0)  # stays in train mode, as the reference leaves it (model_wrap.py:120)
        with torch.no_grad():
            heat = fe.run(sample["image"], onlyHeatmap=True, train=False)
            # export.py:281-284 binds sample["homographies"] to the name `inv_homographies`
            agg = E.combine_heatmap(heat, sample["homographies"].unsqueeze(0), sample["valid_mask"], device="cpu")
        pts_nms = fe.getPtsFromHeatmap(agg.detach().cpu().squeeze())
        fe.heatmap = agg
        pts = fe.soft_argmax_points([pts_nms])[0].transpose()
        if top_k and pts.shape[0] > top_k:
            pts = pts[:top_k]
        conf = pts_nms[2]
        assert len(np.unique(conf)) == len(conf), "G8: tie among kept points, pick another seed"
        hm = npy(agg).squeeze()
        cand = hm[hm >= thr]
        # equal-confidence candidates only matter to the greedy order when they can suppress each other
        cm = np.where(hm >= thr, hm, -1.0)
        for dy in range(-4, 5):
            for dx in range(-4, 5):
                if (dy, dx) > (0, 0):
                    a = cm[max(dy, 0):H + min(dy, 0), max(dx, 0):W + min(dx, 0)]
                    b = cm[max(-dy, 0):H + min(-dy, 0), max(-dx, 0):W + min(-dx, 0)]
                    assert not np.any((a == b) & (a > 0)), "G8: tie inside an NMS window, pick another seed"
        # oracle == reference
        sd = C.to_torch(C.init_state_dict(arch, seed=seed))
        o = C.export_points(sd, sample, arch, conf_thresh=thr, nms_dist=4, top_k=top_k, subpixel=True)
        close(o["views_heatmap"], heat, 1e-6, "G8 views heatmap " + name)
        close(o["heatmap"], agg, 1e-6, "G8 aggregate " + name)
        o_nms = C.get_pts_from_heatmap(hm, thr, 4)  # on the reference's own aggregate: bit-exact
        assert np.array_equal(o_nms, pts_nms), "G8 nms " + name
        o_sub = C.soft_argmax_points(hm, o_nms).transpose()
        assert np.array_equal(o_sub[:len(pts)], pts), "G8 subpixel/top-k " + name
        rsd = fe.net.state_dict()
        for k in ("bnPb.running_mean", "bnPb.running_var", "inc.conv.conv.1.running_var"):
            close(sd[k], rsd[k], 1e-5, "G8 running stats " + k)
        print("  %s: %d candidates, %d kept, %d exported" % (name, len(cand), pts_nms.shape[1], len(pts)))
        np.savez_compressed(os.path.join(OUT, "g8_export_%s.npz" % name), arch=arch, seed=seed, thr=thr,
                            erosion=erosion, top_k=top_k, img=npy(img), homographies=npy(sample["homographies"]),
                            inv_homographies=npy(sample["inv_homographies"]), views=npy(sample["image"]),
                            valid_mask=npy(sample["valid_mask"]), views_heatmap=npy(heat), aggregate=hm,
                            pts_nms=pts_nms, pts=pts, bnPb_running_var=npy(rsd["bnPb.running_var"]))


def g9_logging():
    """heatmap_to_nms + batch_precision_recall of the logging branch (Train_model_heatmap_all.py:574-622,693-707)."""
    R.install()
    import Train_model_heatmap_all as T
    from utils.utils import flattenDetection
    rs = np.random.RandomState(77)
    B, H, W = 3, 64, 96
    semi = torch.from_numpy((2.0 * rs.randn(B, 65, H // 8, W // 8)).astype(np.float32))
    heat = flattenDetection(semi)
    close(C.flatten_detection(semi), heat, 1e-7, "G9 heatmap")
    labels = torch.from_numpy((rs.uniform(size=(B, 1, H, W)) < 0.02).astype(np.float32))
    # make the labels overlap the detections so that precision / recall are not trivially ~0
    nms_ref = np.stack([T.Train_model_heatmap_all.heatmap_nms(h) for h in npy(heat)])
    labels[:, 0][torch.from_numpy(nms_ref).bool() & torch.from_numpy(rs.uniform(size=(B, H, W)) < 0.5)] = 1
    pr_ref = T.Train_model_heatmap_all.batch_precision_recall(torch.from_numpy(nms_ref[:, None]).float(), labels)
    nms_o = np.stack([C.heatmap_nms(h) for h in npy(heat)])
    assert np.array_equal(nms_o, nms_ref), "G9 nms map"
    pr_o = C.batch_precision_recall(torch.from_numpy(nms_o[:, None]), labels)
    close(pr_o["precision"], float(pr_ref["precision"]), 1e-7, "G9 precision")
    close(pr_o["recall"], float(pr_ref["recall"]), 1e-7, "G9 recall")
    np.savez_compressed(os.path.join(OUT, "g9_logging.npz"), semi=npy(semi), heat=npy(heat), labels=npy(labels),
                        nms=nms_ref.astype(np.uint8), precision=float(pr_ref["precision"]),
                        recall=float(pr_ref["recall"]))


def g12_full_size_step():
    """G12: the pair step at the BENCHMARK resolution (240x320, B = 2) on the real reference: scalars of two optimizer
    steps, per-tensor gradient norms + 64-element slices of the first step, post-step eta, and - SURVEY.md section 8c G1 -
    per-channel checksums + strided slices of the first train-mode forward.  Inputs are stored compactly
    (cpu_ref.compact_to_npz: 8-bit images, bit-packed labels / masks, block-constant semantic maps)."""
    for tag, arch in (("sp", "SuperPointNet_gauss2"), ("ssp", "SuperPointNet_gauss2_ssmall")):
        H, W = 240, 320
        semantic = arch.endswith("ssmall")
        cfg = R.base_config(semantic=semantic, H=H, W=W, batch=2, lr=0.001, lambda_loss=1, multi_task=True)
        sd = C.init_state_dict(arch, seed=29)
        sample = C.make_compact_pair(2, H, W, seed=41, semantic=semantic)
        save = C.compact_to_npz(sample)
        # forward checksums (reference network, train mode, fresh running statistics)
        net = ref_net(arch, sd)
        with torch.no_grad():
            o = net(sample["image"])
        po = C.forward(C.to_torch(sd), sample["image"], arch)
        for k in o:
            close(o[k], po[k], 2e-6, "G12 %s forward %s" % (tag, k))
        save["fwd/semi_chsum"] = npy(o["semi"].double().sum(dim=(2, 3)))
        save["fwd/desc_chsum"] = npy(o["desc"].double().sum(dim=(2, 3)))
        save["fwd/semi_s"] = npy(o["semi"][:, ::4, ::3, ::4])
        save["fwd/desc_s"] = npy(o["desc"][:, ::16, ::3, ::4])
        if semantic:
            save["fwd/sem_chsum"] = npy(o["sem"].double().sum(dim=(2, 3)))
            save["fwd/sem_s"] = npy(o["sem"][:, ::19, ::24, ::32])
        # two optimizer steps on the reference, mirrored by the oracle
        agent = R.make_trainer(cfg, sd)
        tr = C.Trainer(arch, sd, lr=0.001, lambda_loss=1.0, multi_task=True)
        for it in range(2):  # the SAME RNG seeds in both steps: one stored index set serves both
            np.random.seed(100); torch.manual_seed(200)
            agent.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=it + 1, train=True)
            sc_ref = {k: float(v) for k, v in agent.scalar_dict.items()}
            np.random.seed(100); torch.manual_seed(200)
            tr.train_val_sample(sample, n_iter=it + 1, train=True)
            for k in sc_ref:
                close(sc_ref[k], tr.scalar_dict[k], 2e-5 * max(1.0, abs(sc_ref[k])), "G12 %s scalar %s it%d" % (tag, k, it))
                save["step%d/%s" % (it, k)] = np.float32(sc_ref[k])
            for i, idx in enumerate(tr.aux["indices"]):
                for nm in ("uv_a", "uv_b", "nm_b"):
                    arr = npy(idx[nm]).astype(np.int16)
                    if it == 0:
                        save["idx/%s%d" % (nm, i)] = arr
                    else:
                        assert np.array_equal(save["idx/%s%d" % (nm, i)], arr)
        save["post/eta"] = npy(agent.multi_task_loss.eta)
        close(agent.multi_task_loss.eta, tr.eta, 1e-5, "G12 eta")
        # gradients of the first step (no optimizer step)
        agent2 = R.make_trainer(cfg, sd)
        agent2.real_batch_size = 10 ** 9
        np.random.seed(100); torch.manual_seed(200)
        agent2.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=1, train=True)
        tr2 = C.Trainer(arch, sd, lr=0.001, lambda_loss=1.0, multi_task=True)
        tr2.real_batch_size = 10 ** 9
        np.random.seed(100); torch.manual_seed(200)
        tr2.train_val_sample(sample, n_iter=1, train=True)
        noisy = {conv + ".bias" for conv, bn, _, _, _ in C.layer_table(arch) if bn is not None}
        for k, p in agent2.net.named_parameters():
            g, go = p.grad, tr2.last_grads[k]
            scale = max(1e-6, float(g.abs().max()))
            if k not in noisy:
                close(g, go, 5e-4 * scale + 1e-6, "G12 %s grad %s" % (tag, k))
            save["grad_norm/" + k] = np.float32(g.norm().item())
            save["grad_slice/" + k] = npy(g.reshape(-1)[:64])
        save["grad/eta"] = npy(agent2.multi_task_loss.eta.grad)
        np.savez_compressed(os.path.join(OUT, "g12_step_%s_240x320.npz" % tag), **save)
        print("G12", tag, "ok;", os.path.getsize(os.path.join(OUT, "g12_step_%s_240x320.npz" % tag)), "bytes")


def magicpoint_config(multi_task, batch=2):
    """configs/magicpoint_shapes_pair.yaml as SHIPPED (read from the reference at generation time) plus the keys it lacks and
    the trainer reads (SURVEY.md section 5 'Shipped-config defects'): model.real_batch_size (:108), data.semantic (:221),
    model.multi_task_loss (:355), an importable front_end_model; batch 64 -> BASELINE configs[0]'s 2."""
    import yaml
    with open(os.path.join(R.REF_ROOT, "configs", "magicpoint_shapes_pair.yaml")) as f:
        cfg = yaml.safe_load(f)
    assert cfg["data"]["warped_pair"]["enable"] is False and cfg["model"]["lambda_loss"] == 0
    assert cfg["data"]["preprocessing"]["resize"] == [120, 160] and cfg["data"]["gaussian_label"]["enable"] is False
    cfg["front_end_model"] = "Train_model_heatmap_all"
    cfg["model"].update({"batch_size": batch, "real_batch_size": batch, "eval_batch_size": batch, "multi_task_loss": bool(multi_task)})
    cfg["data"]["semantic"] = False
    cfg["pretrained"] = None
    return cfg


def g13_single_view_step():
    """G13: BASELINE configs[0] - the SINGLE-VIEW step (`data.warped_pair.enable: false`, Train_model_heatmap_all.py:207,
    237-262, 330-332, 346-353) of the shipped configs/magicpoint_shapes_pair.yaml at 120x160, B = 2, on the real reference:
    scalars of two optimizer steps, gradient norms + slices, post-Adam eta / parameter slices, one validation call
    (train=False: no_grad forward in train-mode BatchNorm + the logging branch's precision / recall).  Both values of the
    key the yaml lacks (model.multi_task_loss).  Also records that `detector_loss.loss_type: l2` RAISES in the reference for
    these models (65 logits against the 64-channel target of add_dustbin=False, :170-172,302-304)."""
    arch, H, W = "SuperPointNet_gauss2", 120, 160
    for tag, mt in (("uniform", False), ("kendall", True)):
        cfg = magicpoint_config(mt)
        sd = C.init_state_dict(arch, seed=37)
        full = C.make_synthetic_pair(2, H, W, seed=43, kp_prob=0.004)
        sample = {k: full[k] for k in ("image", "labels_2D", "valid_mask")}
        sample["valid_mask"] = sample["valid_mask"].clone()
        sample["valid_mask"][0, :, 16:40, 24:56] = 0  # some invalid cells (augmentation border), exercises mask.sum()
        kw = dict(lambda_loss=0.0, multi_task=mt, gaussian=False, warped_pair=False)
        agent = R.make_trainer(cfg, sd)
        tr = C.Trainer(arch, sd, lr=0.001, **kw)
        save = {"in/image_u8": np.round(npy(sample["image"]) * 255).astype(np.uint8),
                "in/labels_2D": np.packbits(npy(sample["labels_2D"]).astype(np.uint8)),
                "in/valid_mask": np.packbits(npy(sample["valid_mask"]).astype(np.uint8))}
        sample["image"] = torch.from_numpy(save["in/image_u8"].astype(np.float32) / 255.0)  # what the test decodes
        for it in range(2):
            agent.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=it + 1, train=True)
            sc_ref = {k: float(v) for k, v in agent.scalar_dict.items()}
            tr.train_val_sample(sample, n_iter=it + 1, train=True)
            for k in sc_ref:
                close(sc_ref[k], tr.scalar_dict[k], 2e-5 * max(1.0, abs(sc_ref[k])), "G13 %s scalar %s it%d" % (tag, k, it))
                save["step%d/%s" % (it, k)] = np.float32(sc_ref[k])
        noisy = {conv + ".bias" for conv, bn, _, _, _ in C.layer_table(arch) if bn is not None}
        named = dict(agent.net.named_parameters())
        for k, p in named.items():
            if k not in noisy and not k.startswith(("convD", "bnD")):  # the descriptor head is not in the graph: untouched
                close_adam(p, tr.sd[k], 0.001, "G13 %s post-step %s" % (tag, k))
        for k in ("convDa.weight", "bnDb.weight"):
            assert torch.equal(named[k].detach(), torch.as_tensor(np.array(sd[k]))), "descriptor head moved"
        close(agent.multi_task_loss.eta, tr.eta, 1e-5, "G13 eta")
        save["post/eta"] = npy(agent.multi_task_loss.eta)
        for k in ("inc.conv.conv.3.weight", "down3.mpconv.1.conv.4.weight", "convPb.weight", "bnPb.bias"):
            save["post_slice/" + k] = npy(named[k].reshape(-1)[:64])
        rsd = agent.net.state_dict()
        for k in rsd:
            if "running_var" in k:
                save["post_state/" + k] = npy(rsd[k])
        # validation call (Train_model_frontend_all.py:340-349): forward under no_grad, BatchNorm still in train mode
        agent.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=7, train=False)
        tr.train_val_sample(sample, n_iter=7, train=False)
        for k in ("loss", "loss_det", "loss_det_warp"):
            close(float(agent.scalar_dict[k]), tr.scalar_dict[k], 2e-5 * max(1.0, abs(float(agent.scalar_dict[k]))), "G13 val " + k)
        for k, v in agent.scalar_dict.items():
            save["val/" + k] = np.float32(float(v))
        # gradients of the first step, no optimizer step
        agent2 = R.make_trainer(cfg, sd)
        agent2.real_batch_size = 10 ** 9
        agent2.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=1, train=True)
        tr2 = C.Trainer(arch, sd, lr=0.001, **kw)
        tr2.real_batch_size = 10 ** 9
        tr2.train_val_sample(sample, n_iter=1, train=True)
        for k, p in agent2.net.named_parameters():
            g, go = p.grad, tr2.last_grads[k]
            if g is None:
                assert go is None and k.startswith(("convD", "bnD")), k
                continue
            scale = max(1e-6, float(g.abs().max()))
            if k not in noisy:
                close(g, go, 5e-4 * scale + 1e-6, "G13 %s grad %s" % (tag, k))
            save["grad_norm/" + k] = np.float32(g.norm().item())
            save["grad_slice/" + k] = npy(g.reshape(-1)[:64])
        eg = agent2.multi_task_loss.eta.grad
        if mt:
            close(eg, tr2.last_grads["eta"], 1e-5, "G13 eta grad")
            save["grad/eta"] = npy(eg)
        else:
            assert eg is None and tr2.last_grads["eta"] is None
        # loss_type l2: the reference itself raises (MSELoss of [B,65,Hc,Wc] logits against the [B,64,Hc,Wc] target)
        cfg_l2 = magicpoint_config(mt)
        cfg_l2["model"]["detector_loss"]["loss_type"] = "l2"
        a3 = R.make_trainer(cfg_l2, sd)
        try:
            a3.train_val_sample({k: v.clone() for k, v in sample.items()}, n_iter=1, train=True)
            raised = ""
        except RuntimeError as e:
            raised = str(e)
        assert "must match the size" in raised, raised
        save["l2_raises"] = np.array(raised)
        path = os.path.join(OUT, "g13_single_view_%s_120x160.npz" % tag)
        np.savez_compressed(path, **save)
        print("G13", tag, "ok;", os.path.getsize(path), "bytes; loss", float(save["step0/loss"]), "->", float(save["step1/loss"]))


def main():
    assert R.available(), "reference not mounted"
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    only = os.environ.get("SSP_GOLDEN_ONLY")  # e.g. SSP_GOLDEN_ONLY=g12 regenerates one family
    for fn in (g1_forward, g2_labels, g3_detector_loss, g5_sem_loss, g4_sparse_loss, g7_warps, g6_train_step, g8_export, g9_logging, g10_dense_loss, g11_pair_labels, g12_full_size_step, g13_single_view_step, g14_sparse_loss_variants):
        if only and not fn.__name__.startswith(only):
            continue
        fn()
        print(fn.__name__, "done")
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()
