"""Drop-in for the reference's trainer plugin `Train_model_heatmap_all` (Train_model_heatmap_all.py:80-572 on top
of Train_model_frontend_all.py:100-439): same constructor, same `loadModel / dataParallel / train /
train_val_sample / saveModel`, same `scalar_dict` and writer calls - but the whole pair step (2 forwards, label
ops, losses, backward) is ONE call into libssp_hip.so (`ssp_pair_step`) followed by the fused Adam kernel.

Deliberate behaviours kept from the reference:
  * the live optimizer's learning rate is CONSTANT (the scheduler built in loadModel drives an orphaned Adam);
  * gradients accumulate un-scaled over micro-batches until ((n_iter+1)*batch) % real_batch_size == 0;
  * scalar_dict["eta_*"] are read AFTER the optimizer step (the reference logs the live parameter);
  * validation (train=False) runs BatchNorm in train mode under no_grad, i.e. it updates running statistics.
Logging branch (every `tensorboard_interval` steps and in validation, Train_model_heatmap_all.py:447-568): the
precision / recall scalars, the NMS maps of both views and the image overlays of `images_dict` are produced on the
device (`log_precision_recall`).
Data parallel (one process per GPU, torch.distributed initialised by the launcher): the gradient all-reduce of the
optimizer step is split in two buckets and overlapped with the tail of the backward pass (parallel.pair_step_overlapped).
"""
import copy
import logging
import os
from pathlib import Path

import numpy as np
import torch

from . import lib as L
from . import parallel
from .models import SuperPointNet_gauss2, SuperPointNet_gauss2_ssmall

_MODELS = {"SuperPointNet_gauss2": SuperPointNet_gauss2, "SuperPointNet_gauss2_ssmall": SuperPointNet_gauss2_ssmall}


def dict_update(d, u):
    """Nested dict merge (utils/tools.py:7-23 semantics)."""
    for k, v in u.items():
        if isinstance(v, dict):
            d[k] = dict_update(d.get(k, {}) or {}, v)
        else:
            d[k] = v
    return d


def sample_sparse_indices_host(homographies, Hc, Wc, n_match, n_non):
    """Reference-faithful HOST sampling of the sparse-loss indices (sparse_loss.py:184-246,
    correspondence_finder.py:29-34,278-280): consumes numpy's and torch's global CPU RNG streams in the
    reference's order, so a run seeded like the reference draws the same indices.  Returns int32 tensors."""
    ma, mb, nm = [], [], []
    for Hn in homographies.detach().cpu().float():
        vs, us = torch.meshgrid(torch.arange(Hc), torch.arange(Wc), indexing="ij")
        uv_a = torch.stack((us.reshape(-1), vs.reshape(-1)), dim=1).float()
        T = torch.tensor([[2.0 / Wc, 0.0, -1.0], [0.0, 2.0 / Hc, -1.0], [0.0, 0.0, 1.0]])
        Hcell = torch.inverse(T) @ Hn @ T
        w = (Hcell @ torch.cat((uv_a, torch.ones(uv_a.shape[0], 1)), dim=1).t()).t()
        uv_b = (w[:, :2] / w[:, 2:]).round()
        keep = (uv_b[:, 0] >= 0) & (uv_b[:, 0] <= Wc - 1) & (uv_b[:, 1] >= 0) & (uv_b[:, 1] <= Hc - 1)
        uv_a, uv_b = uv_a[keep], uv_b[keep]
        n = uv_b.shape[0]
        choice = np.random.permutation(n)
        if n >= n_match:
            choice = choice[:n_match]
        else:
            choice = np.concatenate([choice, np.random.choice(choice, n_match - n, replace=True)])
        choice = torch.as_tensor(choice).long()
        uv_a, uv_b = uv_a[choice], uv_b[choice]
        K = n_match * n_non
        two = torch.rand(2, K)
        nu, nv = torch.floor(two[0] * Wc).long(), torch.floor(two[1] * Hc).long()
        torch.rand(K)   # the no-op "perturbation" of the reference still burns these two draws
        torch.randn(K)
        ma.append((uv_a[:, 0] + uv_a[:, 1] * Wc).int())
        mb.append((uv_b[:, 0] + uv_b[:, 1] * Wc).int())
        nm.append((nu + nv * Wc).int())
    return torch.stack(ma), torch.stack(mb), torch.stack(nm)


class Train_model_heatmap_all(object):
    default_config = {
        "train_iter": 170000, "save_interval": 2000, "tensorboard_interval": 200,
        "model": {"subpixel": {"enable": False}}, "data": {"gaussian_label": {"enable": False}},
    }

    def __init__(self, config, save_path=Path("."), device="cpu", verbose=False):
        self.config = dict_update(copy.deepcopy(self.default_config), copy.deepcopy(config))
        m = self.config["model"]
        self.r = m["real_batch_size"] // m["batch_size"]
        for k in ("train_iter", "validation_interval", "tensorboard_interval", "save_interval"):
            self.config[k] *= self.r
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Train_model_heatmap_all (MI355X build) needs a HIP device, got %s" % device)
        self.save_path = Path(save_path)
        self._train, self._eval = True, True
        self.cell_size = 8
        self.real_batch_size = m["real_batch_size"]
        self.max_iter = self.config["train_iter"]
        self.gaussian = bool(self.config["data"]["gaussian_label"]["enable"])
        if m.get("dense_loss", {}).get("enable", False):  # Train_model_heatmap_all.py:131-137: dense wins over sparse
            self.desc_params = m["dense_loss"].get("params") or {}
            self.desc_loss_type = "dense"
        elif m.get("sparse_loss", {}).get("enable", False):
            self.desc_params = m["sparse_loss"]["params"]   # method / dist: descriptor_loss_sparse's own defaults (sparse_loss.py:76-77)
            self.desc_loss_type = "sparse"
        else:
            raise KeyError("model.dense_loss.enable or model.sparse_loss.enable must be true")
        self.sampler = self.config.get("ssp_sampler", "device")  # "device" | "reference" (host RNG streams)
        self.n_iter = 0
        self.net = None
        self._writer = None
        self.scalar_dict, self.images_dict, self.hist_dict = {}, {}, {}

    # ---- properties of the reference base class ----
    @property
    def writer(self):
        return self._writer

    @writer.setter
    def writer(self, writer):
        self._writer = writer

    @property
    def train_loader(self):
        return self._train_loader

    @train_loader.setter
    def train_loader(self, loader):
        self._train_loader = loader

    @property
    def val_loader(self):
        return self._val_loader

    @val_loader.setter
    def val_loader(self, loader):
        self._val_loader = loader

    # ---- model / optimizer ----
    def loadModel(self):
        name = self.config["model"]["name"]
        params = self.config["model"].get("params") or {}
        if name not in _MODELS:
            raise KeyError("model %r is not on the accelerated path" % name)
        self.net = _MODELS[name](**params).to(self.device)
        if self.desc_loss_type == "dense":  # the engine reserves the [B, cells, cells] coefficient matrix
            self.net._engine_kwargs = {"dense_loss": True}
        else:
            # descriptor_loss_sparse(**desc_params) (sparse_loss.py:65-72): defaults 1000 attempts, 10 masked
            # non-matches per match, lamda_d 250; the shipped configs set 600-1000 x 100 x 1
            n_match = int(self.desc_params.get("num_matching_attempts", 1000))
            n_non = int(self.desc_params.get("num_masked_non_matches_per_match", 10))
            if not (1 <= n_match <= L.SAMPLER_MAX_MATCHES) or not (1 <= n_non <= 4096):
                raise ValueError("sparse_loss.params: num_matching_attempts must be in 1..%d and "
                                 "num_masked_non_matches_per_match in 1..4096 (got %d, %d)"
                                 % (L.SAMPLER_MAX_MATCHES, n_match, n_non))
            self.net._engine_kwargs = {"n_match": n_match, "n_non": n_non}
        # the engine is sized once for the larger of the training / validation batch, so that a validation batch
        # never re-creates it (re-creation carries the optimizer state over, but costs a workspace allocation)
        self.net._engine_min_batch = max(int(self.config["model"].get("batch_size", 1)),
                                         int(self.config["model"].get("eval_batch_size", 1)))
        n_iter = 0
        self._resume = None
        if not self.config.get("retrain", True) and self.config.get("pretrained"):
            path = self.config["pretrained"]
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
            if path[-4:] == ".pth":
                self.net.load_state_dict(ckpt)
            else:
                self.net.load_state_dict(ckpt["model_state_dict"])
                n_iter = ckpt.get("n_iter", 0)
                # The reference loads the optimizer state here (utils/loader.py:190-194) and then THROWS IT AWAY:
                # dataParallel() builds a fresh Adam (Train_model_frontend_all.py:171-181) and MultiTaskLoss.eta is not
                # in the checkpoint at all.  `ssp_restore_optimizer: true` resumes Adam (m, v, step) and eta instead.
                if self.config.get("ssp_restore_optimizer", False):
                    self._resume = {"optimizer_state_dict": ckpt.get("optimizer_state_dict"), "eta": ckpt.get("eta")}
        self.n_iter = 0 if self.config.get("reset_iter", True) else n_iter
        self.learning_rate = self.config["model"]["learning_rate"]
        return self.net

    def dataParallel(self):
        """Reference: re-creates Adam and zeroes the gradients (Train_model_frontend_all.py:171-181).  Here: the
        fused Adam state lives in the engine; replicas (if torch.distributed is initialised) are made identical."""
        self._adam_reset = True

    def _engine_for(self, B, H, W):
        e = self.net.engine(max(B, getattr(self.net, "_engine_min_batch", 1)), H, W, self.device)
        if getattr(self, "_adam_reset", True):
            e.adam_m.zero_(); e.adam_v.zero_(); e.adam_t = 0
            e.zero_grad()
            if getattr(self, "_resume", None):
                L.load_optimizer_state(e, self._resume.get("optimizer_state_dict"), self._resume.get("eta"))
                self._resume = None
            parallel.broadcast_(e.params)
            parallel.broadcast_(e.bn_running)
            self._adam_reset = False
        return e

    # ---- the step ----
    def train_val_sample(self, sample, n_iter=0, train=False):
        task = "train" if train else "val"
        cfg, m = self.config, self.config["model"]
        if_warp = bool(cfg["data"]["warped_pair"]["enable"])  # :207; false = the single-view step (magicpoint_shapes_pair.yaml)
        det_loss_type = m["detector_loss"]["loss_type"]
        if det_loss_type == "l2":
            # The reference builds a 64-channel target (add_dustbin=False, :296-304) and hands it to MSELoss with the models'
            # 65 logits (:170-172): torch raises this RuntimeError on every model of the path (pinned: G13 `l2_raises`).
            raise RuntimeError("The size of tensor a (65) must match the size of tensor b (64) at non-singleton dimension 1 "
                               "(model.detector_loss.loss_type 'l2' is not usable with the 65-logit detector head)")
        if det_loss_type != "softmax":  # :168-178: `loss` is never assigned
            raise UnboundLocalError("local variable 'loss' referenced before assignment (detector_loss.loss_type %r)" % (det_loss_type,))
        img = sample["image"]
        B, _, H, W = img.shape
        self.batch_size = B
        eng = self._engine_for(B, H, W)
        if not if_warp:  # the warped keys of a pair loader are never read (:226-251)
            sample = {k: v for k, v in sample.items() if not k.startswith(("warped_", "homographies", "inv_homographies", "cell_homographies"))}
        dev = {k: (v.to(self.device, non_blocking=True).contiguous() if torch.is_tensor(v) else v) for k, v in sample.items()}
        if if_warp and "cell_homographies" not in dev and not sample["homographies"].is_cuda:
            # the loader's homographies are host tensors: scale them to cell coordinates with the reference's own op
            # sequence here, so that the device sampler's matches round exactly like descriptor_loss_sparse's
            dev["cell_homographies"] = L.scaled_homographies(sample["homographies"], H // 8, W // 8).to(self.device)
        for k in ("image", "warped_img", "labels_2D", "warped_labels", "valid_mask", "warped_valid_mask",
                  "labels_2D_gaussian", "warped_labels_gaussian"):
            if k in dev:
                dev[k] = dev[k].float()
        lam = float(m["lambda_loss"])
        assert if_warp or not lam > 0, "need a pair of images"  # :343
        idx = None
        dense = None
        if self.desc_loss_type == "dense":
            # descriptor_loss(**desc_params): only `descriptor_dist` and `lamda_d` are named parameters, so the shipped
            # `lambda_d: 800` falls into **config and the weight stays 250 (utils/utils.py:779-790)
            dense = {"lamda_d": float(self.desc_params.get("lamda_d", 250)),
                     "descriptor_dist": float(self.desc_params.get("descriptor_dist", 4))}
        if lam > 0 and dense is None and self.sampler == "reference":
            idx = tuple(t.to(self.device) for t in sample_sparse_indices_host(
                sample["homographies"], H // 8, W // 8, eng.n_match, eng.n_non))
        # rank-offset sampler seed; lamda_d defaults to descriptor_loss_sparse's 250
        seed = (int(cfg.get("ssp_seed", 0)) * 1000003 + n_iter) * 64 + parallel.rank()
        kw = dict(indices=idx, seed=seed, train=train, lambda_loss=lam, lamda_d=float(self.desc_params.get("lamda_d", 250)),
                  multi_task=bool(m["multi_task_loss"]), gaussian=self.gaussian, dense=dense)
        if self.desc_loss_type == "sparse":  # descriptor_loss_sparse's own defaults: dist="cos", method="1d" (sparse_loss.py:76-77)
            kw.update(sparse_method=str(self.desc_params.get("method", "1d")), sparse_dist=str(self.desc_params.get("dist", "cos")))
        opt_step = train and ((n_iter + 1) * B) % self.real_batch_size == 0
        if opt_step:  # all-reduce (world > 1) overlapped with the tail of the backward pass, then fused Adam
            sc = parallel.pair_step_overlapped(eng, dev, self.learning_rate, **kw)
            eng.zero_grad()
        else:
            sc = eng.pair_step(dev, **kw)
        vals = sc.cpu().tolist()  # the single host sync of the step (the reference syncs on every .item())
        s = dict(zip(L.SCALAR_NAMES, vals))
        eta = eng.eta.cpu().tolist()
        self.loss = s["loss"]
        self.scalar_dict = {"loss": s["loss"], "loss_det": s["loss_det"], "loss_det_warp": s["loss_det_warp"],
                            "loss_desc": s["loss_desc"], "loss_sem": s["loss_sem"], "loss_sem_warp": s["loss_sem_warp"],
                            "eta_det": eta[0], "eta_desc": eta[1], "positive_dist": s["positive_dist"],
                            "negative_dist": s["negative_dist"]}
        if cfg["data"].get("semantic", False):
            self.scalar_dict["eta_sem"] = eta[2]
        if n_iter % cfg["tensorboard_interval"] == 0 or task == "val":
            self.log_precision_recall(eng, dev, B, H, W)
            self.tb_hist_dict(task, self.hist_dict)   # (:568; `images_dict` stays unwritten like the reference's commented-out :567)
        self.tb_scalar_dict(self.scalar_dict, task)
        return float(s["loss"])

    def log_precision_recall(self, eng, dev, B, H, W):
        """Logging branch (Train_model_heatmap_all.py:447-568): flattenDetection of both views' logits, heatmap_nms
        with its hard-wired defaults (nms_dist=4, conf_thresh=0.015: :693 ignores the config), batch_precision_recall
        of the un-warped view against labels_2D (:555-559, :614-622) and the image overlays of :460-507 in
        `images_dict` (the reference fills the dict and leaves `tb_images_dict` commented out at :565; so does this)."""
        heat = eng.detector_heatmap(0, B, H, W)
        nms, pr = L.op_heatmap_nms(heat, dev["labels_2D"].float().contiguous(), conf_thresh=0.015, nms_dist=4)
        prm = pr.cpu().numpy().mean(axis=0)
        self.scalar_dict.update({"precision": float(prm[0]), "recall": float(prm[1])})
        self.images_dict = {"heatmap_org_nms_batch": nms.unsqueeze(1).cpu().numpy()}
        views = [("original", dev["labels_2D"], heat, nms, dev["image"])]
        if "warped_img" in dev:
            heat_w = eng.detector_heatmap(1, B, H, W)
            nms_w, _ = L.op_heatmap_nms(heat_w, dev["warped_labels"].float().contiguous(), conf_thresh=0.015, nms_dist=4)
            self.images_dict["heatmap_warp_nms_batch"] = nms_w.unsqueeze(1).cpu().numpy()
            views.append(("warped", dev["warped_labels"], heat_w, nms_w, dev["warped_img"]))
        for name, lab, hm, nm, img in views:
            # :473-479 passes heatmap_nms_batch[np.newaxis]: ONE overlay, labels / image of sample 0 over the NMS map of
            # sample 0; the heat-map overlays (:481-487) cover the whole batch
            self.images_dict[name + "_nms_overlap"] = self.img_overlap(lab[:1], nm[:1].unsqueeze(1), img[:1]).cpu().numpy()
            self.images_dict[name + "_heatmap_nms_overlap"] = self.img_overlap(lab, hm.view(B, 1, H, W), img).cpu().numpy()

    @staticmethod
    def img_overlap(img_r, img_g, img_gray):
        """utils/draw.py:50-56 for a batch on any device: [B,1,H,W] x 3 -> [B,3,H,W], gray image in all three channels
        plus img_r on red and img_g on green, clamped to [0, 1]."""
        out = img_gray.float().repeat(1, 3, 1, 1)
        out[:, 0:1] += img_r.float()
        out[:, 1:2] += img_g.float()
        return out.clamp_(0.0, 1.0)

    def tb_scalar_dict(self, losses, task="training"):
        if self._writer is None:
            return
        for element in list(losses):
            self._writer.add_scalar(task + "-" + element, losses[element], self.n_iter // self.r)

    def tb_images_dict(self, task, tb_imgs, max_img=5):
        """Train_model_frontend_all.py:535-566: the first `max_img` entries of every [N,C,H,W] array of the dict as images; with
        `config["semantic"]` the class maps `sem_pred` / `warp_sem_pred` are reduced to their argmax first.  (The reference's
        heat-map trainer leaves its own call commented out, :567; the method is here for callers that want the overlays of
        `images_dict` in their event file.)"""
        if self._writer is None:
            return
        if self.config.get("semantic", False) and "sem_pred" in tb_imgs:
            for key in ("sem_pred", "warp_sem_pred"):
                if key in tb_imgs:
                    a = np.asarray(tb_imgs[key])
                    out = np.zeros((a.shape[0], 1) + a.shape[2:])
                    out[:, 0] = np.argmax(a, axis=1)
                    tb_imgs[key] = out
        for element in list(tb_imgs):
            for idx in range(tb_imgs[element].shape[0]):
                if idx >= max_img:
                    break
                self._writer.add_image(task + "-" + element + "/%d" % idx, tb_imgs[element][idx, ...], self.n_iter // self.r)

    def tb_hist_dict(self, task, tb_dict):
        """Train_model_frontend_all.py:568-571."""
        if self._writer is None:
            return
        for element in list(tb_dict):
            self._writer.add_histogram(task + "-" + element, tb_dict[element], self.n_iter // self.r)

    def printLosses(self, losses, task="training"):
        """Train_model_frontend_all.py:573-582 (the scalars are Python floats here: no .item())."""
        for element in list(losses):
            print(task, "-", element, ": ", float(losses[element]))

    # ---- outer loop / checkpoints (Train_model_frontend_all.py:315-359, 422-439) ----
    def train(self, **options):
        logging.info("n_iter: %d  max_iter: %d", self.n_iter, self.max_iter)
        while self.n_iter < self.max_iter:
            for sample_train in self.train_loader:
                self.train_val_sample(sample_train, self.n_iter, True)
                self.n_iter += 1
                if self._eval and self.n_iter % self.config["validation_interval"] == 0:
                    for j, sample_val in enumerate(self.val_loader):
                        self.train_val_sample(sample_val, self.n_iter + j, False)
                        if j > self.config.get("validation_size", 3):
                            break
                if self.n_iter % self.config["save_interval"] == 0:
                    self.saveModel()
                if self.n_iter > self.max_iter:
                    break

    def saveModel(self):
        """Checkpoint in the reference's wire format {n_iter, model_state_dict, optimizer_state_dict, loss}
        -> <save_path>/superPointNet_<n_iter>_checkpoint.pth.tar (utils/utils.py:134-140)."""
        eng = self.net.engine()
        path = Path(self.save_path) / ("superPointNet_%d_checkpoint.pth.tar" % self.n_iter)  # file: n_iter; dict: n_iter + 1
        if parallel.rank() != 0:  # replicas are identical: one writer
            return path
        os.makedirs(self.save_path, exist_ok=True)
        # optimizer_state_dict has the layout of torch.optim.Adam(list(net.parameters()) + [eta]).state_dict()
        # (Train_model_frontend_all.py:183-198), so reference-side tooling can load it; "eta" is an extra key
        state = {"n_iter": self.n_iter + 1, "model_state_dict": {k: v.detach().cpu() for k, v in self.net.state_dict().items()},
                 "optimizer_state_dict": L.optimizer_state_dict(eng, self.learning_rate),
                 "loss": getattr(self, "loss", None), "eta": eng.eta.cpu()}
        torch.save(state, path)
        return path
