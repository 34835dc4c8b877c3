"""Drop-in for the homography-adaptation branch of the reference's export.py.

Same names as the reference so that its call sites read unchanged:
  combine_heatmap(heatmap, inv_homographies, mask_2D, device)        export.py:49-60
  SuperPointFrontend_torch(...).run / getPtsFromHeatmap / nms_fast / soft_argmax_points
                                                                     models/model_wrap.py:60-293
  export_detector_homoAdapt_gpu(config, output_dir, args)            export.py:192-352
plus `HomoAdaptExporter`, the fused MI355X path the loop uses: one ssp_export_points call per pair of images
(forward of the detector head over the views in train-mode BatchNorm, softmax -> depth-to-space, masked un-warp
accumulation, greedy NMS, soft-argmax refinement, top-k) with no host round trip before the final point list.

Every arithmetic step runs in libssp_hip.so; tensors must live on a HIP device (no CPU fallback).  The descriptor
export of the reference (`export_descriptor`, `run(onlyHeatmap=False)`) is outside this path and raises.
"""
import logging
import os
from pathlib import Path

import numpy as np
import torch

from . import lib as L


def _dev_f32(t, device):
    t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.float32).contiguous()


def combine_heatmap(heatmap, inv_homographies, mask_2D, device="cpu"):
    """export.py:49-60.  heatmap, mask_2D: [N,1,H,W]; inv_homographies: [1,N,3,3] (the reference indexes [0]).
    Returns the [1,H,W] aggregate sum(unwarp(heatmap*mask)) / sum(unwarp(mask)) on the tensors' HIP device."""
    dev = heatmap.device
    mask = _dev_f32(mask_2D, dev)
    heat = (heatmap.to(torch.float32) * mask).contiguous()
    hm = inv_homographies[0] if inv_homographies.dim() == 4 else inv_homographies
    return L.op_combine_heatmap(heat, mask, _dev_f32(hm, dev)).unsqueeze(0)


class SuperPointFrontend_torch(object):
    """The subset of models/model_wrap.py:SuperPointFrontend_torch that the homography-adaptation export uses."""

    def __init__(self, config, weights_path, nms_dist, conf_thresh, nn_thresh, cuda=False, trained=False, device="cpu",
                 grad=False, load=True):
        self.config = config
        self.name = "SuperPoint"
        self.nms_dist, self.conf_thresh, self.nn_thresh = nms_dist, conf_thresh, nn_thresh
        self.cell, self.border_remove = 8, 4  # models/model_wrap.py:69-70
        self.device = torch.device(device)
        self.subpixel = bool(config["model"]["subpixel"]["enable"])
        self.sparsemap = self.pts = self.pts_subpixel = self.patches = None
        self._heatmap = None
        self.net = None
        if load:
            self.loadModel(weights_path)

    @property
    def heatmap(self):
        return self._heatmap

    @heatmap.setter
    def heatmap(self, heatmap):
        self._heatmap = heatmap

    def loadModel(self, weights_path):
        """models/model_wrap.py:84-121: config['model']['name'](**params) + checkpoint['model_state_dict'];
        the network stays in TRAIN mode (the reference never calls .eval() here)."""
        import importlib
        name = self.config["model"]["name"]
        mod = importlib.import_module(__package__ + ".models." + name)
        self.net = getattr(mod, name)(**self.config["model"].get("params", {}))
        if weights_path:
            ckpt = torch.load(weights_path, map_location="cpu")
            self.net.load_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt)
        self.net = self.net.to(self.device)

    def net_parallel(self):
        """nn.DataParallel in the reference (models/model_wrap.py:123-125).  Here: one process per GPU, images
        sharded across ranks by export_detector_homoAdapt_gpu -- nothing to wrap."""
        return None

    def run(self, inp, onlyHeatmap=False, train=True):
        """models/model_wrap.py:338-372 up to `if onlyHeatmap: return heatmap`."""
        if not onlyHeatmap:
            raise NotImplementedError("descriptor export is outside the MI355X hot path (SURVEY.md section 8)")
        inp = inp.to(self.device)
        if train:
            semi = self.net(inp)["semi"]
        else:
            with torch.no_grad():
                semi = self.net(inp)["semi"]
        self.heatmap = L.op_flatten_detection(semi.contiguous())
        return self.heatmap

    def getPtsFromHeatmap(self, heatmap):
        """models/model_wrap.py:266-293: [H,W] heatmap (numpy or tensor, any device) -> float64 3xN (x, y, conf)."""
        hm = _dev_f32(heatmap, self.device).squeeze()
        self.sparsemap = (hm >= float(np.float32(self.conf_thresh))).cpu().numpy()
        return L.op_heatmap_points(hm, self.conf_thresh, self.nms_dist, self.border_remove).T

    def nms_fast(self, in_corners, H, W, dist_thresh):
        """models/model_wrap.py:129-192 for corners at distinct integer pixels: returns (3xN kept corners by
        descending confidence, their indices into in_corners)."""
        n = in_corners.shape[1]
        if n == 0:
            return np.zeros((3, 0)).astype(int), np.zeros(0).astype(int)
        rc = in_corners[:2].round().astype(np.int64)
        flat = rc[1] * W + rc[0]
        if len(np.unique(flat)) != n:
            raise ValueError("nms_fast: corners must round to distinct pixels")
        conf = in_corners[2].astype(np.float32)
        # order-preserving remap to (0, 1]: the device kernel ranks by the fp32 value itself
        grid = torch.full((H * W,), float("nan"))
        grid[torch.from_numpy(flat)] = torch.from_numpy(conf)
        out = L.op_heatmap_points(grid.view(H, W).to(self.device), -np.inf, dist_thresh, 0)
        inds = np.full(H * W, -1, np.int64)
        inds[flat] = np.arange(n)
        keep = inds[(out[:, 1] * W + out[:, 0]).astype(np.int64)]
        return in_corners[:, keep], keep

    def soft_argmax_points(self, pts, patch_size=5):
        """models/model_wrap.py:212-249: pts = [3xN]; returns [3xN] with (x, y) moved by the 5x5 soft-argmax of
        self.heatmap."""
        assert patch_size == 5, "the device kernel implements the 5x5 patch the reference uses"
        p = pts[0].transpose().copy()
        hm = _dev_f32(self.heatmap, self.device).squeeze()
        xy = torch.from_numpy(np.ascontiguousarray(p[:, :2]).astype(np.float32)).to(self.device)
        d = L.op_soft_argmax_points(hm, xy).cpu().numpy()
        p[:, :2] = p[:, :2] + d - patch_size // 2
        self.pts_subpixel = [p.transpose().copy()]
        return self.pts_subpixel.copy()


class HomoAdaptExporter:
    """Fused export of image PAIRS on one GPU.  `net` is one of this package's model drop-ins (its Engine is created
    for n_views x H x W on first use)."""

    def __init__(self, net, device, conf_thresh, nms_dist, top_k, subpixel, border_remove=4):
        self.net, self.device = net, torch.device(device)
        self.args = dict(conf_thresh=conf_thresh, nms_dist=nms_dist, top_k=top_k, subpixel=subpixel,
                         border_remove=border_remove)

    def __call__(self, samples, want_heatmap=False):
        """samples: 1 or 2 dicts with "image" [n,1,H,W] (the warped views), "valid_mask" [n,1,H,W] and
        "homographies" [n,3,3] (the un-warp matrices, see combine_heatmap).  Returns one float64 [N,3] array per
        sample (+ the aggregated heatmaps when asked)."""
        views = [_dev_f32(s["image"], self.device) for s in samples]
        masks = [_dev_f32(s["valid_mask"], self.device) for s in samples]
        hms = [_dev_f32(s["homographies"], self.device) for s in samples]
        n, _, h, w = views[0].shape
        eng = self.net.engine(n, h, w, self.device)
        outs = eng.export_points(views, masks, hms, want_heatmap=want_heatmap, **self.args)
        pts = [L.points_to_numpy(o["pts"], o["count"], self.args["subpixel"]) for o in outs]
        return (pts, [o["heatmap"] for o in outs]) if want_heatmap else pts


def export_detector_homoAdapt_gpu(config, output_dir, args):
    """export.py:192-352: pseudo ground truth by homography adaptation, one `<name>.npz {"pts": [N,3]}` per image.
    The data loader is the HOST repository's (`utils.loader.dataLoader_test`, as in the reference); images are
    sharded over ranks when torch.distributed is initialised (replaces nn.DataParallel)."""
    from utils.loader import dataLoader_test as dataLoader  # the reference's own loader (export.py:236-239)

    task = config["data"]["dataset"]
    export_task = config["data"]["export_folder"]
    if not torch.cuda.is_available():
        raise RuntimeError("export_detector_homoAdapt_gpu needs a HIP device: there is no CPU fallback")
    rank, world = 0, 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    mcfg = config["model"]
    save_output = Path(output_dir) / "predictions" / export_task
    os.makedirs(Path(output_dir) / "checkpoints", exist_ok=True)
    os.makedirs(save_output, exist_ok=True)

    fe = SuperPointFrontend_torch(config=config, weights_path=config["pretrained"], nms_dist=mcfg["nms"],
                                  conf_thresh=mcfg["detection_threshold"], nn_thresh=0.7, cuda=False, device=device)
    fe.net_parallel()
    exporter = HomoAdaptExporter(fe.net, device, fe.conf_thresh, fe.nms_dist, mcfg["top_k"], fe.subpixel,
                                 fe.border_remove)
    if rank == 0:
        with open(save_output / "export.txt", "a") as f:
            f.write("load model: %s\n" % config["pretrained"])
            f.write("homography adaptation: %s\n" % config["data"]["homography_adaptation"]["num"])

    test_loader = dataLoader(config, dataset=task, export_task=export_task)["test_loader"]
    pending, count = [], 0

    def flush():
        nonlocal count
        if not pending:
            return
        for (name, scene), pts in zip([p[0] for p in pending], exporter([p[1] for p in pending])):
            if scene is not None:
                os.makedirs(Path(save_output, scene), exist_ok=True)
            np.savez_compressed(Path(save_output, "{}.npz".format(name)), pts=pts)
            count += 1
        pending.clear()

    for i, sample in enumerate(test_loader):
        if i % world != rank:
            continue
        name = sample["name"][0]
        if Path(save_output, "{}.npz".format(name)).exists():
            logging.info("file %s exists. skip the sample.", name)
            continue
        # loader batch of 1: image [1,n,H,W] -> [n,1,H,W]; the un-warp matrices are sample["homographies"]
        # (export.py:281-284 binds them to the name inv_homographies before combine_heatmap)
        s = {"image": sample["image"].transpose(0, 1), "valid_mask": sample["valid_mask"].transpose(0, 1),
             "homographies": sample["homographies"][0]}
        pending.append(((name, sample["scene_name"][0] if "scene_name" in sample else None), s))
        if len(pending) == 2:
            flush()
    flush()
    logging.info("output pseudo ground truth: %d", count)
    if rank == 0:
        with open(save_output / "export.txt", "a") as f:
            f.write("output pairs: %d\n" % count)
    return count
