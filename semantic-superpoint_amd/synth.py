"""Seeded synthetic image pairs resident in HBM, with the shapes / dtypes / value ranges of the reference's
`sample` dict (Train_model_heatmap_all.py:212-251) and the recipe of SURVEY.md section 8d.

Pair construction follows the dataset side of the reference (datasets/Coco.py:341-392):
  homography  : utils/homographies.py:12-141 (sample_homography_np) with the `warped_pair.params` of
                configs/superpoint_coco_train_heatmap.yaml:41-52, then inverted (Coco.py:342-350)
  warped image: utils/utils.py:347-385 (inv_warp_image_batch: bilinear grid_sample, zeros padding, align_corners=True)
  warped labels: datasets/data_tools.py:37-63 (warpLabels: point warp with the 2/W-scaled homography, round, scatter)
  valid mask  : utils/utils.py:715-742 (nearest warp of ones + elliptical erosion, radius 3)
This runs ONCE before the timed region; torch is used as device-side plumbing for it (it is not part of the
measured hot path).  Default weights: PyTorch's Conv2d / BatchNorm2d default initialisation.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WARP_PARAMS = dict(translation=True, rotation=True, scaling=True, perspective=True, scaling_amplitude=0.2,
                   perspective_amplitude_x=0.2, perspective_amplitude_y=0.2, patch_ratio=0.85, max_angle=1.57,
                   allow_artifacts=True)


def _truncnorm(rs, std):
    while True:
        v = rs.randn()
        if abs(v) <= 2.0:
            return v * std


def sample_homography(rs, shape=(2, 2), shift=-1, perspective=True, scaling=True, rotation=True, translation=True,
                      n_scales=5, n_angles=25, scaling_amplitude=0.2, perspective_amplitude_x=0.2,
                      perspective_amplitude_y=0.2, patch_ratio=0.85, max_angle=1.57, allow_artifacts=True,
                      translation_overflow=0.0):
    """Random homography in normalised [-1,1]^2 coordinates (utils/homographies.py:12-141)."""
    pts1 = np.array([[0., 0.], [0., 1.], [1., 1.], [1., 0.]])
    margin = (1 - patch_ratio) / 2
    pts2 = margin + np.array([[0, 0], [0, patch_ratio], [patch_ratio, patch_ratio], [patch_ratio, 0]])
    if perspective:
        if not allow_artifacts:
            perspective_amplitude_x = min(perspective_amplitude_x, margin)
            perspective_amplitude_y = min(perspective_amplitude_y, margin)
        pd = _truncnorm(rs, perspective_amplitude_y / 2)
        hl = _truncnorm(rs, perspective_amplitude_x / 2)
        hr = _truncnorm(rs, perspective_amplitude_x / 2)
        pts2 = pts2 + np.array([[hl, pd], [hl, -pd], [hr, pd], [hr, -pd]])
    if scaling:
        scales = np.array([1.0] + [1 + _truncnorm(rs, scaling_amplitude / 2) for _ in range(n_scales)])  # :83-84: [1, s_1 .. s_n]
        center = pts2.mean(axis=0, keepdims=True)
        scaled = (pts2 - center)[None] * scales[:, None, None] + center
        valid = np.arange(n_scales) if allow_artifacts else np.where(((scaled >= 0) & (scaled < 1)).all(axis=(1, 2)))[0]
        pts2 = scaled[valid[rs.randint(valid.shape[0])]]
    if translation:
        t_min, t_max = pts2.min(axis=0), (1 - pts2).min(axis=0)
        if allow_artifacts:
            t_min = t_min + translation_overflow
            t_max = t_max + translation_overflow
        pts2 = pts2 + np.array([rs.uniform(-t_min[0], t_max[0]), rs.uniform(-t_min[1], t_max[1])])[None]
    if rotation:
        angles = np.concatenate((np.linspace(-max_angle, max_angle, n_angles), [0.0]))
        center = pts2.mean(axis=0, keepdims=True)
        rot = np.stack([np.cos(angles), -np.sin(angles), np.sin(angles), np.cos(angles)], axis=1).reshape(-1, 2, 2)
        rotated = np.matmul((pts2 - center)[None], rot) + center
        valid = np.arange(n_angles) if allow_artifacts else np.where(((rotated >= 0) & (rotated < 1)).all(axis=(1, 2)))[0]
        pts2 = rotated[valid[rs.randint(valid.shape[0])]]
    sh = np.array(shape[::-1], dtype=np.float64)
    p1, p2 = pts1 * sh[None] + shift, pts2 * sh[None] + shift
    A, b = [], []
    for (x, y), (u, v) in zip(p1, p2):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(A), np.array(b))
    return np.append(h, 1.0).reshape(3, 3).astype(np.float32)


def warp_image(img, inv_h, mode="bilinear"):
    B, C, H, W = img.shape
    dev = img.device
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, H, device=dev), torch.linspace(-1, 1, W, device=dev), indexing="ij")
    pts = torch.stack((gx.reshape(-1), gy.reshape(-1), torch.ones(H * W, device=dev)), dim=0)  # [3, HW]
    w = inv_h.to(dev).float() @ pts  # [B,3,HW]
    src = (w[:, :2] / w[:, 2:]).permute(0, 2, 1).reshape(B, H, W, 2)
    return F.grid_sample(img, src, mode=mode, padding_mode="zeros", align_corners=True)


def _ellipse(r):
    n = 2 * r
    k = torch.zeros(n, n)
    c = n // 2
    for i in range(n):
        dy = i - c
        if abs(dy) <= c:
            dx = int(round(c * math.sqrt(max((c * c - dy * dy) / float(c * c), 0.0))))
            k[i, max(c - dx, 0):min(c + dx + 1, n)] = 1
    return k


def erode(mask, r):
    """Binary erosion with OpenCV's MORPH_ELLIPSE(2r,2r) element, anchor (r,r), out-of-image ignored."""
    if r <= 0:
        return mask
    k = _ellipse(r).to(mask.device)
    n = 2 * r
    padded = F.pad(mask, (r, n - 1 - r, r, n - 1 - r), value=1.0)
    hits = F.conv2d(padded, k.view(1, 1, n, n))
    return (hits >= k.sum() - 0.5).float()


def warp_labels(labels, Hm):
    """labels [B,1,H,W] {0,1}; Hm [B,3,3] normalised.  Integer keypoints -> warped, rounded, scattered."""
    B, _, H, W = labels.shape
    dev = labels.device
    T = torch.tensor([[2.0 / W, 0, -1], [0, 2.0 / H, -1], [0, 0, 1]], device=dev)
    Hp = torch.inverse(T) @ Hm.to(dev).float() @ T
    out = torch.zeros_like(labels)
    for i in range(B):
        yx = torch.nonzero(labels[i, 0])
        if yx.numel() == 0:
            continue
        pts = torch.stack((yx[:, 1].float(), yx[:, 0].float(), torch.ones(yx.shape[0], device=dev)), dim=0)
        w = Hp[i] @ pts
        xy = (w[:2] / w[2:]).t()
        keep = (xy[:, 0] >= 0) & (xy[:, 0] <= W - 1) & (xy[:, 1] >= 0) & (xy[:, 1] <= H - 1)
        q = xy[keep].round().long()
        out[i, 0, q[:, 1], q[:, 0]] = 1
    return out


def _warps(device):
    """(warp_image, erode, warp_labels): the HIP kernels on a GPU; the torch restatements above only serve the
    host-side CPU tests of this generator (they are not a fallback of the training path)."""
    if torch.device(device).type == "cuda":
        from . import lib as L
        return (lambda img, inv, mode="bilinear": L.op_warp_image(img.contiguous(), inv, nearest=(mode == "nearest")),
                lambda m, r: L.op_erode(m.contiguous(), r), lambda lab, Hm: L.op_warp_labels(lab.contiguous(), Hm))
    return warp_image, erode, warp_labels


def make_pair(B, H, W, device, seed=0, semantic=False, kp_prob=0.003, erosion=3, n_classes=133):
    warp_image, erode, warp_labels = _warps(device)
    rs = np.random.RandomState(seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.rand(B, 1, H, W, generator=g).to(device)
    lab = (torch.rand(B, 1, H, W, generator=g) < kp_prob).float().to(device)
    Hs = torch.from_numpy(np.stack([np.linalg.inv(sample_homography(rs, **WARP_PARAMS)) for _ in range(B)])
                          .astype(np.float32)).to(device)
    inv = torch.inverse(Hs).contiguous()
    warped = warp_image(img, inv).contiguous()
    vm = erode(warp_image(torch.ones_like(img), inv, mode="nearest"), erosion).contiguous()
    wl = warp_labels(lab, Hs).contiguous()
    s = {"image": img.contiguous(), "warped_img": warped, "labels_2D": lab, "warped_labels": wl,
         "labels_2D_gaussian": lab.clone(), "warped_labels_gaussian": wl.clone(),
         "valid_mask": torch.ones_like(img), "warped_valid_mask": vm, "homographies": Hs.contiguous(),
         "inv_homographies": inv}
    if torch.device(device).type == "cuda":  # exact cell-space matrices for the device sampler (lib.scaled_homographies)
        from . import lib as L
        s["cell_homographies"] = L.scaled_homographies(Hs, H // 8, W // 8).to(device)
    if semantic:
        sem = torch.randint(0, n_classes + 1, (B, H, W), generator=g).to(device)
        ws = warp_image(sem.float().unsqueeze(1), inv).squeeze(1).long()
        ws[vm.view(B, H, W) == 0] = n_classes
        s["semantic"], s["warped_sem"] = sem.contiguous(), ws.contiguous()
    return s


def default_init_state_dict(layer_table, seed=0):
    """PyTorch default init of Conv2d (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias) and
    BatchNorm2d (gamma 1, beta 0, running 0/1) for the given [(conv, bn, cin, cout, k)] table."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for conv, bn, cin, cout, k in layer_table:
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[conv + ".weight"] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        sd[conv + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
        if bn is not None:
            sd[bn + ".weight"] = torch.ones(cout)
            sd[bn + ".bias"] = torch.zeros(cout)
            sd[bn + ".running_mean"] = torch.zeros(cout)
            sd[bn + ".running_var"] = torch.ones(cout)
            sd[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd
