"""CPU oracle: a plain-PyTorch (fp32, CPU) restatement of the reference's pair-training hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product path (semantic-superpoint_amd/) never does and fails
loudly when its HIP library is missing.

Parity pin: every function below is checked against the REAL reference (imported in the build
container through oracle/ref_harness.py) by oracle/make_goldens.py, which also writes the
fixtures in tests/golden/ that tests/test_oracle_golden.py re-checks on every run (the reference
itself ships no tests or golden vectors for this path: SURVEY.md section 4).
Third-party arithmetic (torch ops) is torch 2.10 semantics; cv2 (erode / getStructuringElement /
getPerspectiveTransform) and torchgeometry (SpatialSoftArgmax2d) are absent from the image: they are restated
from their published sources and pinned against published / hand-evaluated vectors
(test_third_party_definitions_hand_vectors), not against the binaries - "parity unpinned" for those
sub-steps; masks / homographies are *inputs* in every parity test.

All file:line citations are relative to the reference repository root.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Architecture tables (models/SuperPointNet_gauss2.py:15-40, models/SuperPointNet_gauss2_ssmall.py:17-49,
# models/unet_parts.py:10-48).  Each entry: (conv key, bn key | None, cin, cout, ksize)
# --------------------------------------------------------------------------------------
_ENC = [
    ("inc.conv.conv.0", "inc.conv.conv.1", 1, 64, 3),
    ("inc.conv.conv.3", "inc.conv.conv.4", 64, 64, 3),
    ("down1.mpconv.1.conv.0", "down1.mpconv.1.conv.1", 64, 64, 3),
    ("down1.mpconv.1.conv.3", "down1.mpconv.1.conv.4", 64, 64, 3),
    ("down2.mpconv.1.conv.0", "down2.mpconv.1.conv.1", 64, 128, 3),
    ("down2.mpconv.1.conv.3", "down2.mpconv.1.conv.4", 128, 128, 3),
    ("down3.mpconv.1.conv.0", "down3.mpconv.1.conv.1", 128, 128, 3),
    ("down3.mpconv.1.conv.3", "down3.mpconv.1.conv.4", 128, 128, 3),
]
_HEADS_SP = [
    ("convPa", "bnPa", 128, 256, 3),
    ("convPb", "bnPb", 256, 65, 1),
    ("convDa", "bnDa", 128, 256, 3),
    ("convDb", "bnDb", 256, 256, 1),
]


def layer_table(arch, n_classes=133):
    if arch == "SuperPointNet_gauss2":
        return _ENC + _HEADS_SP
    if arch == "SuperPointNet_gauss2_ssmall":
        return _ENC + _HEADS_SP + [("convDS", "bnS1", 128, 256, 3), ("convSout", None, 256, n_classes, 1)]
    raise KeyError(arch)


def state_spec(arch, n_classes=133):
    """(key, shape, dtype) in torch state_dict order."""
    spec = []
    for conv, bn, cin, cout, k in layer_table(arch, n_classes):
        spec.append((conv + ".weight", (cout, cin, k, k), np.float32))
        spec.append((conv + ".bias", (cout,), np.float32))
        if bn is not None:
            spec.append((bn + ".weight", (cout,), np.float32))
            spec.append((bn + ".bias", (cout,), np.float32))
            spec.append((bn + ".running_mean", (cout,), np.float32))
            spec.append((bn + ".running_var", (cout,), np.float32))
            spec.append((bn + ".num_batches_tracked", (), np.int64))
    return spec


def init_state_dict(arch, seed=0, n_classes=133):
    """Deterministic test weights (NOT the reference's default init): conv ~ U(+-sqrt(3/fan_in)),
    conv bias ~ U(+-0.1), BN gamma ~ U(0.5,1.5), beta ~ U(-0.3,0.3) so that parity tests exercise
    the affine terms.  Draw order == state_dict order."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for conv, bn, cin, cout, k in layer_table(arch, n_classes):
        b = math.sqrt(3.0 / (cin * k * k))
        sd[conv + ".weight"] = rs.uniform(-b, b, size=(cout, cin, k, k)).astype(np.float32)
        sd[conv + ".bias"] = rs.uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)
        if bn is not None:
            sd[bn + ".weight"] = rs.uniform(0.5, 1.5, size=(cout,)).astype(np.float32)
            sd[bn + ".bias"] = rs.uniform(-0.3, 0.3, size=(cout,)).astype(np.float32)
            sd[bn + ".running_mean"] = np.zeros((cout,), np.float32)
            sd[bn + ".running_var"] = np.ones((cout,), np.float32)
            sd[bn + ".num_batches_tracked"] = np.zeros((), np.int64)
    return sd


def to_torch(sd, requires_grad=False):
    out = OrderedDict()
    for k, v in sd.items():
        t = torch.as_tensor(np.array(v)).clone()
        if requires_grad and t.dtype == torch.float32 and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


def param_keys(arch, n_classes=133):
    """Keys of trainable tensors in net.parameters() order (== state_dict order minus buffers)."""
    return [k for k, _, _ in state_spec(arch, n_classes)
            if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


# --------------------------------------------------------------------------------------
# Model forward  (models/unet_parts.py:10-48; SuperPointNet_gauss2.py:42-69; _ssmall.py:58-99)
# --------------------------------------------------------------------------------------
def _conv_bn(x, sd, conv, bn, k, train, relu, forced=None):
    y = F.conv2d(x, sd[conv + ".weight"], sd[conv + ".bias"], padding=k // 2)
    if bn is not None:
        # nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1
        y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                         sd[bn + ".bias"], training=train, momentum=0.1, eps=1e-5)
        if train:
            sd[bn + ".num_batches_tracked"] += 1
    if relu and forced is not None:  # test hook: the ReLU gate is dictated by the caller (see forward)
        return y * forced["relu"][conv].to(y.dtype)
    return F.relu(y) if relu else y


def _forced_pool(h, idx):
    """2x2 max-pool whose winner is dictated: idx [N,C,H/2,W/2] in 0..3 (row-major position inside the window)."""
    N, C, H, W = h.shape
    win = h.view(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
    return torch.gather(win, 4, idx.long().unsqueeze(-1)).squeeze(-1)


# --------------------------------------------------------------------------------------
# bf16 leg (BASELINE configs[3]: "bf16 compute / fp32 master").  The reference itself is fp32-only (models/unet_parts.py:14-21);
# this restates the SAME network with the rounding points of the HIP bf16 path (conv algorithm 12, csrc/conv_bf16.hip.h),
# everything else in fp32 or better:
#   * matrix-core operands: bf16(activated input), bf16(weight); products exact, accumulation fp32 (here: torch's fp32 conv);
#   * every 3x3 layer stores its raw output y as bf16; BatchNorm statistics are those of the STORED tensor; the affine is one
#     fp32 fma per element, z = fma(y, scale, shift) with scale = gamma * invstd, shift = beta - mean * scale;
#   * backward: the gradient wrt a stored activation (dOut) and wrt a raw conv output (dY) are bf16 tensors; BatchNorm backward
#     runs in fp32 on them; weight gradients accumulate in fp32 into the fp32 master gradient;
#   * the first layer (K = 9, fp32 vector arithmetic on the fp32 image) rounds only its stored output; the pointwise heads write
#     fp32 logits / descriptors, their fp32 gradients are rounded to bf16 on load by the data- and weight-gradient convolutions.
# --------------------------------------------------------------------------------------
def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _QuantBoth(torch.autograd.Function):  # a tensor that lives in HBM as bf16, and so does its gradient
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _QuantFwd(torch.autograd.Function):  # rounded on the way in (weights; layer-0 output), gradient untouched
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _QuantBwd(torch.autograd.Function):  # fp32 tensor whose GRADIENT is rounded to bf16 by its consumers
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _bn_affine_fma(y, sd, bn, train):
    """BatchNorm2d as the HIP path evaluates it (csrc/bn_kernels.hip.h bn_finalize_kernel + one fp32 fma per element): batch
    statistics in fp64, invstd / scale / shift rounded to fp32, z = fma(y, scale, shift) (emulated exactly in fp64)."""
    yd = y.double()
    if train:
        mean = yd.mean(dim=(0, 2, 3))
        var = yd.var(dim=(0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        with torch.no_grad():
            rm, rv = sd[bn + ".running_mean"], sd[bn + ".running_var"]
            rm.copy_((0.9 * rm.double() + 0.1 * mean).float())
            rv.copy_((0.9 * rv.double() + 0.1 * var * (n / (n - 1) if n > 1 else 1.0)).float())
            sd[bn + ".num_batches_tracked"] += 1
    else:
        mean, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    invstd = (var + 1e-5).rsqrt().float()
    scale = sd[bn + ".weight"] * invstd
    shift = sd[bn + ".bias"] - mean.float() * scale
    z = yd * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    return z.float()


def _forward_bf16(sd, x, arch, train, n_classes, return_x4):
    t = layer_table(arch, n_classes)
    op = x
    for i, (conv, bn, cin, cout, k) in enumerate(t[:8]):
        if i == 0:  # first layer: fp32 arithmetic on the fp32 image, stored as bf16
            y = _QuantFwd.apply(F.conv2d(op, sd[conv + ".weight"], sd[conv + ".bias"], padding=1))
        else:
            y = _QuantBoth.apply(F.conv2d(op, _QuantFwd.apply(sd[conv + ".weight"]), sd[conv + ".bias"], padding=1))
        a = F.relu(_bn_affine_fma(y, sd, bn, train))
        if i in (1, 3, 5):  # the next layer sits behind MaxPool2d(2) (unet_parts.py:41-44)
            a = F.max_pool2d(a, 2)
        op = _QuantBoth.apply(a)
    x4 = op

    def head(c3, b3, c1, b1):
        y = _QuantBoth.apply(F.conv2d(x4, _QuantFwd.apply(sd[c3 + ".weight"]), sd[c3 + ".bias"], padding=1))
        a = _QuantBoth.apply(F.relu(_bn_affine_fma(y, sd, b3, train)))
        o = _QuantBwd.apply(F.conv2d(a, _QuantFwd.apply(sd[c1 + ".weight"]), sd[c1 + ".bias"]))
        return _bn_affine_fma(o, sd, b1, train) if b1 is not None else o

    semi = head("convPa", "bnPa", "convPb", "bnPb")
    desc = head("convDa", "bnDa", "convDb", "bnDb")
    out = {}
    if arch.endswith("ssmall"):
        out["sem"] = F.interpolate(head("convDS", "bnS1", "convSout", None), x.shape[2:], mode="bilinear", align_corners=False)
    dn = torch.norm(desc, p=2, dim=1)
    out["semi"] = semi
    out["desc"] = desc.div(dn.unsqueeze(1))
    if return_x4:
        out["x4"] = x4
    return out


def forward(sd, x, arch="SuperPointNet_gauss2", train=True, n_classes=133, return_x4=False, forced=None, operand_dtype=None):
    """x [N,1,H,W] -> {"semi","desc"[,"sem"]}.  `sd` is a dict of torch tensors; BN running
    statistics are updated in place when train=True (module default; the reference never calls
    .eval() while training: SURVEY.md section 7 'Hard parts').
    forced (test hook, not reference behaviour): {"relu": {conv name: 0/1 gate [N,C,H,W]}, "pool": {layer index:
    winner index [N,C,H/2,W/2]}} replaces the data-dependent ReLU gates and max-pool winners by the given ones, which
    makes the network a smooth function of its parameters: used to show that the end-to-end gradient differences
    between the HIP path and the oracle come from gate flips of activations within rounding distance of 0 only.
    operand_dtype=torch.bfloat16: the bf16 leg above (not reference behaviour: the reference is fp32-only)."""
    if operand_dtype is not None:
        if operand_dtype != torch.bfloat16 or forced is not None:
            raise ValueError("operand_dtype must be None or torch.bfloat16 (without forced gates)")
        return _forward_bf16(sd, x, arch, train, n_classes, return_x4)
    t = layer_table(arch, n_classes)
    h = x
    for i, (conv, bn, cin, cout, k) in enumerate(t[:8]):
        if i in (2, 4, 6):  # down = MaxPool2d(2) -> double_conv   (unet_parts.py:41-44)
            h = F.max_pool2d(h, 2) if forced is None else _forced_pool(h, forced["pool"][i])
        h = _conv_bn(h, sd, conv, bn, k, train, relu=True, forced=forced)
    x4 = h
    cPa = _conv_bn(x4, sd, "convPa", "bnPa", 3, train, relu=True, forced=forced)
    semi = _conv_bn(cPa, sd, "convPb", "bnPb", 1, train, relu=False)
    cDa = _conv_bn(x4, sd, "convDa", "bnDa", 3, train, relu=True, forced=forced)
    desc = _conv_bn(cDa, sd, "convDb", "bnDb", 1, train, relu=False)
    out = {}
    if arch.endswith("ssmall"):
        s = _conv_bn(x4, sd, "convDS", "bnS1", 3, train, relu=True, forced=forced)
        s = _conv_bn(s, sd, "convSout", None, 1, train, relu=False)
        out["sem"] = F.interpolate(s, x.shape[2:], mode="bilinear", align_corners=False)
    dn = torch.norm(desc, p=2, dim=1)  # SuperPointNet_gauss2.py:64-65 (no epsilon)
    desc = desc.div(dn.unsqueeze(1))
    out["semi"] = semi
    out["desc"] = desc
    if return_x4:
        out["x4"] = x4
    return out


# --------------------------------------------------------------------------------------
# Label / mask ops
# --------------------------------------------------------------------------------------
def space_to_depth(x, bs=8):
    """utils/d2s.py:27-44: [B,C,H,W] -> [B,C*bs*bs,H/bs,W/bs], channel = dy*bs+dx for C=1."""
    B, C, H, W = x.shape
    x = x.view(B, C, H // bs, bs, W // bs, bs)
    # reference builds depth index (dy, dx, c): for C == 1 that is dy*bs+dx
    return x.permute(0, 3, 5, 1, 2, 4).reshape(B, bs * bs * C, H // bs, W // bs)


def labels2Dto3D(labels, cell_size=8, add_dustbin=True):
    """utils/utils.py:408-440."""
    B, C, H, W = labels.shape
    Hc, Wc = H // cell_size, W // cell_size
    lab = space_to_depth(labels, cell_size)
    if add_dustbin:
        dustbin = 1 - lab.sum(dim=1)
        dustbin = torch.where(dustbin < 1.0, torch.zeros_like(dustbin), dustbin)  # :431
        lab = torch.cat((lab, dustbin.view(B, 1, Hc, Wc)), dim=1)
        lab = lab / lab.sum(dim=1, keepdim=True)  # :438-439
    return lab


def get_masks(mask_2D, cell_size=8):
    """Train_model_frontend_all.py:373-386: cell valid iff all 64 pixels valid."""
    return torch.prod(labels2Dto3D(mask_2D, cell_size, add_dustbin=False).float(), 1)


def detector_loss(semi, target, mask):
    """Train_model_heatmap_all.py:173-178 (softmax branch): BCELoss(softmax) summed over 65
    channels, masked, / (mask.sum()+1e-5).  BCELoss clamps each log at -100."""
    p = F.softmax(semi, dim=1)
    loss = F.binary_cross_entropy(p, target, reduction="none")
    loss = (loss.sum(dim=1) * mask).sum()
    return loss / (mask.sum() + 1e-5)


def sem_loss(pred, label):
    """Train_model_heatmap_all.py:181-193: CrossEntropyLoss(ignore_index=133), mean over valid."""
    return F.cross_entropy(pred, label, ignore_index=133)


# --------------------------------------------------------------------------------------
# Geometry helpers
# --------------------------------------------------------------------------------------
def scale_homography(Hm, shape, shift=(-1.0, -1.0)):
    """utils/homographies.py:270-276 and utils/utils.py:297-300 (same formula; note 2/W not 2/(W-1))."""
    height, width = shape
    trans = torch.tensor([[2.0 / width, 0.0, shift[0]], [0.0, 2.0 / height, shift[1]], [0.0, 0.0, 1.0]],
                         dtype=torch.float32)
    return torch.inverse(trans) @ Hm @ trans


def warp_points(points, Hm):
    """utils/utils.py:315-343 for one 3x3 homography: points [N,2](x,y) -> [N,2]."""
    pts = torch.cat((points.float(), torch.ones((points.shape[0], 1))), dim=1)
    w = Hm.view(3, 3) @ pts.transpose(0, 1)
    w = w.transpose(0, 1)
    return w[:, :2] / w[:, 2:]


def filter_points(points, shape_wh):
    """utils/utils.py:303-311: keep 0 <= x <= W-1, 0 <= y <= H-1.  Returns (points, mask)."""
    shape = torch.as_tensor(shape_wh).float()
    m = (points >= 0) & (points <= shape - 1)
    m = m[:, 0] & m[:, 1]
    return points[m], m


def cell_matches(Hm, Hc, Wc):
    """sparse_loss.py:184-207: integer cell correspondences under the normalised homography Hm.
    Returns uv_a [n,2], uv_b [n,2] (float tensors holding integers), row-major with v outer."""
    vs, us = torch.meshgrid(torch.arange(Hc), torch.arange(Wc), indexing="ij")
    uv_a = torch.stack((us.reshape(-1), vs.reshape(-1)), dim=1).float()
    H_cell = scale_homography(Hm.float(), (Hc, Wc))
    uv_b = warp_points(uv_a, H_cell)
    uv_b = uv_b.round()  # half-to-even (:197)
    uv_b, m = filter_points(uv_b, (Wc, Hc))
    return uv_a[m], uv_b


def crop_or_pad_choice(n_in, n_out, np_rng):
    """utils/utils.py:964-986 with shuffle=True; `np_rng` exposes permutation/choice
    (np.random module or a RandomState)."""
    choice = np_rng.permutation(n_in)
    if n_in >= n_out:
        return choice[:n_out]
    pad = np_rng.choice(choice, n_out - n_in, replace=True)
    return np.concatenate([choice, pad])


def sample_sparse_indices(Hm, Hc, Wc, n_match=1000, n_non=100, np_rng=np.random, torch_gen=None):
    """The stochastic half of descriptor_loss_sparse (sparse_loss.py:184-246) for one image.
    RNG draw order reproduces the reference: numpy permutation(+choice), torch.rand(2,K),
    torch.rand(K), torch.randn(K)  (correspondence_finder.py:30,278,280).
    Returns dict: uv_a, uv_b [n_match,2] float (integer-valued cells), nm_b [n_match*n_non] int64
    flat cell index (u + v*Wc) in image b.  The a-side of non-match k*n_non+j is match k."""
    uv_a, uv_b = cell_matches(Hm, Hc, Wc)
    choice = torch.as_tensor(crop_or_pad_choice(uv_b.shape[0], n_match, np_rng)).long()
    uv_a, uv_b = uv_a[choice], uv_b[choice]
    K = n_match * n_non
    two = torch.rand(2, K, generator=torch_gen)          # correspondence_finder.py:29-34
    nu = torch.floor(two[0] * Wc).long()
    nv = torch.floor(two[1] * Hc).long()
    # the "perturb near matches" stage is a no-op (`ones = zeros_like`, :269) but burns two draws
    torch.rand(K, generator=torch_gen)
    torch.randn(K, generator=torch_gen)
    return {"uv_a": uv_a, "uv_b": uv_b, "nm_b": nu + nv * Wc}


def norm_pts(pts, shape_wh):
    """utils/utils.py:745-755."""
    return pts / torch.as_tensor(shape_wh).float() * 2 - 1


def descriptor_loss_sparse_given(desc_a, desc_b, idx, lamda_d=1.0, n_non=100, dist="cos", method="2d"):
    """Deterministic half of descriptor_loss_sparse (sparse_loss.py:219-254) for one image given
    the sampled indices.  desc_* [D,Hc,Wc].  Returns (loss, pos, neg).
    method: "2d" = bilinear grid_sample at normPts (pixelwise_contrastive_loss.py:160-184), anything else ("1d") =
    index_select at the integer cell (:185-188).  dist: "cos" = hinge on the dot product (:200-204, :247-256), anything else
    ("euclidean") = squared distance of the matches (:205-206) and (max(0, ||a - b|| - M))^2 of the non-matches (:249-258)."""
    D, Hc, Wc = desc_a.shape
    wh = (Wc, Hc)
    flat_a = desc_a.reshape(D, Hc * Wc).t()
    flat_b = desc_b.reshape(D, Hc * Wc).t()
    if method == "2d":
        ga = norm_pts(idx["uv_a"], wh).view(1, -1, 1, 2)
        gb = norm_pts(idx["uv_b"], wh).view(1, -1, 1, 2)
        da = F.grid_sample(desc_a.unsqueeze(0), ga, mode="bilinear", align_corners=True).squeeze(0).squeeze(-1).t()
        db = F.grid_sample(desc_b.unsqueeze(0), gb, mode="bilinear", align_corners=True).squeeze(0).squeeze(-1).t()
    else:  # sparse_loss.py:224-226,234-236: uv_to_1d, index_select
        da = flat_a[(idx["uv_a"][:, 0] + idx["uv_a"][:, 1] * Wc).long()]
        db = flat_b[(idx["uv_b"][:, 0] + idx["uv_b"][:, 1] * Wc).long()]
    if dist == "cos":
        pos = torch.clamp(1.0 - (da * db).sum(-1), min=0).sum() / da.shape[0]
    else:
        pos = (da - db).pow(2).sum() / da.shape[0]
    # non matches: sparse_loss.py:96-100,245-246 ; pixelwise_contrastive_loss.py:238-263 (M = 0.2, invert=True)
    ia = (idx["uv_a"][:, 0] + idx["uv_a"][:, 1] * Wc).long().repeat_interleave(n_non)
    ib = idx["nm_b"].long()
    if dist == "cos":
        nm = torch.clamp((flat_a[ia] * flat_b[ib]).sum(-1) - 0.2, min=0)
    else:
        nm = torch.clamp((flat_a[ia] - flat_b[ib]).norm(2, 1) - 0.2, min=0).pow(2)
    nnz = int((nm != 0).sum())
    neg = nm.sum() / (nnz + 1)  # sparse_loss.py:154
    return lamda_d * pos + neg, pos, neg


def batch_descriptor_loss_sparse(desc, desc_w, homographies, indices=None, lamda_d=1.0, n_match=1000,
                                 n_non=100, np_rng=np.random, torch_gen=None, dist="cos", method="2d"):
    """sparse_loss.py:267-284.  `indices` (list per image) overrides sampling."""
    ls, ps, ns, used = [], [], [], []
    for i in range(desc.shape[0]):
        idx = indices[i] if indices is not None else sample_sparse_indices(
            homographies[i].float(), desc.shape[2], desc.shape[3], n_match, n_non, np_rng, torch_gen)
        l, p, n = descriptor_loss_sparse_given(desc[i], desc_w[i], idx, lamda_d, n_non, dist, method)
        ls.append(l), ps.append(p), ns.append(n), used.append(idx)
    return torch.stack(ls).mean(), torch.stack(ps).mean(), torch.stack(ns).mean(), used


def multi_task_loss(eta, det, pos, neg, sem=None):
    """Train_model_heatmap_all.py:62-77 (Kendall uncertainty weighting, 'v2_normal')."""
    loss = det * torch.exp(-eta[0]) + eta[0] + 0.5 * (pos + neg) * torch.exp(-eta[1]) + 0.5 * eta[1]
    if sem is not None:
        loss = loss + sem * torch.exp(-eta[2]) + eta[2]
    return loss


# --------------------------------------------------------------------------------------
# The pair step  (Train_model_heatmap_all.py:195-443)
# --------------------------------------------------------------------------------------
def pair_losses(sd, eta, sample, arch="SuperPointNet_gauss2", indices=None, lambda_loss=1.0, lamda_d=1.0,
                multi_task=True, gaussian=True, n_match=1000, n_non=100, np_rng=np.random, torch_gen=None,
                train=True, dense=None, forced=None, operand_dtype=None, warped_pair=True, sparse_dist="cos", sparse_method="2d"):
    """Forward of both views + all losses.  Returns (loss, scalars dict, aux dict).
    sparse_dist / sparse_method: model.sparse_loss.params.dist / method (sparse_loss.py:76-77; every shipped config: cos / 2d).
    dense: None (sparse descriptor loss) or the dict of model.dense_loss.params (Train_model_heatmap_all.py:131-137).
    warped_pair=False: the single-view branch (`data.warped_pair.enable: false`, :207; the shipped
    configs/magicpoint_shapes_pair.yaml:50-51): ONE forward, detector (+ semantic) loss of the image only; the warped
    terms are the constants 0 of :330-332 and lambda_loss must be 0 (:343 asserts "need a pair of images")."""
    semantic = arch.endswith("ssmall")
    if not warped_pair:
        return _single_view_losses(sd, eta, sample, arch, lambda_loss, multi_task, gaussian, train, forced, operand_dtype)
    out = forward(sd, sample["image"], arch, train=train, forced=None if forced is None else forced[0], operand_dtype=operand_dtype)
    out_w = forward(sd, sample["warped_img"], arch, train=train,  # separate BN statistics (:258,262)
                    forced=None if forced is None else forced[1], operand_dtype=operand_dtype)
    lab = sample["labels_2D_gaussian"] if gaussian else sample["labels_2D"]
    lab_w = sample["warped_labels_gaussian"] if gaussian else sample["warped_labels"]
    l3 = labels2Dto3D(lab).float()
    m3 = get_masks(sample["valid_mask"])
    loss_det = detector_loss(out["semi"], l3, m3)
    l3w = labels2Dto3D(lab_w).float()
    m3w = get_masks(sample["warped_valid_mask"])
    loss_det_w = detector_loss(out_w["semi"], l3w, m3w)
    zero = torch.zeros(())
    loss_sem = sem_loss(out["sem"], sample["semantic"]) if semantic else zero
    loss_sem_w = sem_loss(out_w["sem"], sample["warped_sem"]) if semantic else zero
    if lambda_loss > 0 and dense is not None:
        # :340-350: mask_valid = the warped view's cell mask; `lambda_d` of the config is swallowed by **config
        loss_desc, _, pos, neg = descriptor_loss_dense(
            out["desc"], out_w["desc"], sample["homographies"], mask_valid=m3w.unsqueeze(1),
            lamda_d=float(dense.get("lamda_d", 250)), descriptor_dist=float(dense.get("descriptor_dist", 4)))
        used = None
    elif lambda_loss > 0:
        loss_desc, pos, neg, used = batch_descriptor_loss_sparse(
            out["desc"], out_w["desc"], sample["homographies"], indices, lamda_d, n_match, n_non, np_rng, torch_gen,
            sparse_dist, sparse_method)
    else:
        loss_desc, pos, neg, used = zero, zero, zero, None
    if multi_task:
        loss = multi_task_loss(eta, loss_det + loss_det_w, pos, neg, (loss_sem + loss_sem_w) if semantic else None)
    else:  # uniform sum (:363-365)
        loss = loss_det + loss_det_w + loss_sem + loss_sem_w
        if lambda_loss > 0:
            loss = loss + lambda_loss * loss_desc
    scal = {"loss": loss, "loss_det": loss_det, "loss_det_warp": loss_det_w, "loss_desc": loss_desc,
            "loss_sem": loss_sem, "loss_sem_warp": loss_sem_w, "eta_det": eta[0], "eta_desc": eta[1],
            "positive_dist": pos, "negative_dist": neg}
    if semantic:
        scal["eta_sem"] = eta[2]
    return loss, scal, {"out": out, "out_warp": out_w, "indices": used}


def _single_view_losses(sd, eta, sample, arch, lambda_loss, multi_task, gaussian, train, forced, operand_dtype):
    """Train_model_heatmap_all.py:237-262,296-332,346-365 with if_warp == False."""
    semantic = arch.endswith("ssmall")
    assert not lambda_loss > 0, "need a pair of images"  # :343
    out = forward(sd, sample["image"], arch, train=train, forced=None if forced is None else forced[0], operand_dtype=operand_dtype)
    lab = sample["labels_2D_gaussian"] if gaussian else sample["labels_2D"]
    loss_det = detector_loss(out["semi"], labels2Dto3D(lab).float(), get_masks(sample["valid_mask"]))
    zero = torch.zeros(())
    loss_sem = sem_loss(out["sem"], sample["semantic"]) if semantic else zero
    if multi_task:  # pos = neg = 0: the descriptor weight eta[1] still receives its constant 1/2 gradient
        loss = multi_task_loss(eta, loss_det + zero, zero, zero, (loss_sem + zero) if semantic else None)
    else:
        loss = loss_det + zero + loss_sem + zero
    scal = {"loss": loss, "loss_det": loss_det, "loss_det_warp": zero, "loss_desc": zero, "loss_sem": loss_sem,
            "loss_sem_warp": zero, "eta_det": eta[0], "eta_desc": eta[1], "positive_dist": zero, "negative_dist": zero}
    if semantic:
        scal["eta_sem"] = eta[2]
    return loss, scal, {"out": out, "out_warp": None, "indices": None}


class AdamState:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) restated
    (Train_model_frontend_all.py:183-198).  LR is constant: the scheduler drives an orphaned
    optimizer."""

    def __init__(self, params, lr):
        self.params = params
        self.lr = lr
        self.t = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        with torch.no_grad():
            for p, m, v in zip(self.params, self.m, self.v):
                if p.grad is None:
                    continue
                g = p.grad
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class Trainer:
    """Minimal restatement of the Train_model_heatmap_all step state (net params, eta, Adam)."""

    def __init__(self, arch="SuperPointNet_gauss2", sd_np=None, lr=0.001, seed=0, **loss_kw):
        self.arch = arch
        self.sd = to_torch(sd_np if sd_np is not None else init_state_dict(arch, seed), requires_grad=True)
        self.eta = torch.tensor([1.0, 2.0, 1.0], requires_grad=True)  # Train_model_heatmap_all.py:58
        self.pkeys = param_keys(arch)
        self.opt = AdamState([self.sd[k] for k in self.pkeys] + [self.eta], lr)
        self.loss_kw = loss_kw
        self.batch_size = None
        self.real_batch_size = None

    def train_val_sample(self, sample, n_iter=0, train=True, indices=None, **kw):
        args = dict(self.loss_kw)
        args.update(kw)
        B = sample["image"].shape[0]
        real = self.real_batch_size or B
        if train:
            loss, scal, aux = pair_losses(self.sd, self.eta, sample, self.arch, indices, **args)
            loss.backward()
            self.last_grads = OrderedDict(
                (k, None if self.sd[k].grad is None else self.sd[k].grad.clone()) for k in self.pkeys)
            self.last_grads["eta"] = None if self.eta.grad is None else self.eta.grad.clone()
            if ((n_iter + 1) * B) % real == 0:  # :410-413
                self.opt.step()
                self.opt.zero_grad()
        else:
            with torch.no_grad():
                loss, scal, aux = pair_losses(self.sd, self.eta, sample, self.arch, indices, **args)
        self.scalar_dict = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in scal.items()}
        self.aux = aux
        return float(loss)


# --------------------------------------------------------------------------------------
# Pair construction (dataset side)
# --------------------------------------------------------------------------------------
def inv_warp_image_batch(img, mat_homo_inv, mode="bilinear"):
    """utils/utils.py:347-385: sample img at H^-1 * (linspace(-1,1) grid), zeros padding,
    align_corners=True."""
    if img.dim() in (2, 3):
        img = img.view(1, 1, img.shape[-2], img.shape[-1])
    if mat_homo_inv.dim() == 2:
        mat_homo_inv = mat_homo_inv.view(1, 3, 3)
    B, C, H, W = img.shape
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(-1, 1, H), indexing="ij")
    coor = torch.stack((gx, gy), dim=2).transpose(0, 1).contiguous().view(-1, 2)  # [H*W,2] (x,y), row-major
    pts = torch.cat((coor, torch.ones(coor.shape[0], 1)), dim=1)
    w = (mat_homo_inv.float().reshape(B * 3, 3) @ pts.t()).view(B, 3, -1).transpose(2, 1)
    src = (w[:, :, :2] / w[:, :, 2:]).view(B, H, W, 2).float()
    return F.grid_sample(img, src, mode=mode, align_corners=True)


def structuring_element_ellipse(height, width):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (width, height)) restated from OpenCV's published source
    (modules/imgproc/src/morph.dispatch.cpp, 4.x): row i spans columns c - dx .. c + dx with r = height // 2,
    c = width // 2, dx = round(c * sqrt((r^2 - dy^2) / r^2)).  cv2 is absent from the image; the restatement is pinned
    against the 5x5 element printed in OpenCV's "Morphological Transformations" tutorial and hand-evaluated 3x3 / 6x6
    elements (tests/test_oracle_golden.py::test_third_party_definitions_hand_vectors)."""
    k = np.zeros((height, width), np.uint8)
    r, c = height // 2, width // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(height):
        dy = i - r
        if abs(dy) <= r:
            dx = int(round(c * math.sqrt(max((r * r - dy * dy) * inv_r2, 0.0))))  # saturate_cast<int> rounds to nearest
            k[i, max(c - dx, 0):min(c + dx + 1, width)] = 1
    return k


def ellipse_kernel(r):
    """The element of compute_valid_mask (utils/utils.py:736-740): MORPH_ELLIPSE of size (2r, 2r), anchor at (r, r)."""
    return structuring_element_ellipse(2 * r, 2 * r)


def erode_ellipse(mask, r):
    """cv2.erode(mask, ellipse(2r), iterations=1) with the default (constant +inf) border:
    out[y,x] = min over kernel support of mask[y+i-r, x+j-r], out-of-image pixels ignored."""
    if r <= 0:
        return mask
    k = ellipse_kernel(r)
    H, W = mask.shape
    out = mask.clone()
    for i in range(k.shape[0]):
        for j in range(k.shape[1]):
            if not k[i, j]:
                continue
            dy, dx = i - r, j - r
            ys0, ys1 = max(0, -dy), min(H, H - dy)
            xs0, xs1 = max(0, -dx), min(W, W - dx)
            out[ys0:ys1, xs0:xs1] = torch.minimum(out[ys0:ys1, xs0:xs1], mask[ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx])
    return out


def compute_valid_mask(image_shape, inv_homography, erosion_radius=0):
    """utils/utils.py:715-742: nearest warp of ones, then elliptical erosion."""
    if inv_homography.dim() == 2:
        inv_homography = inv_homography.view(-1, 3, 3)
    B = inv_homography.shape[0]
    mask = inv_warp_image_batch(torch.ones(B, 1, image_shape[0], image_shape[1]), inv_homography, mode="nearest")
    mask = mask.view(B, image_shape[0], image_shape[1])
    if erosion_radius > 0:
        mask = torch.stack([erode_ellipse(mask[i], erosion_radius) for i in range(B)])
    return mask


def warp_labels(pnts_xy, H, W, homography):
    """datasets/data_tools.py:37-63 (labels only): warp integer keypoints (x,y) with the
    normalised homography, keep in-range, round half-to-even, scatter 1."""
    Hpix = scale_homography(homography.float(), (H, W))  # utils/utils.py:297-300
    wp = warp_points(pnts_xy.long().float(), Hpix)
    wp, _ = filter_points(wp, (W, H))
    lab = torch.zeros(H, W)
    q = wp.round().long()
    lab[q[:, 1], q[:, 0]] = 1
    return lab.view(1, H, W)


def sample_homography(rs, shape=(2, 2), shift=-1, perspective=True, scaling=True, rotation=True, translation=True,
                      n_scales=5, n_angles=25, scaling_amplitude=0.2, perspective_amplitude_x=0.2,
                      perspective_amplitude_y=0.2, patch_ratio=0.85, max_angle=1.57, allow_artifacts=True,
                      translation_overflow=0.0):
    """utils/homographies.py:12-141 restated with numpy only (truncnorm via rejection, the 4-point
    solve via an 8x8 linear system instead of cv2.getPerspectiveTransform) -- distribution-level
    restatement, parity unpinned (scipy RNG stream / cv2 not reproduced).  Returns 3x3 float32,
    the matrix the reference returns (pts1*shape+shift -> pts2*shape+shift ... solved as in :135-141)."""
    def tn(std):  # truncnorm(-2,2, scale=std)
        while True:
            v = rs.randn()
            if abs(v) <= 2:
                return v * std
    pts1 = np.array([[0., 0.], [0., 1.], [1., 1.], [1., 0.]])
    margin = (1 - patch_ratio) / 2
    pts2 = margin + np.array([[0, 0], [0, patch_ratio], [patch_ratio, patch_ratio], [patch_ratio, 0]])
    if perspective:
        if not allow_artifacts:
            perspective_amplitude_x = min(perspective_amplitude_x, margin)
            perspective_amplitude_y = min(perspective_amplitude_y, margin)
        pd = tn(perspective_amplitude_y / 2)
        hl = tn(perspective_amplitude_x / 2)
        hr = tn(perspective_amplitude_x / 2)
        pts2 += np.array([[hl, pd], [hl, -pd], [hr, pd], [hr, -pd]])
    if scaling:
        scales = np.array([1.0] + [1 + tn(scaling_amplitude / 2) for _ in range(n_scales)])  # :83-84: [1, s_1 .. s_n]
        center = pts2.mean(axis=0, keepdims=True)
        scaled = (pts2 - center)[None] * scales[:, None, None] + center
        valid = np.arange(n_scales) if allow_artifacts else \
            np.where(((scaled >= 0) & (scaled < 1)).all(axis=(1, 2)))[0]
        pts2 = scaled[valid[rs.randint(valid.shape[0])]]
    if translation:
        t_min, t_max = pts2.min(axis=0), (1 - pts2).min(axis=0)
        if allow_artifacts:
            t_min += translation_overflow
            t_max += translation_overflow
        pts2 += np.array([rs.uniform(-t_min[0], t_max[0]), rs.uniform(-t_min[1], t_max[1])])[None]
    if rotation:
        angles = np.concatenate((np.linspace(-max_angle, max_angle, n_angles), [0.0]))
        center = pts2.mean(axis=0, keepdims=True)
        rot = np.stack([np.cos(angles), -np.sin(angles), np.sin(angles), np.cos(angles)], axis=1).reshape(-1, 2, 2)
        rotated = np.matmul((pts2 - center)[None], rot) + center
        valid = np.arange(n_angles) if allow_artifacts else \
            np.where(((rotated >= 0) & (rotated < 1)).all(axis=(1, 2)))[0]
        pts2 = rotated[valid[rs.randint(valid.shape[0])]]
    sh = np.array(shape[::-1], dtype=np.float64)
    p1 = pts1 * sh[None] + shift
    p2 = pts2 * sh[None] + shift
    return get_perspective_transform(p1, p2).astype(np.float32)


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst) restated from its published definition (imgproc/src/imgwarp.cpp): the
    3x3 H with h33 = 1 and dst_i ~ H src_i for four point pairs, i.e. the solution of the 8x8 system
    [x y 1 0 0 0 -ux -uy; 0 0 0 x y 1 -vx -vy] h = [u; v] in float64.  Pinned against homographies with a known closed
    form (tests/test_oracle_golden.py::test_third_party_definitions_hand_vectors); cv2's LU pivoting order is not
    reproduced, the results agree to float64 round-off."""
    A, b = [], []
    for (x, y), (u, v) in zip(np.asarray(src, np.float64), np.asarray(dst, np.float64)):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(A), np.array(b))
    return np.append(h, 1.0).reshape(3, 3)


def make_synthetic_pair(B, H, W, seed=0, semantic=False, kp_prob=0.003, erosion=3, n_classes=133):
    """Seeded synthetic `sample` dict with the shapes/dtypes of SURVEY.md section 8a row a4 and the
    recipe of section 8d (uniform images, Bernoulli keypoints, homographies as in datasets/Coco.py:341-392)."""
    rs = np.random.RandomState(seed)
    img = torch.from_numpy(rs.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    lab = torch.from_numpy((rs.uniform(size=(B, 1, H, W)) < kp_prob).astype(np.float32))
    Hs = np.stack([np.linalg.inv(sample_homography(rs)) for _ in range(B)]).astype(np.float32)  # Coco.py:342-350
    Hs_t = torch.from_numpy(Hs)
    inv_t = torch.inverse(Hs_t).contiguous()
    warped = inv_warp_image_batch(img, inv_t)
    wl = torch.stack([warp_labels(torch.nonzero(lab[i, 0]).flip(1), H, W, Hs_t[i]) for i in range(B)])
    vm = compute_valid_mask((H, W), inv_t, erosion_radius=erosion).view(B, 1, H, W)
    s = {"image": img, "warped_img": warped, "labels_2D": lab, "warped_labels": wl,
         "labels_2D_gaussian": lab.clone(), "warped_labels_gaussian": wl.clone(),
         "valid_mask": torch.ones(B, 1, H, W), "warped_valid_mask": vm,
         "homographies": Hs_t, "inv_homographies": inv_t}
    if semantic:
        sem = torch.from_numpy(rs.randint(0, n_classes + 1, size=(B, H, W)).astype(np.int64))
        ws = inv_warp_image_batch(sem.float().unsqueeze(1), inv_t, mode="bilinear").squeeze(1).long()
        ws[vm.view(B, H, W) == 0] = n_classes
        s["semantic"], s["warped_sem"] = sem, ws
    return s


def make_compact_pair(B, H, W, seed=0, semantic=False, kp_prob=0.003, erosion=3, n_classes=133, block=16):
    """make_synthetic_pair whose tensors survive a compact fixture: images on the 8-bit grid k / 255 (the warped image
    is re-quantised after the warp), semantic maps constant on block x block tiles.  `compact_to_npz` / `compact_from_npz`
    store / restore the sample as uint8 / packed arrays (full-size golden steps stay well below 1 MB)."""
    rs = np.random.RandomState(seed)
    s = make_synthetic_pair(B, H, W, seed=seed, semantic=False, kp_prob=kp_prob, erosion=erosion)
    q = lambda t: torch.round(t.clamp(0, 1) * 255.0) / 255.0  # noqa: E731
    img = q(s["image"])
    s["image"] = img
    s["warped_img"] = q(inv_warp_image_batch(img, s["inv_homographies"]))
    if semantic:
        coarse = rs.randint(0, n_classes + 1, size=(B, (H + block - 1) // block, (W + block - 1) // block))
        sem = torch.from_numpy(np.kron(coarse, np.ones((block, block), dtype=np.int64))[:, :H, :W].astype(np.int64)).contiguous()
        ws = inv_warp_image_batch(sem.float().unsqueeze(1), s["inv_homographies"], mode="nearest").squeeze(1).long()
        ws[s["warped_valid_mask"].view(B, H, W) == 0] = n_classes
        s["semantic"], s["warped_sem"] = sem, ws
    return s


def compact_to_npz(s):
    out = {"in/image_u8": np.round(s["image"].numpy() * 255.0).astype(np.uint8),
           "in/warped_img_u8": np.round(s["warped_img"].numpy() * 255.0).astype(np.uint8),
           "in/homographies": s["homographies"].numpy(), "in/inv_homographies": s["inv_homographies"].numpy()}
    for k in ("labels_2D", "warped_labels", "valid_mask", "warped_valid_mask"):
        out["in/%s_bits" % k] = np.packbits(s[k].numpy().astype(np.uint8))
    if "semantic" in s:
        out["in/semantic_u8"] = s["semantic"].numpy().astype(np.uint8)
        out["in/warped_sem_u8"] = s["warped_sem"].numpy().astype(np.uint8)
    out["in/shape"] = np.array(s["image"].shape, dtype=np.int32)
    return out


def compact_from_npz(z):
    B, _, H, W = (int(v) for v in z["in/shape"])
    s = {"image": torch.from_numpy(z["in/image_u8"].astype(np.float32)) / 255.0,
         "warped_img": torch.from_numpy(z["in/warped_img_u8"].astype(np.float32)) / 255.0,
         "homographies": torch.from_numpy(z["in/homographies"]), "inv_homographies": torch.from_numpy(z["in/inv_homographies"])}
    for k in ("labels_2D", "warped_labels", "valid_mask", "warped_valid_mask"):
        bits = np.unpackbits(z["in/%s_bits" % k])[:B * H * W]
        s[k] = torch.from_numpy(bits.astype(np.float32)).view(B, 1, H, W)
    s["labels_2D_gaussian"], s["warped_labels_gaussian"] = s["labels_2D"].clone(), s["warped_labels"].clone()
    if "in/semantic_u8" in z:
        s["semantic"] = torch.from_numpy(z["in/semantic_u8"].astype(np.int64))
        s["warped_sem"] = torch.from_numpy(z["in/warped_sem_u8"].astype(np.int64))
    return s


# --------------------------------------------------------------------------------------
# Homography-adaptation export: export.py:192-352
# --------------------------------------------------------------------------------------
def flatten_detection(semi):
    """utils/utils.py:515-560 (batch branch): softmax over the 65 channels, drop the dustbin,
    DepthToSpace(8) (utils/d2s.py:8-27: channel c -> pixel (c // 8, c % 8) of the cell)."""
    dense = F.softmax(semi, dim=1)[:, :-1]
    return F.pixel_shuffle(dense, 8)  # [B,1,H,W]


def combine_heatmap(heatmap, unwarp_mats, mask_2D):
    """export.py:49-60.  heatmap, mask_2D: [N,1,H,W]; unwarp_mats: [N,3,3] -- the matrices the
    reference passes as `inv_homographies` (export.py:281-284 swaps the two sample keys, so this is
    sample["homographies"], the inverse of the matrix each view was produced with).  0/0 -> NaN kept."""
    heatmap = inv_warp_image_batch(heatmap * mask_2D, unwarp_mats, mode="bilinear")
    mask_2D = inv_warp_image_batch(mask_2D, unwarp_mats, mode="bilinear")
    return torch.sum(heatmap, dim=0) / torch.sum(mask_2D, dim=0)  # [1,H,W]


def nms_fast(in_corners, H, W, dist_thresh):
    """models/model_wrap.py:129-192: greedy NMS on a grid, highest confidence first, Chebyshev
    radius dist_thresh.  in_corners: 3xN (x, y, conf).  Returns the kept corners sorted by
    descending confidence.  Tie order follows a STABLE sort here (numpy's default quicksort is
    unspecified for ties); fixtures are generated tie-free."""
    n = in_corners.shape[1]
    if n == 0:
        return np.zeros((3, 0))
    order = np.argsort(-in_corners[2], kind="stable")
    corners = in_corners[:, order]
    rc = corners[:2].round().astype(int)
    if n == 1:
        return np.vstack((rc, in_corners[2])).reshape(3, 1)
    pad = dist_thresh
    grid = np.zeros((H + 2 * pad, W + 2 * pad), dtype=np.int8)
    grid[rc[1] + pad, rc[0] + pad] = 1
    keep = []
    for i in range(n):
        x, y = rc[0, i] + pad, rc[1, i] + pad
        if grid[y, x] == 1:
            grid[y - pad:y + pad + 1, x - pad:x + pad + 1] = 0
            grid[y, x] = -1
            keep.append(i)
    return corners[:, keep]


def get_pts_from_heatmap(heatmap, conf_thresh, nms_dist, border_remove=4):
    """models/model_wrap.py:266-293.  heatmap: np [H,W] -> 3xN float64 (x, y, conf), descending conf."""
    H, W = heatmap.shape
    ys, xs = np.where(heatmap >= conf_thresh)
    if len(ys) == 0:
        return np.zeros((3, 0))
    pts = np.zeros((3, len(ys)))
    pts[0], pts[1], pts[2] = xs, ys, heatmap[ys, xs]
    pts = nms_fast(pts, H, W, nms_dist)
    b = border_remove
    drop = (pts[0] < b) | (pts[0] >= W - b) | (pts[1] < b) | (pts[1] >= H - b)
    return pts[:, ~drop]


def spatial_soft_argmax2d(x, eps=1e-6):
    """torchgeometry (requirements.txt:19, unpinned; v0.1.2) contrib.SpatialSoftArgmax2d with
    normalized_coordinates=False, restated from its published source -- the package is absent from
    the image, so THIS function is parity unpinned.  x: [B,C,h,w] -> [B,C,2] (x, y)."""
    B, C, h, w = x.shape
    v = x.reshape(B, C, -1)
    e = torch.exp(v - v.max(dim=-1, keepdim=True)[0])
    inv = 1.0 / (e.sum(dim=-1, keepdim=True) + eps)
    py, px = torch.meshgrid(torch.linspace(0, h - 1, h), torch.linspace(0, w - 1, w), indexing="ij")
    ey = torch.sum((py.reshape(-1) * e) * inv, dim=-1, keepdim=True)
    ex = torch.sum((px.reshape(-1) * e) * inv, dim=-1, keepdim=True)
    return torch.cat([ex, ey], dim=-1)


def soft_argmax_points(heatmap, pts, patch_size=5):
    """models/model_wrap.py:212-249 + utils/losses.py:53-61,64-91,138-142: 5x5 patches of the
    zero-padded heatmap around each point, normalised to sum 1 (+1e-6), log, soft-argmax;
    the point moves by (dx, dy) - patch_size // 2.  pts: 3xN float64 -> 3xN float64."""
    pts = pts.T.copy()  # [N,3]
    if pts.shape[0] == 0:
        return pts.T
    r = patch_size // 2
    hp = np.pad(np.asarray(heatmap, dtype=np.float32), r, "constant")
    patches = np.stack([hp[int(p[1]):int(p[1]) + patch_size, int(p[0]):int(p[0]) + patch_size] for p in pts])
    pt = torch.tensor(patches, dtype=torch.float32).view(-1, 1, patch_size * patch_size)
    pt = pt / (pt.sum(dim=-1, keepdim=True) + 1e-6)
    pt = pt.view(-1, 1, patch_size, patch_size)
    pt[pt < 0] = 1e-6
    dxdy = spatial_soft_argmax2d(torch.log(pt))  # [N,1,2]
    pts[:, :2] = pts[:, :2] + dxdy.numpy().reshape(-1, 2) - r
    return pts.T


def homo_adapt_sample(img, n_views, rs, erosion_radius=0, **params):
    """datasets/Coco.py:258-292: n_views random homographies (the first one replaced by the identity),
    warped copies of `img` [H,W] and their valid masks.  Returns the keys the export loop reads."""
    H, W = img.shape
    hs = np.stack([np.linalg.inv(sample_homography(rs, **params)) for _ in range(n_views)])
    hs[0] = np.identity(3)
    homographies = torch.tensor(hs, dtype=torch.float32)
    inv_h = torch.stack([torch.inverse(homographies[i]) for i in range(n_views)])
    views = inv_warp_image_batch(img.view(1, 1, H, W).repeat(n_views, 1, 1, 1), inv_h, mode="bilinear")
    mask = compute_valid_mask((H, W), inv_h, erosion_radius=erosion_radius)
    return {"image": views, "valid_mask": mask.view(n_views, 1, H, W), "homographies": homographies,
            "inv_homographies": inv_h, "image_2D": img.view(1, H, W)}


def export_points(sd, sample, arch="SuperPointNet_gauss2", conf_thresh=0.015, nms_dist=4, top_k=600, subpixel=True,
                  border_remove=4, n_classes=133):
    """The per-image body of export_detector_homoAdapt_gpu (export.py:274-318).  BatchNorm runs in
    TRAIN mode over the n_views batch (SuperPointFrontend_torch.loadModel leaves `net.eval()`
    commented out, models/model_wrap.py:120) and so also moves the running statistics in `sd`."""
    with torch.no_grad():
        out = forward(sd, sample["image"], arch, train=True, n_classes=n_classes)
        heat = flatten_detection(out["semi"])
        agg = combine_heatmap(heat, sample["homographies"], sample["valid_mask"])  # [1,H,W]
    hm = agg.squeeze().numpy()
    pts = get_pts_from_heatmap(hm, conf_thresh, nms_dist, border_remove)
    if subpixel:
        pts = soft_argmax_points(hm, pts)
    pts = pts.transpose()
    if top_k and pts.shape[0] > top_k:
        pts = pts[:top_k]
    return {"heatmap": agg, "views_heatmap": heat, "pts": pts}


# --------------------------------------------------------------------------------------
# Logging branch of train_val_sample: Train_model_heatmap_all.py:447-568
# --------------------------------------------------------------------------------------
def heatmap_nms(heatmap, nms_dist=4, conf_thresh=0.015):
    """Train_model_heatmap_all.py:693-707 (+ utils/utils.py:581-609): 0/1 map of the NMS survivors."""
    hm = np.asarray(heatmap, dtype=np.float32).squeeze()
    pts = get_pts_from_heatmap(hm, np.float32(conf_thresh), nms_dist, border_remove=4)
    out = np.zeros_like(hm)
    out[pts[1].astype(int), pts[0].astype(int)] = 1
    return out


def precision_recall(pred, labels):
    """utils/utils.py:929-941 precisionRecall_torch."""
    tp = torch.sum(pred * labels)
    return {"precision": tp / (torch.sum(pred) + 1e-6), "recall": tp / (torch.sum(labels) + 1e-6)}


def batch_precision_recall(batch_pred, batch_labels):
    """Train_model_heatmap_all.py:614-622: per-image precision / recall, averaged over the batch."""
    prs = [precision_recall(batch_pred[i], batch_labels[i]) for i in range(batch_labels.shape[0])]
    return {"precision": float(np.mean([float(p["precision"]) for p in prs])),
            "recall": float(np.mean([float(p["recall"]) for p in prs]))}


# --------------------------------------------------------------------------------------
# Dense descriptor loss: utils/utils.py:779-893, selected by
# model.dense_loss.enable (Train_model_heatmap_all.py:131-137, call :348-350)
# --------------------------------------------------------------------------------------
def descriptor_loss_dense(desc, desc_w, homographies, mask_valid=None, cell_size=8, lamda_d=250.0, descriptor_dist=4.0):
    """desc, desc_w: [B,256,Hc,Wc]; homographies [B,3,3] (normalised, image -> warped); mask_valid [B,1,Hc,Wc] = the
    warped view's cell mask.  Returns (loss_desc, mask [B,Hc,Wc,Hc,Wc], pos_sum, neg_sum).
    Reference quirks kept: the shipped configs spell the weight `lambda_d`, which descriptor_loss swallows in **config,
    so lamda_d stays at its default 250; pos_sum / neg_sum (what the multi-task loss sees) ignore mask_valid, only
    loss_desc applies it; normalisation = B * (mask_valid.sum() + 1) * Hc * Wc ("bug in normalization", :884)."""
    B, D, Hc, Wc = desc.shape
    H, W = Hc * cell_size, Wc * cell_size
    with torch.no_grad():
        shape = torch.tensor([H, W], dtype=torch.float32)
        cy, cx = torch.meshgrid(torch.arange(Hc), torch.arange(Wc), indexing="ij")
        coor = torch.stack((cy, cx), dim=2).float() * cell_size + cell_size // 2  # [Hc,Wc,2] (y, x) cell centres
        flat = coor.view(-1, 2)
        nrm = flat / shape * 2 - 1                                               # normPts
        xy = torch.stack((nrm[:, 1], nrm[:, 0]), dim=1)
        pts = torch.cat((xy, torch.ones(xy.shape[0], 1)), dim=1)                 # warp_points (batched)
        w = torch.tensordot(homographies.float(), pts.t(), dims=([2], [0])).transpose(1, 2)  # [B,N,3]
        w = w[:, :, :2] / w[:, :, 2:]
        wyx = torch.stack((w[:, :, 1], w[:, :, 0]), dim=2)
        wyx = (wyx + 1) * shape / 2                                              # denormPts
        wyx = wyx.view(B, Hc, Wc, 1, 1, 2)
        dist = torch.norm(coor.view(1, 1, 1, Hc, Wc, 2) - wyx, dim=-1)
        mask = (dist <= descriptor_dist).float()
    a = desc.permute(0, 2, 3, 1).reshape(B, Hc, Wc, 1, 1, D)
    b = desc_w.permute(0, 2, 3, 1).reshape(B, 1, 1, Hc, Wc, D)
    dot = (a * b).sum(dim=-1)
    pos = torch.clamp(1.0 - dot, min=0.0)
    neg = torch.clamp(dot - 0.2, min=0.0)
    if mask_valid is None:
        mask_valid = torch.ones(B, 1, Hc, Wc)
    mv = mask_valid.view(B, 1, 1, Hc, Wc)
    normalization = B * (mv.sum() + 1) * Hc * Wc
    loss = ((lamda_d * mask * pos + (1 - mask) * neg) * mv).sum() / normalization
    pos_sum = (lamda_d * mask * pos / normalization).sum()
    neg_sum = ((1 - mask) * neg / normalization).sum()
    return loss, mask, pos_sum, neg_sum


# --------------------------------------------------------------------------------------
# Pair construction for real data: the remaining label products of
# datasets/data_tools.py:37-63 warpLabels(bilinear=True) and the semantic map of datasets/Coco_sem.py:406-450
# --------------------------------------------------------------------------------------
def warp_labels_full(pnts_xy, H, W, homography):
    """warpLabels(pnts, H, W, homography, bilinear=True): returns (labels [1,H,W], res [2,H,W] = warped point minus its
    rounded position, written at the rounded position, labels_bi [1,H,W] = the 4-neighbour bilinear splat of
    get_labels_bi, :26-34).  Scatters are last-write-wins (torch index_put), so results are only defined when no two
    points land on the same pixel."""
    Hpix = scale_homography(homography.float(), (H, W))
    wp_all = warp_points(pnts_xy.long().float(), Hpix)
    # get_labels_bi works on ALL warped points, filtering the 4 extrapolated neighbours afterwards
    pi = wp_all.long().float()
    ext = torch.cat((pi, torch.stack((pi[:, 0], pi[:, 1] + 1), 1), torch.stack((pi[:, 0] + 1, pi[:, 1]), 1), pi + 1), 0)
    rx, ry = (wp_all - pi)[:, 0], (wp_all - pi)[:, 1]
    wts = torch.cat(((1 - rx) * (1 - ry), (1 - rx) * ry, rx * (1 - ry), rx * ry), 0)
    ext, keep = filter_points(ext, (W, H))
    bi = torch.zeros(H, W)
    q = ext.round().long()
    bi[q[:, 1], q[:, 0]] = wts[keep]
    wp, _ = filter_points(wp_all, (W, H))
    lab = torch.zeros(H, W)
    q = wp.round().long()
    lab[q[:, 1], q[:, 0]] = 1
    res = torch.zeros(H, W, 2)
    res[q[:, 1], q[:, 0], :] = wp - wp.round()
    return lab.view(1, H, W), res.permute(2, 0, 1).contiguous(), bi.view(1, H, W)


def gaussian_label_u8(x):
    """datasets/Coco.py:378,400 `self.gaussian_blur(...)` = ImgAugTransform (utils/photometric.py:59-78) with GaussianBlur(sigma
    0.2): (x * 255).astype(np.uint8) -> blur -> astype(float32) / 255.  imgaug / cv2 are absent here: the blur is restated as
    the identity on 8-bit data (the 5-tap kernel of sigma 0.2 has off-centre weights exp(-12.5) = 3.7e-6 - below the 8-bit
    fixed-point resolution of the blur); the uint8 quantisation is exact numpy.  Parity unpinned for the blur itself."""
    a = (np.asarray(x, dtype=np.float32) * 255).astype(np.uint8)
    return torch.from_numpy(a.astype(np.float32) / 255)


def warp_semantic(sem, inv_homography, valid_mask, n_classes=133):
    """datasets/Coco_sem.py:406-450: bilinear warp of the class-id map as floats, invalid pixels -> n_classes."""
    w = inv_warp_image_batch(sem.float().view(1, 1, *sem.shape[-2:]), inv_homography.view(1, 3, 3)).view(sem.shape[-2:])
    w = w.clone()
    w[valid_mask.view(sem.shape[-2:]) == 0] = n_classes
    return w
