"""Pair construction for real data on the device: the `warped_pair` branch of
datasets/Coco.py:341-392 / datasets/Coco_sem.py:395-455 for a BATCH that is already resident in HBM.

  homographies      utils/homographies.py:12-141 with `warped_pair.params`, inverted (Coco.py:342-350): device RNG
  warped_img        inv_warp_image_batch, bilinear (utils/utils.py:347-385)
  warped_labels, warped_res, warped_labels_bi   warpLabels(bilinear=True) (datasets/data_tools.py:37-63)
  warped_valid_mask compute_valid_mask (nearest warp of ones + elliptical erosion, utils/utils.py:715-742)
  warped_sem        bilinear warp of the class ids, invalid -> 133 (Coco_sem.py:406-448)

  *_gaussian        ImgAugTransform(GaussianBlur sigma 0.2) of labels_2D / warped_labels_bi (Coco.py:378,400, utils/photometric.py:
                    59-78): float -> uint8 -> blur -> float / 255.  The quantisation floor(x * 255) / 255 is reproduced
                    (`ssp_op_label_quantize`); the blur itself is the identity on 8-bit data (off-centre weights of the 5-tap
                    sigma-0.2 kernel: exp(-12.5) = 3.7e-6) - imgaug / cv2 are absent from the image, so that last statement is
                    restated from the kernel formula, not checked against the binaries.

Not reproduced: the photometric augmentation of the IMAGES (imgaug).  The RNG streams differ from numpy / scipy, so the
homographies are a distribution-level equivalent (like synth.py's host generator)."""
import torch

from . import lib as L


def make_pairs(image, labels_2D, seed, warp_params=None, erosion_radius=3, semantic=None, n_classes=133):
    """image, labels_2D: device tensors [B,1,H,W] (keypoint map: non-zero = keypoint); semantic: int64 [B,H,W] or None.
    Returns the `sample` dict of Train_model_heatmap_all.py:212-251 (device tensors)."""
    if not image.is_cuda:
        raise RuntimeError("make_pairs needs HIP tensors: there is no CPU fallback")
    B, _, H, W = image.shape
    image = image.contiguous().float()
    labels_2D = labels_2D.contiguous().float()
    hs, inv = L.op_sample_homographies(B, seed, image.device, **(warp_params or {}))
    warped = L.op_warp_image(image, inv)
    wl, wres, wbi = L.op_warp_labels_full(labels_2D, hs)
    vm = L.op_erode(L.op_warp_image(torch.ones_like(image), inv, nearest=True), erosion_radius)
    s = {"image": image, "warped_img": warped, "labels_2D": labels_2D, "warped_labels": wl, "warped_res": wres,
         "warped_labels_bi": wbi, "labels_2D_gaussian": L.op_label_quantize(labels_2D),
         "warped_labels_gaussian": L.op_label_quantize(wbi),
         "valid_mask": torch.ones_like(image), "warped_valid_mask": vm, "homographies": hs, "inv_homographies": inv,
         # cell-space matrices in the reference's op order: the device sampler's matches then round like the reference's
         "cell_homographies": L.scaled_homographies(hs, H // 8, W // 8).to(image.device)}
    if semantic is not None:
        sw = L.op_warp_image(semantic.float().view(B, 1, H, W).contiguous(), inv)
        s["semantic"] = semantic
        s["warped_sem"] = L.op_sem_finalize(sw.view(B, H, W), vm.view(B, H, W), n_classes)
    return s
