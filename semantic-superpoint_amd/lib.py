"""ctypes binding of include/ssp_hip.h plus `Engine`, the owner of the torch-allocated device buffers.

PyTorch is plumbing here: it allocates HBM, provides the stream and (in the trainer) RCCL; every
arithmetic step runs in csrc/libssp_hip.so.  There is NO CPU / eager fallback: if the library cannot be
loaded or the tensors are not on a HIP device, calls raise.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import hipbuild as _build

ARCHS = {"SuperPointNet_gauss2": 0, "SuperPointNet_gauss2_ssmall": 1}
SCALAR_NAMES = ["loss", "loss_det", "loss_det_warp", "loss_desc", "loss_sem", "loss_sem_warp", "positive_dist",
                "negative_dist", "eta_det", "eta_desc", "eta_sem"]
N_SCALARS = 16
NREP = 32  # SSP_NREP
SAMPLER_MAX_MATCHES = 8192  # SAMPLER_MAX_CELLS of csrc/sem_kernels.hip.h (bitonic sort capacity of the device sampler)
PROF = {"none": 0, "conv3x3_fwd": 1, "conv3x3_dgrad": 2, "conv3x3_wgrad": 3, "conv_big_fwd": 4, "conv3x3_all": 5,
        "conv3x3_every": 6}
PROF_KERNELS = ["conv_wino4_kernel", "conv_wino_pipe_kernel", "conv_wino_p2_kernel", "wgrad_wino_kernel", "wgrad_wino4_kernel",
                "other", "conv_bf16_kernel", "wgrad_bf16_kernel"]  # SSP_PROF_K_*


class SspConfig(C.Structure):
    _fields_ = [("arch", C.c_int), ("n_classes", C.c_int), ("max_batch", C.c_int), ("height", C.c_int),
                ("width", C.c_int), ("n_match", C.c_int), ("n_non", C.c_int), ("dense_loss", C.c_int)]


class SspBuffers(C.Structure):
    _fields_ = [("params_dev", C.c_void_p), ("grads_dev", C.c_void_p), ("adam_m_dev", C.c_void_p),
                ("adam_v_dev", C.c_void_p), ("bn_running_dev", C.c_void_p), ("num_batches_tracked_dev", C.c_void_p),
                ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class SspPairInputs(C.Structure):
    _fields_ = [("batch", C.c_int), ("image_dev", C.c_void_p), ("warped_image_dev", C.c_void_p),
                ("labels_dev", C.c_void_p), ("warped_labels_dev", C.c_void_p), ("valid_mask_dev", C.c_void_p),
                ("warped_valid_mask_dev", C.c_void_p), ("homographies_dev", C.c_void_p), ("semantic_dev", C.c_void_p),
                ("warped_semantic_dev", C.c_void_p), ("match_a_dev", C.c_void_p), ("match_b_dev", C.c_void_p),
                ("nonmatch_b_dev", C.c_void_p), ("seed", C.c_uint64), ("lambda_loss", C.c_float),
                ("lamda_d", C.c_float), ("multi_task", C.c_int), ("train", C.c_int), ("dense_loss", C.c_int),
                ("dense_lamda_d", C.c_float), ("descriptor_dist", C.c_float), ("cell_homographies_dev", C.c_void_p),
                ("sparse_method", C.c_int), ("sparse_dist", C.c_int)]


class SspHomographyParams(C.Structure):
    _fields_ = [("perspective", C.c_int32), ("scaling", C.c_int32), ("rotation", C.c_int32), ("translation", C.c_int32),
                ("allow_artifacts", C.c_int32), ("n_scales", C.c_int32), ("n_angles", C.c_int32),
                ("scaling_amplitude", C.c_float), ("perspective_amplitude_x", C.c_float),
                ("perspective_amplitude_y", C.c_float), ("patch_ratio", C.c_float), ("max_angle", C.c_float),
                ("translation_overflow", C.c_float)]


class SspExportParams(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("conf_thresh", C.c_float),
                ("nms_dist", C.c_int32), ("border_remove", C.c_int32), ("top_k", C.c_int32), ("subpixel", C.c_int32)]


_lib = None

EXPORTS = ["ssp_last_error", "ssp_create", "ssp_destroy", "ssp_param_count", "ssp_bn_channel_count",
           "ssp_bn_layer_count", "ssp_workspace_bytes", "ssp_bind", "ssp_forward", "ssp_backward", "ssp_zero_grad",
           "ssp_pair_step", "ssp_adam_step", "ssp_sample_indices", "ssp_profile_enable", "ssp_profile_read", "ssp_profile_read_executed",
           "ssp_op_conv", "ssp_op_conv_wgrad", "ssp_op_labels", "ssp_op_sparse_loss", "ssp_op_bn_bwd", "ssp_op_bn_bwd_strided",
           "ssp_debug_buffer", "ssp_debug_conv_knobs", "ssp_set_conv_algo", "ssp_op_warp_image", "ssp_op_erode", "ssp_op_warp_labels",
           "ssp_export_workspace_bytes", "ssp_export_max_points", "ssp_export_points", "ssp_op_homoadapt_views",
           "ssp_op_flatten_detection", "ssp_op_combine_heatmap", "ssp_op_heatmap_points", "ssp_op_soft_argmax_points", "ssp_detector_heatmap", "ssp_op_heatmap_nms", "ssp_op_dense_loss",
           "ssp_op_sample_homographies", "ssp_op_warp_labels_full", "ssp_op_sem_finalize", "ssp_adam_step_scaled",
           "ssp_pair_step_phase", "ssp_grad_early_offset", "ssp_pair_step_graph", "ssp_handle_set_conv_algo",
           "ssp_op_detector_loss", "ssp_debug_occupancy", "ssp_sample_indices_cell", "ssp_op_warp_labels_px",
           "ssp_op_warp_labels_full_px", "ssp_profile_read_kernel", "ssp_op_label_quantize", "ssp_profile_pause", "ssp_op_conv_bf16", "ssp_op_conv_wgrad_bf16", "ssp_op_bn_bwd_bf16", "ssp_build_id",
           "ssp_set_deterministic", "ssp_get_deterministic", "ssp_clock_probe", "ssp_op_sem_loss"]


def load_library(path=None):
    """dlopen csrc/libssp_hip.so.  Without an explicit path (argument or SSP_HIP_LIB) the in-tree library is rebuilt
    first when it is missing or older than its sources (hipbuild.build_locked(): serialised across ranks by a file
    lock).  Raises if the library cannot be produced: there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("SSP_HIP_LIB")  # SSP_HIP_LIB: A/B builds of the kernels (tools/)
    if path is None:
        path = _build.build_locked()
    if not os.path.exists(path):
        raise RuntimeError("libssp_hip.so not found at %s" % path)
    if os.environ.get("SSP_SKIP_ISA_VERIFY") != "1" and not _build.verified(path):  # the binary about to be loaded (e.g. shipped to the GPU box) keeps the register contract
        try:
            _build.verify_and_stamp(path)
        except _build.MissingTool as e:
            raise RuntimeError("%s - cannot check the accumulation-register contract of %s; set SSP_SKIP_ISA_VERIFY=1 to load it "
                               "unchecked" % (e, path)) from e
        except OSError:  # read-only tree: verified, not stamped
            _build.verify_binary(path)
    lib = C.CDLL(path)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    lib.ssp_last_error.restype = C.c_char_p
    try:
        lib.ssp_build_id.restype = C.c_char_p
        lib.ssp_set_deterministic.argtypes = [C.c_int]
        lib.ssp_get_deterministic.restype = C.c_int
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    lib.ssp_create.argtypes = [C.POINTER(SspConfig), C.POINTER(vp)]
    lib.ssp_destroy.argtypes = [vp]
    lib.ssp_destroy.restype = None
    for n in ("ssp_param_count", "ssp_bn_channel_count", "ssp_workspace_bytes"):
        getattr(lib, n).argtypes = [vp]
        getattr(lib, n).restype = C.c_size_t
    lib.ssp_bn_layer_count.argtypes = [vp]
    lib.ssp_bind.argtypes = [vp, C.POINTER(SspBuffers), vp]
    lib.ssp_forward.argtypes = [vp, i, vp, i, i, i, i, vp, vp, vp, vp]
    lib.ssp_backward.argtypes = [vp, i, vp, vp, vp, vp]
    lib.ssp_zero_grad.argtypes = [vp, vp]
    lib.ssp_pair_step.argtypes = [vp, C.POINTER(SspPairInputs), vp, vp]
    lib.ssp_adam_step.argtypes = [vp, f, i, vp]
    lib.ssp_adam_step_scaled.argtypes = [vp, f, i, f, vp]
    lib.ssp_pair_step_phase.argtypes = [vp, C.POINTER(SspPairInputs), vp, i, vp]
    lib.ssp_pair_step_graph.argtypes = [vp, C.POINTER(SspPairInputs), vp, i, i, vp]
    lib.ssp_grad_early_offset.argtypes = [vp]
    lib.ssp_grad_early_offset.restype = C.c_size_t
    lib.ssp_handle_set_conv_algo.argtypes = [vp, i]
    lib.ssp_op_detector_loss.argtypes = [vp, i, vp, vp, i, i, i, vp, C.c_size_t, vp, vp, vp]
    try:  # (newer than an A/B library of an older revision, SSP_HIP_LIB)
        lib.ssp_op_sem_loss.argtypes = [vp, i, vp, i, i, i, i, i, vp, C.c_size_t, vp, vp, vp]
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    lib.ssp_sample_indices.argtypes = [vp, vp, i, C.c_uint64, vp, vp, vp, vp]
    lib.ssp_sample_indices_cell.argtypes = [vp, vp, i, C.c_uint64, vp, vp, vp, vp]
    lib.ssp_op_warp_labels_px.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.ssp_op_warp_labels_full_px.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
    lib.ssp_profile_enable.argtypes = [vp, i]
    lib.ssp_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    lib.ssp_profile_read_executed.argtypes = [vp, C.POINTER(C.c_double)]
    try:
        lib.ssp_profile_pause.argtypes = [vp, i]
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    try:
        lib.ssp_profile_read_kernel.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                                C.POINTER(C.c_double), C.POINTER(C.c_double)]
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    lib.ssp_op_conv.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp, i, vp, C.c_size_t, vp]
    lib.ssp_op_conv_wgrad.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp, C.c_size_t, vp]
    try:  # (the bf16 operators are newer than an A/B library of an older revision, SSP_HIP_LIB)
        lib.ssp_op_conv_bf16.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp, i, i, i, vp, vp, vp, C.c_size_t, vp]
        lib.ssp_op_conv_wgrad_bf16.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, vp, vp, i, vp, C.c_size_t, vp]
        lib.ssp_op_bn_bwd_bf16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    lib.ssp_debug_buffer.argtypes = [vp, i, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.ssp_debug_conv_knobs.argtypes = [i, i]
    lib.ssp_debug_occupancy.argtypes = [i]
    lib.ssp_set_conv_algo.argtypes = [i]
    lib.ssp_op_warp_image.argtypes = [vp, vp, vp, i, i, i, i, vp]
    lib.ssp_op_erode.argtypes = [vp, vp, i, i, i, i, vp]
    lib.ssp_op_warp_labels.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.ssp_op_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    lib.ssp_op_bn_bwd_strided.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    lib.ssp_op_labels.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.ssp_op_sparse_loss.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, C.c_float, C.c_float, vp, vp, vp, vp]
    ep = C.POINTER(SspExportParams)
    lib.ssp_export_workspace_bytes.argtypes = [ep]
    lib.ssp_export_workspace_bytes.restype = C.c_size_t
    lib.ssp_export_max_points.argtypes = [ep]
    lib.ssp_export_points.argtypes = [vp, ep, i] + [C.POINTER(vp)] * 7 + [vp]
    lib.ssp_op_homoadapt_views.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.ssp_op_flatten_detection.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.ssp_op_combine_heatmap.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.ssp_op_heatmap_points.argtypes = [vp, ep, vp, vp, vp, vp]
    lib.ssp_op_soft_argmax_points.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.ssp_detector_heatmap.argtypes = [vp, i, vp, vp]
    lib.ssp_op_heatmap_nms.argtypes = [vp, ep, i, vp, vp, vp, vp, vp]
    lib.ssp_op_sample_homographies.argtypes = [C.c_uint64, C.POINTER(SspHomographyParams), i, vp, vp, vp]
    lib.ssp_op_warp_labels_full.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
    lib.ssp_op_sem_finalize.argtypes = [vp, vp, vp, C.c_size_t, i, vp]
    try:  # (entry points newer than an A/B library of an older revision, SSP_HIP_LIB)
        lib.ssp_op_label_quantize.argtypes = [vp, vp, C.c_size_t, vp]
    except AttributeError:
        if os.environ.get("SSP_HIP_LIB") is None:
            raise
    lib.ssp_op_dense_loss.argtypes = [vp, vp, vp, vp, i, i, i, f, f, i, f, vp, C.c_size_t, vp, vp, vp, vp]
    _lib = lib
    return lib


def set_deterministic(on=True):
    """Bit-reproducible accumulation for Engines created AFTER the call (also SSP_DETERMINISTIC=1): the fp64 accumulators take
    addends rounded to a fixed quantum, the fp32 scatter targets (descriptor / segmentation gradients, bias and first-layer weight
    gradients) go through 64-bit fixed-point shadows.  Two runs from the same state then agree bit for bit; the default mode keeps
    the plain floating-point atomics (run-to-run differences ~1e-6)."""
    _check(load_library().ssp_set_deterministic(1 if on else 0))


def get_deterministic():
    return bool(load_library().ssp_get_deterministic())


def clock_probe(ms=5.0):
    """Shader clock (MHz) the current device sustains under fp32 matrix-core load for ~`ms` milliseconds (ssp_clock_probe; blocking).
    None for an A/B library of an older revision that lacks the entry point."""
    import torch
    lib = load_library()
    if not hasattr(lib, "ssp_clock_probe"):
        return None
    lib.ssp_clock_probe.argtypes = [C.c_float, C.POINTER(C.c_double), C.c_void_p]
    out = C.c_double(0.0)
    _check(lib.ssp_clock_probe(float(ms), C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return float(out.value)


def build_id():
    """Build id of the LOADED library (sha256 of its sources at build time); equals hipbuild.source_id() for the in-tree library."""
    return load_library().ssp_build_id().decode()


def set_conv_algo(algo):
    """Process-wide default (new Engines, handle-less operators): 0 = direct implicit GEMM, 1 = Winograd F(2x2,3x3)
    for the eligible 3x3 convolutions (default).  `Engine.set_conv_algo` changes one engine."""
    _check(load_library().ssp_set_conv_algo(int(algo)))


def _check(rc):
    if rc != 0:
        raise RuntimeError("libssp_hip: %s (code %d)" % (load_library().ssp_last_error().decode(), rc))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on a HIP device: the MI355X path has no CPU fallback" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def scaled_homographies(hn, height, width):
    """T^-1 @ H @ T with T = [[2/width, 0, -1], [0, 2/height, -1], [0, 0, 1]] for a batch of normalised homographies,
    computed on the HOST with the reference's own fp32 op sequence (`torch.inverse(trans) @ H @ trans`, per matrix:
    utils/utils.py:297-300 homography_scaling_torch, utils/homographies.py:270-276 scale_homography_torch), so that the
    matrix - and with it every integer index the device derives from it - is bit-identical to the reference's.
    (height, width) = (H, W) for pixel coordinates (warpLabels), (H/8, W/8) for cell coordinates (sparse-loss matches).
    hn: [B,3,3] on any device (a device tensor costs one small D2H copy); returns a CPU float32 tensor [B,3,3]."""
    h = hn.detach().to("cpu", torch.float32).reshape(-1, 3, 3)
    trans = torch.tensor([[2.0 / width, 0.0, -1.0], [0.0, 2.0 / height, -1.0], [0.0, 0.0, 1.0]])
    inv = torch.inverse(trans)
    return torch.stack([inv @ m @ trans for m in h]).contiguous()


# ------------------------------------------------------------------------------------------------
# parameter layout == reference state_dict layout
# ------------------------------------------------------------------------------------------------
_ENC = [("inc.conv.conv.0", "inc.conv.conv.1", 1, 64, 3), ("inc.conv.conv.3", "inc.conv.conv.4", 64, 64, 3),
        ("down1.mpconv.1.conv.0", "down1.mpconv.1.conv.1", 64, 64, 3),
        ("down1.mpconv.1.conv.3", "down1.mpconv.1.conv.4", 64, 64, 3),
        ("down2.mpconv.1.conv.0", "down2.mpconv.1.conv.1", 64, 128, 3),
        ("down2.mpconv.1.conv.3", "down2.mpconv.1.conv.4", 128, 128, 3),
        ("down3.mpconv.1.conv.0", "down3.mpconv.1.conv.1", 128, 128, 3),
        ("down3.mpconv.1.conv.3", "down3.mpconv.1.conv.4", 128, 128, 3)]
_HEADS = [("convPa", "bnPa", 128, 256, 3), ("convPb", "bnPb", 256, 65, 1), ("convDa", "bnDa", 128, 256, 3),
          ("convDb", "bnDb", 256, 256, 1)]


def layer_table(arch, n_classes=133):
    t = _ENC + _HEADS
    if arch == "SuperPointNet_gauss2_ssmall":
        t = t + [("convDS", "bnS1", 128, 256, 3), ("convSout", None, 256, n_classes, 1)]
    elif arch != "SuperPointNet_gauss2":
        raise KeyError(arch)
    return t


def param_layout(arch, n_classes=133):
    """[(state_dict key, shape, offset)] of the flat parameter vector, net.parameters() order."""
    out, off = [], 0
    for conv, bn, cin, cout, k in layer_table(arch, n_classes):
        for key, shape in ((conv + ".weight", (cout, cin, k, k)), (conv + ".bias", (cout,))):
            out.append((key, shape, off))
            off += int(np.prod(shape))
        if bn is not None:
            for key in (bn + ".weight", bn + ".bias"):
                out.append((key, (cout,), off))
                off += cout
    return out, off


def bn_layout(arch, n_classes=133):
    """[(bn key prefix, C, channel offset)] in layer order."""
    out, off = [], 0
    for conv, bn, cin, cout, k in layer_table(arch, n_classes):
        if bn is not None:
            out.append((bn, cout, off))
            off += cout
    return out, off


class Engine:
    """One libssp handle plus its torch-owned device buffers (one per GPU / stream)."""

    def __init__(self, arch, max_batch, height, width, device, n_classes=133, n_match=1000, n_non=100,
                 with_grad=True, dense_loss=False):
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Engine needs a HIP device (got %s): no CPU fallback exists" % self.device)
        self.arch, self.n_classes = arch, n_classes
        self.max_batch, self.height, self.width = max_batch, height, width
        self.n_match, self.n_non = n_match, n_non
        self.dense_loss = bool(dense_loss)
        cfg = SspConfig(ARCHS[arch], n_classes, max_batch, height, width, n_match, n_non, int(self.dense_loss))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.layout, self.n_params = param_layout(arch, n_classes)
        self.bns, self.n_bn_ch = bn_layout(arch, n_classes)
        assert self.n_params == self.lib.ssp_param_count(h), "parameter layout mismatch with the library"
        assert self.n_bn_ch == self.lib.ssp_bn_channel_count(h)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_params + 3, **f32)
        self.params[self.n_params:] = torch.tensor([1.0, 2.0, 1.0])  # MultiTaskLoss.eta init
        self.grads = torch.zeros(self.n_params + 3, **f32) if with_grad else None
        self.adam_m = torch.zeros(self.n_params + 3, **f32) if with_grad else None
        self.adam_v = torch.zeros(self.n_params + 3, **f32) if with_grad else None
        self.bn_running = torch.cat([torch.zeros(self.n_bn_ch, **f32), torch.ones(self.n_bn_ch, **f32)])
        self.nbt = torch.zeros(len(self.bns), dtype=torch.int64, device=self.device)
        self.ws_bytes = self.lib.ssp_workspace_bytes(h)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        self.scalars = torch.zeros(N_SCALARS, **f32)
        self.adam_t = 0
        self.bind()

    def bind(self):
        b = SspBuffers(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                       _ptr(self.bn_running), _ptr(self.nbt), _ptr(self.workspace), self.ws_bytes)
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_bind(self.h, C.byref(b), _stream()))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ssp_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- state ----
    def load_state_dict(self, sd):
        for key, shape, off in self.layout:
            v = torch.as_tensor(np.asarray(sd[key]) if not torch.is_tensor(sd[key]) else sd[key])
            self.params[off:off + v.numel()] = v.reshape(-1).to(self.device, torch.float32)
        for i, (bn, c, off) in enumerate(self.bns):
            for j, name in enumerate(("running_mean", "running_var")):
                k = bn + "." + name
                if k in sd:
                    v = torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k])
                    self.bn_running[j * self.n_bn_ch + off:j * self.n_bn_ch + off + c] = v.to(self.device, torch.float32)
            k = bn + ".num_batches_tracked"
            if k in sd:
                self.nbt[i] = int(sd[k])

    def state_dict(self):
        out = OrderedDict()
        bn_iter = {bn: (i, c, off) for i, (bn, c, off) in enumerate(self.bns)}
        for key, shape, off in self.layout:
            n = int(np.prod(shape))
            out[key] = self.params[off:off + n].view(shape).clone()
            prefix = key.rsplit(".", 1)[0]
            if key.endswith(".bias") and prefix in bn_iter:
                i, c, boff = bn_iter[prefix]
                out[prefix + ".running_mean"] = self.bn_running[boff:boff + c].clone()
                out[prefix + ".running_var"] = self.bn_running[self.n_bn_ch + boff:self.n_bn_ch + boff + c].clone()
                out[prefix + ".num_batches_tracked"] = self.nbt[i].clone()
        return out

    def grad_dict(self):
        out = OrderedDict()
        for key, shape, off in self.layout:
            out[key] = self.grads[off:off + int(np.prod(shape))].view(shape)
        out["eta"] = self.grads[self.n_params:]
        return out

    @property
    def eta(self):
        return self.params[self.n_params:]

    # ---- compute ----
    def forward(self, x, slot=0, train=True, want=("semi", "desc")):
        _need_gpu(x, "x")
        n, c, hh, ww = x.shape
        assert c == 1 and x.dtype == torch.float32
        f32 = dict(dtype=torch.float32, device=self.device)
        out = {}
        semi = torch.empty(n, 65, hh // 8, ww // 8, **f32) if "semi" in want else None
        desc = torch.empty(n, 256, hh // 8, ww // 8, **f32) if "desc" in want else None
        sem = torch.empty(n, self.n_classes, hh, ww, **f32) if "sem" in want else None
        if not hasattr(self, "_x"):
            self._x = [None, None]
        self._x[slot] = x  # the first-layer weight gradient re-reads the image in backward: keep it alive
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_forward(self.h, slot, _ptr(x), n, hh, ww, int(bool(train)), _ptr(semi), _ptr(desc),
                                        _ptr(sem), _stream()))
        if semi is not None:
            out["semi"] = semi
        if desc is not None:
            out["desc"] = desc
        if sem is not None:
            out["sem"] = sem
        return out

    def backward(self, slot, dsemi=None, ddesc=None, dsem=None):
        for t, nm in ((dsemi, "dsemi"), (ddesc, "ddesc"), (dsem, "dsem")):
            if t is not None:
                _need_gpu(t, nm)
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_backward(self.h, slot, _ptr(dsemi), _ptr(ddesc), _ptr(dsem), _stream()))

    def zero_grad(self):
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_zero_grad(self.h, _stream()))

    def adam_step(self, lr, grad_scale=None):
        """Fused Adam over the flat vector; grad_scale (e.g. 1 / world_size after an all-reduce SUM) scales the
        gradient inside the kernel without touching `grads`."""
        self.adam_t += 1
        with torch.cuda.device(self.device):
            if grad_scale is None:
                _check(self.lib.ssp_adam_step(self.h, float(lr), self.adam_t, _stream()))
            else:
                _check(self.lib.ssp_adam_step_scaled(self.h, float(lr), self.adam_t, float(grad_scale), _stream()))

    def set_conv_algo(self, algo):
        _check(self.lib.ssp_handle_set_conv_algo(self.h, int(algo)))

    @property
    def early_offset(self):
        """First element of the gradient bucket that is final after phase 1 of a split pair step."""
        return int(self.lib.ssp_grad_early_offset(self.h))

    def sample_indices(self, homographies, seed, cell_homographies=None):
        """Device sampler of the sparse-loss indices.  cell_homographies ([B,3,3], host or device): the reference's
        scale_homography_torch matrices (`scaled_homographies(H, Hc, Wc)`): the sampled matches are then exact
        correspondences of the reference (bit-identical rounding); without them T^-1 H T is derived on the device."""
        _need_gpu(homographies, "homographies")
        B = homographies.shape[0]
        i32 = dict(dtype=torch.int32, device=self.device)
        ma = torch.empty(B, self.n_match, **i32)
        mb = torch.empty(B, self.n_match, **i32)
        nm = torch.empty(B, self.n_match * self.n_non, **i32)
        with torch.cuda.device(self.device):
            if cell_homographies is not None:
                hc = cell_homographies.to(self.device, torch.float32).contiguous()
                assert tuple(hc.shape) == (B, 3, 3)
                self._keep_hc = hc
                _check(self.lib.ssp_sample_indices_cell(self.h, _ptr(hc), B, int(seed), _ptr(ma), _ptr(mb), _ptr(nm), _stream()))
            else:
                _check(self.lib.ssp_sample_indices(self.h, _ptr(homographies), B, int(seed), _ptr(ma), _ptr(mb), _ptr(nm),
                                                   _stream()))
        return ma, mb, nm

    def pair_step(self, sample, indices=None, seed=0, train=True, lambda_loss=1.0, lamda_d=1.0, multi_task=True,
                  gaussian=True, dense=None, phase=0, graph=False, sparse_method="2d", sparse_dist="cos"):
        """`sample`: dict of device tensors with the reference's keys (Train_model_heatmap_all.py:212-251).
        indices: (match_a, match_b, nonmatch_b) int32 device tensors or None (device sampler with `seed`).
        sparse_method / sparse_dist: model.sparse_loss.params.method / dist (sparse_loss.py:76-77): "2d" = bilinear grid_sample of the
        matches (every shipped config), anything else = index_select at the cell; "cos" = hinges on the dot product, anything else =
        the euclidean forms (pixelwise_contrastive_loss.py:140,185-210,247-258).
        dense: None (sparse descriptor loss) or the model.dense_loss.params dict (dense descriptor loss,
        utils/utils.py:779-893; keys lamda_d (default 250: the shipped `lambda_d` spelling is ignored by the reference
        too) and descriptor_dist (4)); needs an Engine created with dense_loss=True.
        phase: 0 whole step; 1 / 2 = the two halves of ssp_pair_step_phase (data-parallel overlap; call 2 with the
        same arguments).  graph=True replays the step as a hipGraph (ssp_pair_step_graph; needs a non-default
        current stream; the device sampler then fills persistent index buffers inside the graph).
        A sample WITHOUT "warped_img" is the single-view step of `data.warped_pair.enable: false`
        (Train_model_heatmap_all.py:207,237-262,330-332; configs/magicpoint_shapes_pair.yaml): one forward, detector
        (+ segmentation) loss of the image only; lambda_loss must be 0 (the reference asserts "need a pair of images").
        Returns the device tensor of SSP_N_SCALARS floats (no host sync)."""
        img = sample["image"]
        B, c1, H, W = img.shape
        single = sample.get("warped_img") is None
        if single and lambda_loss > 0:
            raise AssertionError("need a pair of images")  # Train_model_heatmap_all.py:343
        lab = sample["labels_2D_gaussian"] if gaussian else sample["labels_2D"]
        if single:
            imgw = labw = maskw = None
            req = [img, lab, sample["valid_mask"]]
        else:
            imgw, maskw = sample["warped_img"], sample["warped_valid_mask"]
            labw = sample["warped_labels_gaussian"] if gaussian else sample["warped_labels"]
            req = [img, imgw, lab, labw, sample["valid_mask"], maskw]
        for t in req:
            _need_gpu(t, "sample tensor")
            if t.dtype != torch.float32 or tuple(t.shape) != (B, 1, H, W):
                raise ValueError("pair-step image / label / mask tensors must be float32 [B,1,H,W] = %s, got %s %s"
                                 % ((B, 1, H, W), t.dtype, tuple(t.shape)))
        if (H, W) != (self.height, self.width) or B > self.max_batch:
            raise ValueError("pair step [%d,1,%d,%d] does not match the engine (%d x %dx%d)"
                             % (B, H, W, self.max_batch, self.height, self.width))
        Hm = None
        if not single:
            Hm = sample["homographies"].to(torch.float32)
            if not Hm.is_contiguous():
                Hm = Hm.contiguous()
            _need_gpu(Hm, "homographies")
            if tuple(Hm.shape) != (B, 3, 3):
                raise ValueError("homographies must be [B,3,3]")
        if dense is not None and not self.dense_loss:
            raise RuntimeError("create the Engine with dense_loss=True to use the dense descriptor loss")
        sample_in_graph = False
        if lambda_loss > 0 and indices is None and dense is None:
            if graph:  # persistent buffers: the captured sampler writes them
                if getattr(self, "_graph_idx", None) is None or self._graph_idx[0].shape[0] != B:
                    i32 = dict(dtype=torch.int32, device=self.device)
                    self._graph_idx = (torch.zeros(B, self.n_match, **i32), torch.zeros(B, self.n_match, **i32),
                                       torch.zeros(B, self.n_match * self.n_non, **i32))
                indices = self._graph_idx
                sample_in_graph = True
            elif phase != 2:
                indices = self._last_idx = self.sample_indices(Hm, seed, sample.get("cell_homographies"))
            else:
                indices = self._last_idx
        if indices is not None:
            ma, mb, nm = indices
            for t, shp in ((ma, (B, self.n_match)), (mb, (B, self.n_match)), (nm, (B, self.n_match * self.n_non))):
                _need_gpu(t, "sparse-loss indices")
                if t.dtype != torch.int32 or tuple(t.shape) != shp:
                    raise ValueError("sparse-loss indices must be int32 %s, got %s %s" % (shp, t.dtype, tuple(t.shape)))
        else:
            ma = mb = nm = None
        sem = semw = None
        if self.arch.endswith("ssmall"):
            sem, semw = sample.get("semantic"), (None if single else sample.get("warped_sem"))
            for t in ((sem,) if single else (sem, semw)):
                if t is None:
                    raise KeyError("the ssmall model needs sample['semantic'] and sample['warped_sem']")
                _need_gpu(t, "semantic labels")
                if t.dtype != torch.int64 or tuple(t.shape) != (B, H, W):
                    raise ValueError("semantic labels must be int64 [B,H,W] = %s, got %s %s"
                                     % ((B, H, W), t.dtype, tuple(t.shape)))
            if getattr(self, "check_label_range", False):  # one host sync: off by default (torch raises here too)
                for t in ((sem,) if single else (sem, semw)):
                    if int(t.min()) < 0 or int(t.max()) > self.n_classes:
                        raise ValueError("semantic label outside [0, %d]" % self.n_classes)
        inp = SspPairInputs(B, _ptr(img), _ptr(imgw), _ptr(lab), _ptr(labw), _ptr(sample["valid_mask"]),
                            _ptr(maskw), _ptr(Hm), _ptr(sem), _ptr(semw), _ptr(ma), _ptr(mb),
                            _ptr(nm), int(seed) & 0xFFFFFFFFFFFFFFFF, float(lambda_loss), float(lamda_d),
                            int(bool(multi_task)), int(bool(train)), int(dense is not None),
                            float((dense or {}).get("lamda_d", 250.0)), float((dense or {}).get("descriptor_dist", 4.0)),
                            None, int(sparse_method != "2d"), int(sparse_dist != "cos"))
        hcell = sample.get("cell_homographies")
        if hcell is not None:  # the reference's own cell-space matrices (scaled_homographies): exact match indices
            _need_gpu(hcell, "cell_homographies")
            if hcell.dtype != torch.float32 or tuple(hcell.shape) != (B, 3, 3):
                raise ValueError("cell_homographies must be float32 [B,3,3]")
            inp.cell_homographies_dev = hcell.data_ptr()
        self._keep = (req, Hm, indices, sem, semw, hcell)  # keep alive until the stream has consumed them
        with torch.cuda.device(self.device):
            if graph:
                _check(self.lib.ssp_pair_step_graph(self.h, C.byref(inp), _ptr(self.scalars), int(phase),
                                                    int(sample_in_graph), _stream()))
            elif phase == 0:
                _check(self.lib.ssp_pair_step(self.h, C.byref(inp), _ptr(self.scalars), _stream()))
            else:
                _check(self.lib.ssp_pair_step_phase(self.h, C.byref(inp), _ptr(self.scalars), int(phase), _stream()))
        return self.scalars

    def export_points(self, views, masks, unwarp_h, conf_thresh=0.015, nms_dist=4, top_k=600, subpixel=True,
                      border_remove=4, want_heatmap=False):
        """Homography-adaptation export of 1 or 2 images (export.py:296-309).  views/masks: lists of [n,1,H,W] (or
        [n,H,W]) device tensors -- one BatchNorm batch each; unwarp_h: list of [n,3,3] (sample["homographies"]).
        Returns a list of dicts {"pts": device [count,5] rows (x, y, conf, sx, sy), "count": device int32,
        "heatmap": [H,W] or None}; no host synchronisation happens here."""
        k = len(views)
        n, hh, ww = views[0].shape[0], views[0].shape[-2], views[0].shape[-1]
        p = SspExportParams(n, hh, ww, float(np.float32(conf_thresh)), int(nms_dist), int(border_remove),
                            int(top_k or 0), int(bool(subpixel)))
        for t in list(views) + list(masks) + list(unwarp_h):
            _need_gpu(t, "export tensor")
            assert t.dtype == torch.float32 and t.shape[0] == n
        wsb = self.lib.ssp_export_workspace_bytes(C.byref(p))
        cap = self.lib.ssp_export_max_points(C.byref(p))
        if wsb == 0 or cap < 0:
            _check(-1)
        key = (n, hh, ww, int(nms_dist), int(top_k or 0))
        if getattr(self, "_export_ws_key", None) != key:
            self._export_ws = [torch.empty(wsb, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self._export_ws_key = key
        f32 = dict(dtype=torch.float32, device=self.device)
        pts = [torch.empty(max(cap, 1), 5, **f32) for _ in range(k)]
        cnt = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(k)]
        hm = [torch.empty(hh, ww, **f32) if want_heatmap else None for _ in range(k)]
        arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() if t is not None else None for t in ts])  # noqa: E731
        self._keep_export = (views, masks, unwarp_h)
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_export_points(self.h, C.byref(p), k, arr(views), arr(masks), arr(unwarp_h),
                                              arr(self._export_ws[:k]), arr(hm), arr(pts), arr(cnt), _stream()))
        return [{"pts": pts[j], "count": cnt[j], "heatmap": hm[j]} for j in range(k)]

    def detector_heatmap(self, slot, n, hh, ww):
        """flattenDetection of the detector logits left in `slot` by the last forward / pair step -> [n,1,hh,ww]."""
        out = torch.empty(n, 1, hh, ww, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.ssp_detector_heatmap(self.h, slot, _ptr(out), _stream()))
        return out

    def debug_buffer(self, slot, name, shape, dtype=torch.float32):
        """Test hook: copy of an internal NHWC buffer as a torch tensor of `shape` (dtype bfloat16 for the activation /
        gradient tensors of the bf16 path)."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.ssp_debug_buffer(self.h, slot, name.encode(), C.byref(p), C.byref(n)))
        numel = int(np.prod(shape))
        es = 2 if dtype == torch.bfloat16 else 4
        base = self.workspace.data_ptr()
        off = p.value - base
        assert 0 <= off and off + numel * es <= self.ws_bytes
        return self.workspace[off:off + numel * es].view(dtype).view(*shape).clone()

    def profile_enable(self, family):
        _check(self.lib.ssp_profile_enable(self.h, PROF[family] if isinstance(family, str) else int(family)))

    def profile_pause(self, paused):
        if hasattr(self.lib, "ssp_profile_pause"):
            _check(self.lib.ssp_profile_pause(self.h, int(bool(paused))))

    def profile_read(self):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _check(self.lib.ssp_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        ex = C.c_double()
        _check(self.lib.ssp_profile_read_executed(self.h, C.byref(ex)))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value, "exec_flops": ex.value}

    def profile_read_kernels(self):
        """Per-kernel split of profile_read(): {kernel name: {ms, launches, flops, exec_flops, bytes}} (launched kernels only)."""
        out = {}
        if not hasattr(self.lib, "ssp_profile_read_kernel"):  # (an A/B library of an older revision, SSP_HIP_LIB)
            return out
        for k, name in enumerate(PROF_KERNELS):
            ms, n, fl, ex, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _check(self.lib.ssp_profile_read_kernel(self.h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(ex), C.byref(by)))
            if n.value > 0:
                out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "exec_flops": ex.value, "bytes": by.value}
        return out


# ---- optimizer state in torch.optim.Adam's wire format ----
def optimizer_state_dict(eng, lr):
    """state_dict() of the reference's optimizer, torch.optim.Adam(list(net.parameters()) + [eta], lr, betas=(0.9, 0.999))
    (Train_model_frontend_all.py:183-198), filled from the engine's flat m / v vectors."""
    state, idx = {}, 0
    step = torch.tensor(float(eng.adam_t))
    entries = [(shape, off) for _, shape, off in eng.layout] + [((3,), eng.n_params)]
    for shape, off in entries:
        n = int(np.prod(shape))
        if eng.adam_t > 0:
            state[idx] = {"step": step.clone(), "exp_avg": eng.adam_m[off:off + n].view(shape).detach().cpu().clone(),
                          "exp_avg_sq": eng.adam_v[off:off + n].view(shape).detach().cpu().clone()}
        idx += 1
    group = {"lr": float(lr), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
             "params": list(range(idx))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state(eng, osd, eta=None):
    """Inverse of optimizer_state_dict (also accepts round 1's {"adam_m", "adam_v", "step"} form); eta: 3 floats."""
    if osd is not None and "state" in osd:
        entries = [(shape, off) for _, shape, off in eng.layout] + [((3,), eng.n_params)]
        steps = []
        for idx, (shape, off) in enumerate(entries):
            st = osd["state"].get(idx)
            if st is None:
                continue
            n = int(np.prod(shape))
            eng.adam_m[off:off + n] = torch.as_tensor(st["exp_avg"]).reshape(-1).to(eng.device, torch.float32)
            eng.adam_v[off:off + n] = torch.as_tensor(st["exp_avg_sq"]).reshape(-1).to(eng.device, torch.float32)
            steps.append(int(float(st["step"])))
        if steps:
            if len(set(steps)) != 1:
                raise ValueError("per-parameter Adam steps differ: the fused kernel keeps one step count")
            eng.adam_t = steps[0]
    elif osd is not None and "adam_m" in osd:
        eng.adam_m.copy_(torch.as_tensor(osd["adam_m"]).to(eng.device))
        eng.adam_v.copy_(torch.as_tensor(osd["adam_v"]).to(eng.device))
        eng.adam_t = int(osd.get("step", 0))
    if eta is not None:
        eng.params[eng.n_params:] = torch.as_tensor(eta).reshape(3).to(eng.device, torch.float32)


# ---- operator-level wrappers (tests) ----
def op_conv(x_nhwc, w_oihw, bias, ksize, in_mode=0, in_scale=None, in_shift=None, stats=None, transpose_flip=False,
            out_hw=None):
    lib = load_library()
    _need_gpu(x_nhwc, "x")
    N, Hin, Win, cin = x_nhwc.shape
    H, W = (Hin // 2, Win // 2) if in_mode == 2 else (Hin, Win)
    cout = w_oihw.shape[1] if transpose_flip else w_oihw.shape[0]
    out = torch.empty(N, H, W, cout, dtype=torch.float32, device=x_nhwc.device)
    # packed weight image of the kernel family (3x3: Winograd components; 1x1: 32 x 32 operand tiles + the work-queue counters)
    ws = torch.empty(max(((cin + 15) // 16) * ((cout + 63) // 64) * (36 if ksize == 3 else 1) * 16 * 64 * 4,
                         ((cin + 31) // 32) * ((cout + 31) // 32) * 4096 + 512) + 1024, dtype=torch.uint8, device=x_nhwc.device)
    with torch.cuda.device(x_nhwc.device):
        _check(lib.ssp_op_conv(_ptr(x_nhwc), _ptr(w_oihw), _ptr(bias), _ptr(out), N, H, W, cin, cout, ksize, in_mode,
                               _ptr(in_scale), _ptr(in_shift), _ptr(stats), int(transpose_flip), _ptr(ws), ws.numel(),
                               _stream()))
    return out


def op_conv_bf16(x_nhwc, w_oihw, bias, ksize, in_mode=0, in_scale=None, in_shift=None, stats=None, transpose_flip=False,
                 out_f32=False, pool_gamma=None):
    """conv_bf16_kernel as an operator: x bf16 (or fp32) NHWC, fp32 OIHW weights -> bf16 (or fp32) NHWC output
    (+ the raw pooled copy when pool_gamma is given)."""
    lib = load_library()
    _need_gpu(x_nhwc, "x")
    N, H, W, cin = x_nhwc.shape
    cout = w_oihw.shape[1] if transpose_flip else w_oihw.shape[0]
    in_f32 = x_nhwc.dtype == torch.float32
    if not in_f32 and x_nhwc.dtype != torch.bfloat16:
        raise RuntimeError("x must be bfloat16 or float32")
    out = torch.empty(N, H, W, cout, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x_nhwc.device)
    pool = torch.empty(N, H // 2, W // 2, cout, dtype=torch.bfloat16, device=x_nhwc.device) if pool_gamma is not None else None
    ws = torch.empty(((cout + 63) // 64) * ((cin + 31) // 32) * ksize * ksize * 4096, dtype=torch.uint8, device=x_nhwc.device)
    with torch.cuda.device(x_nhwc.device):
        _check(lib.ssp_op_conv_bf16(_ptr(x_nhwc), _ptr(w_oihw), _ptr(bias), _ptr(out), N, H, W, cin, cout, ksize, in_mode,
                                    _ptr(in_scale), _ptr(in_shift), _ptr(stats), int(transpose_flip), int(in_f32), int(out_f32),
                                    _ptr(pool), _ptr(pool_gamma), _ptr(ws), ws.numel(), _stream()))
    return (out, pool) if pool_gamma is not None else out


def op_conv_wgrad_bf16(x_nhwc, dy_nhwc, ksize, in_mode=0, in_scale=None, in_shift=None, dw=None):
    """wgrad_bf16_kernel as an operator: x bf16 NHWC, dY bf16 (or fp32) NHWC -> fp32 OIHW gradient (accumulated into dw)."""
    lib = load_library()
    _need_gpu(x_nhwc, "x")
    N, H, W, cin = x_nhwc.shape
    cout = dy_nhwc.shape[3]
    if x_nhwc.dtype != torch.bfloat16:
        raise RuntimeError("x must be bfloat16")
    dy_f32 = dy_nhwc.dtype == torch.float32
    if dw is None:
        dw = torch.zeros(cout, cin, ksize, ksize, dtype=torch.float32, device=x_nhwc.device)
    ws = torch.empty(512 * ksize * ksize * 4096 * 4, dtype=torch.uint8, device=x_nhwc.device)
    with torch.cuda.device(x_nhwc.device):
        _check(lib.ssp_op_conv_wgrad_bf16(_ptr(x_nhwc), _ptr(dy_nhwc), _ptr(dw), N, H, W, cin, cout, ksize, in_mode,
                                          _ptr(in_scale), _ptr(in_shift), int(dy_f32), _ptr(ws), ws.numel(), _stream()))
    return dw


def op_conv_wgrad(x_nhwc, dout_nhwc, ksize, in_mode=0, in_scale=None, in_shift=None):
    lib = load_library()
    N, Hin, Win, cin = x_nhwc.shape
    _, H, W, cout = dout_nhwc.shape
    dw = torch.zeros(cout, cin, ksize, ksize, dtype=torch.float32, device=x_nhwc.device)
    # partial slabs: 512 x taps x [64][64] (3x3 / generic 1x1) or 512 x [256][128] (grouped pointwise kernel, 256 input channels)
    ws = torch.empty(512 * max(ksize * ksize * 4096, 32768 if ksize == 1 else 0) * 4, dtype=torch.uint8, device=x_nhwc.device)
    with torch.cuda.device(x_nhwc.device):
        _check(lib.ssp_op_conv_wgrad(_ptr(x_nhwc), _ptr(dout_nhwc), _ptr(dw), N, H, W, cin, cout, ksize, in_mode,
                                     _ptr(in_scale), _ptr(in_shift), _ptr(ws), ws.numel(), _stream()))
    return dw


def op_labels(labels2d=None, mask2d=None):
    """labels2Dto3D(add_dustbin=True) and getMasks on the device; returns (target [B,65,Hc,Wc], cellmask [B,Hc,Wc])."""
    lib = load_library()
    ref = labels2d if labels2d is not None else mask2d
    _need_gpu(ref, "labels/mask")
    B, _, H, W = ref.shape
    f32 = dict(dtype=torch.float32, device=ref.device)
    tgt = torch.empty(B, 65, H // 8, W // 8, **f32) if labels2d is not None else None
    cm = torch.empty(B, H // 8, W // 8, **f32) if mask2d is not None else None
    with torch.cuda.device(ref.device):
        _check(lib.ssp_op_labels(_ptr(labels2d), _ptr(mask2d), _ptr(tgt), _ptr(cm), B, H, W, _stream()))
    return tgt, cm


def op_sem_loss(sout_nchw, labels, grad=True, algo=0, cs=None):
    """sem_loss of public NCHW logits [B,C,Hc,Wc] at 1/8 resolution against int64 labels [B,8Hc,8Wc] (C = ignore index): the fused
    bilinear upsample + cross entropy of the training step; returns (loss, d loss / d logits NCHW or None).  algo: see ssp_op_sem_loss."""
    lib = load_library()
    _need_gpu(sout_nchw, "sout")
    B, c, Hc, Wc = sout_nchw.shape
    cs = cs or (c + 7) // 8 * 8
    dev = sout_nchw.device
    x = torch.zeros(B, Hc, Wc, cs, dtype=torch.float32, device=dev)
    x[..., :c] = sout_nchw.permute(0, 2, 3, 1)
    d = torch.full_like(x, float("nan")) if grad else None   # (the operator overwrites it)
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    scratch = torch.empty(65536, dtype=torch.uint8, device=dev)
    lab = labels.to(device=dev, dtype=torch.int64).contiguous()
    assert tuple(lab.shape) == (B, Hc * 8, Wc * 8)
    with torch.cuda.device(dev):
        _check(lib.ssp_op_sem_loss(_ptr(x), cs, _ptr(lab), B, Hc * 8, Wc * 8, c, algo, _ptr(scratch), scratch.numel(), _ptr(out), _ptr(d),
                                   _stream()))
    return float(out.item()), (d[..., :c].permute(0, 3, 1, 2).contiguous() if grad else None)


def op_detector_loss(semi_nchw, labels2d, mask2d, grad=True):
    """detector_loss (softmax + BCE) of public NCHW logits [B,65,Hc,Wc]; returns (loss, d loss / d semi NCHW or None)."""
    lib = load_library()
    _need_gpu(semi_nchw, "semi")
    B, c, Hc, Wc = semi_nchw.shape
    assert c == 65
    dev = semi_nchw.device
    x = torch.zeros(B, Hc, Wc, 80, dtype=torch.float32, device=dev)
    x[..., :65] = semi_nchw.permute(0, 2, 3, 1)
    d = torch.empty_like(x) if grad else None
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    scratch = torch.empty(65536 + 4 * (B * Hc * Wc + 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.ssp_op_detector_loss(_ptr(x), 80, _ptr(labels2d.contiguous()), _ptr(mask2d.contiguous()), B, Hc * 8, Wc * 8,
                                        _ptr(scratch), scratch.numel(), _ptr(out), _ptr(d), _stream()))
    return float(out.item()), (d[..., :65].permute(0, 3, 1, 2).contiguous() if grad else None)


def op_sparse_loss(desc_a_nchw, desc_b_nchw, match_a, match_b, nonmatch_b, method="2d", dist="cos", grad=None):
    """(positive_dist, negative_dist) of the sparse descriptor loss for NCHW descriptor maps and explicit indices; with
    grad = (coef_pos, coef_neg) also the gradients of coef_pos * positive_dist + coef_neg * negative_dist wrt both maps (NCHW).
    method / dist: sparse_loss.params ("2d" / "cos" in every shipped config; see Engine.pair_step)."""
    lib = load_library()
    _need_gpu(desc_a_nchw, "desc")
    B, D, Hc, Wc = desc_a_nchw.shape
    a = desc_a_nchw.permute(0, 2, 3, 1).contiguous()
    b = desc_b_nchw.permute(0, 2, 3, 1).contiguous()
    out = torch.zeros(2, dtype=torch.float32, device=a.device)
    da = torch.full_like(a, float("nan")) if grad is not None else None   # (the operator overwrites them)
    db = torch.full_like(b, float("nan")) if grad is not None else None
    cp, cn = grad if grad is not None else (0.0, 0.0)
    with torch.cuda.device(a.device):
        _check(lib.ssp_op_sparse_loss(_ptr(a), _ptr(b), _ptr(match_a), _ptr(match_b), _ptr(nonmatch_b), B, Hc, Wc,
                                      match_a.shape[1], nonmatch_b.shape[1] // match_a.shape[1], int(method != "2d"), int(dist != "cos"),
                                      float(cp), float(cn), _ptr(da), _ptr(db), _ptr(out), _stream()))
    torch.cuda.synchronize()
    if grad is None:
        return float(out[0]), float(out[1])
    return float(out[0]), float(out[1]), da.permute(0, 3, 1, 2).contiguous(), db.permute(0, 3, 1, 2).contiguous()


def op_dense_loss(desc_a_nchw, desc_b_nchw, homographies, mask_valid, lamda_d=250.0, descriptor_dist=4.0, grad=None):
    """Dense descriptor loss (utils/utils.py:779-893) of NCHW descriptor maps.  Returns (loss, pos_sum, neg_sum) and,
    with grad = ("loss", scale) or ("multi_task", scale), the gradients of scale * loss_desc resp.
    scale * (pos_sum + neg_sum) wrt both maps (NCHW)."""
    lib = load_library()
    _need_gpu(desc_a_nchw, "desc")
    B, D, Hc, Wc = desc_a_nchw.shape
    dev = desc_a_nchw.device
    a = desc_a_nchw.permute(0, 2, 3, 1).contiguous()
    b = desc_b_nchw.permute(0, 2, 3, 1).contiguous()
    hm = homographies.to(dev, torch.float32).contiguous()
    mv = mask_valid.to(dev, torch.float32).reshape(B, Hc * Wc).contiguous()
    out = torch.zeros(3, dtype=torch.float32, device=dev)
    da = torch.empty_like(a) if grad else None
    db = torch.empty_like(b) if grad else None
    scratch = torch.empty(65536 + (B * (Hc * Wc) ** 2 * 4 if grad else 0), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.ssp_op_dense_loss(_ptr(a), _ptr(b), _ptr(hm), _ptr(mv), B, Hc, Wc, float(lamda_d), float(descriptor_dist),
                                     int(bool(grad and grad[0] == "multi_task")), float(grad[1]) if grad else 0.0,
                                     _ptr(scratch), scratch.numel(), _ptr(out), _ptr(da), _ptr(db), _stream()))
    o = out.cpu().tolist()
    if not grad:
        return o[0], o[1], o[2]
    return (o[0], o[1], o[2]), da.permute(0, 3, 1, 2).contiguous(), db.permute(0, 3, 1, 2).contiguous()


def op_bn_bwd(y_nhwc, dout_nhwc, gamma, scale, shift, mean, invstd, relu=True, pool=False):
    """Backward of BatchNorm2d(train)+ReLU(+MaxPool2d(2)); returns (dy, dgamma, dbeta, dbias).  The channel count is
    len(gamma); the NHWC tensors may be channel-padded (pixel stride = their last dimension, a multiple of 4)."""
    lib = load_library()
    _need_gpu(y_nhwc, "y")
    N, H, W, cs = y_nhwc.shape
    Cc = gamma.numel()
    dev = y_nhwc.device
    stats4 = torch.cat([scale, shift, mean, invstd]).contiguous()
    dy = torch.zeros_like(y_nhwc) if cs != Cc else torch.empty_like(y_nhwc)
    dg, db, dbias = (torch.zeros(Cc, dtype=torch.float32, device=dev) for _ in range(3))
    sums = torch.zeros(NREP * 2 * Cc, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _check(lib.ssp_op_bn_bwd_strided(_ptr(y_nhwc), _ptr(dout_nhwc), _ptr(gamma), _ptr(stats4), _ptr(dy), _ptr(dg),
                                         _ptr(db), _ptr(dbias), _ptr(sums), N, H, W, Cc, cs, int(relu), int(pool), _stream()))
    return dy, dg, db, dbias


def op_bn_bwd_bf16(y_nhwc, dout_nhwc, gamma, scale, shift, mean, invstd, pool=False):
    """op_bn_bwd on bf16 tensors (BatchNorm + ReLU (+ MaxPool2d(2)) backward of the bf16 path); dy is bf16."""
    lib = load_library()
    _need_gpu(y_nhwc, "y")
    if y_nhwc.dtype != torch.bfloat16 or dout_nhwc.dtype != torch.bfloat16:
        raise RuntimeError("y and dout must be bfloat16")
    N, H, W, cs = y_nhwc.shape
    Cc = gamma.numel()
    dev = y_nhwc.device
    stats4 = torch.cat([scale, shift, mean, invstd]).contiguous()
    dy = torch.zeros_like(y_nhwc)
    dg, db, dbias = (torch.zeros(Cc, dtype=torch.float32, device=dev) for _ in range(3))
    sums = torch.zeros(NREP * 2 * Cc, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _check(lib.ssp_op_bn_bwd_bf16(_ptr(y_nhwc), _ptr(dout_nhwc), _ptr(gamma), _ptr(stats4), _ptr(dy), _ptr(dg), _ptr(db),
                                      _ptr(dbias), _ptr(sums), N, H, W, Cc, cs, 1, int(pool), _stream()))
    return dy, dg, db, dbias


def op_warp_image(img, inv_h, nearest=False):
    """inv_warp_image_batch on the device: img [B,1,H,W], inv_h [B,3,3] -> warped [B,1,H,W]."""
    lib = load_library()
    _need_gpu(img, "img")
    inv_h = inv_h.to(img.device, torch.float32).contiguous()
    B, _, H, W = img.shape
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _check(lib.ssp_op_warp_image(_ptr(img), _ptr(inv_h), _ptr(out), B, H, W, int(bool(nearest)), _stream()))
    return out


def op_erode(mask, radius):
    lib = load_library()
    _need_gpu(mask, "mask")
    B, _, H, W = mask.shape
    out = torch.empty_like(mask)
    with torch.cuda.device(mask.device):
        _check(lib.ssp_op_erode(_ptr(mask), _ptr(out), B, H, W, int(radius), _stream()))
    return out


def op_warp_labels(labels, hn, exact=True):
    """warpLabels on a keypoint map.  exact=True (default): the pixel-space homography is computed on the host like the
    reference (`scaled_homographies`) -> bit-identical indices; exact=False: analytic T^-1 H T on the device (no D2H copy)."""
    lib = load_library()
    _need_gpu(labels, "labels")
    B, _, H, W = labels.shape
    out = torch.empty_like(labels)
    with torch.cuda.device(labels.device):
        if exact:
            hpx = scaled_homographies(hn, H, W).to(labels.device)
            _check(lib.ssp_op_warp_labels_px(_ptr(labels), _ptr(hpx), _ptr(out), B, H, W, _stream()))
        else:
            hn = hn.to(labels.device, torch.float32).contiguous()
            _check(lib.ssp_op_warp_labels(_ptr(labels), _ptr(hn), _ptr(out), B, H, W, _stream()))
    return out


# ---- homography-adaptation export operators ----
def op_homoadapt_views(img, inv_h):
    """datasets/Coco.py:279-288 on the device: img [H,W], inv_h [n,3,3] -> (views [n,1,H,W], masks [n,1,H,W])."""
    lib = load_library()
    _need_gpu(img, "img")
    inv_h = inv_h.to(img.device, torch.float32).contiguous()
    n, (H, W) = inv_h.shape[0], img.shape[-2:]
    views = torch.empty(n, 1, H, W, dtype=torch.float32, device=img.device)
    masks = torch.empty_like(views)
    with torch.cuda.device(img.device):
        _check(lib.ssp_op_homoadapt_views(_ptr(img), _ptr(inv_h), _ptr(views), _ptr(masks), n, H, W, _stream()))
    return views, masks


def op_flatten_detection(semi, mask=None):
    """flattenDetection (utils/utils.py:515-560) of semi [n,65,Hc,Wc] -> [n,1,8Hc,8Wc], times mask if given."""
    lib = load_library()
    _need_gpu(semi, "semi")
    n, c, Hc, Wc = semi.shape
    assert c == 65
    out = torch.empty(n, 1, Hc * 8, Wc * 8, dtype=torch.float32, device=semi.device)
    if mask is not None:
        _need_gpu(mask, "mask")
    with torch.cuda.device(semi.device):
        _check(lib.ssp_op_flatten_detection(_ptr(semi), _ptr(mask), _ptr(out), n, Hc, Wc, _stream()))
    return out


def op_combine_heatmap(heat, mask, unwarp_h):
    """combine_heatmap (export.py:49-60); heat must already be heatmap*mask.  -> [H,W]"""
    lib = load_library()
    _need_gpu(heat, "heat")
    _need_gpu(mask, "mask")
    unwarp_h = unwarp_h.to(heat.device, torch.float32).contiguous()
    n, (H, W) = heat.shape[0], heat.shape[-2:]
    out = torch.empty(H, W, dtype=torch.float32, device=heat.device)
    with torch.cuda.device(heat.device):
        _check(lib.ssp_op_combine_heatmap(_ptr(heat), _ptr(mask), _ptr(unwarp_h), _ptr(out), n, H, W, _stream()))
    return out


def op_heatmap_points(heatmap, conf_thresh, nms_dist=4, border_remove=4, top_k=0, subpixel=False):
    """getPtsFromHeatmap (+ soft_argmax_points, top-k) on the device.  Returns a float64 numpy [N,3] array of
    (x, y, conf), assembled exactly like models/model_wrap.py:245 (x + sx - 2 in float64)."""
    lib = load_library()
    _need_gpu(heatmap, "heatmap")
    H, W = heatmap.shape
    p = SspExportParams(1, H, W, float(np.float32(conf_thresh)), int(nms_dist), int(border_remove), int(top_k or 0),
                        int(bool(subpixel)))
    wsb, cap = lib.ssp_export_workspace_bytes(C.byref(p)), lib.ssp_export_max_points(C.byref(p))
    if wsb == 0 or cap < 0:
        _check(-1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=heatmap.device)
    pts = torch.empty(max(cap, 1), 5, dtype=torch.float32, device=heatmap.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=heatmap.device)
    with torch.cuda.device(heatmap.device):
        _check(lib.ssp_op_heatmap_points(_ptr(heatmap), C.byref(p), _ptr(ws), _ptr(pts), _ptr(cnt), _stream()))
    return points_to_numpy(pts, cnt, subpixel)


def op_heatmap_nms(heat, labels=None, conf_thresh=0.015, nms_dist=4, border_remove=4, want_map=True):
    """heatmap_to_nms + the per-image terms of batch_precision_recall on the device.  heat, labels: [B,1,H,W] (or
    [B,H,W]).  Returns (nms_map [B,H,W] or None, pr [B,2] = (precision, recall) or None) as device tensors."""
    lib = load_library()
    _need_gpu(heat, "heat")
    B, (H, W) = heat.shape[0], heat.shape[-2:]
    p = SspExportParams(1, H, W, float(np.float32(conf_thresh)), int(nms_dist), int(border_remove), 0, 0)
    wsb = lib.ssp_export_workspace_bytes(C.byref(p))
    if wsb == 0:
        _check(-1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=heat.device)
    if labels is not None:
        _need_gpu(labels, "labels")
        assert labels.numel() == heat.numel() and labels.dtype == torch.float32
    nms = torch.empty(B, H, W, dtype=torch.float32, device=heat.device) if want_map else None
    pr = torch.empty(B, 2, dtype=torch.float32, device=heat.device) if labels is not None else None
    with torch.cuda.device(heat.device):
        _check(lib.ssp_op_heatmap_nms(_ptr(heat), C.byref(p), B, _ptr(ws), _ptr(labels), _ptr(nms), _ptr(pr), _stream()))
    return nms, pr


def op_soft_argmax_points(heatmap, xy):
    """(sx, sy) in [0,4] of the 5x5 soft-argmax around each (x, y) of xy [n,2] (device float32) -> device [n,2]."""
    lib = load_library()
    _need_gpu(heatmap, "heatmap")
    _need_gpu(xy, "xy")
    H, W = heatmap.shape
    out = torch.empty(xy.shape[0], 2, dtype=torch.float32, device=heatmap.device)
    with torch.cuda.device(heatmap.device):
        _check(lib.ssp_op_soft_argmax_points(_ptr(heatmap), _ptr(xy), _ptr(out), xy.shape[0], H, W, _stream()))
    return out


def points_to_numpy(pts, count, subpixel):
    """Device rows (x, y, conf, sx, sy) -> the reference's float64 [N,3] `pts` (one host synchronisation)."""
    n = int(count.item())
    a = pts[:n].cpu().numpy()
    out = np.zeros((n, 3), dtype=np.float64)
    out[:, 0], out[:, 1], out[:, 2] = a[:, 0], a[:, 1], a[:, 2]
    if subpixel:
        out[:, :2] = out[:, :2] + a[:, 3:5] - 2
    return out


# ---- pair construction for real data ----
def op_sample_homographies(B, seed, device, perspective=True, scaling=True, rotation=True, translation=True, n_scales=5,
                           n_angles=25, scaling_amplitude=0.1, perspective_amplitude_x=0.1, perspective_amplitude_y=0.1,
                           patch_ratio=0.5, max_angle=np.pi / 2, allow_artifacts=False, translation_overflow=0.0):
    """B homographies with the keyword names / defaults of utils/homographies.py:sample_homography_np, drawn on the
    device.  Returns (homographies, inv_homographies) [B,3,3] as the dataset stores them (Coco.py:342-350)."""
    lib = load_library()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("op_sample_homographies needs a HIP device")
    p = SspHomographyParams(int(perspective), int(scaling), int(rotation), int(translation), int(allow_artifacts),
                            int(n_scales), int(n_angles), float(scaling_amplitude), float(perspective_amplitude_x),
                            float(perspective_amplitude_y), float(patch_ratio), float(max_angle), float(translation_overflow))
    h = torch.empty(B, 3, 3, dtype=torch.float32, device=device)
    inv = torch.empty_like(h)
    with torch.cuda.device(device):
        _check(lib.ssp_op_sample_homographies(int(seed), C.byref(p), B, _ptr(h), _ptr(inv), _stream()))
    return h, inv


def op_warp_labels_full(labels, hn, exact=True):
    """warpLabels(bilinear=True) on a keypoint map [B,1,H,W]: (labels [B,1,H,W], res [B,2,H,W], labels_bi [B,1,H,W]).
    exact: see op_warp_labels."""
    lib = load_library()
    _need_gpu(labels, "labels")
    B, _, H, W = labels.shape
    lab = torch.empty_like(labels)
    res = torch.empty(B, 2, H, W, dtype=torch.float32, device=labels.device)
    bi = torch.empty_like(labels)
    with torch.cuda.device(labels.device):
        if exact:
            hpx = scaled_homographies(hn, H, W).to(labels.device)
            _check(lib.ssp_op_warp_labels_full_px(_ptr(labels), _ptr(hpx), _ptr(lab), _ptr(res), _ptr(bi), B, H, W, _stream()))
        else:
            hn = hn.to(labels.device, torch.float32).contiguous()
            _check(lib.ssp_op_warp_labels_full(_ptr(labels), _ptr(hn), _ptr(lab), _ptr(res), _ptr(bi), B, H, W, _stream()))
    return lab, res, bi


def op_label_quantize(labels):
    """The reference's `*_gaussian` label maps (datasets/Coco.py:378,400): uint8 quantisation floor(x * 255) / 255 of a float map
    (the sigma-0.2 blur behind it is the identity on 8-bit data)."""
    lib = load_library()
    _need_gpu(labels, "labels")
    out = torch.empty_like(labels)
    with torch.cuda.device(labels.device):
        _check(lib.ssp_op_label_quantize(_ptr(labels), _ptr(out), labels.numel(), _stream()))
    return out


def op_sem_finalize(sem_warped, valid, n_classes=133):
    lib = load_library()
    _need_gpu(sem_warped, "sem")
    _need_gpu(valid, "valid")
    out = torch.empty(sem_warped.shape, dtype=torch.int64, device=sem_warped.device)
    with torch.cuda.device(sem_warped.device):
        _check(lib.ssp_op_sem_finalize(_ptr(sem_warped), _ptr(valid), _ptr(out), sem_warped.numel(), int(n_classes), _stream()))
    return out
