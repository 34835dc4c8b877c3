"""Container-only harness that imports the REAL reference (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_goldens.py to generate the committed
fixtures under tests/golden/ and by tests that are skipped when /root/reference is absent.
Nothing here is shipped, nothing here runs on the GPU box, and nothing of the reference's
source is copied: we only put stub modules into sys.modules for third-party packages the
image lacks (SURVEY.md section 8c lists them) and then `import` the reference in place.
"""
import collections
import collections.abc
import os
import sys
import types

REF_ROOT = os.environ.get("SSP_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


class _Writer:
    """tensorboardX.SummaryWriter stand-in: records scalars."""

    def __init__(self, *a, **k):
        self.scalars = {}

    def add_scalar(self, name, value, n_iter=0):
        self.scalars[name] = float(value)

    def add_image(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Install stubs + sys.path so that `import Train_model_heatmap_all` works here."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    if not hasattr(collections, "Mapping"):  # utils/tools.py:18 uses the py<3.10 alias
        collections.Mapping = collections.abc.Mapping
    import numpy as _np
    if not hasattr(_np, "int"):  # Train_model_heatmap_all.py:706 uses the alias numpy removed in 1.24
        _np.int = int
    if "cv2" not in sys.modules:
        _stub("cv2", __version__="0.0-stub")
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.ops = _stub("torchvision.ops")
    if "torch_poly_lr_decay" not in sys.modules:

        class PolynomialLRDecay:  # only .step() is reached (Train_model_heatmap_all.py:412)
            def __init__(self, optimizer, max_decay_steps=1, end_learning_rate=0.0, power=1.0):
                self.optimizer = optimizer

            def step(self, *a):
                pass

        _stub("torch_poly_lr_decay", PolynomialLRDecay=PolynomialLRDecay)
    if "tensorboardX" not in sys.modules:
        _stub("tensorboardX", SummaryWriter=_Writer)
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            _stub("tqdm", tqdm=lambda x, *a, **k: x)
    if "imageio" not in sys.modules:  # export.py:18 (only imread, never reached by the fixtures)
        try:
            import imageio  # noqa: F401
        except Exception:
            _stub("imageio", imread=None)
    if "torchgeometry" not in sys.modules:
        # utils/losses.py:129-135 calls tgm.contrib.SpatialSoftArgmax2d; the package is absent, so the stub
        # forwards to the oracle's restatement of its published algorithm: that ONE sub-step of the export
        # fixtures is therefore self-referential ("parity unpinned", see oracle/cpu_ref.py).
        class SpatialSoftArgmax2d:
            def __init__(self, normalized_coordinates=True):
                assert not normalized_coordinates

            def __call__(self, x):
                from oracle import cpu_ref
                return cpu_ref.spatial_soft_argmax2d(x)

        tgm = _stub("torchgeometry")
        tgm.contrib = _stub("torchgeometry.contrib", SpatialSoftArgmax2d=SpatialSoftArgmax2d)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # the trainer imports a module that the reference repo itself does not contain
    # (Train_model_heatmap_all.py:19)
    import utils.loss_functions  # noqa: F401  (namespace package of the reference)

    _stub("utils.loss_functions.min_norm_solvers", MinNormSolver=object)
    _installed = True


def base_config(semantic=False, H=240, W=320, batch=2, lr=0.001, lambda_loss=1, warped_pair=True,
                multi_task=True, gaussian=True, real_batch=None):
    """The keys of configs/superpoint_coco_train_(wsem_)heatmap.yaml that the step reads, plus the
    defaults the shipped yaml files lack (SURVEY.md section 5 'Shipped-config defects')."""
    return {
        "data": {
            "dataset": "Coco", "semantic": bool(semantic),
            "gaussian_label": {"enable": bool(gaussian), "params": {"GaussianBlur": {"sigma": 0.2}}},
            "preprocessing": {"resize": [H, W]},
            "warped_pair": {"enable": bool(warped_pair), "valid_border_margin": 3},
        },
        "front_end_model": "Train_model_heatmap_all",
        "model": {
            "name": "SuperPointNet_gauss2_ssmall" if semantic else "SuperPointNet_gauss2",
            "params": {},
            "detector_loss": {"loss_type": "softmax"},
            "batch_size": batch, "real_batch_size": real_batch or batch, "eval_batch_size": batch,
            "learning_rate": lr, "detection_threshold": 0.015, "lambda_loss": lambda_loss, "nms": 4,
            "dense_loss": {"enable": False, "params": {"descriptor_dist": 4, "lambda_d": 800}},
            "sparse_loss": {"enable": True, "params": {
                "num_matching_attempts": 1000, "num_masked_non_matches_per_match": 100,
                "lamda_d": 1, "dist": "cos", "method": "2d"}},
            "multi_task_loss": bool(multi_task),
            "seg_head": {"loss": "cross_entropy"},
        },
        "retrain": True, "reset_iter": True, "train_iter": 200000, "validation_interval": 1000,
        "tensorboard_interval": 200, "save_interval": 5000, "validation_size": 10, "pretrained": None,
    }


def make_trainer(config, state_dict=None):
    """Instantiate the reference trainer on CPU exactly as train4.py:81-94 does
    (ctor -> loadModel -> dataParallel), optionally loading our deterministic weights."""
    install()
    import copy
    import torch
    import Train_model_heatmap_all as T

    T.Train_model_heatmap_all.default_config = copy.deepcopy(T.Train_model_heatmap_all.default_config)
    agent = T.Train_model_heatmap_all(copy.deepcopy(config), device="cpu")
    agent.writer = _Writer()
    agent.loadModel()
    if state_dict is not None:
        agent.net.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()})
    agent.dataParallel()
    return agent
