"""MI355X-native pair-training path of Semantic-SuperPoint (hand-written HIP for gfx950 behind the
reference's own plugin API).  Import as `semantic_superpoint_amd` (alias package at the repo root).

  lib.Engine                         ctypes owner of one libssp_hip handle + its torch-allocated HBM
  models.SuperPointNet_gauss2        drop-in for the reference's models/SuperPointNet_gauss2.py
  models.SuperPointNet_gauss2_ssmall drop-in for models/SuperPointNet_gauss2_ssmall.py
  Train_model_heatmap_all            drop-in trainer plugin (train_val_sample)
"""
from . import hipbuild  # noqa: F401
from .hipbuild import build  # noqa: F401
from . import lib  # noqa: F401
from .lib import Engine, load_library  # noqa: F401

__all__ = ["build", "lib", "Engine", "load_library"]
