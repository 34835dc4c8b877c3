"""Helpers shared by the oracle-vs-golden (CPU) and HIP-vs-oracle (GPU) tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def g4_full_inputs(seed=17, B=2, Hc=30, Wc=40):
    """Regenerates the descriptor maps of g4_sparse_loss_full.npz (same draws as oracle/make_goldens.py)."""
    rs = np.random.RandomState(seed)
    d = rs.randn(B, 256, Hc, Wc).astype(np.float32)
    dw = (0.6 * d + 0.8 * rs.randn(B, 256, Hc, Wc)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dw /= np.linalg.norm(dw, axis=1, keepdims=True)
    return d, dw


def indices_from(g, prefix, B):
    out = []
    for i in range(B):
        out.append({"uv_a": torch.from_numpy(g["%suv_a%d" % (prefix, i)].astype(np.float32)),
                    "uv_b": torch.from_numpy(g["%suv_b%d" % (prefix, i)].astype(np.float32)),
                    "nm_b": torch.from_numpy(g["%snm_b%d" % (prefix, i)].astype(np.int64))})
    return out


def sample_from(g):
    s = {}
    for k, v in g.items():
        if k.startswith("in/"):
            s[k[3:]] = torch.from_numpy(v)
    return s


def g10_inputs(seed, B, Hc, Wc):
    """Regenerates the descriptor maps of g10_dense_loss_*.npz (same draws as oracle/make_goldens.py:g10_dense_loss).
    NOTE: the generator draws the homographies and the mask AFTER the maps from the same stream; those are stored."""
    rs = np.random.RandomState(seed)
    d = rs.randn(B, 256, Hc, Wc).astype(np.float32)
    dw = (0.8 * d + 0.6 * rs.randn(B, 256, Hc, Wc)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dw /= np.linalg.norm(dw, axis=1, keepdims=True)
    return d, dw


def g13_sample(g, H=120, W=160, B=2):
    """Decodes the single-view inputs of g13_single_view_*.npz (8-bit image, bit-packed labels / valid mask)."""
    n = B * H * W
    return {"image": torch.from_numpy(g["in/image_u8"].astype(np.float32) / 255.0),
            "labels_2D": torch.from_numpy(np.unpackbits(g["in/labels_2D"])[:n].reshape(B, 1, H, W).astype(np.float32)),
            "valid_mask": torch.from_numpy(np.unpackbits(g["in/valid_mask"])[:n].reshape(B, 1, H, W).astype(np.float32))}
