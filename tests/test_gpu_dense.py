"""GPU: dense descriptor loss (utils/utils.py:779-893) -- the HIP kernels through the
C ABI against the G10 fixtures (reference values and autograd gradients) and, inside the pair step, against the G6
dense training-step fixtures of the real Train_model_heatmap_all (multi-task and uniform-sum)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as C
from tests import golden_util as G

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_dense_loss_operator_small_with_gradients():
    from semantic_superpoint_amd import lib as L
    g = G.load("g10_dense_loss_small.npz")
    dev = _dev()
    a, b = t(g["desc"]).to(dev), t(g["desc_w"]).to(dev)
    hm, mv = t(g["homographies"]), t(g["mask_valid"])
    loss, pos, neg = L.op_dense_loss(a, b, hm, mv)
    assert abs(loss - float(g["loss"])) < 1e-5 and abs(pos - float(g["pos_sum"])) < 1e-5
    assert abs(neg - float(g["neg_sum"])) < 1e-7
    _, da, db = L.op_dense_loss(a, b, hm, mv, grad=("loss", 1.0))
    assert (da.cpu() - t(g["g_loss_a"])).abs().max() < 1e-6 and (db.cpu() - t(g["g_loss_b"])).abs().max() < 1e-6
    _, da, db = L.op_dense_loss(a, b, hm, mv, grad=("multi_task", 0.5))
    assert (da.cpu() - t(g["g_mt_a"])).abs().max() < 1e-6 and (db.cpu() - t(g["g_mt_b"])).abs().max() < 1e-6


def test_dense_loss_operator_full_cell_grid():
    """30x40 cells (1200 x 1200 Gram matrix per image): values, gradient norms and strided gradient samples."""
    from semantic_superpoint_amd import lib as L
    f = G.load("g10_dense_loss_full.npz")
    dev = _dev()
    d, dw = G.g10_inputs(int(f["seed"]), 2, 30, 40)
    a, b = t(d).to(dev), t(dw).to(dev)
    hm, mv = t(f["homographies"]), t(f["mask_valid"])
    loss, pos, neg = L.op_dense_loss(a, b, hm, mv)
    assert abs(loss - float(f["loss"])) < 1e-5 * max(1.0, float(f["loss"]))
    assert abs(pos - float(f["pos_sum"])) < 1e-5 and abs(neg - float(f["neg_sum"])) < 1e-7
    for mode, scale, pre in (("loss", 1.0, "g_loss"), ("multi_task", 0.5, "g_mt")):
        _, da, db = L.op_dense_loss(a, b, hm, mv, grad=(mode, scale))
        for mine, nm in ((da, pre + "_a"), (db, pre + "_b")):
            mine = mine.cpu()
            n_ref = float(f[nm + "_norm"])
            assert abs(float(mine.norm()) - n_ref) < 1e-4 * n_ref, nm
            assert (mine.reshape(-1)[::977][:256] - t(f[nm + "_slice"])).abs().max() < 1e-5 * float(mine.abs().max()) + 1e-9, nm
    # oracle on the same inputs: mask decisions and sums agree
    o = C.descriptor_loss_dense(t(d), t(dw), hm, mv)
    assert abs(float(o[0]) - loss) < 1e-5 and float(o[1].sum()) == float(f["mask_sum"])


@pytest.mark.parametrize("tag", ["sp_dense_64x96", "sp_dense_uniform_64x96"])
def test_pair_step_with_dense_loss_golden(tag):
    """The full step with model.dense_loss.enable against the reference trainer's scalars and gradients (G6)."""
    from semantic_superpoint_amd.lib import Engine, SCALAR_NAMES
    arch = "SuperPointNet_gauss2"
    g = G.load("g6_step_%s.npz" % tag)
    sample = {k: v.to(_dev()).contiguous() for k, v in G.sample_from(g).items()}
    B, _, H, W = sample["image"].shape
    e = Engine(arch, B, H, W, _dev(), dense_loss=True)
    e.load_state_dict(C.init_state_dict(arch, seed=23))
    e.zero_grad()
    mt = "uniform" not in tag
    sc = e.pair_step(sample, train=True, lambda_loss=1.0, multi_task=mt, dense={"descriptor_dist": 4, "lambda_d": 800})
    torch.cuda.synchronize()
    sc = dict(zip(SCALAR_NAMES, sc.cpu().tolist()))
    for name in ("loss", "loss_det", "loss_det_warp", "loss_desc", "positive_dist", "negative_dist"):
        ref = float(g["step0/" + name])
        assert abs(sc[name] - ref) < TOL * max(1.0, abs(ref)), (name, sc[name], ref)
    gd = e.grad_dict()
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
    for k in C.param_keys(arch):
        if k in noisy or ("grad_norm/" + k) not in g:
            continue
        n_ref = float(g["grad_norm/" + k])
        mine = gd[k].cpu().reshape(-1)
        assert abs(float(mine.norm()) - n_ref) < 5e-3 * n_ref + 1e-6, (k, float(mine.norm()), n_ref)
        assert (mine[:64] - t(g["grad_slice/" + k])).abs().max() < 2e-2 * float(mine.abs().max()) + 1e-6, k
    if mt:
        assert (gd["eta"].cpu() - t(g["grad/eta"])).abs().max() < 1e-3
    # the sparse-loss engine refuses the dense request loudly
    e2 = Engine(arch, B, H, W, _dev())
    with pytest.raises(RuntimeError):
        e2.pair_step(sample, train=True, dense={"descriptor_dist": 4})


def test_trainer_plugin_with_dense_loss_config():
    """model.dense_loss.enable through the drop-in trainer: two steps vs the oracle trainer."""
    from semantic_superpoint_amd.Train_model_heatmap_all import Train_model_heatmap_all
    dev = _dev()
    B, H, W = 2, 64, 96
    cfg = {"data": {"dataset": "Coco", "semantic": False, "gaussian_label": {"enable": False},
                    "warped_pair": {"enable": True}},
           "model": {"name": "SuperPointNet_gauss2", "params": {}, "batch_size": B, "real_batch_size": B,
                     "learning_rate": 1e-3, "lambda_loss": 1, "multi_task_loss": True, "detector_loss": {"loss_type": "softmax"},
                     "dense_loss": {"enable": True, "params": {"descriptor_dist": 4, "lambda_d": 800}},
                     "sparse_loss": {"enable": True, "params": {"lamda_d": 1, "dist": "cos", "method": "2d"}}},
           "retrain": True, "reset_iter": True, "train_iter": 10, "validation_interval": 5, "tensorboard_interval": 100,
           "save_interval": 100}
    agent = Train_model_heatmap_all(cfg, device=dev)
    assert agent.desc_loss_type == "dense"
    agent.loadModel()
    sd = C.init_state_dict("SuperPointNet_gauss2", seed=4)
    agent.net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    agent.dataParallel()
    sample = C.make_synthetic_pair(B, H, W, seed=8)
    tr = C.Trainer("SuperPointNet_gauss2", sd, lr=1e-3, gaussian=False, dense={"descriptor_dist": 4, "lambda_d": 800})
    for it in (1, 2):
        agent.train_val_sample(sample, n_iter=it, train=True)
        tr.train_val_sample(sample, n_iter=it, train=True)
        for name in ("loss", "loss_desc", "positive_dist", "negative_dist"):
            ref = tr.scalar_dict[name]
            assert abs(agent.scalar_dict[name] - ref) < 2e-3 * max(1.0, abs(ref)), (it, name, agent.scalar_dict[name], ref)
