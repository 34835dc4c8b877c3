"""GPU: the drop-in boundary (nn.Module classes + trainer plugin) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as C

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("arch", ["SuperPointNet_gauss2", "SuperPointNet_gauss2_ssmall"])
def test_module_forward_backward_autograd(arch):
    """net(x) twice then loss.backward(): exactly the call pattern of Train_model_heatmap_all.py:258-262,407."""
    from semantic_superpoint_amd import models
    dev = _dev()
    sd = C.init_state_dict(arch, seed=2)
    net = getattr(models, arch)()
    net.load_state_dict({k: torch.as_tensor(np.array(v)) for k, v in sd.items()})
    net = net.to(dev)
    rs = np.random.RandomState(0)
    x1 = torch.from_numpy(rs.uniform(0, 1, (2, 1, 64, 96)).astype(np.float32))
    x2 = torch.from_numpy(rs.uniform(0, 1, (2, 1, 64, 96)).astype(np.float32))
    tsd = C.to_torch(sd, requires_grad=True)
    r1, r2 = C.forward(tsd, x1, arch), C.forward(tsd, x2, arch)
    o1, o2 = net(x1.to(dev)), net(x2.to(dev))
    assert set(o1.keys()) == set(r1.keys())
    for k in r1:
        assert (o1[k].cpu() - r1[k]).abs().max() < TOL and (o2[k].cpu() - r2[k]).abs().max() < TOL, k
    gs = {k: torch.from_numpy(rs.randn(*r1[k].shape).astype(np.float32)) * (0.05 if k == "sem" else 1.0) for k in r1}
    ref_loss = sum((r1[k] * gs[k]).sum() + 0.5 * (r2[k] * gs[k]).sum() for k in r1)
    ref_loss.backward()
    loss = sum((o1[k] * gs[k].to(dev)).sum() + 0.5 * (o2[k] * gs[k].to(dev)).sum() for k in o1)
    loss.backward()
    torch.cuda.synchronize()
    noisy = {c + ".bias" for c, bn, _, _, _ in C.layer_table(arch) if bn is not None}
    for k, p in net.named_parameters():
        if k in noisy:
            continue
        r = tsd[k].grad.double().reshape(-1)
        m = p.grad.detach().cpu().double().reshape(-1)
        assert float((m - r).norm() / (r.norm() + 1e-30)) < 1.5e-2, k
    # buffers are views of the engine: running statistics advanced twice, state_dict round-trips
    st = net.state_dict()
    assert int(st["bnPb.num_batches_tracked"]) == 2
    assert (st["inc.conv.conv.1.running_mean"].cpu() - tsd["inc.conv.conv.1.running_mean"]).abs().max() < TOL
    # torch.optim.Adam on net.parameters() (the reference's optimizer) updates the engine's flat buffer in place
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = net.engine().params.clone()
    opt.step()
    assert not torch.equal(before, net.engine().params)
    net.eval()
    with torch.no_grad():
        oe = net(x1.to(dev))
    assert oe["semi"].shape == (2, 65, 8, 12)


@pytest.mark.parametrize("arch,semantic,method,dist", [
    ("SuperPointNet_gauss2", False, "2d", "cos"), ("SuperPointNet_gauss2_ssmall", True, "2d", "cos"),
    ("SuperPointNet_gauss2", False, "1d", "euclidean"),   # the other values descriptor_loss_sparse accepts (sparse_loss.py:76-77)
    ("SuperPointNet_gauss2", False, None, None)])          # keys absent from the config: the function's own defaults ("1d", "cos")
def test_trainer_plugin_two_steps_vs_oracle(arch, semantic, method, dist, tmp_path):
    """Train_model_heatmap_all.train_val_sample with the reference-faithful host sampler and the same numpy/torch
    seeds as the oracle trainer: scalar_dict (incl. post-step eta) after two optimizer steps."""
    from semantic_superpoint_amd.Train_model_heatmap_all import Train_model_heatmap_all as T
    B, H, W = 2, 64, 96
    cfg = {"data": {"semantic": semantic, "gaussian_label": {"enable": True}, "warped_pair": {"enable": True}},
           "model": {"name": arch, "params": {}, "batch_size": B, "real_batch_size": B, "learning_rate": 1e-3,
                     "lambda_loss": 1, "multi_task_loss": True, "dense_loss": {"enable": False},
                     "detector_loss": {"loss_type": "softmax"},
                     "sparse_loss": {"enable": True, "params": {"num_matching_attempts": 1000,
                                                                "num_masked_non_matches_per_match": 100, "lamda_d": 1}}},
           "validation_interval": 1000, "retrain": True, "reset_iter": True, "ssp_sampler": "reference"}
    if method is not None:
        cfg["model"]["sparse_loss"]["params"].update({"dist": dist, "method": method})

    class W_:
        def __init__(self):
            self.s = {}

        def add_scalar(self, name, v, it):
            self.s[name] = v

    agent = T(cfg, save_path=tmp_path, device="cuda:0")
    agent.writer = W_()
    agent.loadModel()
    sd = C.init_state_dict(arch, seed=6)
    agent.net.load_state_dict({k: torch.as_tensor(np.array(v)) for k, v in sd.items()})
    agent.dataParallel()
    sample = C.make_synthetic_pair(B, H, W, seed=8, semantic=semantic, kp_prob=0.01)
    tr = C.Trainer(arch, sd, lr=1e-3, sparse_method=method or "1d", sparse_dist=dist or "cos")
    for it in range(2):
        np.random.seed(10 + it); torch.manual_seed(20 + it)
        tr.train_val_sample(sample, n_iter=it, train=True)
        np.random.seed(10 + it); torch.manual_seed(20 + it)
        loss = agent.train_val_sample(sample, n_iter=it, train=True)
        for k, ref in tr.scalar_dict.items():
            assert abs(agent.scalar_dict[k] - ref) < 2e-3 * max(1.0, abs(ref)), (it, k, agent.scalar_dict[k], ref)
        assert abs(loss - tr.scalar_dict["loss"]) < 2e-3 * max(1.0, abs(loss))
    assert "train-loss" in agent.writer.s and "train-eta_det" in agent.writer.s
    v = agent.train_val_sample(sample, n_iter=2, train=False)  # validation path
    assert np.isfinite(v)
    p = agent.saveModel()
    ck = torch.load(p, map_location="cpu")
    assert set(ck) >= {"n_iter", "model_state_dict", "optimizer_state_dict", "loss"}
    assert list(ck["model_state_dict"].keys()) == [k for k, _, _ in C.state_spec(arch)]


def test_loaded_library_is_built_from_the_checked_out_sources():
    """The library the GPU tests run on carries the sha256 of exactly the csrc/ + include/ sources of this checkout
    (ssp_build_id(), compiled in by hipbuild.build): the tested kernels are HEAD's by content, not by file time."""
    import semantic_superpoint_amd as ssp
    assert ssp.lib.build_id() == ssp.hipbuild.source_id()
